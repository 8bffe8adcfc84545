"""Drop-in ``VectorQuantizer`` / ``VectorQuantizerEMA`` for the reference's ``vqvae/vqvae.py``.

Same constructor signatures (vqvae/vqvae.py:7, :80), same ``forward(inputs, srl=None)`` returning
``(loss, quantized, perplexity, encodings)`` (:76, :165), same attributes and ``state_dict`` keys
(``_embedding.weight``, ``_ema_w``, ``_ema_cluster_size``), same train/eval gate (``self.training``
only, :131), same ``get_latent / get_indices / get_hidden`` (:167-238).  All arithmetic runs in the
hand-written HIP kernels behind ``include/t5vq.h``; this file only orders the launches and keeps
autograd informed.  There is no CPU path: tensors must live on a ROCm device.

Differences that are visible and deliberate:
* EMA state is updated in place under ``no_grad`` instead of re-binding fresh Parameters
  (:132,137,142,144) -- identical values, stable tensor identities for optimizers/checkpoints.
* ``encodings`` (the 4th return value, discarded by every caller of the reference) is the dense fp32
  one-hot by default; set ``module.encodings = "indices"`` to get the ``[N, 1]`` int64 indices instead and
  skip the N x K materialisation (the benchmark does).
* ``process_group`` / ``world tokens``: see ``t5vqvae_amd/sharded.py``.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
from torch import nn

from . import _lib, ops
from .roles import INDEX_MODES

BUCKET_MIN_TOKENS = 1024     # ranged search: tokens are visited role by role from this many on (= the library's rule)

_KEY_NONE = 0x7FFFFFFFFFFFFFFF
# per-module state of the search="auto" back-off (events and pinned buffers: kept out of the module's __dict__ so
# that copy.deepcopy / pickling of the module are unaffected)
_FILTER_STATE = weakref.WeakKeyDictionary()


def _filter_state(module):
    st = _FILTER_STATE.get(module)
    if st is None:
        st = _FILTER_STATE[module] = {"skip": 0, "pending": None}
    return st


# |e_k|^2 of the codebook, kept between calls (the reference recomputes it every call, vqvae/vqvae.py:107).  The entry
# is (padded norms, data_ptr, _version) of the weight it was computed from: the finalize kernel that writes the
# codebook also writes its norms, so in EMA training no call reads the codebook for them.  Any tracked in-place change
# of the Parameter (optimizer step, load_state_dict, copy_, .to()) changes the version or the pointer and drops the
# entry.  Writes through ``weight.data`` are invisible to the version counter: call ``module.invalidate_code_norms()``
# after one (the constructor's own ``.data`` initialisation happens before any entry exists).
_EE_STATE = weakref.WeakKeyDictionary()


# fp16 image of the codebook for the filtered search (ops.code_image), kept between calls like |e|^2 and under the same
# key: (block, data_ptr, _version).  The kernel that writes the codebook (EMA update) writes its image too, a search that
# finds no valid image makes it in the module's block instead of its own workspace -- either way the NEXT search's
# preparation pass reads no codebook row.  Only the base module's own flows keep one.
_IMG_STATE = weakref.WeakKeyDictionary()


# State block of the single-launch small-shape forward (tickets; zeroed once, left zeroed by every launch), one per
# module and device: calls of one module are ordered, calls of different modules may overlap.
_SMALL_STATE = weakref.WeakKeyDictionary()


class _Plan:
    """Per-call configuration handed to the autograd function (plain Python, no tensors)."""
    __slots__ = ("ema", "training", "cc", "decay", "eps", "loss", "batch", "group", "world_tokens", "module", "xx",
                 "grad_scale", "ximg")


def _scalar_of(buf: torch.Tensor, i: int) -> torch.Tensor:
    """0-dim tensor over element i of ``buf`` that is NOT an autograd view (callers may do ``loss += ...``
    on it, which a view created inside a custom Function would refuse), and costs no kernel."""
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(buf.untyped_storage(), buf.storage_offset() + i, ())


class _VQFunction(torch.autograd.Function):
    """(flat inputs [N,D], codebook [K,D]) -> (loss, straight-through output, perplexity, indices)."""

    @staticmethod
    def forward(ctx, x, weight, ranges, plan: _Plan):
        mod = plan.module
        need_gx = ctx.needs_input_grad[0]
        need_gw = ctx.needs_input_grad[1] and not plan.ema
        mse = plan.loss == "mse"
        r = mod._launch(x, weight.detach(), ranges, plan, want_diff=(need_gx or need_gw) and mse,
                        want_q=(need_gx or need_gw) and not mse)
        ctx.plan = plan
        ctx.dims = r["dims"]                      # (global tokens, D, rows of the codebook gradient, batch)
        ctx.need_gw = need_gw
        ctx.save_for_backward(r["diff"], r["q"], x if not mse else None, r["idx"])
        ctx.mark_non_differentiable(r["ppl"], r["idx"])
        return r["loss"], r["ste"], r["ppl"], r["idx"]

    @staticmethod
    def backward(ctx, g_loss, g_ste, g_ppl, g_idx):
        diff, q, x, idx = ctx.saved_tensors
        plan = ctx.plan
        n_total, d, k, batch = ctx.dims
        gs = plan.grad_scale      # token-sharded: see _QuantizerBase.gradient_reduction (1.0 on a single GPU)
        gx = gw = None
        if plan.loss == "mse":
            if ctx.needs_input_grad[0]:
                gx = ops.ste_backward(g_ste, diff, g_loss, gs * plan.cc * 2.0 / (n_total * d))
            if ctx.need_gw:
                gw = ops.scatter_rows(diff, idx, k, scale_dev=g_loss, scale_host=gs * 2.0 / (n_total * d)) \
                    if g_loss is not None else torch.zeros(k, d, dtype=torch.float32, device=diff.device)
        else:
            gx_, gq = ops.cosine_backward(x, q, g_ste, g_loss, gs * plan.cc / batch, gs / batch, ctx.need_gw)
            if ctx.needs_input_grad[0]:
                gx = gx_
            if ctx.need_gw:
                gw = ops.scatter_rows(gq, idx, k)
        return gx, gw, None, None


class _QuantizerBase(nn.Module):
    _is_ema = False
    encodings = "dense"          # "dense": fp32 one-hot [N,K] (reference behaviour) | "indices": int64 [N,1]
    process_group = None         # set by sharded.TokenSharded*: all-reduce the EMA statistics over this group
    world_tokens = 0             # global token count when ranks hold unequal shares (0 = n_local * world)

    # Token-sharded training (process_group set): the loss VALUE is that of the concatenated batch on every rank, but
    # each rank back-propagates through its own tokens only.  "mean": the caller averages parameter gradients over the
    # group afterwards (DistributedDataParallel's default) -- the local gradients (to the inputs, and to the codebook
    # of the plain variant) are scaled by the group size, so that the average equals the single-process gradient on
    # the concatenated batch.  "sum": the caller sums them (or all-reduces with SUM) -- no scaling.
    gradient_reduction = "mean"
    # Token-sharded exchange of the EMA statistics.  "serial": forward (search, gather + statistics in one pass over x)
    # -> all-reduce -> finalize, everything on one stream.  "overlap": the statistics pass runs first and alone, the
    # all-reduce of its head starts at once (async_op) and runs under the second pass, which writes the per-token
    # outputs (straight-through values, q - x, indices); finalize waits for it.  Same results; costs one more read of x
    # ("overlap the all-reduce with K2").  bench.py --gpus N times both.
    stats_exchange = "serial"

    def __init__(self):
        super().__init__()
        # both libraries are loaded here, on the constructing thread: the first caller of a backward-only op is the
        # autograd worker thread, and a dlopen (or a failing one, retried) must never happen there
        _lib.load_all()

    def _flatten(self, inputs, srl):
        flat = inputs.view(-1, self._embedding_dim)          # same view rule as vqvae/vqvae.py:103
        ops.require_device(flat, "inputs")
        if flat.dtype != torch.float32:
            raise RuntimeError(f"inputs must be float32 (got {flat.dtype}); the reference path is fp32")
        if self._embedding_dim % 4 or self._embedding_dim > 2048:
            raise RuntimeError(f"embedding_dim must be a multiple of 4 and at most 2048 (got {self._embedding_dim}): "
                               "limits of the gather / statistics kernels")
        ranges = None
        if self.disentanglement:
            ranges = srl.view(-1, 2)                         # vqvae/vqvae.py:111
            if ranges.dtype != torch.int64:
                ranges = ranges.to(torch.int64)
            ranges = ranges.contiguous()
        return flat.contiguous(), ranges

    def _check_loss_name(self):
        if self.loss not in ops.LOSS_KINDS:
            exit("Error: wrong loss name")                   # reference: vqvae/vqvae.py:67,155 (SystemExit)

    bucket_roles = True          # disentangled mode: visit tokens role by role (see _nearest_keys); False is a
    #                              diagnostic (tools/bench_ranged.py) and takes the fine-grained entry points
    # How the nearest code is found.  "scan": fused fp32 MFMA distance + argmin over all codes.  "filter": an fp16
    # MFMA pass with a proven error bound leaves the few codes that can be the fp32 argmin and the scan's fp32
    # arithmetic decides among them -- the SAME indices, bit for bit, several times faster on large problems.
    # "auto": filter when it applies (D % 64 == 0) and pays (N*K >= filter_min_work), with a
    # back-off to the scan while the filter leaves too many tokens unsettled (degenerate codebooks: all rows equal).
    search = "auto"
    filter_min_work = 1 << 23    # N*K from which the filter pays (tools/bench_crossover.py: 0.85x at 2^21, 1.15x at
                                 # 2^22, 1.7x at 2^23, 2.8x at 2^24; C2: 2e6 -> scan, C1: 2.9e7 -> filter)
    filter_backoff_calls = 32    # calls to stay on the scan after a call with > 1/8 of the tokens unsettled
    # EMA statistics.  False: embed_sum by fp32 atomics (fastest; last bits depend on arrival order).  True: 64-bit
    # fixed-point sums (t5vq_embed_sum_fixed_f32): bit-reproducible from run to run and for any token sharding, and
    # exact up to one final rounding; costs one more pass over x.
    deterministic = False
    # True: small problems (fewer than 256 tiles of 128 x 128, D % 256 == 0) without an EMA update -- evaluation, or the
    # plain variant -- take ONE launch (t5vq_vq_forward_small_f32): same results bit for bit.  Off by default: measured
    # at C2 it runs level with the three-launch chain (45.0 against 45.7 us back to back), so the
    # default stays the path with the longer record.
    small_forward = False
    index_mode = "reference"     # disentangled mode: "reference" = slice-relative indices used as rows (the
    #                              reference's behaviour, vqvae/vqvae.py:116-119) | "absolute" = start + argmin

    def _nearest_keys(self, x, w, ranges, code_offset=0, xx=None, exact_keys=True, use_filter=None, images=None):
        """Packed (distance, index) keys of the nearest admissible codeword per token.

        Disentangled mode with many tokens: the search visits the tokens role by role (``ops.order_by_range``: a
        counting sort of the [start, end) ranges on the device; no row of x is moved), so that each token tile of the
        distance kernel carries (mostly) one role and skips every code tile outside it: a grouped codebook then
        costs 2 D sum_r N_r K_r flops instead of 2 D N K.  Keys come out in token order either way."""
        if self.index_mode not in INDEX_MODES:
            raise ValueError(f"index_mode must be one of {INDEX_MODES}, got {self.index_mode!r}")
        if use_filter is None:
            use_filter = self._use_filter(x, w, ranges)
        ee = self._code_norms(w)             # cached |e|^2 of this codebook, or None
        order = None
        if ranges is not None and self.bucket_roles and x.shape[0] >= BUCKET_MIN_TOKENS:
            order = ops.order_by_range(ranges, w.shape[0], code_offset)
        if use_filter:
            # norms that are not given are written by the search's own first pass (bit-identical to row_sqnorms)
            return self._filtered_keys(x, w, exact_keys or code_offset != 0, ranges, code_offset, xx=xx, ee=ee,
                                       order=order, images=images)
        if xx is None or ee is None:
            xx_, ee_ = ops.row_sqnorms(x if xx is None else None, w if ee is None else None)
            xx, ee = (xx_ if xx is None else xx), (ee_ if ee is None else ee)
        return ops.dist_argmin(x, w, xx, ee, ranges, code_offset=code_offset, order=order)

    # --- |e|^2 kept between calls ------------------------------------------------------------------------
    code_norm_cache = True       # False: recompute the codebook norms in every call, as the reference does (:107)

    def invalidate_code_norms(self):
        """Drop the cached |e|^2 (needed only after an in-place write through ``_embedding.weight.data``, which the
        Parameter's version counter does not see)."""
        _EE_STATE.pop(self, None)
        _IMG_STATE.pop(self, None)

    # The codebook's fp16 image for the filtered search, kept between calls.  "always" (default): the update kernel also
    # writes the image of the codebook it writes, and a search that finds no valid image makes it in the module's block,
    # so the next search's preparation pass reads no codebook row (C3: preparation 75.6 -> 71.6 us, update 18.4 -> 20.3 us;
    # the saving grows with K / N).  "eval": only calls WITHOUT an EMA update keep it, an update drops it.  False: every
    # search makes its own in its workspace.
    code_image_cache = "always"

    def _images(self, w, ee, use_filter, update, ximg):
        """Keyword arguments about fp16 images for this call's ops: ``code_image`` / ``code_image_valid`` for the search
        (and, with ``update``, the block the new codebook's image goes to), ``token_image`` from the producer of x."""
        out = {}
        if ximg is not None and use_filter:
            out["token_image"] = ximg
        param = self._embedding.weight
        mode = "always" if self.code_image_cache is True else self.code_image_cache
        if mode not in (False, None, "eval", "always"):
            raise ValueError(f"code_image_cache must be False, 'eval' or 'always', got {self.code_image_cache!r}")
        if update and mode != "always":
            _IMG_STATE.pop(self, None)                       # the update rewrites the codebook behind the image's back
            return out
        if not mode or ee is None or w.data_ptr() != param.data_ptr() or w.shape[1] % 64:
            return out
        ent = _IMG_STATE.get(self)
        if ent is not None and (ent[0].device != w.device or ent[3] != tuple(w.shape)):
            ent = None
        if ent is None:
            if not use_filter:
                return out                                   # nobody has asked for an image yet
            ent = (ops.code_image(w.shape[0], w.shape[1], w.device), None, None, tuple(w.shape))
        valid = ent[1] == param.data_ptr() and ent[2] == param._version
        if use_filter or update:
            out["code_image"] = ent[0]
            out["code_image_valid"] = bool(valid and use_filter)
            _IMG_STATE[self] = (ent[0], None, None, ent[3])     # in the making: _image_done() stamps it
        return out

    def _image_done(self, images):
        """After the call's last op: the block holds the image of the codebook as the call leaves it (made by the
        search's first pass, or by the update) -- stamp it with the Parameter's pointer and version as they are NOW."""
        if "code_image" in images:
            param = self._embedding.weight
            ent = _IMG_STATE.get(self)
            if ent is not None and ent[0] is images["code_image"]:
                _IMG_STATE[self] = (ent[0], param.data_ptr(), param._version, ent[3])

    def _code_norms(self, w, make=False):
        """The cached padded |e|^2 of ``w`` (this module's codebook tensor) if still valid, else None -- or, with
        ``make``, a fresh one computed now (one launch) and remembered."""
        if not self.code_norm_cache:
            return None
        param = self._embedding.weight
        if w.data_ptr() != param.data_ptr():
            return None                                  # not this module's codebook (e.g. a gathered one)
        ent = _EE_STATE.get(self)
        if ent is not None and ent[1] == param.data_ptr() and ent[2] == param._version and ent[0].device == w.device:
            return ent[0]
        if not make:
            return None
        ee = ops.code_sqnorms(w)
        _EE_STATE[self] = (ee, param.data_ptr(), param._version)
        return ee

    def _use_filter(self, x, w, ranges) -> bool:
        if self.search not in ("auto", "scan", "filter"):
            raise ValueError(f"search must be 'auto', 'scan' or 'filter', got {self.search!r}")
        n, d = x.shape
        eligible = d % 64 == 0 and d <= 2048
        if self.search == "scan" or not eligible:
            if self.search == "filter":
                raise RuntimeError("search='filter' needs D % 64 == 0 and D <= 2048")
            return False
        if self.search == "filter":
            return True
        if n * w.shape[0] < self.filter_min_work:
            return False
        st = _filter_state(self)
        if st["pending"] is not None and not torch.cuda.is_current_stream_capturing():
            event, host, n_prev = st["pending"]
            if event.query():                          # the previous call's count has landed: no host wait
                st["pending"] = None
                if int(host[0]) * 8 > n_prev:
                    st["skip"] = self.filter_backoff_calls
        if st["skip"] > 0:
            st["skip"] -= 1
            return False
        return True

    def _filtered_keys(self, x, w, exact_keys, ranges=None, code_offset=0, xx=None, ee=None, order=None, images=None):
        keys, ovf = ops.dist_argmin_filtered(x, w, xx=xx, ee=ee, want_overflow=True, exact_keys=exact_keys,
                                             ranges=ranges, code_offset=code_offset, order=order, **(images or {}))
        self._note_unsettled(ovf, x.shape[0])
        return keys

    def _note_unsettled(self, ovf, n_tokens):
        """Remember the filter's unsettled-token count (device) and, in "auto", start its asynchronous read-back."""
        if self.search == "auto" and not torch.cuda.is_current_stream_capturing():
            st = _filter_state(self)
            if st["pending"] is None:                  # asynchronous read-back of the unsettled-token count
                host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                host.copy_(ovf, non_blocking=True)
                event = torch.cuda.Event()
                event.record()
                st["pending"] = (event, host, n_tokens)
        self.last_unsettled = ovf

    def _absolute(self, keys, ranges):
        """index_mode "absolute": slice-relative index + start (the index is the key's low word; rows whose range
        holds no code keep the 'none' key)."""
        if ranges is None or self.index_mode != "absolute":
            return keys
        return torch.where(keys != _KEY_NONE, keys + ranges[:, 0], keys)

    def _launch(self, x, w, ranges, plan, want_diff, want_q):
        """Order the kernels of one call (single GPU, or token-sharded when ``plan.group`` is set).

        Single GPU, whole-codebook search: ONE op, t5vq::vq_step (forward + finalize).  Token-sharded: t5vq::vq_forward,
        ONE all-reduce of the statistics head [embed_sum | counts | loss_sum], t5vq::vq_finalize.  Disentangled
        codebooks go the same way (the composite forward orders the tokens by role itself, on the device); only
        ``index_mode="absolute"``, the diagnostic ``bucket_roles=False`` and the order-independent statistics chain the
        fine-grained entry points."""
        if self.index_mode not in INDEX_MODES:
            raise ValueError(f"index_mode must be one of {INDEX_MODES}, got {self.index_mode!r}")
        n, d = x.shape
        k = w.shape[0]
        dev = x.device
        update = plan.ema and plan.training
        mse = plan.loss == "mse"
        fixed = update and self.deterministic
        use_filter = self._use_filter(x, w, ranges)
        # the composite forward visits a ranged batch role by role from BUCKET_MIN_TOKENS on: switching that off, or
        # shifting the indices by their range start, needs the keys between the search and the gather
        bucketed = ranges is not None and ((not self.bucket_roles and n >= BUCKET_MIN_TOKENS)
                                           or self.index_mode == "absolute")
        if self.stats_exchange not in ("serial", "overlap"):
            raise ValueError(f"stats_exchange must be 'serial' or 'overlap', got {self.stats_exchange!r}")
        overlap = plan.group is not None and self.stats_exchange == "overlap" and not fixed

        n_total, batch, world = n, plan.batch, 1
        if plan.group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(plan.group)
            n_total = plan.world_tokens if plan.world_tokens else n * world
            batch = batch * world
        denom = float(n_total * d) if mse else float(batch)
        # EMA training keeps |e|^2 current: the finalize kernel writes the norms of the codebook it writes.  The plain
        # variant's codebook moves with the optimizer every step, so its norms are computed in the forward's own launch.
        ee = self._code_norms(w, make=plan.ema or not plan.training)
        hyper = dict(decay=plan.decay, one_minus_decay=1 - plan.decay, epsilon=plan.eps, k_epsilon=k * plan.eps,
                     commitment_cost=plan.cc, plain_variant=not plan.ema)
        cs = self._ema_cluster_size if update else None
        ema_w = self._ema_w.data if update else None
        dims = (n_total, d, k, batch)
        images = self._images(w, ee, use_filter, update, plan.ximg)
        search_images = {key: v for key, v in images.items() if use_filter}
        code_image_out = images.get("code_image") if update else None

        if not bucketed and not fixed and plan.group is None:        # the whole call behind one op
            small = None
            if (self.small_forward and not update and ranges is None and mse and not use_filter
                    and ops.small_supported(n, k, d)):
                ent = _SMALL_STATE.get(self)
                if ent is not None and ent.device == dev:
                    small = ent
                elif not torch.cuda.is_current_stream_capturing():    # never allocated (and zeroed) inside a capture
                    small = _SMALL_STATE[self] = ops.small_state(k, dev)
            ste, idx, diff, q, scalars, ovf = ops.fused_step(
                x, w, ranges, plan.loss, xx=plan.xx, ee=ee, cluster_size=cs, ema_w=ema_w, filtered=use_filter,
                want_diff=want_diff, want_q=want_q, loss_denom=denom, update=update, ee_out=ee if update else None,
                small_state=small, **images, **hyper)
            self._image_done(images)
            if ovf is not None:
                self._note_unsettled(ovf, n)
            self._last_scalars = scalars
            return {"loss": _scalar_of(scalars, 0), "ppl": _scalar_of(scalars, 1), "ste": ste, "idx": idx, "diff": diff,
                    "q": q, "dims": dims}

        with_es = update and not fixed
        reduce_loss = plan.group is not None
        kd, counts_off, loss_off, terms_off, n_terms = ops.stats_layout(n, k, d, with_es)
        work = None
        if not bucketed and not overlap:
            ste, idx, diff, q, stats, ovf = ops.fused_forward(
                x, w, ranges, plan.loss, xx=plan.xx, ee=ee, filtered=use_filter, want_diff=want_diff, want_q=want_q,
                want_embed_sum=with_es, reduce_loss=reduce_loss, **search_images)
            if ovf is not None:
                self._note_unsettled(ovf, n)
        else:
            # the search on its own (scan or filter), the keys adjusted, then the fine-grained calls
            stats = torch.empty(terms_off + n_terms, dtype=torch.float32, device=dev)
            ste = torch.empty_like(x)
            idx = torch.empty(n, dtype=torch.int64, device=dev)
            diff = torch.empty_like(x) if want_diff else None
            q = torch.empty_like(x) if want_q else None
            keys = self._absolute(self._nearest_keys(x, w, ranges, xx=plan.xx, exact_keys=False,
                                                     use_filter=use_filter, images=search_images), ranges)
            # embed_sum wanted and enough tokens: walk them grouped by code (one flush per run of equal codes)
            perm = ops.order_by_code(keys, k) if (with_es and n >= ops.ORDER_MIN_TOKENS) else None
            out = dict(ste=ste, q=q, diff=diff, idx=idx)
            st_ = dict(counts=stats[counts_off:counts_off + k], embed_sum=stats[:kd].view(k, d) if kd else None,
                       loss_partial=stats[terms_off:terms_off + n_terms], perm=perm)
            if overlap:
                # statistics first; their all-reduce runs under the pass that writes the per-token outputs
                ops.gather_stats(x, w, keys, plan.loss, **st_)
                ops.reduce_partials(stats[terms_off:terms_off + n_terms], stats[loss_off:loss_off + 1])
                work = dist.all_reduce(stats[:loss_off + 1], op=dist.ReduceOp.SUM, group=plan.group, async_op=True)
                ops.gather_stats(x, w, keys, plan.loss, **out)
            else:
                ops.gather_stats(x, w, keys, plan.loss, **out, **st_)
                if reduce_loss:
                    ops.reduce_partials(stats[terms_off:terms_off + n_terms], stats[loss_off:loss_off + 1])

        esum = max_bits = None
        if fixed:
            # order-independent statistics: every rank quantises with the same scale (global max norm), the int64
            # sums add exactly in any order
            xx = plan.xx if plan.xx is not None else ops.row_sqnorms(x, None)[0]
            max_bits = ops.max_norm(xx)
            if plan.group is not None:
                dist.all_reduce(max_bits, op=dist.ReduceOp.MAX, group=plan.group)
            esum = ops.embed_sum_fixed(x, idx, max_bits, k)      # idx doubles as keys (index = low word)
            if plan.group is not None:
                dist.all_reduce(esum, op=dist.ReduceOp.SUM, group=plan.group)
        if work is not None:
            work.wait()                      # the current stream waits for the collective's stream; no host wait
        elif plan.group is not None:
            dist.all_reduce(stats[:loss_off + 1], op=dist.ReduceOp.SUM, group=plan.group)

        # q / loss above used the pre-update codebook (vqvae/vqvae.py:128 precedes :131-144)
        if fixed:
            scalars = torch.empty(4, dtype=torch.float32, device=dev)
            loss_in = stats[loss_off:loss_off + 1] if reduce_loss else stats[terms_off:terms_off + n_terms]
            ops.ema_cluster(self._ema_cluster_size, stats[counts_off:counts_off + k], loss_in, k, n_total, denom,
                            plan.decay, 1 - plan.decay, plan.eps, k * plan.eps, plan.cc, not plan.ema, 1, scalars)
            ops.ema_embed_fixed(self._ema_w.data, self._embedding.weight.data, esum, max_bits,
                                self._ema_cluster_size, plan.decay, 1 - plan.decay, ee_out=ee,
                                code_image=code_image_out)
        else:
            scalars = ops.fused_finalize(cs, ema_w, w if update else None, stats, n, k, d, with_es, reduce_loss,
                                         n_total, denom, update=update, ee_out=ee if update else None,
                                         code_image=code_image_out, **hyper)
        self._image_done(images)
        self._last_scalars = scalars
        return {"loss": _scalar_of(scalars, 0), "ppl": _scalar_of(scalars, 1), "ste": ste, "idx": idx, "diff": diff,
                "q": q, "dims": dims}

    def forward(self, inputs, srl=None, _xx=None, _ximg=None):
        """``_xx`` (not part of the reference interface): |x_i|^2 [N] if the producer of ``inputs`` already has them
        (t5vqvae_amd.encoder_tail); they must be the row norms of exactly this tensor.  ``_ximg``: its fp16 image block
        (``ops.token_image``) from the same producer, for the filtered search (needs ``_xx``)."""
        self._check_loss_name()
        flat, ranges = self._flatten(inputs, srl)
        plan = _Plan()
        plan.xx = _xx
        plan.ximg = _ximg if _xx is not None else None
        plan.ema, plan.training = self._is_ema, self.training
        plan.cc = float(self._commitment_cost)
        plan.decay = float(getattr(self, "_decay", 0.0))
        plan.eps = float(getattr(self, "_epsilon", 0.0))
        plan.loss, plan.batch = self.loss, int(inputs.shape[0])
        plan.group, plan.world_tokens, plan.module = self.process_group, int(self.world_tokens), self
        plan.grad_scale = 1.0
        if plan.group is not None:
            if self.gradient_reduction not in ("mean", "sum"):
                raise ValueError(f"gradient_reduction must be 'mean' or 'sum', got {self.gradient_reduction!r}")
            if self.gradient_reduction == "mean":
                import torch.distributed as dist
                plan.grad_scale = float(dist.get_world_size(plan.group))
        weight = self._embedding.weight if not self._is_ema else self._embedding.weight.detach()
        if torch.is_grad_enabled() and (flat.requires_grad or weight.requires_grad):
            loss, ste, ppl, idx = _VQFunction.apply(flat, weight, ranges, plan)
        else:       # nothing to differentiate (evaluation, no_grad): no autograd node, ~10 us less host time per call
            r = self._launch(flat, weight.detach(), ranges, plan, want_diff=False, want_q=False)
            loss, ste, ppl, idx = r["loss"], r["ste"], r["ppl"], r["idx"]
        self.last_indices = idx
        quantized = ste.view(inputs.shape)
        if self.encodings == "dense":
            enc = ops.one_hot(idx, self._num_embeddings)     # vqvae/vqvae.py:124-125
        else:
            enc = idx.unsqueeze(1)
        return loss, quantized, ppl, enc

    # --- evaluation helpers (vqvae/vqvae.py:167-238) -------------------------------------------------
    @torch.no_grad()
    def get_latent(self, inputs):
        flat = inputs.view(-1, self._embedding_dim).contiguous()
        return ops.distances(flat, self._embedding.weight.detach())

    @torch.no_grad()
    def _keys(self, inputs, srl):
        flat, ranges = self._flatten(inputs, srl)
        w = self._embedding.weight.detach()
        return flat, w, self._absolute(self._nearest_keys(flat, w, ranges), ranges)

    @torch.no_grad()
    def get_indices(self, inputs, srl=None):
        _, _, keys = self._keys(inputs, srl)
        return ops.unpack_keys(keys)[0].unsqueeze(1)

    @torch.no_grad()
    def get_hidden(self, inputs, srl=None):
        flat, w, keys = self._keys(inputs, srl)
        q = torch.empty_like(flat)
        ops.gather_stats(flat, w, keys, "mse", q=q)
        return q.view(inputs.shape)

    # --- evaluation-side searches (callers in evaluate_vqvae.py; no counterpart method in vqvae/vqvae.py) ----
    @torch.no_grad()
    def topk_codes(self, inputs, top_k: int):
        """(distances [N, top_k], indices [N, top_k]) of the top_k nearest codes per token, nearest first: the
        ``torch.sort(get_latent(x)[dim])[:top_k]`` of the traversal (evaluate_vqvae.py:129-135) for every row at
        once, without the N x K matrix."""
        flat = inputs.view(-1, self._embedding_dim).contiguous()
        w = self._embedding.weight.detach()
        xx, ee = ops.row_sqnorms(flat, w)
        keys = ops.dist_topk(flat, w, xx, ee, int(top_k))
        idx, dmin = ops.unpack_keys(keys.view(-1), want_idx=True, want_dmin=True)
        n = flat.shape[0]
        return dmin.view(top_k, n).t().contiguous(), idx.view(top_k, n).t().contiguous()

    @torch.no_grad()
    def blended_indices(self, source_code, target_code, source_t: float, target_t: float):
        """Indices [N, 1] of ``argmin(source_t * d(source_code, W) + target_t * d(target_code, W))``: one step of
        the latent-space interpolation (evaluate_vqvae.py:186-200).  The two [N, K] distance matrices are formed
        (as the reference does; N is one sentence there) and blended in the reference's rounding order."""
        a = source_code.reshape(-1, self._embedding_dim).contiguous()
        b = target_code.reshape(-1, self._embedding_dim).contiguous()
        if a.shape != b.shape:
            raise RuntimeError("source_code and target_code must have the same number of rows")
        w = self._embedding.weight.detach()
        aa, ee = ops.row_sqnorms(a, w)
        bb, _ = ops.row_sqnorms(b, None)
        keys = ops.blend_argmin(ops.distances(a, w, aa, ee), ops.distances(b, w, bb, ee), float(source_t),
                                float(target_t))
        return ops.unpack_keys(keys)[0].unsqueeze(1)


class VectorQuantizer(_QuantizerBase):
    """Codebook learned by gradient (vqvae/vqvae.py:6-76).  ``**kwargs`` are accepted and ignored, as in the
    reference (train_t5.py:322 passes ``decay=`` and the role dict here)."""

    def __init__(self, num_embeddings, embedding_dim, commitment_cost, disentanglement=False, loss="mse", **kwargs):
        super().__init__()
        self._embedding_dim = embedding_dim
        self._num_embeddings = num_embeddings
        self._embedding = nn.Embedding(self._num_embeddings, self._embedding_dim)
        self._embedding.weight.data.uniform_(-1 / self._num_embeddings, 1 / self._num_embeddings)  # :14
        self._commitment_cost = commitment_cost
        self.disentanglement = disentanglement
        self.loss = loss


class VectorQuantizerEMA(_QuantizerBase):
    """Codebook learned by exponential moving averages (vqvae/vqvae.py:79-165)."""
    _is_ema = True

    def __init__(self, num_embeddings, embedding_dim, commitment_cost, decay, epsilon=1e-5, disentanglement=False,
                 loss="mse", **kwargs):
        super().__init__()
        self._embedding_dim = embedding_dim
        self._num_embeddings = num_embeddings
        self._embedding = nn.Embedding(self._num_embeddings, self._embedding_dim)
        self._embedding.weight.data.normal_()                                          # :87
        self._commitment_cost = commitment_cost
        self.register_buffer("_ema_cluster_size", torch.zeros(num_embeddings))         # :90
        self._ema_w = nn.Parameter(torch.Tensor(num_embeddings, self._embedding_dim))  # :91
        self._ema_w.data.normal_()                                                     # :92 (independent draw)
        self._decay = decay
        self._epsilon = epsilon
        self.disentanglement = disentanglement
        self.loss = loss
