"""Multi-GPU forms of the VQ bottleneck: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests of the host logic).

Token sharding (BASELINE config 4) -- ``shard_tokens(module)``.
    Every rank holds N/G tokens and a full replica of the codebook state.  Distances, argmin, gather and the
    loss terms are per token, so the only exchange is ONE all-reduce(sum) of the fused statistics buffer
    [embed_sum K*D | counts K | loss_sum 1] that ``vqvae._QuantizerBase._launch`` lays out contiguously;
    every rank then runs the identical finalize kernels on identical reduced values, so replicas stay
    bit-identical without a broadcast.  Loss and perplexity are those of the concatenated batch.

Code sharding (BASELINE config 5) -- ``CodeShardedVectorQuantizerEMA``.
    Every rank holds K/G codes (rows [rank*K/G, (rank+1)*K/G) of ``_embedding.weight``, ``_ema_w``,
    ``_ema_cluster_size``) and sees ALL tokens.  Exchanges per call:
      1. all-reduce(min) of the packed int64 (distance, global index) keys [N] -- RCCL has no MINLOC; min of
         the packed key is "lowest distance, then lowest index" = torch.argmin's tie rule across shards;
      2. all-gather of the codebook slices [K, D] so the winning rows can be gathered locally (the
         disentangled mode's slice-relative indices may point into ANY shard, vqvae/vqvae.py:116-119).  The
         gathered codebook is KEPT: it is refreshed only when a slice changed -- in evaluation never, in
         training right after the EMA update, asynchronously (``exchange = "prefetch"``: the 201 MB of C5 move
         on RCCL's stream while the next call's distance search runs, and are waited for only before that
         call's row gather).  ``exchange = "sync"`` gathers inside every call (the round-1 behaviour);
      3. all-reduce(sum) of two scalars: the Laplace normaliser sum(cluster_size) and the perplexity's
         entropy partial sum.
    EMA statistics need no reduction: tokens are replicated, so each rank accumulates counts / embed_sum for
    its own code slice only (``stats_begin / stats_count`` of t5vq_gather_stats_f32).
"""
from __future__ import annotations

import weakref

import torch
import torch.distributed as dist
from torch import nn

from . import vqvae
from .vqvae import _QuantizerBase

# The gathered codebook a code-sharded module keeps between calls: module -> (full [K, D], local data_ptr, local
# _version, pending c10d work | None).  A weak side table, like vqvae._EE_STATE / _FILTER_STATE: a c10d Work and a K x D
# tensor in the module's __dict__ would break copy.deepcopy / torch.save(module) or bloat them.
_FULL_STATE = weakref.WeakKeyDictionary()


def shard_tokens(module, group=None, world_tokens: int = 0):
    """Turn a VectorQuantizerEMA / VectorQuantizer replica into a token-sharded one (in place).

    ``group``: process group to reduce over (default: the world group).  ``world_tokens``: the global
    token count if ranks hold unequal shares (default: local count x world size)."""
    if not dist.is_initialized():
        raise RuntimeError("shard_tokens needs an initialised torch.distributed process group")
    module.process_group = group if group is not None else dist.group.WORLD
    module.world_tokens = int(world_tokens)
    return module


def broadcast_state(module, src: int = 0, group=None):
    """Make replicas identical at start-up (the reference seeds one process; here rank `src` wins)."""
    for t in module.state_dict().values():
        dist.broadcast(t, src=src, group=group)
    return module


class CodeShardedVectorQuantizerEMA(_QuantizerBase):
    """``VectorQuantizerEMA`` (vqvae/vqvae.py:79-165) with the codebook sharded by code over a process group.

    Constructor arguments are the reference's, with ``num_embeddings`` the GLOBAL codebook size; the local
    tensors hold ``num_embeddings / world`` rows.  ``forward`` takes the same (replicated) inputs on every
    rank and returns the same values on every rank as the unsharded module would."""
    _is_ema = True

    def __init__(self, num_embeddings, embedding_dim, commitment_cost, decay, epsilon=1e-5, disentanglement=False,
                 loss="mse", group=None, **kwargs):
        super().__init__()
        if not dist.is_initialized():
            raise RuntimeError("CodeShardedVectorQuantizerEMA needs an initialised process group")
        self.group = group if group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        if num_embeddings % self.world:
            raise ValueError(f"num_embeddings={num_embeddings} must be divisible by the group size {self.world}")
        self._embedding_dim = embedding_dim
        self._num_embeddings = num_embeddings                    # global K, as callers read it
        self.local_codes = num_embeddings // self.world
        self.code_begin = self.rank * self.local_codes
        self._embedding = nn.Embedding(self.local_codes, embedding_dim)
        self._embedding.weight.data.normal_()
        self._commitment_cost = commitment_cost
        self.register_buffer("_ema_cluster_size", torch.zeros(self.local_codes))
        self._ema_w = nn.Parameter(torch.Tensor(self.local_codes, embedding_dim))
        self._ema_w.data.normal_()
        self._decay = decay
        self._epsilon = epsilon
        self.disentanglement = disentanglement
        self.loss = loss
        self.exchange = "prefetch"       # "prefetch" | "sync": when the gathered codebook is refreshed (module docstring)

    # --- the kept gathered codebook: when it is dropped --------------------------------------------------
    # Keep-or-re-gather is a COLLECTIVE decision (a rank that re-gathers alone waits for the others forever), so it is
    # taken from things that happen on every rank alike: this module's own EMA update (always refreshes), a switch
    # between train() and eval(), a state-dict load, a move (.to / .cuda), and invalidate_code_norms(), which the
    # caller must run on every rank after writing through ``_embedding.weight.data``.  A tracked in-place change of the
    # local slice (copy_, an optimizer step) is seen through the Parameter's version as well; like any collective call
    # it has to be made by every rank of the group.
    def invalidate_code_norms(self):
        """Drop the cached |e|^2 AND the kept gathered codebook.  Call it on EVERY rank of the group (the next forward
        re-gathers collectively) after a write through ``_embedding.weight.data``, which no version counter sees."""
        super().invalidate_code_norms()
        _FULL_STATE.pop(self, None)

    def __deepcopy__(self, memo):
        """A copy shares the process group (a communicator is not copyable) and starts without a kept codebook."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for name, value in self.__dict__.items():
            new.__dict__[name] = value if name == "group" else copy.deepcopy(value, memo)
        return new

    def train(self, mode: bool = True):
        if mode != self.training:
            _FULL_STATE.pop(self, None)
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        _FULL_STATE.pop(self, None)
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        _FULL_STATE.pop(self, None)
        return super()._load_from_state_dict(*args, **kwargs)

    # --- state in the reference's (unsharded) layout -------------------------------------------------
    @torch.no_grad()
    def load_full_state_dict(self, full: dict):
        """Take this rank's rows out of a reference-layout state_dict (vqvae/vqvae.py:86-92)."""
        sl = slice(self.code_begin, self.code_begin + self.local_codes)
        self._embedding.weight.copy_(full["_embedding.weight"][sl])
        self._ema_w.copy_(full["_ema_w"][sl])
        self._ema_cluster_size.copy_(full["_ema_cluster_size"][sl])
        self.invalidate_code_norms()

    @torch.no_grad()
    def full_state_dict(self) -> dict:
        """All-gather the shards back into the reference's state_dict layout (every rank gets it)."""
        out = {}
        for name, t in (("_ema_w", self._ema_w.data), ("_ema_cluster_size", self._ema_cluster_size),
                        ("_embedding.weight", self._embedding.weight.data)):
            full = torch.empty((self._num_embeddings,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(full, t.contiguous(), group=self.group)
            out[name] = full
        return out

    def _full_codebook(self, w_local):
        """The whole codebook [K, D] on this rank.  Kept between calls while the local slice is unchanged (pointer and
        version of the Parameter, as for the code norms; every rank takes the same decision because every rank's
        slice changes in the same calls); a refresh started by ``_refresh_full_codebook`` is waited for here, on the
        stream (no host wait)."""
        param = self._embedding.weight
        ent = _FULL_STATE.get(self)
        if (self.exchange == "prefetch" and ent is not None and ent[1] == param.data_ptr()
                and ent[2] == param._version and ent[0].device == w_local.device):
            if ent[3] is not None:
                ent[3].wait()                    # current stream waits for RCCL's stream
                _FULL_STATE[self] = (ent[0], ent[1], ent[2], None)
            return ent[0]
        if ent is not None and ent[3] is not None:
            ent[3].wait()                        # a refresh in flight writes the buffer that is gathered into below
        full = ent[0] if (ent is not None and ent[0].device == w_local.device) else \
            torch.empty(self._num_embeddings, self._embedding_dim, dtype=w_local.dtype, device=w_local.device)
        dist.all_gather_into_tensor(full, w_local.contiguous(), group=self.group)
        _FULL_STATE[self] = (full, param.data_ptr(), param._version, None)
        return full

    def _refresh_full_codebook(self):
        """After the EMA update of the local slice: start the all-gather of the new slices into the kept buffer
        (in place: the collective is ordered after this call's row gather, which was the buffer's last reader) and
        return at once; the next ``_full_codebook`` waits for it."""
        ent = _FULL_STATE.get(self)
        if self.exchange != "prefetch" or ent is None:
            return
        work = dist.all_gather_into_tensor(ent[0], self._embedding.weight.data, group=self.group, async_op=True)
        _FULL_STATE[self] = (ent[0], ent[1], ent[2], work)

    def _global_keys(self, x, w_local, ranges):
        # a code shard of a disentangled codebook: most tokens' roles live on other shards (a hint for the filtered search)
        sparse = {"range_sparse": True} if (ranges is not None and self.world >= 4) else None
        keys = self._nearest_keys(x, w_local, ranges, code_offset=self.code_begin, images=sparse)
        dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=self.group)
        return self._absolute(keys, ranges)

    def _launch(self, x, w_local, ranges, plan, want_diff, want_q):
        ops = vqvae.ops
        n, d = x.shape
        kl, k = self.local_codes, self._num_embeddings
        dev = x.device
        update = plan.training
        mse = plan.loss == "mse"

        keys = self._global_keys(x, w_local, ranges)
        w_full = self._full_codebook(w_local)

        fixed = update and self.deterministic      # order-independent fixed-point sums (vqvae._QuantizerBase)
        counts = torch.empty(kl, dtype=torch.float32, device=dev)
        embed_sum = torch.empty(kl, d, dtype=torch.float32, device=dev) if (update and not fixed) else None
        perm = (ops.order_by_code(keys, w_full.shape[0])
                if (embed_sum is not None and n >= ops.ORDER_MIN_TOKENS) else None)
        partial = torch.empty(n if perm is not None else ops.gather_blocks(n), dtype=torch.float32, device=dev)
        ste = torch.empty_like(x)
        idx = torch.empty(n, dtype=torch.int64, device=dev)
        diff = torch.empty_like(x) if want_diff else None
        q = torch.empty_like(x) if want_q else None
        ops.gather_stats(x, w_full, keys, plan.loss, ste=ste, q=q, diff=diff, idx=idx, counts=counts,
                         embed_sum=embed_sum, loss_partial=partial, stats_begin=self.code_begin, stats_count=kl,
                         perm=perm)
        # tokens are replicated: the loss needs no reduction across ranks, and the cluster kernel adds the terms up

        scalars = torch.empty(4, dtype=torch.float32, device=dev)
        denom = float(n * d) if mse else float(plan.batch)
        cs = self._ema_cluster_size if update else None
        tail = (kl, n, denom, plan.decay, 1 - plan.decay, plan.eps, k * plan.eps, plan.cc, False)
        ops.ema_cluster(cs, counts, None, *tail, 2, scalars)                     # local partial sums (no loss yet)
        dist.all_reduce(scalars[1:3], op=dist.ReduceOp.SUM, group=self.group)    # entropy, sum(cluster_size)
        ops.ema_cluster(cs, counts, partial, *tail, 3, scalars)                  # smoothing + perplexity + loss
        if fixed:
            max_bits = ops.max_norm(ops.row_sqnorms(x, None)[0])                 # tokens are replicated: same scale
            esum = ops.embed_sum_fixed(x, idx, max_bits, k, stats_begin=self.code_begin, stats_count=kl)
            ops.ema_embed_fixed(self._ema_w.data, self._embedding.weight.data, esum, max_bits,
                                self._ema_cluster_size, plan.decay, 1 - plan.decay)
        elif update:
            ops.ema_embed(self._ema_w.data, self._embedding.weight.data, embed_sum, self._ema_cluster_size,
                          plan.decay, 1 - plan.decay)
        if update:
            self._refresh_full_codebook()
        self._last_scalars = scalars
        return {"loss": vqvae._scalar_of(scalars, 0), "ppl": vqvae._scalar_of(scalars, 1), "ste": ste, "idx": idx,
                "diff": diff, "q": q, "dims": (n, d, k, plan.batch)}

    # --- evaluation helpers over the sharded codebook ------------------------------------------------
    @torch.no_grad()
    def get_latent(self, inputs):
        """Full [N, K] distance matrix (vqvae/vqvae.py:167-178): local slab per rank, all-gathered by column."""
        ops = vqvae.ops
        flat = inputs.view(-1, self._embedding_dim).contiguous()
        w = self._embedding.weight.detach()
        xx, ee = ops.row_sqnorms(flat, w)
        local = ops.distances(flat, w, xx, ee)                                   # [N, K/G]
        slabs = torch.empty(self.world * flat.shape[0], self.local_codes, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(slabs, local, group=self.group)          # rank-major concatenation
        return slabs.view(self.world, flat.shape[0], self.local_codes).permute(1, 0, 2) \
            .reshape(flat.shape[0], self._num_embeddings)

    @torch.no_grad()
    def get_indices(self, inputs, srl=None):
        flat, ranges = self._flatten(inputs, srl)
        keys = self._global_keys(flat, self._embedding.weight.detach(), ranges)
        return vqvae.ops.unpack_keys(keys)[0].unsqueeze(1)

    @torch.no_grad()
    def get_hidden(self, inputs, srl=None):
        flat, ranges = self._flatten(inputs, srl)
        w = self._embedding.weight.detach()
        keys = self._global_keys(flat, w, ranges)
        q = torch.empty_like(flat)
        vqvae.ops.gather_stats(flat, self._full_codebook(w), keys, "mse", q=q)
        return q.view(inputs.shape)
