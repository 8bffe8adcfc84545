"""Tensor-level wrappers of the kernels: argument checks, output/workspace allocation (torch's caching allocator:
plumbing), launch on torch's current HIP stream.  No arithmetic happens here; every function enqueues kernels and
returns without synchronising.

Two routes to the same C ABI (include/t5vq.h):
* ``fused_step / fused_forward / fused_finalize`` and the backward / helper ops go through the PyTorch-ROCm extension
  ``torch.ops.t5vq`` (csrc/t5vq_torch.cpp) -- what the nn.Modules call: one op per module call;
* the fine-grained functions bind the C ABI directly with ctypes (code-sharded codebooks, role-bucketed search, the
  C-ABI tests).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib

LOSS_KINDS = {"mse": 0, "cosine": 1}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need(t: torch.Tensor, name: str, dtype=torch.float32, ndim: Optional[int] = None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a ROCm device (got {t.device}); t5vqvae_amd has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name} must have {ndim} dims, got shape {tuple(t.shape)}")
    return t


def require_device(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a ROCm device (got {t.device}); t5vqvae_amd has no CPU path")


def padded_codes(k: int) -> int:
    return int(_lib.load().t5vq_padded_codes(k))


ORDER_MIN_TOKENS = 32768    # tokens from which ordering them by code pays for itself (GS_SMALL_N in the kernels)


def gather_blocks(n: int) -> int:
    return int(_lib.load().t5vq_gather_blocks(n))


def row_sqnorms(x: Optional[torch.Tensor], w: Optional[torch.Tensor]):
    """(xx [N], ee [padded(K)]) -- either input may be None."""
    lib = _lib.load()
    ref = x if x is not None else w
    d = ref.shape[1]
    xx = ee = None
    n = k = 0
    if x is not None:
        _need(x, "x", ndim=2)
        n = x.shape[0]
        xx = torch.empty(n, dtype=torch.float32, device=x.device)
    if w is not None:
        _need(w, "w", ndim=2)
        k = w.shape[0]
        ee = torch.empty(padded_codes(k), dtype=torch.float32, device=w.device)
    with torch.cuda.device(ref.device):
        _lib.check(lib.t5vq_row_sqnorms_f32(_p(x), n, _p(xx), _p(w), k, _p(ee), d, _stream(ref)), "row_sqnorms")
    return xx, ee


def _check_order(order, ranges, n):
    if order is None:
        return
    if ranges is None:
        raise RuntimeError("order is an ordering for the ranged search: ranges is None")
    _need(order, "order", torch.int32, 1)
    if order.shape[0] != n:
        raise RuntimeError(f"order must hold {n} token ids, got {order.shape[0]}")


def order_by_range(ranges, k: int, code_offset: int = 0):
    """int32 [N]: the token ids grouped by their (start, end) code range (t5vq_order_by_range) -- the visiting order
    that lets every token tile of a ranged search skip the code tiles outside its one or two roles."""
    lib = _lib.load()
    _need(ranges, "ranges", torch.int64, 2)
    n = ranges.shape[0]
    if ranges.shape[1] != 2:
        raise RuntimeError(f"ranges must be [N, 2], got {tuple(ranges.shape)}")
    order = torch.empty(n, dtype=torch.int32, device=ranges.device)
    work = torch.empty(k + 8, dtype=torch.int32, device=ranges.device)
    with torch.cuda.device(ranges.device):
        _lib.check(lib.t5vq_order_by_range(_p(ranges), n, k, code_offset, _p(order), _p(work), _stream(ranges)),
                   "order_by_range")
    return order


def dist_argmin(x, w, xx, ee, ranges=None, code_offset: int = 0, keys=None, merge: bool = False, order=None):
    """Packed (distance, index) keys [N] int64 of the nearest codeword per token.  ``order`` (ranged search only):
    the visiting order of the tokens (``order_by_range``); the keys do not depend on it."""
    lib = _lib.load()
    _need(x, "x", ndim=2), _need(w, "w", ndim=2), _need(xx, "xx", ndim=1), _need(ee, "ee", ndim=1)
    n, d = x.shape
    k = w.shape[0]
    if w.shape[1] != d or xx.shape[0] != n or ee.shape[0] != padded_codes(k):
        raise RuntimeError("dist_argmin: inconsistent shapes")
    if ranges is not None:
        _need(ranges, "ranges", torch.int64, 2)
        if tuple(ranges.shape) != (n, 2):
            raise RuntimeError(f"ranges must be [{n}, 2], got {tuple(ranges.shape)}")
    _check_order(order, ranges, n)
    if keys is None:
        if merge:
            raise RuntimeError("merge=True needs existing keys")
        keys = torch.empty(n, dtype=torch.int64, device=x.device)
    else:
        _need(keys, "keys", torch.int64, 1)
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_dist_argmin_f32(_p(x), _p(w), _p(xx), _p(ee), _p(ranges), _p(order), n, k, d, code_offset,
                                            1 if merge else 0, _p(keys), _stream(x)), "dist_argmin")
    return keys


def dist_argmin_filtered(x, w, xx=None, ee=None, keys=None, want_overflow: bool = False, exact_keys: bool = True,
                         ranges=None, code_offset: int = 0, return_norms: bool = False, order=None,
                         code_image=None, code_image_valid: bool = False, token_image=None, range_sparse: bool = False):
    """Same keys as ``dist_argmin(x, w, xx, ee)``, found by filtering in fp16 and deciding in fp32
    (t5vq_dist_argmin_filtered_f32).  ``xx`` / ``ee`` None (either or both): that norm is computed by the call's own
    first pass (no separate ``row_sqnorms``); a given norm is only read.  ``exact_keys=False``: only the index word is
    guaranteed (cheaper).  ``order`` (ranged search only): the visiting order of the tokens (``order_by_range``).
    Returns keys, or (keys, n_overflow int32[1]) with ``want_overflow``; ``return_norms`` appends (xx, ee) -- the
    norms the search used, whether given or computed.  ``code_image`` (a ``code_image()`` block): the codebook's fp16
    image is read from it when ``code_image_valid`` (``ee`` must be given), else this call writes it there;
    ``token_image``: a valid image of x from its producer (``xx`` must be given; ignored with ``order``).
    ``range_sparse``: a hint that most tokens' ranges miss this codebook slice (a code shard): more, smaller code splits."""
    lib = _lib.load()
    _need(x, "x", ndim=2), _need(w, "w", ndim=2)
    n, d = x.shape
    k = w.shape[0]
    if w.shape[1] != d:
        raise RuntimeError("dist_argmin_filtered: x and w disagree on D")
    if d % 64:
        raise RuntimeError("the filtered search needs D % 64 == 0")
    if ranges is not None:
        _need(ranges, "ranges", torch.int64, 2)
        if tuple(ranges.shape) != (n, 2):
            raise RuntimeError(f"ranges must be [{n}, 2], got {tuple(ranges.shape)}")
    _check_order(order, ranges, n)
    # compute_norms of the C ABI: 0 both given, 1 both computed (outputs), 2 xx given / ee computed, 3 the reverse.
    # A norm that is not given is allocated here and written by the search's first pass; a given one is an input and
    # is never overwritten.
    compute = (1 if ee is None else 3) if xx is None else (2 if ee is None else 0)
    if xx is None:
        xx = torch.empty(n, dtype=torch.float32, device=x.device)
    if ee is None:
        ee = torch.empty(padded_codes(k), dtype=torch.float32, device=x.device)
    _need(xx, "xx", ndim=1), _need(ee, "ee", ndim=1)
    if xx.shape[0] != n or ee.shape[0] != padded_codes(k):
        raise RuntimeError("dist_argmin_filtered: inconsistent shapes")
    if keys is None:
        keys = torch.empty(n, dtype=torch.int64, device=x.device)
    else:
        _need(keys, "keys", torch.int64, 1)
    nbytes = int(lib.t5vq_filter_workspace_bytes(n, k, d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    ovf = torch.empty(1, dtype=torch.int32, device=x.device) if want_overflow else None
    _check_images(lib, code_image, token_image, n, k, d)
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_dist_argmin_filtered_img_f32(_p(x), _p(w), _p(xx), _p(ee), compute, _p(ranges), _p(order),
                                                         n, k, d, code_offset, 1 if exact_keys else 0, _p(ws), nbytes,
                                                         _p(keys), _p(ovf), _p(code_image),
                                                         1 if code_image_valid else 0, _p(token_image),
                                                         1 if range_sparse else 0, _stream(x)),
                   "dist_argmin_filtered")
    out = (keys, ovf) if want_overflow else (keys,)
    if return_norms:
        out = out + (xx, ee)
    return out if len(out) > 1 else keys


def code_image(k: int, d: int, device) -> torch.Tensor:
    """An (uninitialised) block for the fp16 image of a [k, d] codebook (t5vq_code_image_bytes): filled by a filtered
    search called with ``code_image_valid=False`` or by the codebook update, read by later searches."""
    return torch.empty(int(_lib.load().t5vq_code_image_bytes(k, d)), dtype=torch.uint8, device=device)


def token_image(n: int, d: int, device) -> torch.Tensor:
    """An (uninitialised) block for the fp16 image of [n, d] inputs (t5vq_token_image_bytes), filled by
    ``rmsnorm_sqnorms(..., token_image=...)``."""
    return torch.empty(int(_lib.load().t5vq_token_image_bytes(n, d)), dtype=torch.uint8, device=device)


def _check_images(lib, code_img, token_img, n: int, k: int, d: int):
    if code_img is not None and _need(code_img, "code_image", torch.uint8).numel() < int(lib.t5vq_code_image_bytes(k, d)):
        raise RuntimeError("code_image smaller than t5vq_code_image_bytes(K, D)")
    if token_img is not None and _need(token_img, "token_image", torch.uint8).numel() < int(lib.t5vq_token_image_bytes(n, d)):
        raise RuntimeError("token_image smaller than t5vq_token_image_bytes(N, D)")


def unpack_keys(keys, want_idx=True, want_dmin=False):
    lib = _lib.load()
    _need(keys, "keys", torch.int64, 1)
    n = keys.shape[0]
    idx = torch.empty(n, dtype=torch.int64, device=keys.device) if want_idx else None
    dmin = torch.empty(n, dtype=torch.float32, device=keys.device) if want_dmin else None
    with torch.cuda.device(keys.device):
        _lib.check(lib.t5vq_unpack_keys(_p(keys), n, _p(idx), _p(dmin), _stream(keys)), "unpack_keys")
    return idx, dmin


def distances(x, w, xx=None, ee=None):
    """Full [N, K] distance matrix (get_latent).  With norms given: the C ABI call; without: one op that computes them."""
    if xx is None or ee is None:
        return _lib.load_torch_ops().distances(x, w)
    lib = _lib.load()
    _need(x, "x", ndim=2), _need(w, "w", ndim=2), _need(xx, "xx", ndim=1), _need(ee, "ee", ndim=1)
    n, d = x.shape
    k = w.shape[0]
    if w.shape[1] != d or xx.shape[0] != n or ee.shape[0] != padded_codes(k):
        raise RuntimeError("distances: inconsistent shapes")
    out = torch.empty(n, k, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_distances_f32(_p(x), _p(w), _p(xx), _p(ee), n, k, d, _p(out), _stream(x)), "distances")
    return out


def dist_topk(x, w, xx, ee, topk: int):
    """Packed keys [topk, N] of the topk nearest codes per token, nearest first (t5vq_dist_topk_f32)."""
    lib = _lib.load()
    _need(x, "x", ndim=2), _need(w, "w", ndim=2), _need(xx, "xx", ndim=1), _need(ee, "ee", ndim=1)
    n, d = x.shape
    k = w.shape[0]
    if w.shape[1] != d or xx.shape[0] != n or ee.shape[0] != padded_codes(k):
        raise RuntimeError("dist_topk: inconsistent shapes")
    if not 1 <= topk <= k:
        raise RuntimeError(f"topk must be in [1, {k}], got {topk}")
    keys = torch.empty(topk, n, dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_dist_topk_f32(_p(x), _p(w), _p(xx), _p(ee), n, k, d, topk, _p(keys), _stream(x)),
                   "dist_topk")
    return keys


def blend_argmin(dist_left, dist_right, weight_left: float, weight_right: float):
    """Packed keys [N] of argmin_k (weight_left * dist_left + weight_right * dist_right) (t5vq_blend_argmin_f32)."""
    lib = _lib.load()
    _need(dist_left, "dist_left", ndim=2), _need(dist_right, "dist_right", ndim=2)
    if dist_left.shape != dist_right.shape:
        raise RuntimeError("blend_argmin: the two distance matrices differ in shape")
    n, k = dist_left.shape
    keys = torch.empty(n, dtype=torch.int64, device=dist_left.device)
    with torch.cuda.device(dist_left.device):
        _lib.check(lib.t5vq_blend_argmin_f32(_p(dist_left), _p(dist_right), weight_left, weight_right, n, k,
                                             _p(keys), _stream(dist_left)), "blend_argmin")
    return keys


def gather_stats(x, w, keys, loss: str = "mse", *, ste=None, q=None, diff=None, idx=None, counts=None,
                 embed_sum=None, loss_partial=None, stats_begin: int = 0, stats_count: Optional[int] = None,
                 perm=None):
    """Every output is optional and caller-allocated (None = not produced).  ``perm`` (order_by_code): walk the
    tokens grouped by code; ``loss_partial`` then holds one term per token (length n) instead of one per workgroup."""
    lib = _lib.load()
    _need(x, "x", ndim=2), _need(w, "w", ndim=2), _need(keys, "keys", torch.int64, 1)
    n, d = x.shape
    k = w.shape[0]
    if w.shape[1] != d or keys.numel() != n:
        raise RuntimeError("gather_stats: x, w and keys disagree on N or D")
    if stats_count is None:
        stats_count = k - stats_begin
    if stats_begin < 0 or stats_count < 0 or stats_begin + stats_count > k:
        raise RuntimeError(f"gather_stats: statistics slice [{stats_begin}, {stats_begin + stats_count}) outside [0, {k})")
    for name, t, shape in (("ste", ste, (n, d)), ("q", q, (n, d)), ("diff", diff, (n, d)),
                           ("counts", counts, (stats_count,)), ("embed_sum", embed_sum, (stats_count, d)),
                           ("loss_partial", loss_partial, (n if perm is not None else gather_blocks(n),))):
        if t is not None:
            _need(t, name)
            if t.numel() != int(torch.Size(shape).numel()):
                raise RuntimeError(f"{name}: expected {shape}, got {tuple(t.shape)}")
    if idx is not None:
        _need(idx, "idx", torch.int64)
        if idx.numel() != n:
            raise RuntimeError(f"idx: expected {n} elements")
    if perm is not None:
        _need(perm, "perm", torch.int32, 1)
        if perm.numel() != n:
            raise RuntimeError(f"perm: expected {n} token ids, got {perm.numel()}")
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_gather_stats_f32(_p(x), _p(w), _p(keys), n, k, d, LOSS_KINDS[loss], _p(ste), _p(q),
                                             _p(diff), _p(idx), stats_begin, stats_count, _p(counts),
                                             _p(embed_sum), _p(loss_partial), _p(perm), _stream(x)), "gather_stats")


def order_by_code(keys, k: int):
    """int32 [N]: the token ids grouped by the code in their key (t5vq_order_by_code), for gather_stats(perm=)."""
    lib = _lib.load()
    _need(keys, "keys", torch.int64, 1)
    n = keys.numel()
    perm = torch.empty(n, dtype=torch.int32, device=keys.device)
    work = torch.empty(k + 2, dtype=torch.int32, device=keys.device)
    with torch.cuda.device(keys.device):
        _lib.check(lib.t5vq_order_by_code(_p(keys), n, k, _p(perm), _p(work), _stream(keys)), "order_by_code")
    return perm


def forward_workspace_bytes(n: int, k: int) -> int:
    return int(_lib.load().t5vq_forward_workspace_bytes(n, k))


def forward_loss_terms(n: int, with_embed_sum: bool) -> int:
    return int(_lib.load().t5vq_forward_loss_terms(n, 1 if with_embed_sum else 0))


def _check_forward_args(what, x, w, ranges, xx, ee, ste, q, diff, idx, counts, embed_sum, loss_terms, loss_sum):
    _need(x, "x", ndim=2), _need(w, "w", ndim=2)
    n, d = x.shape
    k = w.shape[0]
    if w.shape[1] != d:
        raise RuntimeError(f"{what}: x and w disagree on D")
    if ranges is not None:
        _need(ranges, "ranges", torch.int64, 2)
        if tuple(ranges.shape) != (n, 2):
            raise RuntimeError(f"ranges must be [{n}, 2], got {tuple(ranges.shape)}")
    for name, t, numel in (("xx", xx, n), ("ee", ee, padded_codes(k)), ("ste", ste, n * d), ("q", q, n * d),
                           ("diff", diff, n * d), ("counts", counts, k), ("embed_sum", embed_sum, k * d),
                           ("loss_terms", loss_terms, forward_loss_terms(n, embed_sum is not None)),
                           ("loss_sum", loss_sum, 1)):
        if t is not None:
            _need(t, name)
            if t.numel() != numel:
                raise RuntimeError(f"{name}: expected {numel} elements, got shape {tuple(t.shape)}")
    if idx is not None:
        _need(idx, "idx", torch.int64)
        if idx.numel() != n:
            raise RuntimeError(f"idx: expected {n} elements")
    return n, k, d


def vq_forward(x, w, ranges, loss: str, *, xx=None, ee=None, ste=None, q=None, diff=None, idx=None, counts=None,
               embed_sum=None, loss_terms=None, loss_sum=None):
    """The whole forward behind one host call (t5vq_vq_forward_f32, ctypes route): row norms, distance + argmin,
    gather / straight-through / loss terms / EMA statistics (, loss_sum).  Outputs are optional and caller-allocated."""
    lib = _lib.load()
    n, k, d = _check_forward_args("vq_forward", x, w, ranges, xx, ee, ste, q, diff, idx, counts, embed_sum, loss_terms,
                                  loss_sum)
    nbytes = forward_workspace_bytes(n, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_vq_forward_f32(_p(x), _p(w), _p(ranges), _p(xx), _p(ee), n, k, d, LOSS_KINDS[loss], _p(ws),
                                           nbytes, _p(ste), _p(q), _p(diff), _p(idx), _p(counts), _p(embed_sum),
                                           _p(loss_terms), _p(loss_sum), _stream(x)), "vq_forward")


def vq_forward_filtered(x, w, ranges, loss: str, *, xx=None, ee=None, ste=None, q=None, diff=None, idx=None,
                        counts=None, embed_sum=None, loss_terms=None, loss_sum=None):
    """vq_forward with the filtered search (t5vq_vq_forward_filtered_f32): same outputs, one host call.  Returns the
    device int32[1] count of tokens the filter left to the scan kernel."""
    lib = _lib.load()
    n, k, d = _check_forward_args("vq_forward_filtered", x, w, ranges, xx, ee, ste, q, diff, idx, counts, embed_sum,
                                  loss_terms, loss_sum)
    if d % 64 or d > 4096:
        raise RuntimeError("the filtered search needs D % 64 == 0, D <= 4096")
    nbytes = int(lib.t5vq_forward_filtered_workspace_bytes(n, k, d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    ovf = torch.empty(1, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_vq_forward_filtered_f32(_p(x), _p(w), _p(ranges), _p(xx), _p(ee), n, k, d, LOSS_KINDS[loss],
                                                    _p(ws), nbytes, _p(ste), _p(q), _p(diff), _p(idx), _p(counts),
                                                    _p(embed_sum), _p(loss_terms), _p(loss_sum), _p(ovf), _stream(x)),
                   "vq_forward_filtered")
    return ovf


def small_state(k: int, device) -> torch.Tensor:
    """A zeroed state block for ``vq_forward_small`` (tickets + the call's counts); every call leaves it zeroed again."""
    return torch.zeros(int(_lib.load().t5vq_small_state_bytes(k)), dtype=torch.uint8, device=device)


def small_supported(n: int, k: int, d: int) -> bool:
    return bool(_lib.load().t5vq_vq_forward_small_supported(n, k, d))


def vq_forward_small(x, w, state, *, xx=None, ee=None, ste=None, q=None, diff=None, idx=None, counts=None,
                     embed_sum=None, loss_terms=None, scalars=None, loss_denom: float = 1.0,
                     commitment_cost: float = 0.0, plain_variant: bool = False):
    """The single-launch forward for small problems (t5vq_vq_forward_small_f32, ctypes route).  ``state``: a block from
    ``small_state`` that only calls ordered after one another share.  With ``scalars`` ([4]) the launch also writes
    loss / perplexity (a call without an EMA update)."""
    lib = _lib.load()
    _need(x, "x", ndim=2), _need(w, "w", ndim=2), _need(state, "state", torch.uint8)
    n, d = x.shape
    k = w.shape[0]
    if w.shape[1] != d:
        raise RuntimeError("vq_forward_small: x and w disagree on D")
    if state.numel() < int(lib.t5vq_small_state_bytes(k)):
        raise RuntimeError("state block smaller than t5vq_small_state_bytes(K)")
    for name, t, numel in (("xx", xx, n), ("ee", ee, padded_codes(k)), ("ste", ste, n * d), ("q", q, n * d),
                           ("diff", diff, n * d), ("counts", counts, k), ("embed_sum", embed_sum, k * d),
                           ("loss_terms", loss_terms, n), ("scalars", scalars, 4)):
        if t is not None:
            _need(t, name)
            if t.numel() != numel:
                raise RuntimeError(f"{name}: expected {numel} elements, got shape {tuple(t.shape)}")
    if idx is not None and (_need(idx, "idx", torch.int64).numel() != n):
        raise RuntimeError(f"idx: expected {n} elements")
    nbytes = forward_workspace_bytes(n, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_vq_forward_small_f32(_p(x), _p(w), _p(xx), _p(ee), n, k, d, _p(ws), nbytes, _p(state), _p(ste),
                                                 _p(q), _p(diff), _p(idx), _p(counts), _p(embed_sum), _p(loss_terms),
                                                 _p(scalars), loss_denom, commitment_cost, 1 if plain_variant else 0,
                                                 _stream(x)), "vq_forward_small")


def ema_finalize(cluster_size, ema_w, w, counts, embed_sum, loss_sum, n_tokens: int, loss_denom: float, decay: float,
                 one_minus_decay: float, epsilon: float, k_epsilon: float, commitment_cost: float,
                 plain_variant: bool, update: bool, scalars, ee_out=None, code_image=None):
    """EMA update (if ``update``) + loss / perplexity scalars behind one host call (t5vq_vq_ema_finalize_f32, ctypes
    route).  ``loss_sum``: one ready sum, or the forward's unreduced loss terms (all its elements are added up)."""
    lib = _lib.load()
    _need(scalars, "scalars"), _need(counts, "counts"), _need(loss_sum, "loss_sum")
    k = counts.numel()
    d = 0
    if update:
        _need(ema_w, "ema_w", ndim=2), _need(w, "w", ndim=2), _need(embed_sum, "embed_sum"), _need(cluster_size, "cs")
        d = w.shape[1]
        if tuple(ema_w.shape) != (k, d) or tuple(w.shape) != (k, d) or embed_sum.numel() != k * d \
                or cluster_size.numel() != k:
            raise RuntimeError("ema_finalize: inconsistent shapes")
    if ee_out is not None and _need(ee_out, "ee_out").numel() < k:
        raise RuntimeError("ee_out must hold at least K norms")
    if update:
        _check_images(lib, code_image, None, 0, k, d)
    with torch.cuda.device(scalars.device):
        _lib.check(lib.t5vq_vq_ema_finalize_img_f32(_p(cluster_size) if update else None, _p(ema_w) if update else None,
                                                _p(w) if update else None, _p(counts),
                                                _p(embed_sum) if update else None, _p(loss_sum), loss_sum.numel(), k, d,
                                                n_tokens, loss_denom, decay, one_minus_decay, epsilon, k_epsilon,
                                                commitment_cost, 1 if plain_variant else 0, 1 if update else 0,
                                                _p(scalars), _p(ee_out) if update else None,
                                                _p(code_image) if update else None, _stream(scalars)),
                   "ema_finalize")


# ---- the module's route: torch.ops.t5vq (csrc/t5vq_torch.cpp) ---------------------------------------------------
def stats_layout(n: int, k: int, d: int, with_embed_sum: bool):
    """(embed_sum elements, counts offset, loss_sum offset, loss-terms offset, number of loss terms) of the statistics
    buffer the fused ops return: [embed_sum K*D | counts K | loss_sum 1 | pad | loss terms].  Mirrors StatsLayout in
    csrc/t5vq_torch.cpp; the head [0, K*D + K + 1) is what a token-sharded run all-reduces."""
    kd = k * d if with_embed_sum else 0
    loss_off = kd + k
    terms_off = (loss_off + 1 + 3) // 4 * 4
    return kd, kd, loss_off, terms_off, forward_loss_terms(n, with_embed_sum)


def code_sqnorms(w):
    """|e_k|^2 in the padded layout the searches take (+inf beyond K)."""
    return _lib.load_torch_ops().code_sqnorms(w)


def fused_forward(x, w, ranges, loss: str, *, xx=None, ee=None, filtered=False, want_diff=False, want_q=False,
                  want_embed_sum=False, reduce_loss=False, code_image=None, code_image_valid=False, token_image=None):
    """(ste, idx, diff | None, q | None, stats, unsettled | None): t5vq::vq_forward.  Image blocks: filtered search only
    (see ``dist_argmin_filtered``)."""
    return _lib.load_torch_ops().vq_forward(x, w, ranges, xx, ee, LOSS_KINDS[loss], filtered, want_diff, want_q,
                                            want_embed_sum, reduce_loss, code_image, code_image_valid, token_image)


def fused_finalize(cluster_size, ema_w, w, stats, n_local: int, k: int, d: int, with_embed_sum: bool,
                   reduced_loss: bool, n_tokens: int, loss_denom: float, decay: float, one_minus_decay: float,
                   epsilon: float, k_epsilon: float, commitment_cost: float, plain_variant: bool, update: bool,
                   ee_out=None, code_image=None):
    """scalars [loss, perplexity, n, raw latent loss]: t5vq::vq_finalize.  ``code_image``: with ``update`` the fp16 image
    of the codebook just written goes there."""
    return _lib.load_torch_ops().vq_finalize(cluster_size, ema_w, w, stats, n_local, k, d, with_embed_sum, reduced_loss,
                                             n_tokens, loss_denom, decay, one_minus_decay, epsilon, k_epsilon,
                                             commitment_cost, plain_variant, update, ee_out, code_image)


def fused_step(x, w, ranges, loss: str, *, xx=None, ee=None, cluster_size=None, ema_w=None, filtered=False,
               want_diff=False, want_q=False, loss_denom: float, decay: float, one_minus_decay: float, epsilon: float,
               k_epsilon: float, commitment_cost: float, plain_variant: bool, update: bool, ee_out=None,
               small_state=None, code_image=None, code_image_valid=False, token_image=None):
    """(ste, idx, diff | None, q | None, scalars, unsettled | None): the whole module call on one GPU behind one op
    (t5vq::vq_step = vq_forward + vq_finalize); scalars = [loss, perplexity, n, raw latent loss].  ``small_state`` (a
    block from ``small_state()``): calls without an EMA update on small problems then take ONE launch."""
    return _lib.load_torch_ops().vq_step(x, w, ranges, xx, ee, cluster_size, ema_w, LOSS_KINDS[loss], filtered,
                                         want_diff, want_q, loss_denom, decay, one_minus_decay, epsilon, k_epsilon,
                                         commitment_cost, plain_variant, update, ee_out, small_state, code_image,
                                         code_image_valid, token_image)


def rmsnorm_sqnorms(hidden, gamma, keep, scale: float, eps: float, want_rms: bool = True, token_image=None):
    """(x, xx, rms): T5LayerNorm + dropout mask + row norms of the result in one pass (t5vq_rmsnorm_sqnorms_f32).
    ``token_image`` (a ``token_image()`` block): the same pass writes the fp16 image of x the filtered search takes."""
    lib = _lib.load()
    _need(hidden, "hidden", ndim=2), _need(gamma, "gamma", ndim=1)
    n, d = hidden.shape
    if gamma.shape[0] != d:
        raise RuntimeError(f"gamma must have {d} elements")
    if keep is not None:
        _need(keep, "keep", torch.uint8, 2)
        if tuple(keep.shape) != (n, d):
            raise RuntimeError(f"keep must be [{n}, {d}]")
    x = torch.empty_like(hidden)
    xx = torch.empty(n, dtype=torch.float32, device=hidden.device)
    rms = torch.empty(n, dtype=torch.float32, device=hidden.device) if want_rms else None
    _check_images(lib, None, token_image, n, 1, d)
    with torch.cuda.device(hidden.device):
        _lib.check(lib.t5vq_rmsnorm_sqnorms_img_f32(_p(hidden), _p(gamma), _p(keep), scale, eps, n, d, _p(x), _p(xx),
                                                    _p(rms), _p(token_image), _stream(hidden)), "rmsnorm_sqnorms")
    return x, xx, rms


def rmsnorm_backward(grad_x, hidden, gamma, keep, scale: float, rms, want_grad_gamma: bool = True):
    """(grad_hidden, grad_gamma) of rmsnorm_sqnorms (t5vq_rmsnorm_backward_f32)."""
    lib = _lib.load()
    grad_x = _need(grad_x.contiguous(), "grad_x", ndim=2)
    _need(hidden, "hidden", ndim=2), _need(gamma, "gamma", ndim=1), _need(rms, "rms", ndim=1)
    n, d = hidden.shape
    gh = torch.empty_like(hidden)
    gg = ws = None
    if want_grad_gamma:
        gg = torch.empty(d, dtype=torch.float32, device=hidden.device)
        ws = torch.empty(int(lib.t5vq_rmsnorm_backward_blocks(n)) * d, dtype=torch.float32, device=hidden.device)
    with torch.cuda.device(hidden.device):
        _lib.check(lib.t5vq_rmsnorm_backward_f32(_p(grad_x), _p(hidden), _p(gamma), _p(keep), scale, _p(rms), n, d,
                                                 _p(gh), _p(gg), _p(ws), _stream(hidden)), "rmsnorm_backward")
    return gh, gg


def reduce_partials(partials, out):
    lib = _lib.load()
    _need(partials, "partials"), _need(out, "out")
    with torch.cuda.device(partials.device):
        _lib.check(lib.t5vq_reduce_partials_f32(_p(partials), partials.numel(), _p(out), _stream(partials)),
                   "reduce_partials")


def ema_cluster(cluster_size, counts, loss_sum, k: int, n_tokens: int, loss_denom: float, decay: float,
                one_minus_decay: float, epsilon: float, k_epsilon: float, commitment_cost: float, plain_variant: bool,
                mode: int, scalars):
    """``loss_sum``: one ready sum or unreduced loss terms (every element is added up, fixed order)."""
    lib = _lib.load()
    _need(scalars, "scalars")
    with torch.cuda.device(scalars.device):
        _lib.check(lib.t5vq_ema_cluster_f32(_p(cluster_size), _p(counts), _p(loss_sum),
                                            loss_sum.numel() if loss_sum is not None else 0, k, n_tokens, loss_denom,
                                            decay, one_minus_decay, epsilon, k_epsilon, commitment_cost,
                                            1 if plain_variant else 0, mode, _p(scalars), _stream(scalars)),
                   "ema_cluster")


def ema_embed(ema_w, w, embed_sum, cluster_size, decay: float, one_minus_decay: float, ee_out=None, code_image=None):
    lib = _lib.load()
    _need(ema_w, "ema_w", ndim=2), _need(w, "w", ndim=2), _need(embed_sum, "embed_sum"), _need(cluster_size, "cs")
    k, d = ema_w.shape
    if ee_out is not None and _need(ee_out, "ee_out").numel() < k:
        raise RuntimeError("ee_out must hold at least K norms")
    _check_images(lib, code_image, None, 0, k, d)
    with torch.cuda.device(w.device):
        _lib.check(lib.t5vq_ema_embed_img_f32(_p(ema_w), _p(w), _p(embed_sum), _p(cluster_size), k, d, decay,
                                              one_minus_decay, _p(ee_out), _p(code_image), _stream(w)), "ema_embed")


def max_norm(xx):
    """int32[1]: bit image of max(xx) -- the scale source of the fixed-point statistics (t5vq_max_norm_f32)."""
    lib = _lib.load()
    _need(xx, "xx", ndim=1)
    out = torch.empty(1, dtype=torch.int32, device=xx.device)
    with torch.cuda.device(xx.device):
        _lib.check(lib.t5vq_max_norm_f32(_p(xx), xx.shape[0], _p(out), _stream(xx)), "max_norm")
    return out


def embed_sum_fixed(x, keys, max_bits, k: int, stats_begin: int = 0, stats_count: Optional[int] = None):
    """int64 [stats_count, D]: order-independent fixed-point sums of the inputs per code (t5vq_embed_sum_fixed_f32)."""
    lib = _lib.load()
    _need(x, "x", ndim=2), _need(keys, "keys", torch.int64, 1), _need(max_bits, "max_bits", torch.int32)
    n, d = x.shape
    if stats_count is None:
        stats_count = k - stats_begin
    out = torch.empty(stats_count, d, dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.t5vq_embed_sum_fixed_f32(_p(x), _p(keys), _p(max_bits), n, k, d, stats_begin, stats_count,
                                                _p(out), _stream(x)), "embed_sum_fixed")
    return out


def ema_embed_fixed(ema_w, w, embed_sum_i64, max_bits, cluster_size, decay: float, one_minus_decay: float,
                    ee_out=None, code_image=None):
    lib = _lib.load()
    _need(ema_w, "ema_w", ndim=2), _need(w, "w", ndim=2), _need(embed_sum_i64, "embed_sum", torch.int64)
    _need(max_bits, "max_bits", torch.int32), _need(cluster_size, "cs")
    k, d = ema_w.shape
    if embed_sum_i64.numel() != k * d:
        raise RuntimeError("ema_embed_fixed: embed_sum must be [K, D]")
    if ee_out is not None and _need(ee_out, "ee_out").numel() < k:
        raise RuntimeError("ee_out must hold at least K norms")
    _check_images(lib, code_image, None, 0, k, d)
    with torch.cuda.device(w.device):
        _lib.check(lib.t5vq_ema_embed_fixed_img_f32(_p(ema_w), _p(w), _p(embed_sum_i64), _p(max_bits), _p(cluster_size),
                                                    k, d, decay, one_minus_decay, _p(ee_out), _p(code_image),
                                                    _stream(w)), "ema_embed_fixed")


def ste_backward(g_out, diff, g_loss, coef: float):
    return _lib.load_torch_ops().ste_backward(g_out, diff, g_loss, coef)


def scatter_rows(rows, idx, k: int, scale_dev=None, scale_host: float = 1.0):
    return _lib.load_torch_ops().scatter_rows(rows, idx, scale_dev, scale_host, k)


def cosine_backward(x, q, g_out, g_loss, coef_x: float, coef_q: float, want_grad_q: bool):
    return _lib.load_torch_ops().cosine_backward(x, q, g_out, g_loss, coef_x, coef_q, want_grad_q)


def one_hot(idx, k: int):
    return _lib.load_torch_ops().one_hot(idx.reshape(-1), k)
