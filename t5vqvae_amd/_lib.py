"""Bindings of the two in-tree libraries: ``csrc/libt5vq.so`` (the C ABI of include/t5vq.h, through ctypes: the
fine-grained entry points, the C-ABI tests) and ``csrc/libt5vq_torch.so`` (the PyTorch-ROCm extension over it:
``torch.ops.t5vq.*``, what the nn.Modules call).

There is no fallback: if a library is missing or fails to load, the ops raise.  Both are built in-tree by
``__graft_entry__.build()`` / ``make -C t5vqvae_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# T5VQ_LIB: another build of the same library (variant builds for A/B timing); there is still no CPU path behind it
LIB_PATH = os.environ.get("T5VQ_LIB") or os.path.join(_HERE, "csrc", "libt5vq.so")

P = c_void_p
I64 = c_int64

# name -> (restype, argtypes); mirrors include/t5vq.h one to one
SIGNATURES = {
    "t5vq_version": (c_int, []),
    "t5vq_last_error": (c_char_p, []),
    "t5vq_set_tuning": (c_int, [c_int]),
    "t5vq_padded_codes": (I64, [I64]),
    "t5vq_gather_blocks": (I64, [I64]),
    "t5vq_row_sqnorms_f32": (c_int, [P, I64, P, P, I64, P, I64, P]),
    "t5vq_dist_argmin_f32": (c_int, [P, P, P, P, P, P, I64, I64, I64, I64, c_int, P, P]),
    "t5vq_order_by_range": (c_int, [P, I64, I64, I64, P, P, P]),
    "t5vq_unpack_keys": (c_int, [P, I64, P, P, P]),
    "t5vq_distances_f32": (c_int, [P, P, P, P, I64, I64, I64, P, P]),
    "t5vq_gather_stats_f32": (c_int, [P, P, P, I64, I64, I64, c_int, P, P, P, P, I64, I64, P, P, P, P, P]),
    "t5vq_order_by_code": (c_int, [P, I64, I64, P, P, P]),
    "t5vq_reduce_partials_f32": (c_int, [P, I64, P, P]),
    "t5vq_ema_cluster_f32": (c_int, [P, P, P, I64, I64, I64, c_float, c_float, c_float, c_float, c_float, c_float,
                                     c_int, c_int, P, P]),
    "t5vq_ema_embed_f32": (c_int, [P, P, P, P, I64, I64, c_float, c_float, P, P]),
    "t5vq_max_norm_f32": (c_int, [P, I64, P, P]),
    "t5vq_embed_sum_fixed_f32": (c_int, [P, P, P, I64, I64, I64, I64, I64, P, P]),
    "t5vq_ema_embed_fixed_f32": (c_int, [P, P, P, P, P, I64, I64, c_float, c_float, P, P]),
    "t5vq_ste_backward_f32": (c_int, [P, P, P, c_float, I64, P, P]),
    "t5vq_scatter_rows_f32": (c_int, [P, P, P, c_float, I64, I64, I64, P, P]),
    "t5vq_cosine_backward_f32": (c_int, [P, P, P, P, c_float, c_float, I64, I64, P, P, P]),
    "t5vq_one_hot_f32": (c_int, [P, I64, I64, P, P]),
    "t5vq_filter_workspace_bytes": (I64, [I64, I64, I64]),
    "t5vq_dist_argmin_filtered_f32": (c_int, [P, P, P, P, c_int, P, P, I64, I64, I64, I64, c_int, P, I64, P, P, P]),
    "t5vq_dist_topk_f32": (c_int, [P, P, P, P, I64, I64, I64, I64, P, P]),
    "t5vq_blend_argmin_f32": (c_int, [P, P, c_float, c_float, I64, I64, P, P]),
    "t5vq_rmsnorm_sqnorms_f32": (c_int, [P, P, P, c_float, c_float, I64, I64, P, P, P, P]),
    "t5vq_rmsnorm_sqnorms_img_f32": (c_int, [P, P, P, c_float, c_float, I64, I64, P, P, P, P, P]),
    "t5vq_code_image_bytes": (I64, [I64, I64]),
    "t5vq_token_image_bytes": (I64, [I64, I64]),
    "t5vq_dist_argmin_filtered_img_f32": (c_int, [P, P, P, P, c_int, P, P, I64, I64, I64, I64, c_int, P, I64, P, P, P,
                                                  c_int, P, c_int, P]),
    "t5vq_ema_embed_img_f32": (c_int, [P, P, P, P, I64, I64, c_float, c_float, P, P, P]),
    "t5vq_ema_embed_fixed_img_f32": (c_int, [P, P, P, P, P, I64, I64, c_float, c_float, P, P, P]),
    "t5vq_rmsnorm_backward_blocks": (I64, [I64]),
    "t5vq_rmsnorm_backward_f32": (c_int, [P, P, P, P, c_float, P, I64, I64, P, P, P, P]),
    "t5vq_forward_workspace_bytes": (I64, [I64, I64]),
    "t5vq_forward_loss_terms": (I64, [I64, c_int]),
    "t5vq_vq_forward_f32": (c_int, [P, P, P, P, P, I64, I64, I64, c_int, P, I64, P, P, P, P, P, P, P, P, P]),
    "t5vq_vq_forward_noalloc_f32": (c_int, [P, P, P, P, P, I64, I64, I64, c_int, P, P, P, P, P, P, P, P, P]),
    "t5vq_small_state_bytes": (I64, [I64]),
    "t5vq_vq_forward_small_supported": (c_int, [I64, I64, I64]),
    "t5vq_vq_forward_small_f32": (c_int, [P, P, P, P, I64, I64, I64, P, I64, P, P, P, P, P, P, P, P, P, c_float, c_float,
                                          c_int, P]),
    "t5vq_forward_filtered_workspace_bytes": (I64, [I64, I64, I64]),
    "t5vq_vq_forward_filtered_f32": (c_int, [P, P, P, P, P, I64, I64, I64, c_int, P, I64, P, P, P, P, P, P, P, P, P,
                                             P]),
    "t5vq_vq_ema_finalize_f32": (c_int, [P, P, P, P, P, P, I64, I64, I64, I64, c_float, c_float, c_float, c_float,
                                         c_float, c_float, c_int, c_int, P, P, P]),
    "t5vq_vq_forward_filtered_img_f32": (c_int, [P, P, P, P, P, I64, I64, I64, c_int, P, I64, P, P, P, P, P, P, P, P, P,
                                                 P, c_int, P, P]),
    "t5vq_vq_ema_finalize_img_f32": (c_int, [P, P, P, P, P, P, I64, I64, I64, I64, c_float, c_float, c_float, c_float,
                                             c_float, c_float, c_int, c_int, P, P, P, P]),
}

TORCH_LIB_PATH = os.environ.get("T5VQ_TORCH_LIB") or os.path.join(_HERE, "csrc", "libt5vq_torch.so")

_lib = None
_torch_ops = None
# A failed load is remembered and re-raised: a second ``dlopen`` / ``torch.ops.load_library`` of a library whose first
# load threw half-way (its static registrations partly run) is undefined -- in round 2 such a retry, made by the autograd
# worker thread of a backward-only flow, hung inside ``run_backward``.
_lib_error = None
_torch_ops_error = None


_load_lock = threading.RLock()

ABI_VERSION = 310     # t5vq_version() of the library these signatures were written against (image-emitting _img entry points)


class T5VQError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises if it is missing -- there is no CPU path."""
    global _lib, _lib_error
    if _lib is not None:
        return _lib
    if _lib_error is not None:
        raise _lib_error
    with _load_lock:
        if _lib is not None:
            return _lib
        if _lib_error is not None:
            raise _lib_error
        try:
            _lib = _load_c_abi()
        except BaseException as e:      # sticky: never dlopen twice
            _lib_error = e
            raise
    return _lib


def _load_c_abi() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise T5VQError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C t5vqvae_amd/csrc`.  t5vqvae_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # a library older than these signatures would be called with shifted arguments: refuse it instead
    if int(lib.t5vq_version()) < ABI_VERSION:
        raise T5VQError(f"{LIB_PATH} is version {int(lib.t5vq_version())}, this binding needs >= {ABI_VERSION}: "
                        "rebuild it (`make -C t5vqvae_amd/csrc all`)")
    return lib


def load_torch_ops():
    """``torch.ops.t5vq`` (once): the TORCH_LIBRARY extension csrc/t5vq_torch.cpp.  Raises if it is missing."""
    global _torch_ops, _torch_ops_error
    if _torch_ops is not None:
        return _torch_ops
    if _torch_ops_error is not None:
        raise _torch_ops_error
    with _load_lock:
        if _torch_ops is not None:
            return _torch_ops
        if _torch_ops_error is not None:
            raise _torch_ops_error
        try:
            import torch
            load()                                   # libt5vq.so first: the extension links against it
            if not os.path.exists(TORCH_LIB_PATH):
                raise T5VQError(
                    f"{TORCH_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C t5vqvae_amd/csrc`.  t5vqvae_amd has no CPU fallback.")
            torch.ops.load_library(TORCH_LIB_PATH)
            _torch_ops = torch.ops.t5vq
        except BaseException as e:      # sticky: torch.ops.load_library is never retried on a half-registered library
            _torch_ops_error = e
            raise
    return _torch_ops


def load_all():
    """Load both libraries NOW, on the calling thread (the modules' constructors call this): no ``dlopen`` can then
    happen later on another thread -- the first caller of a backward-only op is the autograd worker thread."""
    load()
    return load_torch_ops()


def check(status: int, what: str):
    if status != 0:
        msg = load().t5vq_last_error().decode("utf-8", "replace")
        raise T5VQError(f"{what} failed (code {status}): {msg}")
