"""Deterministic, platform-independent synthetic data for the VQ bottleneck.

The golden fixtures under ``tests/golden/`` are produced in one container (where
the reference imports) and checked on another (the GPU box, where it does not
exist).  Inputs therefore come from a generator whose output bits depend on
nothing but (seed, stream, index): a 32-bit integer mixer evaluated in int64
with every intermediate product kept below 2**63 (no reliance on wrap-around),
an Irwin-Hall(8) sum of 16-bit fields (exact integer arithmetic), one IEEE
float64 subtraction and division, and one rounding to float32.  No
transcendental function, no library RNG -- so CPU and GPU, numpy-free, any torch
version, give identical float32 bits.

Shapes/semantics follow the VQ input contract of the reference:
``inputs`` is fp32 ``[B, S, D]`` (vqvae/vqvae.py:99-103), codebooks are fp32
``[K, D]`` (vqvae/vqvae.py:86-92), ``srl`` is int64 ``[B, S, 2]`` of
``[start, end)`` code ranges (t5/load_data.py:76-99).
"""
from __future__ import annotations

import math

import torch

_M32 = 0xFFFFFFFF
# 1 / (65536 * sqrt(8/12)): Irwin-Hall(8) of 16-bit uniforms has variance 8 * 2^32 / 12.
_IH8_STD = 65536.0 * math.sqrt(8.0 / 12.0)
_IH8_MEAN = 8 * 32767.5


def _mul32(x: torch.Tensor, c: int) -> torch.Tensor:
    """(x * c) mod 2**32 for 0 <= x < 2**32, without any product reaching 2**63."""
    lo = x * (c & 0xFFFF)                       # < 2**48
    hi = (x * (c >> 16)) & 0xFFFF               # product < 2**48; keep the 16 bits that survive << 16
    return (lo + (hi << 16)) & _M32


def _mix32(x: torch.Tensor) -> torch.Tensor:
    """Bijective 32-bit avalanche mixer (xor-shift-multiply), int64 carrier."""
    x = x ^ (x >> 16)
    x = _mul32(x, 0x7FEB352D)
    x = x ^ (x >> 15)
    x = _mul32(x, 0x846CA68B)
    x = x ^ (x >> 16)
    return x


def _mix32_int(x: int) -> int:
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def _keys(seed: int, stream: int) -> list[int]:
    base = _mix32_int(_mix32_int(seed * 0x9E3779B1 + 0x85EBCA6B) ^ (stream * 0xC2B2AE35 + 0x27D4EB2F))
    return [_mix32_int(base + 0x9E3779B9 * (j + 1)) for j in range(4)]


def hash_u32(n: int, seed: int, stream: int = 0, word: int = 0, device="cpu", start: int = 0) -> torch.Tensor:
    """n 32-bit hash words (int64 tensor, values in [0, 2**32)) for counters start..start+n-1."""
    if start + n > (1 << 32):
        raise ValueError("synth counters are 32-bit")
    i = torch.arange(start, start + n, dtype=torch.int64, device=device)
    return _mix32(i ^ _keys(seed, stream)[word])


def _ih8(n: int, seed: int, stream: int, device, lo: int = 0) -> torch.Tensor:
    """Irwin-Hall(8) integer sums (int64, in [0, 8*65535]) for counters lo..lo+n-1."""
    i = torch.arange(lo, lo + n, dtype=torch.int64, device=device)
    acc = torch.zeros(n, dtype=torch.int64, device=device)
    for k in _keys(seed, stream):
        h = _mix32(i ^ k)
        acc += (h & 0xFFFF) + (h >> 16)
    return acc


def normal(shape, seed: int, stream: int = 0, device="cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """Approximately N(0,1) fp32 tensor of ``shape`` (Irwin-Hall(8), |z| < 4.9), bit-reproducible."""
    n = 1
    for s in shape:
        n *= int(s)
    if n > (1 << 32):
        raise ValueError("synth counters are 32-bit")
    out = torch.empty(n, dtype=torch.float32, device=device)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        z = (_ih8(m, seed, stream, device, lo).to(torch.float64) - _IH8_MEAN) / _IH8_STD
        out[lo:lo + m] = z.to(torch.float32)
    return out.view(*shape)


def uniform_int(n: int, high: int, seed: int, stream: int = 0, device="cpu") -> torch.Tensor:
    """n integers in [0, high) (int64); high <= 2**31.  Slight modulo bias is irrelevant here."""
    return hash_u32(n, seed, stream, 0, device) % int(high)


def c1_activations(n: int, d: int, p: float, seed: int, stream: int = 7, device="cpu") -> torch.Tensor:
    """C1 activation model: N(0,1) rows scaled to unit RMS (T5 final RMS-norm,
    python_transformers/modeling_t5.py:191-207,864), then inverted dropout with rate p (:865).
    The row RMS is taken from the exact integer sums, so the result is bit-reproducible."""
    a = _ih8(n * d, seed, stream, device).view(n, d)
    c = 2 * a - int(2 * _IH8_MEAN)                       # exact: 2*(acc - mean)
    ss = (c * c).sum(-1, keepdim=True)                    # exact int64, < 2**48
    rms = (ss.to(torch.float64) / (4.0 * _IH8_STD * _IH8_STD * d)).sqrt()
    y = ((a.to(torch.float64) - _IH8_MEAN) / _IH8_STD / rms).to(torch.float32)
    keep = hash_u32(n * d, seed, stream, 1, device).view(n, d) >= int(p * 4294967296.0)
    scale = torch.tensor(1.0 / (1.0 - p), dtype=torch.float32, device=device)
    return torch.where(keep, y * scale, torch.zeros((), dtype=torch.float32, device=device))


def clustered(n: int, codebook: torch.Tensor, sigma: float, seed: int, stream: int = 11):
    """x_i = W[c_i] + sigma * noise with c_i uniform over the codebook (SURVEY 8d 'uniform' regime).
    Returns (x [n, D] fp32, c [n] int64)."""
    k, d = codebook.shape
    c = uniform_int(n, k, seed, stream, codebook.device)
    noise = normal((n, d), seed, stream + 1, codebook.device)
    x = codebook.index_select(0, c) + noise * torch.tensor(sigma, dtype=torch.float32, device=codebook.device)
    return x, c


def role_ranges(n: int, bounds: list[tuple[int, int]], seed: int, stream: int = 13, device="cpu") -> torch.Tensor:
    """Per-token [start, end) ranges (int64 [n, 2]) drawn from a list of role ranges, as the
    reference's loader emits from ``dimension_dict`` (t5/load_data.py:76-99, train_t5.py:309)."""
    r = uniform_int(n, len(bounds), seed, stream, device)
    table = torch.tensor(bounds, dtype=torch.int64, device=device)
    return table.index_select(0, r)
