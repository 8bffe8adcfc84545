"""Grouped (disentangled) nearest-code search: scan and filtered search, with and without role bucketing.

    python3 tools/bench_ranged.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t5vqvae_amd import VectorQuantizerEMA, synth  # noqa: E402

dev = torch.device("cuda", 0)
d = 768


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n, k, roles, name in [
    (65536, 8192, [(i * 1024, (i + 1) * 1024) for i in range(8)], "8 equal roles of 1024"),
    (65536, 10000, [(0, 2000), (2000, 4000), (4000, 6000), (6000, 8000), (8000, 10000)], "5 roles of 2000 (evaluate_vqvae.py:460)"),
    (65536, 8192, [(0, 1000), (1000, 8192)], "V / residual (train_t5.py:309 style)"),
]:
    m = VectorQuantizerEMA(k, d, 0.25, decay=0.99, disentanglement=True).to(dev).eval()
    x = synth.normal((1, n, d), 5, 0, device=dev)
    srl = synth.role_ranges(n, roles, 5, 1, device=dev).view(1, n, 2)
    flops = 2.0 * d * sum(float((srl.view(-1, 2)[:, 0] == a).sum()) * (b - a) for a, b in roles)
    res = {}
    ref = None
    for search in ("scan", "filter"):
        m.search = search
        for bucket in (False, True):
            m.bucket_roles = bucket
            idx = m.get_indices(x, srl)
            ref = idx if ref is None else ref
            assert torch.equal(idx, ref)
            res[(search, bucket)] = timed(lambda: m.get_indices(x, srl))
    full = 2.0 * n * k * d
    print(f"N={n} K={k} {name}: scan unbucketed {res[('scan', False)]:.3f} ms, bucketed {res[('scan', True)]:.3f} ms | "
          f"filtered unbucketed {res[('filter', False)]:.3f} ms, bucketed {res[('filter', True)]:.3f} ms "
          f"(grouped flops {flops/1e9:.0f} G of dense {full/1e9:.0f} G; identical indices)", flush=True)
