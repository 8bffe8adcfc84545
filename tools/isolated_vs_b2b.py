"""Isolated call (GPU idle before it) against back-to-back calls, for the module step and for the distance kernel alone.
    python tools/isolated_vs_b2b.py c1 [scan|filter]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t5vqvae_amd import ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
search = sys.argv[2] if len(sys.argv) > 2 else "scan"
dev = torch.device("cuda", 0)
n, k, d, update, _ = bench.WORKLOADS[name]
w, ema_w, cs, x = bench.build_state(name, "uniform", 0, 1, dev)
mod = bench.make_module(k, d, w, ema_w, cs, update, search, dev)
flat = x.reshape(-1, d)
xx, ee = ops.row_sqnorms(flat, w)
keys = torch.empty(n, dtype=torch.int64, device=dev)


def isolated(fn, reps=20):
    ts = []
    for _ in range(reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts[3:]), min(ts[3:])


def b2b(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


with torch.no_grad():
    step = lambda: mod(x)
    dist = lambda: ops.dist_argmin(flat, w, xx, ee, keys=keys)
    print(f"{name} {search}: step isolated median/min {isolated(step)} us, back-to-back {b2b(step):.1f} us")
    print(f"{name}: distance kernel isolated median/min {isolated(dist)} us, back-to-back {b2b(dist):.1f} us")
    g = lambda: ops.gather_stats(flat, w, keys, "mse", ste=torch.empty_like(flat))
    print(f"{name}: gather isolated {isolated(g)} us, back-to-back {b2b(g):.1f} us")
