"""A/B the tile configurations of the distance kernel in ONE process, interleaved rounds (needs a build made
with `make EXTRA=-DT5VQ_TUNING`).   python3 tools/tune_dist.py [workload] [variants...]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t5vqvae_amd import _lib, ops  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
variants = [int(v) for v in sys.argv[2:]] or list(range(8))
lib = _lib.load()
dev = torch.device("cuda", 0)
if "x" in workload:                       # custom "NxK" shape, D = 768
    n, k = (int(v) for v in workload.split("x"))
    bench.WORKLOADS[workload] = (n, k, 768, True, "custom")
n, k, d, _, _ = bench.WORKLOADS[workload]
w, _, _, x = bench.build_state(workload, "uniform", 0, 1, dev)
flat = x.reshape(-1, d)
xx, ee = ops.row_sqnorms(flat, w)
ref = None
times = {v: [] for v in variants}
for v in variants:
    lib.t5vq_set_tuning(v)
    keys = ops.dist_argmin(flat, w, xx, ee)
    torch.cuda.synchronize()
    if ref is None:
        ref = keys.clone()
    else:
        if v % 1000 not in (8, 9, 10, 11, 12, 13, 24, 25):    # these are timing-only ablations with wrong results
            assert torch.equal(keys, ref), f"variant {v} changes the keys"
reps = 3 if n * k > 2 ** 28 else 20
for rnd in range(5):
    for v in variants:
        lib.t5vq_set_tuning(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.dist_argmin(flat, w, xx, ee, keys=keys)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / reps)
flops = 2.0 * n * k * d
for v in variants:
    med, mn = statistics.median(times[v]), min(times[v])
    print(f"{workload} variant {v}: median {med:8.4f} ms  min {mn:8.4f} ms  {flops / med / 1e9:7.1f} TF/s "
          f"({flops / med / 1e9 / 157.3 * 100:5.1f}% of fp32 MFMA peak)", flush=True)
