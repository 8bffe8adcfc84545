"""Soak: the filtered search against the scan, many seeds and shapes, full keys (looks for rare staging races)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from t5vqvae_amd import ops, synth  # noqa: E402

dev = torch.device("cuda", 0)
bad = 0
runs = 0
for n, k, d, reps in ((65536, 8192, 768, 30), (2880, 10000, 768, 150), (9000, 2049, 256, 80), (131072, 1024, 128, 30)):
    for i in range(reps):
        w = synth.normal((k, d), 1000 + i, 0, device=dev)
        x = synth.clustered(n, w, 0.4 + 0.02 * (i % 10), 2000 + i)[0].to(dev) if i % 2 else \
            synth.normal((n, d), 2000 + i, 0, device=dev)
        xx, ee = ops.row_sqnorms(x, w)
        scan = ops.dist_argmin(x, w, xx, ee)
        filt = ops.dist_argmin_filtered(x, w, xx, ee)
        diff = int((scan != filt).sum())
        runs += 1
        if diff:
            bad += 1
            print(f"MISMATCH n={n} k={k} d={d} rep={i}: {diff} keys differ", flush=True)
    print(f"shape ({n}, {k}, {d}): {reps} runs done", flush=True)
print(f"{runs} runs, {bad} with mismatches")
sys.exit(1 if bad else 0)
