"""BASELINE config 4 whole (N = 2^20, K = 16384) on ONE GPU through both searches.
Measured: identical keys, scan 180.5 ms (146 TFLOP/s fp32), filtered 30.9 ms."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from t5vqvae_amd import ops, synth
dev = torch.device("cuda", 0)
n, k, d = 1 << 20, 16384, 768
w = synth.normal((k, d), 1, 0, device=dev)
x, _ = synth.clustered(n, w, 0.5, 2)
xx, ee = ops.row_sqnorms(x, w)
def t(fn, reps=2):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ks = ops.dist_argmin(x, w, xx, ee)
kf, ovf = ops.dist_argmin_filtered(x, w, xx, ee, want_overflow=True)
print("identical", bool(torch.equal(ks, kf)), "unsettled", int(ovf))
print("scan ms", t(lambda: ops.dist_argmin(x, w, xx, ee)), "filter ms", t(lambda: ops.dist_argmin_filtered(x, w, exact_keys=False)))
