"""The ranged form of the distance kernel against the plain one on the same work."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from t5vqvae_amd import ops, synth  # noqa: E402

dev = torch.device("cuda", 0)
n, d = 65536, 768


def t(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


x = synth.normal((n, d), 5, 0, device=dev)
for k, lo, hi in ((1024, 0, 1024), (8192, 0, 1024), (8192, 1024, 2048), (8192, 7168, 8192), (8192, 0, 8192)):
    w = synth.normal((k, d), 6, 0, device=dev)
    xx, ee = ops.row_sqnorms(x, w)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    rg = torch.tensor([[lo, hi]], dtype=torch.int64, device=dev).repeat(n, 1).contiguous()
    tp = t(lambda: ops.dist_argmin(x, w, xx, ee, keys=keys)) if (lo, hi) == (0, k) else float("nan")
    tr = t(lambda: ops.dist_argmin(x, w, xx, ee, rg, keys=keys))
    print(f"K={k} range [{lo},{hi}): plain {tp:.0f} us, ranged {tr:.0f} us", flush=True)
