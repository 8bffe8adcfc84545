"""Where the role-bucketed search spends its time (N = 65536, 8 roles of 1024): the pieces around the kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from t5vqvae_amd import VectorQuantizerEMA, ops, synth  # noqa: E402
from t5vqvae_amd.roles import bucket_by_role  # noqa: E402

dev = torch.device("cuda", 0)
n, k, d = 65536, 8192, 768
roles = [(i * 1024, (i + 1) * 1024) for i in range(8)]
m = VectorQuantizerEMA(k, d, 0.25, decay=0.99, disentanglement=True).to(dev).eval()
x = synth.normal((n, d), 5, 0, device=dev)
rg = synth.role_ranges(n, roles, 5, 1, device=dev).view(n, 2)
w = m._embedding.weight.data


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


order, rs = bucket_by_role(rg)
xs = x.index_select(0, order)
ee = ops.code_sqnorms(w)
print("argsort + ranges select %.0f us" % timed(lambda: bucket_by_role(rg)))
print("x.index_select          %.0f us" % timed(lambda: x.index_select(0, order)))
ks, _ = ops.dist_argmin_filtered(xs, w, ee=ee, want_overflow=True, exact_keys=False, ranges=rs)
print("filtered search         %.0f us" % timed(lambda: ops.dist_argmin_filtered(xs, w, ee=ee, want_overflow=True, exact_keys=False, ranges=rs)))
xx, _ = ops.row_sqnorms(xs, None)
print("row norms               %.0f us" % timed(lambda: ops.row_sqnorms(xs, None)))
print("scan search             %.0f us" % timed(lambda: ops.dist_argmin(xs, w, xx, ee, rs)))
print("keys scatter back       %.0f us" % timed(lambda: torch.empty_like(ks).index_copy_(0, order, ks)))
st = rg[:, 0].contiguous()
print("counting sort by start  %.0f us" % timed(lambda: ops.order_by_code(st, k)))
for search in ("scan", "filter"):
    m.search = search
    print(search, "module.get_indices  %.0f us" % timed(lambda: m.get_indices(x.view(1, n, d), rg.view(1, n, 2))))
