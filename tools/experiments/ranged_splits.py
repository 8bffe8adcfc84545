"""Role-bucketed ranged search at N=65536, K=8192, 8 roles of 1024: kernel time against the number of code splits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from t5vqvae_amd import _lib, ops, synth  # noqa: E402
from t5vqvae_amd.roles import bucket_by_role  # noqa: E402

dev = torch.device("cuda", 0)
n, k, d = 65536, 8192, 768
lib = _lib.load()
x = synth.normal((n, d), 5, 0, device=dev)
w = synth.normal((k, d), 6, 0, device=dev)
roles = [(i * 1024, (i + 1) * 1024) for i in range(8)]
rg = synth.role_ranges(n, roles, 5, 1, device=dev)
order, rs = bucket_by_role(rg)
xs = x.index_select(0, order)
xx, ee = ops.row_sqnorms(xs, w)
keys = torch.empty(n, dtype=torch.int64, device=dev)
ref = None
for s in (0, 1, 2, 4, 8, 16, 32, 64):
    lib.t5vq_set_tuning(1000 * s)
    ops.dist_argmin(xs, w, xx, ee, rs, keys=keys)
    torch.cuda.synchronize()
    ref = keys.clone() if ref is None else ref
    assert torch.equal(ref, keys)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.dist_argmin(xs, w, xx, ee, rs, keys=keys)
    e1.record()
    torch.cuda.synchronize()
    print(f"splits {s}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per search (ideal 1/8 of the dense scan: ~715 us)", flush=True)
lib.t5vq_set_tuning(0)
