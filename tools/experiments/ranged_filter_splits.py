"""Filtered ranged search: the split count of the ranged form (T5VQ_RANGED_SPLITS, experiment builds) on one code shard
of grouped C5 and on whole disentangled codebooks.   python3 tools/experiments/ranged_filter_splits.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from t5vqvae_amd import ops, synth  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = []
for rs in bench.C5_ROLE_SETS:
    n, kl, d, off, w, x, rg, flops = bench.c5_grouped_setup(dev, rs)
    cases.append((f"shard {rs}", x, w, rg, off, flops))
d = 768
for n, k, roles, name in [
    (65536, 8192, [(i * 1024, (i + 1) * 1024) for i in range(8)], "K=8192, 8 equal roles of 1024"),
    (65536, 10000, [(0, 2000), (2000, 4000), (4000, 6000), (6000, 8000), (8000, 10000)], "K=10000, 5 roles of 2000"),
    (65536, 8192, [(0, 1000), (1000, 8192)], "K=8192, V / residual"),
    (65536, 65536, bench.C5_ROLE_SETS["5_unequal_roles_straddling_shards"], "K=65536, 5 unequal roles"),
    (4096, 10000, [(0, 2000), (2000, 4000), (4000, 6000), (6000, 8000), (8000, 10000)], "N=4096 K=10000, 5 roles of 2000"),
]:
    w = synth.normal((k, d), 5, 3, device=dev)
    rg = synth.role_ranges(n, roles, 5, 1, device=dev).contiguous()
    span = rg[:, 1] - rg[:, 0]
    c = rg[:, 0] + synth.uniform_int(n, 1 << 30, 6, 5, dev) % span
    x = w.index_select(0, c) + synth.normal((n, d), 5, 0, device=dev) * 0.5
    cases.append((name, x, w, rg, 0, 2.0 * d * float(span.sum())))

for name, x, w, rg, off, flops in cases:
    k = w.shape[0]
    xx, ee = ops.row_sqnorms(x, w)
    order = ops.order_by_range(rg, k, off)
    ref = ops.dist_argmin(x, w, xx, ee, rg, code_offset=off, order=order)
    t_scan = timed(lambda: ops.dist_argmin(x, w, xx, ee, rg, code_offset=off, order=order))
    line = f"{name}: grouped {flops / 1e9:.0f} GF | scan {t_scan:.0f} us"
    for s in (0, 4, 8, 16, 32):
        os.environ["T5VQ_RANGED_SPLITS"] = str(s)
        keys, ovf = ops.dist_argmin_filtered(x, w, xx=xx, ee=ee, ranges=rg, code_offset=off, order=order, exact_keys=True,
                                             want_overflow=True)
        assert torch.equal(keys, ref), (name, s)
        t = timed(lambda: ops.dist_argmin_filtered(x, w, xx=xx, ee=ee, ranges=rg, code_offset=off, order=order,
                                                   exact_keys=True))
        line += f" | S={s}: {t:.0f} us (ovf {int(ovf)})"
    print(line, flush=True)
