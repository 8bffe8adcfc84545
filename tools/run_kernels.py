"""Small driver for profiler passes: a few launches of the hot kernels at a BASELINE shape.

    python3 tools/run_kernels.py [c3|c1|c2|c4] [reps] [dist|step] [scan|filter|auto]
    python3 tools/run_kernels.py [ranged8|ranged5] [reps] search [scan|filter]   one code shard of grouped C5, one rank
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t5vqvae_amd import VectorQuantizerEMA, ops  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
what = sys.argv[3] if len(sys.argv) > 3 else "step"
search = sys.argv[4] if len(sys.argv) > 4 else "scan"   # the benchmark's headline mode
from t5vqvae_amd import _lib  # noqa: E402
_lib.load().t5vq_set_tuning(int(os.environ.get("T5VQ_TUNE", "0")))
dev = torch.device("cuda", 0)
if workload.startswith("ranged"):
    role_set = [r for r in bench.C5_ROLE_SETS if r.startswith(workload[len("ranged"):] + "_")][0]
    n, kl, d, off, w, x, rg, flops = bench.c5_grouped_setup(dev, role_set)
    xx, ee = ops.row_sqnorms(x, w)
    for _ in range(reps):
        order = ops.order_by_range(rg, kl, off)
        if search == "scan":
            ops.dist_argmin(x, w, xx, ee, rg, code_offset=off, order=order)
        else:
            ops.dist_argmin_filtered(x, w, xx=xx, ee=ee, ranges=rg, code_offset=off, order=order, exact_keys=True,
                                     range_sparse=os.environ.get("T5VQ_RANGE_SPARSE", "1") == "1")
    torch.cuda.synchronize()
    print(f"done {role_set} {search} algorithmic_flops={flops:.6g}")
    sys.exit(0)
n, k, d, update, _ = bench.WORKLOADS[workload]
w, ema_w, cs, x = bench.build_state(workload, os.environ.get("T5VQ_REGIME", "uniform"), 0, 1, dev)
if what == "dist":
    flat = x.reshape(-1, d)
    xx, ee = ops.row_sqnorms(flat, w)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    for _ in range(reps):
        ops.dist_argmin(flat, w, xx, ee, keys=keys)
else:
    mod = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev)
    with torch.no_grad():
        mod._embedding.weight.copy_(w)
        mod._ema_w.copy_(ema_w)
        mod._ema_cluster_size.copy_(cs)
        mod.encodings = "indices"
        mod.search = search
        if os.environ.get("T5VQ_CODE_IMAGE"):               # "always": the update writes the codebook's fp16 image too
            mod.code_image_cache = os.environ["T5VQ_CODE_IMAGE"]
        mod.train(update)
        for _ in range(reps):
            mod(x)
torch.cuda.synchronize()
print("done")
