"""Where the filtered search overtakes the fp32 scan (module.filter_min_work): search time of both, index-only keys.

    python3 tools/bench_crossover.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t5vqvae_amd import ops, synth  # noqa: E402


def timed(fn, iters=25):
    ts = []
    for i in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


def main():
    dev = torch.device("cuda", 0)
    d = 768
    for n, k in ((1024, 1024), (2048, 1024), (1024, 4096), (4096, 1024), (2048, 4096), (8192, 1024), (1024, 8192),
                 (4096, 4096), (2880, 10000), (16384, 1024), (8192, 4096)):
        w = synth.normal((k, d), 1, 0, device=dev)
        x, _ = synth.clustered(n, w, 0.5, 2)
        x = x.to(dev)

        def scan():
            xx, ee = ops.row_sqnorms(x, w)
            return ops.dist_argmin(x, w, xx, ee)

        def filt():
            return ops.dist_argmin_filtered(x, w, exact_keys=False)
        same = bool(torch.equal(scan() & 0xFFFFFFFF, filt() & 0xFFFFFFFF))
        ts, tf = timed(scan), timed(filt)
        print(f"N={n:6d} K={k:6d} N*K=2^{(n * k).bit_length() - 1:2d} ({n * k / 1e6:6.1f} M)  scan {ts:8.1f} us  "
              f"filtered {tf:8.1f} us  ratio {ts / tf:5.2f}  identical indices: {same}", flush=True)


if __name__ == "__main__":
    main()
