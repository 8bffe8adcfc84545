"""Filtered search (bf16 filter + fp32 decision) against the fp32 scan: identical keys, time of each.

    python3 tools/bench_filter.py [c1 c2 c3 c4]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t5vqvae_amd import ops  # noqa: E402


def timed(fn, iters=15):
    ts = []
    for i in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


def main():
    dev = torch.device("cuda", 0)
    for name in (sys.argv[1:] or ["c2", "c1", "c3", "c4"]):
        n, k, d, update, desc = bench.WORKLOADS[name]
        for regime in ("uniform", "cold"):
            w, ema_w, cs, x = bench.build_state(name, regime, 0, 1, dev)
            flat = x.reshape(-1, d)
            xx, ee = ops.row_sqnorms(flat, w)
            k_scan = ops.dist_argmin(flat, w, xx, ee)
            k_filt, ovf = ops.dist_argmin_filtered(flat, w, xx, ee, want_overflow=True)
            torch.cuda.synchronize()
            same = bool(torch.equal(k_scan, k_filt))
            nbad = int((k_scan != k_filt).sum())
            t_scan = timed(lambda: ops.dist_argmin(flat, w, xx, ee))
            t_filt = timed(lambda: ops.dist_argmin_filtered(flat, w, xx, ee))
            k_idx = ops.dist_argmin_filtered(flat, w, xx, ee, exact_keys=False)
            same_idx = bool(torch.equal(k_idx & 0xFFFFFFFF, k_scan & 0xFFFFFFFF))
            t_idx = timed(lambda: ops.dist_argmin_filtered(flat, w, xx, ee, exact_keys=False))
            print(f"{name} {regime:8s} keys identical: {same} (differing {nbad}), index-only identical: {same_idx} | "
                  f"overflow tokens {int(ovf)} | scan {t_scan:9.1f} us | filtered {t_filt:9.1f} us "
                  f"(x{t_scan / t_filt:5.2f}) | index-only {t_idx:9.1f} us (x{t_scan / t_idx:5.2f})", flush=True)


if __name__ == "__main__":
    main()
