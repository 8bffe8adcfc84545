"""every single-GPU BASELINE config x assignment regime, one VQ call per measurement with
the module state reset OUTSIDE the timed region (HIP events around the call, median of `iters`), next to the
CPU reference path (oracle's faithful three-GEMM restatement) on the host cores.

    python3 tools/bench_regimes.py [--cpu] [--out profiles/r1_regimes.md]
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from t5vqvae_amd import VectorQuantizerEMA, synth  # noqa: E402

PEAK = bench.PEAK_FP32_MFMA_TFLOPS


def states(workload, regime, dev):
    n, k, d, update, _ = bench.WORKLOADS[workload]
    if regime == "uniform":
        return bench.build_state(workload, "uniform", 0, 1, dev)
    w, ema_w, cs, x = bench.build_state(workload, "cold", 0, 1, dev)
    if regime == "collapsed":          # state after two reference-faithful EMA steps from the cold start
        m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev).train()
        with torch.no_grad():
            m._embedding.weight.copy_(w), m._ema_w.copy_(ema_w), m._ema_cluster_size.copy_(cs)
            m.encodings = "indices"
            m(x), m(x)
        w, ema_w, cs = m._embedding.weight.data.clone(), m._ema_w.data.clone(), m._ema_cluster_size.clone()
    return w, ema_w, cs, x


def gpu_time(workload, regime, dev, iters, search="scan"):
    n, k, d, update, _ = bench.WORKLOADS[workload]
    w, ema_w, cs, x = states(workload, regime, dev)
    m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev).train(update)
    m.encodings = "indices"
    m.search = search
    ts, used = [], 0
    with torch.no_grad():
        for i in range(iters + 2):
            m._embedding.weight.copy_(w), m._ema_w.copy_(ema_w), m._ema_cluster_size.copy_(cs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = m(x)
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        used = int(torch.unique(out[3]).numel())
    return statistics.median(ts), min(ts), used


def cpu_time(workload, regime, budget=25.0):
    from oracle import vq_oracle as vo
    n, k, d, update, _ = bench.WORKLOADS[workload]
    ns = min(n, 8192)
    w, ema_w, cs, x = (t.cpu() for t in states(workload, regime, torch.device("cuda", 0)))
    x = x.reshape(-1, d)[:ns].view(1, ns, d)
    st0 = vo.VQState(w, ema_w, cs)
    ts, t_start = [], time.perf_counter()
    with torch.no_grad():
        for i in range(6):
            st = st0.clone()
            t0 = time.perf_counter()
            vo.faithful_forward(st, x, ema=True, training=update, commitment_cost=0.25)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget and i >= 1:
                break
    return statistics.median(ts[1:] or ts), ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true", help="also time the CPU reference path (slow)")
    ap.add_argument("--out", default="")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lines = ["| config | regime | codes used | scan: GPU ms (median / min) | M vectors/s | % fp32 MFMA roofline (2NKD) |"
             " filtered search: GPU ms | M vectors/s | CPU ref vectors/s (slice) | scan speed-up |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for wl in ("c1", "c2", "c3", "c4"):
        n, k, d, update, desc = bench.WORKLOADS[wl]
        for regime in ("uniform", "cold", "collapsed"):
            med, mn, used = gpu_time(wl, regime, dev, args.iters)
            vps = n / med * 1e3
            frac = 2.0 * n * k * d / (med * 1e-3) / 1e12 / PEAK
            fmed = gpu_time(wl, regime, dev, args.iters, "filter")[0] if d % 64 == 0 else float("nan")
            cpu = ""
            sp = ""
            if args.cpu and regime == "uniform":
                ct, ns = cpu_time(wl, regime)
                cpu = f"{ns / ct:,.0f} ({ns} tokens, {torch.get_num_threads()} threads)"
                sp = f"{vps / (ns / ct):,.0f}x"
            lines.append(f"| {wl.upper()} N={n} K={k} {'fwd+EMA' if update else 'fwd'} | {regime} | {used} | "
                         f"{med:.4f} / {mn:.4f} | {vps / 1e6:.3f} | {100 * frac:.1f} | {fmed:.4f} | "
                         f"{n / fmed * 1e3 / 1e6:.3f} | {cpu} | {sp} |")
            print(lines[-1], flush=True)
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write("# Regime table (tools/bench_regimes.py): one call per measurement, state reset outside the "
                    "timed region, HIP events, median of %d\n\n" % args.iters + text)


if __name__ == "__main__":
    main()
