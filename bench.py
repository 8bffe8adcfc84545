"""Benchmark of the VQ bottleneck hot path (fused forward + EMA codebook update) on MI355X.

    python bench.py [--gpus N --steps K --warmup W --workload c3|c4|c1|c2 --regime uniform|cold]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one ``VectorQuantizerEMA.forward`` in training mode (distance + argmin + gather + loss + perplexity
+ EMA update) over one resident batch of synthetic tokens.  Default workload: BASELINE config 3 per GPU
(N=65536 tokens/GPU, K=8192, D=768); with N GPUs the tokens are sharded (weak scaling: 65536 tokens per
rank) and the EMA statistics are all-reduced over RCCL each step.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA, same guide
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (tokens per GPU, K, D, update, description)
    "c1": (2880, 10000, 768, True, "C1 published training shape N=2880 K=10000 fwd+EMA"),
    "c2": (4096, 512, 768, False, "C2 N=4096 K=512 fwd only"),
    "c3": (65536, 8192, 768, True, "C3 N=65536/GPU K=8192 D=768 fused fwd+EMA"),
    "c4": (131072, 16384, 768, True, "C4 share N=2^20/8 per GPU K=16384 D=768 fused fwd+EMA"),
}


def build_state(workload, regime, rank, world, dev):
    from t5vqvae_amd import synth
    n, k, d, update, _ = WORKLOADS[workload]
    w = synth.normal((k, d), 1000, 0, device=dev)
    if regime == "uniform":                      # steady-state proxy: every code in use
        cs = torch.full((k,), float(n * world) / k, device=dev)
        ema_w = w * cs.unsqueeze(1)
        x, _ = synth.clustered(n, w, 0.5, 2000 + rank)
    else:                                        # reference cold start on N(0,1) inputs
        cs = torch.zeros(k, device=dev)
        ema_w = synth.normal((k, d), 1000, 1, device=dev)
        x = synth.normal((n, d), 2000 + rank, 0, device=dev)
    x = x.view(64, n // 64, d) if n % 64 == 0 else x.view(1, n, d)
    return w, ema_w, cs, x


def cpu_baseline(workload, dev, budget_s=20.0):
    """CPU leg: the oracle's faithful op-for-op path (dense one-hot, three matmuls) timed on
    the host cores on a bounded token slice of the workload -- and, on the same slice and the same initial state,
    used as the checker of one GPU step (the parity report SURVEY.md 8d asks for beside every benchmark)."""
    from oracle import vq_oracle as vo
    from t5vqvae_amd import VectorQuantizerEMA, synth
    n, k, d, update, _ = WORKLOADS[workload]
    ns = min(n, 8192)
    w = synth.normal((k, d), 1000, 0)
    cs = torch.full((k,), float(n) / k)
    st0 = vo.VQState(w, w * cs.unsqueeze(1), cs)
    x, _ = synth.clustered(ns, w, 0.5, 2000)
    x = x.view(1, ns, d)
    times = []
    t_start = time.perf_counter()
    with torch.no_grad():
        for i in range(6):
            st = st0.clone()
            t0 = time.perf_counter()
            ref = vo.faithful_forward(st, x, ema=True, training=update, commitment_cost=0.25)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s and i >= 1:
                break
        # parity of one GPU step on the same slice / state
        mod = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev)
        mod._embedding.weight.copy_(st0.weight)
        mod._ema_w.copy_(st0.ema_w)
        mod._ema_cluster_size.copy_(st0.cluster_size)
        mod.encodings = "indices"
        mod.train(update)
        loss, q, ppl, idx = mod(x.to(dev))
        idx = idx.squeeze(1).cpu()
        outside, inside = vo.classify_mismatches(x.view(-1, d), st0.weight, idx, ref.indices)
        chk = vo.algorithmic_forward(st0.clone(), x, ema=True, training=update, commitment_cost=0.25, indices=idx)

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double()
        return float((a - b).norm() / b.norm())
    stc = st0.clone()
    vo.algorithmic_forward(stc, x, ema=True, training=update, commitment_cost=0.25, indices=idx)
    parity = {"tokens": ns, "idx_mismatch_outside_band": outside, "idx_mismatch_inside_band": inside,
              "band": "|d64(a)-d64(b)| <= 1e-6*(|x|^2+max|e|^2)",
              "rel_err": {"quantized": rel(q, chk.quantized), "loss": rel(loss, chk.loss),
                          "perplexity": rel(ppl, chk.perplexity)}}
    if update:
        parity["rel_err"].update({"cluster_size": rel(mod._ema_cluster_size, stc.cluster_size),
                                  "ema_w": rel(mod._ema_w.data, stc.ema_w),
                                  "weight": rel(mod._embedding.weight.data, stc.weight)})
    tt = times[1:] or times
    med = sorted(tt)[len(tt) // 2]
    return {"value": ns / med, "unit": "vectors/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{ns}-token slice of the workload (K={k}, D={d}, fwd{'+EMA' if update else ''}), "
                      f"median of {len(tt)} calls, torch CPU fp32, {os.cpu_count()} host CPUs visible",
            "parity": parity}


def other_search(mod, x, w, ema_w, cs, args, n, k, d):
    """The same step with the other search mode (rank 0, single GPU): 'scan' = fused fp32 MFMA distance + argmin;
    'filter' = fp16 MFMA filter with a proven error bound + the scan's fp32 arithmetic on the surviving codes.
    Both produce the same indices (checked here on this batch, same initial state)."""
    from t5vqvae_amd import ops
    other = "filter" if args.search == "scan" else "scan"
    if other == "filter" and d % 64:
        return None

    def reset():
        with torch.no_grad():
            mod._embedding.weight.copy_(w), mod._ema_w.copy_(ema_w), mod._ema_cluster_size.copy_(cs)
    with torch.no_grad():
        reset()
        mod.search = args.search
        idx_a = mod(x)[3].clone()
        reset()
        mod.search = other
        idx_b = mod(x)[3].clone()
        for _ in range(max(2, args.warmup)):
            mod(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mod(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        flat = x.reshape(-1, d)
        wq = mod._embedding.weight.detach()
        xx, ee = ops.row_sqnorms(flat, wq)
        fn = (lambda: ops.dist_argmin_filtered(flat, wq, xx, ee, exact_keys=False)) if other == "filter" \
            else (lambda: ops.dist_argmin(flat, wq, xx, ee))
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        s_ms = e0.elapsed_time(e1) / 10
    mod.search = args.search
    flops = 2.0 * n * k * d
    peak = PEAK_F16_MFMA_TFLOPS if other == "filter" else PEAK_FP32_MFMA_TFLOPS
    return {"search": other, "value": n / (ms * 1e-3), "unit": "vectors/s", "ms_per_step": ms,
            "indices_identical_to_headline_mode": bool(torch.equal(idx_a, idx_b)),
            "search_ms": s_ms, "search_algorithmic_tflops": flops / (s_ms * 1e-3) / 1e12,
            "search_frac_of_pipe_peak": flops / (s_ms * 1e-3) / 1e12 / peak,
            "pipe": "fp16 MFMA filter (2500 TF) + fp32 decision" if other == "filter" else "fp32 MFMA (157.3 TF)"}


def step_bytes(n, k, d, update):
    b = 4.0 * (n * d + k * d + n * d) + 8.0 * n + 4.0 * k
    if update:
        b += 4.0 * n * d + 12.0 * k * d + 12.0 * k
    return b


def measured_traffic(workload):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json; counters cannot be read from inside this process)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)[workload]["dist_mfma_v2_kernel"]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--regime", default="uniform", choices=["uniform", "cold"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--search", default="scan", choices=["scan", "filter", "auto"],
                    help="how the module finds the nearest code for the headline number: 'scan' = the fused fp32 MFMA "
                         "kernel (the north star's design, default); 'filter'/'auto' = fp16 filter + fp32 decision "
                         "(identical indices); the other mode is always reported beside it")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # T5VQ_BENCH_FORCE_DIST=1 takes the RCCL path even at world size 1 (rehearsal on a one-GPU box)
    use_dist = world > 1 or os.environ.get("T5VQ_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from t5vqvae_amd import VectorQuantizerEMA, ops
    from t5vqvae_amd.sharded import shard_tokens
    n, k, d, update, desc = WORKLOADS[args.workload]
    w, ema_w, cs, x = build_state(args.workload, args.regime, rank, world, dev)
    mod = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev)
    with torch.no_grad():
        mod._embedding.weight.copy_(w)
        mod._ema_w.copy_(ema_w)
        mod._ema_cluster_size.copy_(cs)
    mod.encodings = "indices"
    mod.search = args.search
    mod.train(update)
    if use_dist:
        shard_tokens(mod)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = mod(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = mod(x)
        barrier()
        elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed / args.steps * 1e3
    value = n * world * args.steps / elapsed

    result = None
    if rank == 0:
        # roofline of the dominant kernel (fused distance + argmin), measured live with HIP events on the
        # stream the kernel runs on (torch's current stream is the one handed to the C ABI)
        flat = x.reshape(-1, d)
        wq = mod._embedding.weight.detach()
        xx, ee = ops.row_sqnorms(flat, wq)
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        filt = args.search == "filter" or (args.search == "auto" and d % 64 == 0 and n * k >= mod.filter_min_work)
        search = (lambda: ops.dist_argmin_filtered(flat, wq, xx, ee, keys=keys, exact_keys=False)) if filt \
            else (lambda: ops.dist_argmin(flat, wq, xx, ee, keys=keys))
        for _ in range(2):
            search()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            search()
        e1.record()
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / reps
        peak = PEAK_F16_MFMA_TFLOPS if filt else PEAK_FP32_MFMA_TFLOPS
        kname = "filtered search: filter_f16_kernel + prep/collect/refine" if filt \
            else "dist_mfma_v2_kernel (fused distance+argmin)"
        flops = 2.0 * n * k * d
        achieved = flops / (k_ms * 1e-3) / 1e12
        result = {
            "metric": "quantized vectors/sec (fwd+EMA)" if update else "quantized vectors/sec (fwd)",
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "tokens_per_gpu": n, "global_tokens": n * world, "K": k, "D": d,
                       "regime": args.regime, "parallelism": f"token-sharded x{world}" if world > 1 else "single GPU",
                       "encodings": "indices (dense one-hot excluded, SURVEY 8b)"},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "kernel_ms": k_ms,
                         "flops_per_launch": flops, "traffic": None if filt else measured_traffic(args.workload),
                         "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
                         "algorithmic_bytes": 4.0 * (n * d + k * d) + 12.0 * n + 4.0 * k},
            "step_frac_of_mfma_roofline": (flops / PEAK_FP32_MFMA_TFLOPS / 1e12) / (ms_per_step * 1e-3),
            # SURVEY 8d algorithmic bytes of the whole step (fwd: x, W, q, idx, ee; EMA: x again, ema_w r/w, W w)
            "step_algorithmic_bytes": step_bytes(n, k, d, update),
            "step_frac_of_hbm_roofline": step_bytes(n, k, d, update) / (PEAK_HBM_GBS * 1e9) / (ms_per_step * 1e-3),
        }
    if rank == 0 and world == 1:
        result["config"]["search"] = args.search
        result["other_search"] = other_search(mod, x, w, ema_w, cs, args, n, k, d)
    if use_dist:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.workload, dev)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
