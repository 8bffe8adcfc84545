"""Benchmark of the VQ bottleneck hot path (fused forward + EMA codebook update) on MI355X.

    python bench.py [--gpus N --steps K --warmup W --workload auto|c1|c2|c3|c4|c5 --regime uniform|cold|collapsed]

A step = one ``VectorQuantizerEMA.forward`` in training mode (distance + argmin + gather + loss + perplexity + EMA
update) over one resident batch of synthetic tokens.  Prints ONE JSON line (rank 0).

Launching.  ``--gpus 1`` runs in this process.  ``--gpus N`` (N > 1) without a launcher: this process starts N fresh
child processes BEFORE it touches the GPU (it never re-executes itself), one rank per GPU, rendezvous on 127.0.0.1 over
RCCL, and relays rank 0's line.  Under ``python -m torch.distributed.run`` (RANK / WORLD_SIZE in the environment) the
process is a rank and does not spawn.  ``T5VQ_BENCH_FORCE_DIST=1`` takes the RCCL path at world size 1 (rehearsal on a
one-GPU box); ``--dry-run`` drives spawn + rendezvous + the statistics all-reduce on CPU over gloo with a test stand-in
for the kernels (no number it prints is a measurement).

Workloads (BASELINE.json configs; D = 768, fp32):
  c3  N = 65536 tokens per GPU, K = 8192 (config 3): the headline at --gpus 1; with N GPUs token-sharded, WEAK scaling
  c4  N = 2^20 tokens in total, N/G per GPU, K = 16384 (config 4): token-sharded, STRONG scaling -- the default for N > 1
  c5  K = 65536 codes sharded K/G per GPU, N = 65536 tokens replicated (config 5): global argmin across shards, and the
      disentangled grouped forms (8 role ranges aligned to the shards, 5 unequal ranges that straddle them)
  c1  N = 2880, K = 10000 (published training shape, config 1);  c2  N = 4096, K = 512 forward only (config 2)
At --gpus 1 the line also carries `configs`: every other single-GPU configuration (and the assignment regimes of c3)
measured one call at a time -- HIP events around the call, state reset outside the timed region, median / min -- with
its roofline fractions and a parity check against the CPU oracle on a token slice.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA, same guide
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (tokens per GPU, K, D, update, description)  -- single-GPU shapes; c4 here is ONE GPU's share at 8 GPUs
    "c1": (2880, 10000, 768, True, "C1 published training shape N=2880 K=10000 fwd+EMA"),
    "c2": (4096, 512, 768, False, "C2 N=4096 K=512 fwd only"),
    "c3": (65536, 8192, 768, True, "C3 N=65536/GPU K=8192 D=768 fused fwd+EMA"),
    "c4": (131072, 16384, 768, True, "C4 share N=2^20/8 per GPU K=16384 D=768 fused fwd+EMA"),
}
C4_TOTAL_TOKENS = 1 << 20
WARMUP_CALLS = 5          # untimed calls in front of every per-call series (>= 20 iterations after >= 5 warm-ups)
C5_TOKENS, C5_CODES = 65536, 65536


# =====================================================================================================
# launcher (parent process: no GPU call, no torch import)
# =====================================================================================================
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n_ranks, argv):
    """Start one child per rank (fresh interpreters, this script, same arguments) and relay their output; rank 0 prints
    the JSON line.  The parent has not initialised the GPU and never replaces itself with another program."""
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), T5VQ_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    deadline = time.time() + float(os.environ.get("T5VQ_BENCH_TIMEOUT", "1500"))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is not None:
                    pending.remove(p)
                    if code != 0:
                        rc = rc or code
            if rc or time.time() > deadline:
                rc = rc or 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:                      # exactly the children started above
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc


# =====================================================================================================
# state, regimes
# =====================================================================================================
def build_state(workload, regime, rank, world, dev, n=None, k=None):
    """(w, ema_w, cluster_size, x) of one rank.  uniform: steady-state proxy, every code in use; cold: the
    reference's cold start on N(0,1) inputs; collapsed: the state two reference-faithful steps after the cold start."""
    import torch
    from t5vqvae_amd import VectorQuantizerEMA, synth
    wn, wk, d, update, _ = WORKLOADS[workload]
    n, k = n or wn, k or wk
    w = synth.normal((k, d), 1000, 0, device=dev)
    if regime == "uniform":
        cs = torch.full((k,), float(n * world) / k, device=dev)
        ema_w = w * cs.unsqueeze(1)
        x, _ = synth.clustered(n, w, 0.5, 2000 + rank)
    else:
        cs = torch.zeros(k, device=dev)
        ema_w = synth.normal((k, d), 1000, 1, device=dev)
        x = synth.normal((n, d), 2000 + rank, 0, device=dev)
        if regime == "collapsed":
            m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev).train()
            m.encodings = "indices"
            with torch.no_grad():
                m._embedding.weight.copy_(w), m._ema_w.copy_(ema_w), m._ema_cluster_size.copy_(cs)
                m(x.view(1, n, d)), m(x.view(1, n, d))
            w, ema_w, cs = m._embedding.weight.data.clone(), m._ema_w.data.clone(), m._ema_cluster_size.clone()
    x = x.view(64, n // 64, d) if n % 64 == 0 else x.view(1, n, d)
    return w, ema_w, cs, x


def make_module(k, d, w, ema_w, cs, update, search, dev):
    import torch
    from t5vqvae_amd import VectorQuantizerEMA
    mod = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev)
    with torch.no_grad():
        mod._embedding.weight.copy_(w), mod._ema_w.copy_(ema_w), mod._ema_cluster_size.copy_(cs)
    mod.encodings = "indices"
    mod.search = search
    mod.train(update)
    return mod


def step_bytes(n, k, d, update):
    """SURVEY 8d algorithmic bytes of the whole step (fwd: x, W, q, idx, ee; EMA: x again, ema_w r/w, W w)."""
    b = 4.0 * (n * d + k * d + n * d) + 8.0 * n + 4.0 * k
    if update:
        b += 4.0 * n * d + 12.0 * k * d + 12.0 * k
    return b


def roofline_fracs(n, k, d, update, ms):
    flops = 2.0 * n * k * d
    return {"frac_of_mfma_roofline": (flops / PEAK_FP32_MFMA_TFLOPS / 1e12) / (ms * 1e-3),
            "frac_of_hbm_roofline": step_bytes(n, k, d, update) / (PEAK_HBM_GBS * 1e9) / (ms * 1e-3),
            "algorithmic_flops": flops, "algorithmic_bytes": step_bytes(n, k, d, update)}


def back_to_back_ms(mod, x, w, ema_w, cs, iters):
    """The same call queued back to back (no host wait between calls, state reset once before): what a training loop
    that never synchronises sees.  HIP events around the whole run."""
    import torch
    with torch.no_grad():
        mod._embedding.weight.copy_(w), mod._ema_w.copy_(ema_w), mod._ema_cluster_size.copy_(cs)
        for _ in range(3):
            mod(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            mod(x)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def per_call_ms(mod, x, w, ema_w, cs, iters):
    """One call per measurement: state reset OUTSIDE the timed region, HIP events around the call on the stream the
    kernels run on (torch's current stream is the one the extension launches on).  (median, min) in ms."""
    import torch
    ts = []
    with torch.no_grad():
        for i in range(iters + WARMUP_CALLS):
            mod._embedding.weight.copy_(w), mod._ema_w.copy_(ema_w), mod._ema_cluster_size.copy_(cs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            out = mod(x)
            e1.record()
            torch.cuda.synchronize()
            if i >= WARMUP_CALLS:
                ts.append(e0.elapsed_time(e1))
    return statistics.median(ts), min(ts), out


def graph_replay_ms(mod, x, w, ema_w, cs, iters):
    """The per-call protocol with the call captured once in a HIP graph and replayed: what a caller sees who takes the
    host's enqueue work (Python, dispatcher, one hipLaunchKernel per node) out of a launch-bound step.  Same kernels,
    same state reset outside the timed region, HIP events around the replay.  (median, min) in ms."""
    import torch
    ts = []
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream, as capture requires
            for _ in range(3):
                mod(x)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            mod(x)
        for i in range(iters + WARMUP_CALLS):
            mod._embedding.weight.copy_(w), mod._ema_w.copy_(ema_w), mod._ema_cluster_size.copy_(cs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            if i >= WARMUP_CALLS:
                ts.append(e0.elapsed_time(e1))
    del graph
    return statistics.median(ts), min(ts)


# =====================================================================================================
# CPU leg: the oracle as baseline and as checker (the ONLY place that imports oracle/)
# =====================================================================================================
def cpu_leg(k, d, update, search, state, dev, slice_tokens, time_budget_s, desc):
    """The oracle's faithful op-for-op path (dense one-hot, three matmuls) on the host cores on
    a bounded token slice -- timed when time_budget_s > 0 -- and, on the same slice and the same initial state, the
    checker of one GPU step taken with the benchmark's own search mode (the parity report of SURVEY.md 8d)."""
    import torch
    from oracle import vq_oracle as vo
    w, ema_w, cs, x = (t.detach().cpu() for t in state)
    x = x.reshape(-1, d)
    ns = min(x.shape[0], slice_tokens)
    x = x[:ns].contiguous().view(1, ns, d)
    st0 = vo.VQState(w, ema_w, cs)
    times = []
    t_start = time.perf_counter()
    with torch.no_grad():
        for i in range(6 if time_budget_s > 0 else 1):
            st = st0.clone()
            t0 = time.perf_counter()
            ref = vo.faithful_forward(st, x, ema=True, training=update, commitment_cost=0.25)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > time_budget_s and i >= 1:
                break
        mod = make_module(k, d, st0.weight, st0.ema_w, st0.cluster_size, update, search, dev)
        loss, q, ppl, idx = mod(x.to(dev))
        idx = idx.squeeze(1).cpu()
        outside, inside = vo.classify_mismatches(x.view(-1, d), st0.weight, idx, ref.indices)
        stc = st0.clone()
        chk = vo.algorithmic_forward(stc, x, ema=True, training=update, commitment_cost=0.25, indices=idx)

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double()
        return float((a - b).norm() / b.norm())
    parity = {"tokens": ns, "search": search, "idx_mismatch_outside_band": outside, "idx_mismatch_inside_band": inside,
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
            "sample": f"{ns}-token slice of {desc} (K={k}, D={d}, fwd{'+EMA' if update else ''}), "
                      f"median of {len(tt)} call(s), torch CPU fp32, {os.cpu_count()} host CPUs visible",
            "parity": parity}


def c5_shard_config(dev, search, iters, check):
    """ONE code shard of config 5 on one GPU: 8192 of 65536 codes (code_offset = shard 3), all 65536 tokens, statistics
    for the local slice only, the two-phase cluster update (without its scalar all-reduce: no peer here).  Parity: the
    shard's indices minus its offset against the oracle's argmin over the same code slice."""
    import torch
    from t5vqvae_amd import ops, synth
    n, kl, d, k = C5_TOKENS, C5_CODES // 8, 768, C5_CODES
    off = 3 * kl
    w = synth.normal((kl, d), 1000, 3, device=dev)
    x, _ = synth.clustered(n, w, 0.5, 2000)
    cs0 = torch.full((kl,), float(n) / kl, device=dev)
    ew0 = w * cs0.unsqueeze(1)
    cs, ew, wl = cs0.clone(), ew0.clone(), w.clone()
    use_filter = search != "scan"

    def call():
        ee = ops.code_sqnorms(wl)
        if use_filter:
            keys = ops.dist_argmin_filtered(x, wl, ee=ee, code_offset=off, exact_keys=True)
        else:
            keys = ops.dist_argmin(x, wl, ops.row_sqnorms(x, None)[0], ee, code_offset=off)
        # (the all-reduce(MIN) of the keys over the shards sits here; every token of this synthetic batch is won locally)
        local = keys - off
        counts = torch.empty(kl, device=dev)
        es = torch.empty(kl, d, device=dev)
        perm = ops.order_by_code(local, kl)
        part = torch.empty(n, device=dev)
        ste = torch.empty_like(x)
        idx = torch.empty(n, dtype=torch.int64, device=dev)
        ops.gather_stats(x, wl, local, "mse", ste=ste, idx=idx, counts=counts, embed_sum=es, loss_partial=part, perm=perm)
        sc = torch.empty(4, device=dev)
        args = (counts, part, kl, n, float(n * d), 0.99, 1 - 0.99, 1e-5, k * 1e-5, 0.25, False)
        ops.ema_cluster(cs, *args, 2, sc)
        ops.ema_cluster(cs, *args, 3, sc)
        ops.ema_embed(ew, wl, es, cs, 0.99, 1 - 0.99)
        return keys

    ts = []
    with torch.no_grad():
        for i in range(iters + WARMUP_CALLS):
            cs.copy_(cs0), ew.copy_(ew0), wl.copy_(w)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            keys = call()
            e1.record()
            torch.cuda.synchronize()
            if i >= WARMUP_CALLS:
                ts.append(e0.elapsed_time(e1))
    med = statistics.median(ts)
    out = {"name": "c5_shard", "workload": f"one code shard of C5: {kl} of {k} codes (code_offset {off}), N={n} tokens, "
                                           "fwd + local-slice EMA, fine-grained entry points, no peer exchange",
           "search": search, "regime": "uniform", "ms_per_call_median": med, "ms_per_call_min": min(ts),
           "vectors_per_s": n / (med * 1e-3)}
    if check:                                  # part of the CPU leg: the oracle as checker
        from oracle import vq_oracle as vo
        ns = 4096
        idx_gpu = ((keys[:ns] & 0xFFFFFFFF) - off).cpu()
        xs, wc = x[:ns].cpu(), w.cpu()
        ref = torch.argmin(vo.distances(xs, wc), dim=1)
        outside, inside = vo.classify_mismatches(xs, wc, idx_gpu, ref)
        out["parity"] = {"tokens": ns, "idx_mismatch_outside_band": outside, "idx_mismatch_inside_band": inside,
                         "against": "oracle argmin over the shard's code slice"}
    out.update(roofline_fracs(n, kl, d, True, med))
    return out


C5_ROLE_SETS = {
    "8_roles_aligned_to_shards": [(i * (C5_CODES // 8), (i + 1) * (C5_CODES // 8)) for i in range(8)],
    "5_unequal_roles_straddling_shards": [(0, C5_CODES // 10), (C5_CODES // 10, C5_CODES // 3),
                                          (C5_CODES // 3, C5_CODES // 2 + 5),
                                          (C5_CODES // 2 + 5, C5_CODES - C5_CODES // 7), (C5_CODES - C5_CODES // 7, C5_CODES)],
}


def c5_grouped_setup(dev, role_set):
    """One code shard of the GROUPED form of config 5: codes [3 K/8, 4 K/8) of K = 65536, all 65536 tokens, every token
    restricted to its role's [start, end) range of the whole codebook (vqvae/vqvae.py:110-119)."""
    import torch
    from t5vqvae_amd import synth
    n, kl, d = C5_TOKENS, C5_CODES // 8, 768
    off = 3 * kl
    w = synth.normal((kl, d), 1000, 3, device=dev)
    roles = C5_ROLE_SETS[role_set]
    rg = synth.role_ranges(n, roles, 77, 0, device=dev).contiguous()
    # codes of the shard each token may take: its range cut to [off, off + kl)
    lo, hi = rg[:, 0].clamp(off, off + kl), rg[:, 1].clamp(off, off + kl)
    flops = 2.0 * d * float((hi - lo).clamp_min(0).sum())
    # 'uniform' regime: every token sits near a code of its own range (tokens whose range misses the shard: near any code)
    span = (hi - lo).clamp_min(0)
    pick = synth.uniform_int(n, 1 << 30, 2000, 5, dev)
    c = torch.where(span > 0, lo - off + pick % span.clamp_min(1), pick % kl)
    x = w.index_select(0, c) + synth.normal((n, d), 2000, 0, device=dev) * 0.5
    return n, kl, d, off, w, x, rg, flops


def c5_grouped_shard_config(dev, role_set, search, iters, check):
    """The ranged search of one code shard (role order on the device + fused distance / argmin over each token's own
    range), per call; roofline on the grouped flops 2 D sum_i |range_i within the shard|."""
    import torch
    from t5vqvae_amd import ops
    n, kl, d, off, w, x, rg, flops = c5_grouped_setup(dev, role_set)
    xx, ee = ops.row_sqnorms(x, w)

    def call(mode):
        order = ops.order_by_range(rg, kl, off)
        if mode == "scan":
            return ops.dist_argmin(x, w, xx, ee, rg, code_offset=off, order=order)
        return ops.dist_argmin_filtered(x, w, xx=xx, ee=ee, ranges=rg, code_offset=off, order=order, exact_keys=True,
                                        range_sparse=True)     # as the code-sharded module calls it on a shard

    def timed(mode):
        ts = []
        for i in range(iters + WARMUP_CALLS):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            keys = call(mode)
            e1.record()
            torch.cuda.synchronize()
            if i >= WARMUP_CALLS:
                ts.append(e0.elapsed_time(e1))
        return statistics.median(ts), min(ts), keys

    med, mn, keys = timed(search)
    other = "filter" if search == "scan" else "scan"
    omed, _, okeys = timed(other)
    out = {"name": f"c5_grouped_shard_{role_set}",
           "workload": f"one code shard of grouped C5 ({role_set}): codes [{off}, {off + kl}) of {C5_CODES}, N={n} tokens, "
                       "role order on the device + ranged search, fine-grained entry points, no peer exchange",
           "search": search, "ms_per_call_median": med, "ms_per_call_min": mn, "vectors_per_s": n / (med * 1e-3),
           "algorithmic_flops": flops, "flops_rule": "2 D sum over tokens of |range within the shard|",
           "dense_flops": 2.0 * n * kl * d,
           "frac_of_mfma_roofline": (flops / PEAK_FP32_MFMA_TFLOPS / 1e12) / (med * 1e-3),
           "other_search": {"search": other, "ms_per_call_median": omed,
                            "indices_identical_to_headline_mode": bool(torch.equal(keys & 0xFFFFFFFF,
                                                                                    okeys & 0xFFFFFFFF))}}
    if check:
        from oracle import vq_oracle as vo
        ns = 2048
        xs, wc, rs = x[:ns].cpu(), w.cpu(), rg[:ns].cpu()
        dist = vo.distances(xs, wc)                                  # [ns, kl] against the shard's codes
        codes = torch.arange(off, off + kl)
        valid = (codes.unsqueeze(0) >= rs[:, :1]) & (codes.unsqueeze(0) < rs[:, 1:])
        has = valid.any(1)
        ref_local = torch.where(valid, dist, torch.full_like(dist, float("inf"))).argmin(1)
        got = keys[:ns].cpu()
        none = got == 0x7FFFFFFFFFFFFFFF
        got_local = (got & 0xFFFFFFFF) + rs[:, 0] - off              # the key's index word is slice-relative
        outside, inside = vo.classify_mismatches(xs[has], wc, got_local[has], ref_local[has])
        out["parity"] = {"tokens": ns, "tokens_with_codes_in_this_shard": int(has.sum()),
                         "none_keys_where_the_range_misses_the_shard": bool(torch.equal(none, ~has)),
                         "idx_mismatch_outside_band": outside, "idx_mismatch_inside_band": inside,
                         "against": "oracle distances to the shard's codes, argmin over range ∩ shard, slice-relative"}
    return out


def single_gpu_configs(args, dev):
    """Every BASELINE configuration one GPU can hold besides the headline, plus the assignment regimes of c3."""
    import torch
    out = []
    plan = [("c1", "uniform"), ("c2", "uniform"), ("c4", "uniform"), ("c3", "uniform"), ("c3", "cold"),
            ("c3", "collapsed")]
    for wl, regime in plan:
        n, k, d, update, desc = WORKLOADS[wl]
        state = build_state(wl, regime, 0, 1, dev)
        w, ema_w, cs, x = state
        mod = make_module(k, d, w, ema_w, cs, update, args.search, dev)
        med, mn, res = per_call_ms(mod, x, w, ema_w, cs, args.config_iters)
        entry = {"name": wl if regime == "uniform" else f"{wl}_{regime}", "workload": desc, "regime": regime,
                 "search": args.search, "ms_per_call_median": med, "ms_per_call_min": mn,
                 "vectors_per_s": n / (med * 1e-3), "codes_used": int(res[3].unique().numel())}
        entry.update(roofline_fracs(n, k, d, update, med))
        if regime == "uniform":           # queued back to back the state is not reset: only the stationary regime
            b2b = back_to_back_ms(mod, x, w, ema_w, cs, 2 * args.config_iters)
            entry["ms_back_to_back"] = b2b
            entry["frac_of_mfma_roofline_back_to_back"] = (2.0 * n * k * d / PEAK_FP32_MFMA_TFLOPS / 1e12) / (b2b * 1e-3)
        if wl in ("c1", "c2"):            # launch-bound shapes: the same call replayed from a captured HIP graph
            gmed, gmin = graph_replay_ms(mod, x, w, ema_w, cs, args.config_iters)
            entry["ms_per_call_graph_replay_median"] = gmed
            entry["ms_per_call_graph_replay_min"] = gmin
            entry["frac_of_mfma_roofline_graph_replay"] = (2.0 * n * k * d / PEAK_FP32_MFMA_TFLOPS / 1e12) / (gmed * 1e-3)
        if not args.no_cpu_baseline:
            leg = cpu_leg(k, d, update, args.search, state, dev, 2880 if wl == "c1" else 4096, 0.0, desc)
            entry["parity"] = leg["parity"]
            entry["cpu_ref_vectors_per_s_one_call"] = leg["value"]
        # the module's default search ("auto") takes the other mode from N*K = 2^23 on: the same call, same protocol, with
        # that mode, and whether it returns the headline mode's indices
        other = "filter" if args.search == "scan" else "scan"
        if d % 64 == 0 and n * k >= mod.filter_min_work:
            mod.search = other
            omed, omn, ores = per_call_ms(mod, x, w, ema_w, cs, args.config_iters)
            entry["other_search"] = {"search": other, "ms_per_call_median": omed, "ms_per_call_min": omn,
                                     "vectors_per_s": n / (omed * 1e-3),
                                     "indices_identical_to_headline_mode": bool(torch.equal(ores[3], res[3]))}
            if regime == "uniform":
                entry["other_search"]["ms_back_to_back"] = back_to_back_ms(mod, x, w, ema_w, cs, 2 * args.config_iters)
            mod.search = args.search
        out.append(entry)
        del mod, state, w, ema_w, cs, x
    out.append(c5_shard_config(dev, args.search, args.config_iters, not args.no_cpu_baseline))
    for role_set in C5_ROLE_SETS:
        out.append(c5_grouped_shard_config(dev, role_set, args.search, args.config_iters, not args.no_cpu_baseline))
    return out


# =====================================================================================================
# ranks
# =====================================================================================================
def other_search(mod, x, w, ema_w, cs, args, n, k, d):
    """The same step with the other search mode (rank 0, single GPU): 'scan' = fused fp32 MFMA distance + argmin;
    'filter' = fp16 MFMA filter with a proven error bound + the scan's fp32 arithmetic on the surviving codes.
    Both produce the same indices (checked here on this batch, same initial state)."""
    import torch
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


def measured_traffic(workload):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json; counters cannot be read from inside this process)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)[workload]["dist_mfma_v2_kernel"]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def search_kernel_roofline(mod, x, n, k, d, args, workload):
    """Roofline of the dominant kernel (fused distance + argmin), measured live with HIP events on the stream the
    kernel runs on (torch's current stream is the one handed to the C ABI)."""
    import torch
    from t5vqvae_amd import ops
    dev = x.device
    flat = x.reshape(-1, d)
    wq = mod._embedding.weight.detach()
    xx, ee = ops.row_sqnorms(flat, wq)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    filt = args.search == "filter" or (args.search == "auto" and d % 64 == 0 and n * k >= mod.filter_min_work)
    search = (lambda: ops.dist_argmin_filtered(flat, wq, xx, ee, keys=keys, exact_keys=False)) if filt \
        else (lambda: ops.dist_argmin(flat, wq, xx, ee, keys=keys))
    for _ in range(3):
        search()
    reps = 30
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:                     # one event pair per launch, all queued back to back, read after one sync
        a.record()
        search()
        b.record()
    torch.cuda.synchronize()
    per_launch = sorted(a.elapsed_time(b) for a, b in evs)
    # the median launch: a single stall of the box (seen once: one launch of 14 ms among 5.7 ms ones) moves a mean of
    # few launches by several per cent and says nothing about the kernel; mean, min and max are reported beside it
    k_ms = statistics.median(per_launch)
    peak = PEAK_F16_MFMA_TFLOPS if filt else PEAK_FP32_MFMA_TFLOPS
    kname = "filtered search: filter_f16_kernel + prep/collect/refine" if filt \
        else "dist_mfma_v2_kernel (fused distance+argmin)"
    flops = 2.0 * n * k * d
    achieved = flops / (k_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "kernel_ms": k_ms, "kernel_ms_stat": f"median of {reps} launches, HIP events per launch",
            "kernel_ms_mean": sum(per_launch) / reps, "kernel_ms_min": per_launch[0], "kernel_ms_max": per_launch[-1],
            "flops_per_launch": flops,
            "traffic": None if filt else measured_traffic(workload),
            "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
            "algorithmic_bytes": 4.0 * (n * d + k * d) + 12.0 * n + 4.0 * k}


def series_point(workload, n, k, d, args, rank, world, dev, dist, barrier, steps):
    """A second token-sharded measurement inside the same run (same protocol as the headline: warm-up, K steps
    between barriers, MAX over ranks), so that every line carries a point of BOTH multi-GPU series: config 3 weak
    (65536 tokens per GPU, K = 8192) and config 4 strong (2^20 tokens in total, K = 16384)."""
    import torch
    from t5vqvae_amd.sharded import shard_tokens
    w, ema_w, cs, x = build_state(workload, args.regime, rank, world, dev, n=n, k=k)
    mod = make_module(k, d, w, ema_w, cs, True, args.search, dev)
    if dist is not None:
        shard_tokens(mod)
    with torch.no_grad():
        for _ in range(2):
            mod(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            mod(x)
        barrier()
        el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t)
    ms = el / steps * 1e3
    return {"n_gpus": world, "tokens_per_gpu": n, "global_tokens": n * world, "K": k, "steps": steps, "ms_per_step": ms,
            "value": n * world / (ms * 1e-3), "unit": "vectors/s",
            "step_frac_of_mfma_roofline": (2.0 * n * k * d / PEAK_FP32_MFMA_TFLOPS / 1e12) / (ms * 1e-3)}


def run_c5(args, rank, world, dev, dist, barrier):
    """Config 5: K = 65536 codes sharded K/G per rank, N = 65536 replicated tokens; global argmin, then the grouped
    (disentangled) forms.  Both codebook exchanges are timed: kept + refreshed asynchronously after the update
    ("prefetch", the default) and gathered inside every call ("sync")."""
    import torch
    from t5vqvae_amd import synth
    from t5vqvae_amd.sharded import CodeShardedVectorQuantizerEMA
    n, k, d = (512, 1024, 64) if args.dry_run else (C5_TOKENS, C5_CODES, 768)
    kl = k // world
    w_full = synth.normal((k, d), 1000, 0, device=dev)
    x, _ = synth.clustered(n, w_full, 0.5, 2000)
    x = x.view(64, n // 64, d)
    cs_full = torch.full((k,), float(n) / k, device=dev)
    full = {"_embedding.weight": w_full, "_ema_w": w_full * cs_full.unsqueeze(1), "_ema_cluster_size": cs_full}

    def build(roles=None):
        m = CodeShardedVectorQuantizerEMA(k, d, 0.25, decay=0.99, disentanglement=roles is not None).to(dev)
        m.load_full_state_dict(full)
        m.encodings = "indices"
        m.search = args.search
        return m.train()

    def timed(m, srl, steps, warmup):
        with torch.no_grad():
            for _ in range(warmup):
                m(x, srl)
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                m(x, srl)
            barrier()
            el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
        return el / steps * 1e3

    mod = build()
    ms = timed(mod, None, args.steps, args.warmup)
    mod.exchange = "sync"
    ms_sync = timed(mod, None, max(3, args.steps // 2), 2)
    grouped = {}
    role_sets = {"8_roles_aligned_to_shards": [(i * (k // 8), (i + 1) * (k // 8)) for i in range(8)],
                 "5_unequal_roles_straddling_shards": [(0, k // 10), (k // 10, k // 3), (k // 3, k // 2 + 5),
                                                       (k // 2 + 5, k - k // 7), (k - k // 7, k)]}
    for name, roles in role_sets.items():
        m = build(roles)
        srl = synth.role_ranges(n, roles, 77, 0, device=dev).view(64, n // 64, 2)
        g_ms = timed(m, srl, max(3, args.steps // 2), 2)
        flops = 2.0 * d * sum(float((srl.view(-1, 2)[:, 0] == a).sum()) * (b - a) for a, b in roles)
        grouped[name] = {"ms_per_step": g_ms, "vectors_per_s": n / (g_ms * 1e-3), "algorithmic_flops": flops,
                         "frac_of_mfma_roofline": flops / (PEAK_FP32_MFMA_TFLOPS * 1e12 * world) / (g_ms * 1e-3)}
        del m
    if rank != 0:
        return None
    flops = 2.0 * n * k * d
    return {"metric": "quantized vectors/sec (fwd+EMA)", "value": n / (ms * 1e-3), "unit": "vectors/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C5 K={k} code-sharded {kl} codes/GPU, N={n} replicated tokens, D={d}, fwd+EMA",
                       "K": k, "codes_per_gpu": kl, "tokens": n, "D": d, "regime": "uniform", "search": args.search,
                       "parallelism": f"code-sharded x{world}",
                       "exchange": "all-reduce(MIN) of packed int64 keys [N]; gathered codebook kept and refreshed "
                                   "asynchronously after the update; 2-scalar all-reduce"},
            "step_frac_of_mfma_roofline": flops / (PEAK_FP32_MFMA_TFLOPS * 1e12 * world) / (ms * 1e-3),
            "codebook_exchange": {"prefetch_ms_per_step": ms, "sync_allgather_ms_per_step": ms_sync,
                                  "allgather_bytes": 4.0 * k * d},
            "grouped": grouped, "roofline": None, "cpu_baseline": None}


def run_rank(args):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and "RANK" in os.environ:
        args.gpus = world                      # under a launcher the launcher's world size is the truth
    dry = args.dry_run
    if dry:                                    # CPU rehearsal of the launch / rendezvous / all-reduce path
        dev = torch.device("cpu")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import fake_ops
        from t5vqvae_amd import vqvae
        vqvae.ops = fake_ops
        torch.cuda.synchronize = lambda *a, **kw: None
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # T5VQ_BENCH_FORCE_DIST=1 takes the RCCL path even at world size 1 (rehearsal on a one-GPU box)
    use_dist = world > 1 or os.environ.get("T5VQ_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    workload = args.workload
    if workload == "auto":
        workload = "c3" if args.gpus == 1 else "c4"
    if workload == "c5":
        if not use_dist:
            raise SystemExit("--workload c5 is the code-sharded form: run it with --gpus N > 1 "
                             "(or T5VQ_BENCH_FORCE_DIST=1 for a one-rank rehearsal)")
        result = run_c5(args, rank, world, dev, dist, barrier)
        if rank == 0:
            result["rccl_world_size"] = dist.get_world_size()
            result["dry_run"] = dry
            print(json.dumps(result), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if workload in ("c1", "c2") and world > 1:
        raise SystemExit(f"--workload {workload} is a single-GPU configuration")

    from t5vqvae_amd.sharded import shard_tokens
    n, k, d, update, desc = WORKLOADS[workload]
    strong = workload == "c4" and use_dist
    if strong:                                 # config 4 as written: 2^20 tokens in total, N/G per rank
        n = C4_TOTAL_TOKENS // world
        desc = f"C4 N=2^20 tokens in total ({n}/GPU) K=16384 D=768 token-sharded fused fwd+EMA"
    if dry:
        n, k, d = 256, 128, 64
        WORKLOADS[workload] = (n, k, d, update, desc)
    w, ema_w, cs, x = build_state(workload, args.regime, rank, world, dev, n=n, k=k)
    mod = make_module(k, d, w, ema_w, cs, update, args.search, dev)
    if use_dist:
        shard_tokens(mod)

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

    dist_info = None
    if use_dist:
        # the exchange step alone: K all-reduces of the statistics head [embed_sum | counts | loss_sum]
        buf = torch.zeros(k * d + k + 1, dtype=torch.float32, device=dev)
        for _ in range(2):
            dist.all_reduce(buf)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dist.all_reduce(buf)
        barrier()
        ar = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        dist_info = {"rccl_world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "allreduce_ms_per_step": float(ar) / args.steps * 1e3, "allreduce_bytes": 4.0 * (k * d + k + 1)}

        # The exchange inside the step, both forms, and the step with no exchange at all (same module, same tokens, the
        # statistics left local): what the all-reduce leaves exposed = step - step without exchange.
        def timed_steps(m, steps):
            with torch.no_grad():
                m._embedding.weight.copy_(w), m._ema_w.copy_(ema_w), m._ema_cluster_size.copy_(cs)
                for _ in range(2):
                    m(x)
                barrier()
                t0 = time.perf_counter()
                for _ in range(steps):
                    m(x)
                barrier()
                el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            return float(el) / steps * 1e3
        mod.stats_exchange = "overlap"
        ms_overlap = timed_steps(mod, args.steps)
        mod.stats_exchange = "serial"
        group = mod.process_group
        mod.process_group = None                      # local statistics only: the step's kernels without the exchange
        ms_local = timed_steps(mod, args.steps)
        mod.process_group = group
        dist_info["exchange"] = {
            "serial_ms_per_step": ms_per_step, "overlap_ms_per_step": ms_overlap, "no_exchange_ms_per_step": ms_local,
            "allreduce_exposed_ms_serial": ms_per_step - ms_local, "allreduce_exposed_ms_overlap": ms_overlap - ms_local,
            "note": "serial = forward -> all-reduce -> finalize on one stream (module default, the headline); overlap = "
                    "statistics pass -> asynchronous all-reduce under the output pass -> finalize (one more read of x)"}
        if strong and world > 1 and not dry:
            # the SAME workload on one GPU (rank 0 alone, the others wait): the reference point of strong scaling
            if rank == 0:
                w1, e1_, c1_, x1 = build_state("c4", args.regime, 0, 1, dev, n=C4_TOTAL_TOKENS, k=k)
                m1 = make_module(k, d, w1, e1_, c1_, update, args.search, dev)
                with torch.no_grad():
                    m1(x1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        m1(x1)
                    torch.cuda.synchronize()
                    one_ms = (time.perf_counter() - t0) / 3 * 1e3
                del m1, w1, e1_, c1_, x1
                one = C4_TOTAL_TOKENS / (one_ms * 1e-3)
                dist_info["single_gpu_same_workload"] = {"ms_per_step": one_ms, "value": one, "unit": "vectors/s",
                                                         "steps": 3}
                dist_info["speedup_vs_single_gpu"] = value / one
                dist_info["parallel_efficiency_vs_single_gpu"] = value / one / world
            dist.barrier()

    result = None
    if rank == 0:
        flops = 2.0 * n * k * d
        result = {
            "metric": "quantized vectors/sec (fwd+EMA)" if update else "quantized vectors/sec (fwd)",
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "tokens_per_gpu": n, "global_tokens": n * world, "K": k, "D": d,
                       "regime": args.regime, "search": args.search,
                       "parallelism": f"token-sharded x{world}" if use_dist else "single GPU",
                       "encodings": "indices (dense one-hot excluded, SURVEY 8b)",
                       "like_for_like_across_n_gpus": (
                           "this line's top-level value is config 4 (2^20 tokens in total, strong scaling); compare it with "
                           "single_gpu_same_workload.value (speedup_vs_single_gpu, parallel_efficiency_vs_single_gpu) and "
                           "with series.c4_strong of the --gpus 1 line; series.c3_weak is the config-3 point to hold "
                           "against the --gpus 1 headline") if strong else (
                           "config 3, 65536 tokens per GPU (weak scaling): compare with series.c3_weak of the --gpus N "
                           "lines; series.c4_strong is the one-GPU point of the --gpus N headline (config 4)")},
            "roofline": None if dry else search_kernel_roofline(mod, x, n, k, d, args, workload),
            "step_frac_of_mfma_roofline": (flops / PEAK_FP32_MFMA_TFLOPS / 1e12) / (ms_per_step * 1e-3),
            "step_algorithmic_bytes": step_bytes(n, k, d, update),
            "step_frac_of_hbm_roofline": step_bytes(n, k, d, update) / (PEAK_HBM_GBS * 1e9) / (ms_per_step * 1e-3),
            "dry_run": dry,
        }
        if dist_info:
            result.update(dist_info)
    # Both multi-GPU series in every line of a default (--workload auto) run: the one-GPU line adds the config-4 point
    # (the whole 2^20 tokens on one GPU), the N-GPU lines add the config-3 weak point -- a reader (or the driver) can then
    # compare like with like across N whichever of the two it follows.
    series = None
    if args.workload == "auto" and not dry:
        if world == 1 and not use_dist:
            series = {"c4_strong": series_point("c4", C4_TOTAL_TOKENS, 16384, d, args, 0, 1, dev, None, barrier, 3)}
        elif use_dist:
            series = {"c3_weak": series_point("c3", 65536, 8192, d, args, rank, world, dev, dist, barrier,
                                              args.steps)}
    if rank == 0 and series:
        result["series"] = series
        result["series_note"] = ("config 3 (this line's headline at --gpus 1) scales WEAKLY: 65536 tokens per GPU; "
                                 "config 4 (the headline at --gpus N > 1) scales STRONGLY: 2^20 tokens in total. "
                                 "Compare c3 points with c3 points and c4 points with c4 points across N.")
    if rank == 0 and world == 1 and not dry:
        med, mn, _ = per_call_ms(mod, x, w, ema_w, cs, args.config_iters)
        result["per_call"] = {"ms_median": med, "ms_min": mn, "protocol": "HIP events around one call, state reset "
                              "outside the timed region", "frac_of_mfma_roofline":
                              (flops / PEAK_FP32_MFMA_TFLOPS / 1e12) / (med * 1e-3)}
        result["other_search"] = other_search(mod, x, w, ema_w, cs, args, n, k, d)
    if use_dist:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not dry:
            result["cpu_baseline"] = cpu_leg(k, d, update, args.search, (w, ema_w, cs, x), dev, 8192, 20.0, desc)
        else:
            result["cpu_baseline"] = None
        if world == 1 and not dry and not args.no_configs and not use_dist:
            del mod
            result["configs"] = single_gpu_configs(args, dev)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", choices=["auto", "c1", "c2", "c3", "c4", "c5"],
                    help="auto: c3 on one GPU (BASELINE config 3, the headline), c4 strong scaling on several")
    ap.add_argument("--regime", default="uniform", choices=["uniform", "cold", "collapsed"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration table of the one-GPU run")
    ap.add_argument("--config-iters", type=int, default=20, help="timed calls per entry of the per-call protocol")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU + gloo + a test stand-in for the kernels: exercises the launcher and the collectives only")
    ap.add_argument("--search", default="scan", choices=["scan", "filter", "auto"],
                    help="how the module finds the nearest code for the headline number: 'scan' = the fused fp32 MFMA "
                         "kernel (the north star's design, default); 'filter'/'auto' = fp16 filter + fp32 decision "
                         "(identical indices); the other mode is always reported beside it")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
