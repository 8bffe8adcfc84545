"""The single-launch small-shape forward (t5vq_vq_forward_small_f32) against the three-launch chain
(t5vq_vq_forward_f32 + t5vq_vq_ema_finalize_f32): bit-identity of every output, then kernel time (HIP events around
back-to-back calls through ctypes).   python tools/bench_small.py [N K D] ..."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from t5vqvae_amd import ops, synth

DEV = torch.device("cuda", 0)


def chain(x, w, ee, want_stats):
    n, d = x.shape
    k = w.shape[0]
    o = dict(ste=torch.empty_like(x), diff=torch.empty_like(x), idx=torch.empty(n, dtype=torch.int64, device=DEV),
             counts=torch.empty(k, device=DEV))
    if want_stats:
        o["embed_sum"] = torch.empty(k, d, device=DEV)
    terms = torch.empty(ops.forward_loss_terms(n, want_stats), device=DEV)
    ops.vq_forward(x, w, None, "mse", ee=ee, loss_terms=terms, **o)
    sc = torch.empty(4, device=DEV)
    ops.ema_finalize(None, None, None, o["counts"], None, terms, n, float(n * d), 0.99, 0.01, 1e-5, k * 1e-5, 0.25,
                     False, False, sc)
    o["terms"], o["scalars"] = terms, sc
    return o


def small(x, w, ee, state, want_stats, fused_scalars=True):
    n, d = x.shape
    k = w.shape[0]
    o = dict(ste=torch.empty_like(x), diff=torch.empty_like(x), idx=torch.empty(n, dtype=torch.int64, device=DEV),
             counts=torch.empty(k, device=DEV))
    if want_stats:
        o["embed_sum"] = torch.empty(k, d, device=DEV)
    terms = torch.empty(n, device=DEV)
    sc = torch.empty(4, device=DEV) if fused_scalars and not want_stats else None
    ops.vq_forward_small(x, w, state, ee=ee, loss_terms=terms, scalars=sc, loss_denom=float(n * d),
                         commitment_cost=0.25, **o)
    if sc is None:
        sc = torch.empty(4, device=DEV)
        ops.ema_finalize(None, None, None, o["counts"], None, terms, n, float(n * d), 0.99, 0.01, 1e-5, k * 1e-5,
                         0.25, False, False, sc)
    o["terms"], o["scalars"] = terms, sc
    return o


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    shapes = [(4096, 512, 768)]
    if len(sys.argv) > 1:
        v = [int(a) for a in sys.argv[1:]]
        shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
    for n, k, d in shapes:
        assert ops.small_supported(n, k, d), (n, k, d)
        w = synth.normal((k, d), 1000, 0, device=DEV)
        x, _ = synth.clustered(n, w.cpu(), 0.5, 2000)
        x = x.to(DEV)
        ee = ops.code_sqnorms(w)
        state = ops.small_state(k, DEV)
        for want_stats in (False, True):
            a = chain(x, w, ee, want_stats)
            for rep in range(3):                       # repeated: the state block must come back zeroed
                b = small(x, w, ee, state, want_stats)
                torch.cuda.synchronize()
                bad = [name for name in ("idx", "ste", "diff", "counts", "terms", "scalars")
                       if not torch.equal(a[name], b[name])]
                if want_stats:
                    err = float((a["embed_sum"].double() - b["embed_sum"].double()).norm() / a["embed_sum"].double().norm())
                    if err > 1e-6:
                        bad.append(f"embed_sum {err:.2e}")
                print(f"N={n} K={k} D={d} stats={want_stats} rep={rep}: " + ("IDENTICAL" if not bad else f"DIFFERS: {bad}"))
                if bad:
                    for name in bad[:3]:
                        if name in a:
                            ne = (a[name] != b[name]).nonzero()
                            print("   ", name, "first diffs at", ne[:5].flatten().tolist(), a[name].flatten()[:4].tolist(),
                                  b[name].flatten()[:4].tolist())
            assert float(state.float().abs().sum()) == 0.0, "state block not left zeroed"
        flops = 2.0 * n * k * d
        # kernel time without the host: everything preallocated, the raw C-ABI call in the loop
        import ctypes
        from t5vqvae_amd import _lib
        lib = _lib.load()
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        o = dict(ste=torch.empty_like(x), idx=torch.empty(n, dtype=torch.int64, device=DEV))
        terms, sc = torch.empty(n, device=DEV), torch.empty(4, device=DEV)
        nbytes = ops.forward_workspace_bytes(n, k)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        counts = torch.empty(k, device=DEV)

        def raw_small(scalars=True):
            lib.t5vq_vq_forward_small_f32(P(x), P(w), None, P(ee), n, k, d, P(ws), nbytes, P(state), P(o["ste"]), None,
                                          None, P(o["idx"]), None if scalars else P(counts), None, P(terms),
                                          P(sc) if scalars else None, float(n * d), 0.25, 0, stream)

        def raw_chain():
            lib.t5vq_vq_forward_f32(P(x), P(w), None, None, P(ee), n, k, d, 0, P(ws), nbytes, P(o["ste"]), None, None,
                                    P(o["idx"]), P(counts), None, P(terms), None, stream)
            lib.t5vq_vq_ema_finalize_f32(None, None, None, P(counts), None, P(terms), n, k, d, n, float(n * d), 0.99,
                                         0.01, 1e-5, k * 1e-5, 0.25, 0, 0, P(sc), None, stream)

        for name, fn in (("raw chain (3 launches)", raw_chain), ("raw small, scalars fused", raw_small),
                         ("raw small, no scalars (+memset of counts)", lambda: raw_small(False))):
            us = timed(fn, 500)
            print(f"N={n} K={k} D={d} {name}: {us:.1f} us per call back to back, {flops / us / 1e6 / 157.3:.3f} of fp32 MFMA peak")
        for name, fn in (("chain (3 launches)", lambda: chain(x, w, ee, False)),
                         ("small, scalars fused (1 launch)", lambda: small(x, w, ee, state, False)),
                         ("small + finalize (2 launches)", lambda: small(x, w, ee, state, False, False))):
            us = timed(fn)
            print(f"N={n} K={k} D={d} {name}: {us:.1f} us per call back to back, {flops / us / 1e6 / 157.3:.3f} of fp32 MFMA peak")


if __name__ == "__main__":
    main()
