"""Time the filtered search of a variant build of the library (A/B of kernel edits in one GPU call).

    python3 tools/experiments/filter_variant.py path/to/libt5vq_variant.so [more.so ...]
One process per library (the binding loads one library per process): the parent starts itself once per path.
Diagnostic builds of t5vq_filter.hip: -DT5VQ_AB_NOEPI (no epilogue), -DT5VQ_AB_NOSLOW (no per-element path in the
epilogue); their results are wrong
by construction, only tools/experiments/ab_prof.sh's kernel time means anything.
"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(path):
    sys.path.insert(0, ROOT)
    import torch
    from t5vqvae_amd import _lib
    _lib.LIB_PATH = os.path.abspath(path)
    from t5vqvae_amd import ops, synth
    dev = torch.device("cuda", 0)
    n, k, d = 65536, 8192, 768
    w = synth.normal((k, d), 21, 0, device=dev)
    x, _ = synth.clustered(n, w, 0.5, 22)
    x = x.to(dev)
    ref = ops.dist_argmin(x, w, *ops.row_sqnorms(x, w))
    ts = []
    for i in range(25):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        keys = ops.dist_argmin_filtered(x, w, exact_keys=False)
        e1.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1) * 1e3)
    same = bool(((keys & 0xFFFFFFFF) == (ref & 0xFFFFFFFF)).all())
    for regime in ("uniform", "cold"):
        import bench
        wr, _, _, xr = bench.build_state("c3", regime, 0, 1, dev)
        _, ovf = ops.dist_argmin_filtered(xr.reshape(-1, d), wr, exact_keys=False, want_overflow=True)
        print(f"    {regime}: tokens sent to the exact scan {int(ovf)}", flush=True)
    print(f"{os.path.basename(path):32s} filtered search (index-only) {statistics.median(ts):8.1f} us "
          f"(min {min(ts):8.1f}); indices == scan: {same}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for p in sys.argv[1:]:
            rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", p], timeout=150).returncode
            if rc != 0:
                print(f"{p}: exit {rc}; stopping")
                sys.exit(rc)
