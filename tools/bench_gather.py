"""Time t5vq_gather_stats_f32 alone (HIP events, median of 20) over assignment patterns.

    python3 tools/bench_gather.py
uniform: every token its own random code; hot21 / hot1: all tokens on 21 / 1 codes (collapsed codebook);
runs8: runs of 8 consecutive tokens share a code (clustered batches)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t5vqvae_amd import ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for n, k, d, update in ((65536, 8192, 768, True), (2880, 10000, 768, True), (4096, 512, 768, False),
                            (131072, 16384, 768, True)):
        x = synth.normal((n, d), 1, 0, device=dev)
        w = synth.normal((k, d), 2, 0, device=dev)
        pats = {"uniform": synth.uniform_int(n, k, 3, 0, dev),
                "runs8": synth.uniform_int(n, k, 3, 1, dev)[(torch.arange(n, device=dev) // 8)],
                "hot21": synth.uniform_int(n, 21, 3, 2, dev) * (k // 21),
                "hot1": torch.full((n,), 5, dtype=torch.int64, device=dev)}
        ste = torch.empty_like(x)
        idx = torch.empty(n, dtype=torch.int64, device=dev)
        counts = torch.empty(k, device=dev)
        es = torch.empty(k, d, device=dev) if update else None
        part = torch.empty(ops.gather_blocks(n), device=dev)
        for name, codes in pats.items():
            keys = codes.to(torch.int64)          # distance word 0, index word = code
            ts, ts_sort, ts_ord = [], [], []
            part_tok = torch.empty(n, device=dev)
            for i in range(25):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e2.record()
                perm = ops.order_by_code(keys, k)
                e3.record()
                ops.gather_stats(x, w, keys, "mse", ste=ste, idx=idx, counts=counts, embed_sum=es,
                                 loss_partial=part_tok, perm=perm)
                e0.record()
                ops.gather_stats(x, w, keys, "mse", ste=ste, idx=idx, counts=counts, embed_sum=es, loss_partial=part)
                e1.record()
                torch.cuda.synchronize()
                if i >= 5:
                    ts.append(e0.elapsed_time(e1) * 1e3)
                    ts_sort.append(e2.elapsed_time(e3) * 1e3)
                    ts_ord.append(e3.elapsed_time(e0) * 1e3)
            assert torch.equal(idx, codes)
            ref = torch.bincount(codes, minlength=k).float()
            assert torch.equal(counts, ref)
            if update:
                chk = torch.zeros(k, d, device=dev).index_add_(0, codes, x)
                err = float((es - chk).norm() / chk.norm())
                assert err < 1e-5, err
            mb = (2 * n * d + (n * d if update else 0)) * 4 / 1e6
            print(f"N={n:7d} K={k:6d} {'fwd+stats' if update else 'fwd      '} {name:8s} "
                  f"{statistics.median(ts):8.1f} us (min {min(ts):7.1f}) | ordered: sort {statistics.median(ts_sort):6.1f} + "
                  f"gather {statistics.median(ts_ord):7.1f} us  x+ste{'+atomics' if update else ''} {mb:7.1f} MB",
                  flush=True)


if __name__ == "__main__":
    main()
