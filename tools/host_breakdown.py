"""Where the host time of one module call goes: raw op, module wrapper, single launches.

    python tools/host_breakdown.py [c2|c1]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t5vqvae_amd import VectorQuantizerEMA, ops  # noqa: E402


def host_us(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    return t * 1e6


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    dev = torch.device("cuda", 0)
    n, k, d, update, _ = bench.WORKLOADS[name]
    w, ema_w, cs, x = bench.build_state(name, "uniform", 0, 1, dev)
    mod = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev)
    with torch.no_grad():
        mod._embedding.weight.copy_(w); mod._ema_w.copy_(ema_w); mod._ema_cluster_size.copy_(cs)
    mod.encodings = "indices"
    mod.search = "scan"
    mod.train(update)
    flat = x.reshape(-1, d)
    wd = mod._embedding.weight.detach()
    ee = ops.code_sqnorms(wd)
    hyper = dict(loss_denom=float(n * d), decay=0.99, one_minus_decay=0.01, epsilon=1e-5, k_epsilon=k * 1e-5,
                 commitment_cost=0.25, plain_variant=False)
    with torch.no_grad():
        print(f"{name}: module call                {host_us(lambda: mod(x)):7.1f} us host")
        print(f"{name}: ops.fused_step             "
              f"{host_us(lambda: ops.fused_step(flat, wd, None, 'mse', ee=ee, update=False, **hyper)):7.1f} us host")
        xx = torch.empty(n, device=dev)
        print(f"{name}: one launch via ctypes      {host_us(lambda: ops.row_sqnorms(flat, None)):7.1f} us host")
        print(f"{name}: one launch via torch op    {host_us(lambda: ops.code_sqnorms(wd)):7.1f} us host")
        print(f"{name}: torch.empty x6             "
              f"{host_us(lambda: [torch.empty(16, device=dev) for _ in range(6)]):7.1f} us host")
        a = torch.empty(1024, device=dev)
        print(f"{name}: aten add_ (one launch)     {host_us(lambda: a.add_(1.0)):7.1f} us host")


if __name__ == "__main__":
    main()
