"""Cost of module.deterministic = True (fixed-point, order-independent EMA statistics) at C3 / C1.

    python3 tools/bench_deterministic.py"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t5vqvae_amd import VectorQuantizerEMA  # noqa: E402

dev = torch.device("cuda", 0)
for name in ("c3", "c1"):
    n, k, d, update, _ = bench.WORKLOADS[name]
    w, ema_w, cs, x = bench.build_state(name, "uniform", 0, 1, dev)
    for search in ("scan", "filter"):
        for det in (False, True):
            m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev).train()
            with torch.no_grad():
                m._embedding.weight.copy_(w), m._ema_w.copy_(ema_w), m._ema_cluster_size.copy_(cs)
                m.encodings, m.search, m.deterministic = "indices", search, det
                for _ in range(3):
                    m(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    m(x)
                torch.cuda.synchronize()
            print(f"{name} search={search:6s} deterministic={det!s:5s} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms/step",
                  flush=True)
