"""Host-side cost of one module call against its GPU time, eager and replayed from a HIP graph.

    python tools/host_overhead.py [c1|c2|c3]

Prints per workload: isolated call latency (GPU idle at the start, HIP events), back-to-back throughput time per
call (queue kept full), the host enqueue time per call, and the same call captured once into a HIP graph
(torch.cuda.CUDAGraph over the current stream -- the C ABI only enqueues on the stream it is given) and replayed."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t5vqvae_amd import VectorQuantizerEMA  # noqa: E402


def main():
    names = sys.argv[1:] or ["c2", "c1", "c3"]
    dev = torch.device("cuda", 0)
    for name in names:
        n, k, d, update, desc = bench.WORKLOADS[name]
        w, ema_w, cs, x = bench.build_state(name, "uniform", 0, 1, dev)
        mod = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev)
        with torch.no_grad():
            mod._embedding.weight.copy_(w); mod._ema_w.copy_(ema_w); mod._ema_cluster_size.copy_(cs)
        mod.encodings = "indices"
        mod.train(update)
        reps = 50
        with torch.no_grad():
            for _ in range(5):
                mod(x)
            torch.cuda.synchronize()
            # isolated latency
            lat = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record(); mod(x); e1.record()
                torch.cuda.synchronize()
                lat.append(e0.elapsed_time(e1))
            # back to back
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                mod(x)
            t_host = (time.perf_counter() - t0) / reps
            torch.cuda.synchronize()
            t_b2b = (time.perf_counter() - t0) / reps
            # graph
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    mod(x)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                out = mod(x)
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            t_graph = (time.perf_counter() - t0) / reps
        lat.sort()
        print(f"{name}: isolated {lat[len(lat)//2]*1e3:8.1f} us | back-to-back {t_b2b*1e6:8.1f} us | "
              f"host enqueue {t_host*1e6:8.1f} us | graph replay {t_graph*1e6:8.1f} us | loss {float(out[0]):.6f}",
              flush=True)


if __name__ == "__main__":
    main()
