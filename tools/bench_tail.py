"""Encoder tail + VQ: eager torch T5LayerNorm + dropout feeding the quantizer vs the fused prologue.

    python3 tools/bench_tail.py [c1|c3]
HIP events around the whole call (tail + VQ training step), median of 20, dropout mask drawn inside both."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t5vqvae_amd import VectorQuantizerEMA  # noqa: E402
from t5vqvae_amd.encoder_tail import EncoderTailVQ  # noqa: E402


class EagerT5LayerNorm(torch.nn.Module):      # the reference's op sequence (modeling_t5.py:200-207), on the GPU
    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(d))
        self.variance_epsilon = eps

    def forward(self, x):
        variance = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x / torch.sqrt(variance + self.variance_epsilon)
        return self.weight * x


def timed(fn, iters=25):
    ts = []
    for i in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


def main():
    dev = torch.device("cuda", 0)
    for name in (sys.argv[1:] or ["c1", "c3"]):
        n, k, d, update, desc = bench.WORKLOADS[name]
        w, ema_w, cs, x = bench.build_state(name, "uniform", 0, 1, dev)
        h = x * 3.0

        def make():
            m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(dev)
            with torch.no_grad():
                m._embedding.weight.copy_(w), m._ema_w.copy_(ema_w), m._ema_cluster_size.copy_(cs)
            m.encodings = "indices"
            return m.train(update)
        vq_a, vq_b = make(), make()
        ln, drop = EagerT5LayerNorm(d).to(dev), torch.nn.Dropout(0.1)
        tail = EncoderTailVQ(vq_b, hidden_size=d, dropout_rate=0.1).to(dev).train()
        with torch.no_grad():
            t_tail_eager = timed(lambda: drop(ln(h)))
            t_eager = timed(lambda: vq_a(drop(ln(h))))
            t_fused = timed(lambda: tail(h))
            t_vq = timed(lambda: vq_a(x))
        print(f"{name}: eager tail alone {t_tail_eager:8.1f} us | eager tail + VQ {t_eager:8.1f} us | fused tail + VQ "
              f"{t_fused:8.1f} us | VQ alone {t_vq:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
