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
        vq_a, vq_b, vq_c = make(), make(), make()
        vq_b.code_image_cache = "always"      # the update writes the codebook's image too (default: evaluation only)
        vq_c.code_image_cache = False
        ln, drop = EagerT5LayerNorm(d).to(dev), torch.nn.Dropout(0.1)
        tail = EncoderTailVQ(vq_b, hidden_size=d, dropout_rate=0.1).to(dev).train()
        tail_c = EncoderTailVQ(vq_c, hidden_size=d, dropout_rate=0.1).to(dev).train()   # no images from the producers
        tail_c._wants_image = lambda *a: False
        tail.token_image_min_elements = 0
        with torch.no_grad():
            t_tail_eager = timed(lambda: drop(ln(h)))
            t_eager = timed(lambda: vq_a(drop(ln(h))))
            t_fused = timed(lambda: tail(h))
            t_fused_c = timed(lambda: tail_c(h))
            t_vq = timed(lambda: vq_b(x))
            t_vq_c = timed(lambda: vq_c(x))
            vq_a.eval(), vq_c.eval()
            t_ev, t_ev_c = timed(lambda: vq_a(x)), timed(lambda: vq_c(x))
        print(f"{name} (search={vq_a.search}): eager tail alone {t_tail_eager:8.1f} us | eager tail + VQ {t_eager:8.1f} us | "
              f"fused tail + VQ {t_fused:8.1f} us (without fp16 images from the producers {t_fused_c:8.1f}) | VQ step alone, "
              f"code image written by the update {t_vq:8.1f} us (no kept image {t_vq_c:8.1f}) | evaluation call, kept code "
              f"image {t_ev:8.1f} us (none {t_ev_c:8.1f})", flush=True)


if __name__ == "__main__":
    main()
