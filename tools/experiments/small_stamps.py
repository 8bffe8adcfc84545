"""Where a workgroup of the single-launch small-shape forward spends its time: a diagnostic build
(-DT5VQ_SMALL_STAMPS, tools/experiments/build_small_stamps.sh -> /tmp or gpurun_out) stamps the wall clock (100 MHz)
at its phase borders.   T5VQ_LIB=.../libt5vq_stamps.so python tools/experiments/small_stamps.py [N K D]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from t5vqvae_amd import _lib, ops, synth

DEV = torch.device("cuda", 0)
n, k, d = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (4096, 512, 768)
scalars_on = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
lib = _lib.load()
lib.t5vq_small_set_stamps.argtypes = [ctypes.c_void_p]
w = synth.normal((k, d), 1000, 0, device=DEV)
x, _ = synth.clustered(n, w.cpu(), 0.5, 2000)
x = x.to(DEV)
ee = ops.code_sqnorms(w)
state = ops.small_state(k, DEV)
blocks = 4096
stamps = torch.zeros(blocks, 8, dtype=torch.int64, device=DEV)
lib.t5vq_small_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
ste = torch.empty_like(x)
idx = torch.empty(n, dtype=torch.int64, device=DEV)
terms = torch.empty(n, device=DEV)
sc = torch.empty(4, device=DEV) if scalars_on else None
counts = None if scalars_on else torch.empty(k, device=DEV)
b2b = int(os.environ.get("B2B", "1"))     # calls queued back to back before the one that is read (clocks up, caches warm)
for it in range(3):
    stamps.zero_()
    torch.cuda.synchronize()
    for _ in range(b2b):
        ops.vq_forward_small(x, w, state, ee=ee, ste=ste, idx=idx, loss_terms=terms, scalars=sc, counts=counts,
                             loss_denom=float(n * d), commitment_cost=0.25)
    torch.cuda.synchronize()
st = stamps.cpu()
used = st[:, 0] > 0
st = st[used]
t0 = int(st[:, 0].min())
names = ["start", "prologue done", "k loop done", "epilogue+merge", "exchange done (last arrivers)", "gather done",
         "last of all: ticket+acquire", "scalars written"]
print(f"N={n} K={k} D={d} b2b={b2b}: {int(used.sum())} workgroups; times in us since the first workgroup started")
for i, name in enumerate(names):
    col = st[:, i]
    have = col > 0
    if not bool(have.any()):
        continue
    v = (col[have] - t0).double() / 100.0
    print(f"  {name:32s} n={int(have.sum()):4d}  min {float(v.min()):7.2f}  median {float(v.median()):7.2f}  max {float(v.max()):7.2f}")
dur = (st[:, 2] - st[:, 1]).double() / 100.0
print(f"  k loop per workgroup: min {float(dur.min()):.2f} median {float(dur.median()):.2f} max {float(dur.max()):.2f} us "
      f"(matrix work {2.0 * n * k * d / 157.3e6 :.2f} us at peak over {int(used.sum())} workgroups -> ideal per workgroup "
      f"{2.0 * 32 * (k / (int(used.sum()) / ((n + 31) // 32))) * d / (157.3e6 / 256):.2f} us)")
