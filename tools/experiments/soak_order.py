"""Soak of the role-ordered search: random shapes and range tables, keys with and without the visiting order, scan and
filtered search, and the composite forward (which orders by itself) against the fine-grained chain.

    python3 tools/experiments/soak_order.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from t5vqvae_amd import ops, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(1234)
dev = torch.device("cuda", 0)
bad = 0
for r in range(rounds):
    n = int(torch.randint(1024, 20000, (1,), generator=g))
    k = int(torch.randint(3, 6000, (1,), generator=g))
    d = int(torch.randint(1, 6, (1,), generator=g)) * 64
    n_roles = int(torch.randint(1, 12, (1,), generator=g))
    cuts = torch.sort(torch.randint(0, k + 1, (n_roles + 1,), generator=g)).values.tolist()
    roles = [(cuts[i], cuts[i + 1]) for i in range(n_roles)] + [(0, k), (k // 3, k // 3)]
    x = synth.normal((n, d), 500 + r, 0, device=dev)
    w = synth.normal((k, d), 500 + r, 1, device=dev)
    rg = synth.role_ranges(n, roles, 500 + r, 2, device=dev)
    xx, ee = ops.row_sqnorms(x, w)
    order = ops.order_by_range(rg, k)
    ref = ops.dist_argmin(x, w, xx, ee, rg)
    a = ops.dist_argmin(x, w, xx, ee, rg, order=order)
    b = ops.dist_argmin_filtered(x, w, xx, ee, ranges=rg)
    c = ops.dist_argmin_filtered(x, w, xx, ee, ranges=rg, order=order)
    ok = torch.equal(ref, a) and torch.equal(ref, b) and torch.equal(ref, c)
    # composite forward (orders the tokens itself) against gather over the reference keys
    ste = torch.empty_like(x)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    ops.vq_forward(x, w, rg, "mse", ste=ste, idx=idx)
    ste2 = torch.empty_like(x)
    idx2 = torch.empty(n, dtype=torch.int64, device=dev)
    ops.gather_stats(x, w, ref, "mse", ste=ste2, idx=idx2)
    ok = ok and torch.equal(idx, idx2) and torch.equal(ste, ste2)
    if d % 64 == 0:
        ops.vq_forward_filtered(x, w, rg, "mse", ste=ste, idx=idx)
        ok = ok and torch.equal(idx, idx2) and torch.equal(ste, ste2)
    if not ok:
        bad += 1
        print(f"round {r}: MISMATCH n={n} k={k} d={d} roles={roles}", flush=True)
    elif r % 10 == 0:
        print(f"round {r}: ok (n={n} k={k} d={d}, {len(roles)} roles)", flush=True)
print(f"{rounds} rounds, {bad} mismatching")
sys.exit(1 if bad else 0)
