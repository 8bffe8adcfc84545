"""GPU checks of the pieces the multi-GPU modes rely on.

* single process, through the C ABI: codebook shards processed one after another with ``code_offset`` and
  ``merge`` must give the unsharded keys bit for bit; statistics restricted to a code slice must equal the
  slice of the full statistics; the two-phase cluster update (modes 2 + 3) must equal the fused mode 1;
* two processes on the one GPU of the box over gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL
  run is the driver's): token-sharded and code-sharded modules against the CPU oracle.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vq_oracle as vo
from t5vqvae_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


@pytest.mark.parametrize("ranged", [False, True])
def test_shard_merge_equals_unsharded_keys(ranged):
    from t5vqvae_amd import ops
    n, k, d, g = 3000, 4096, 768, 4
    x = synth.normal((n, d), 71, 0).cuda()
    w = synth.normal((k, d), 71, 1).cuda()
    rg = synth.role_ranges(n, [(0, 1024), (1000, 3000), (2047, 2049), (0, 4096)], 72).cuda() if ranged else None
    xx, ee = ops.row_sqnorms(x, w)
    full = ops.dist_argmin(x, w, xx, ee, rg)
    keys = None
    for s in range(g):
        ws = w[s * (k // g):(s + 1) * (k // g)].contiguous()
        _, ees = ops.row_sqnorms(None, ws)
        keys = ops.dist_argmin(x, ws, xx, ees, rg, code_offset=s * (k // g), keys=keys, merge=s > 0)
    assert torch.equal(keys, full)


def test_c5_full_size_shards_equal_unsharded():
    """BASELINE config 5 at full size on one GPU: K = 65536 as 8 shards of 8192 codes processed one after another
    (code_offset + merge) must give the keys of the unsharded call bit for bit -- global argmin and the grouped
    (8 role ranges aligned to the shards, and 5 unequal ranges that straddle them) forms."""
    from t5vqvae_amd import ops
    n, k, d, g = 65536, 65536, 768, 8
    x = synth.normal((n, d), 81, 0, device="cuda")
    w = synth.normal((k, d), 81, 1, device="cuda")
    xx, ee = ops.row_sqnorms(x, w)
    aligned = [(i * 8192, (i + 1) * 8192) for i in range(8)]
    straddle = [(0, 20000), (20000, 20001), (20001, 40000), (40000, 65536), (8000, 60000)]
    for roles in (None, aligned, straddle):
        rg = synth.role_ranges(n, roles, 82, device="cuda") if roles else None
        full = ops.dist_argmin(x, w, xx, ee, rg)
        keys = None
        for s in range(g):
            ws = w[s * (k // g):(s + 1) * (k // g)]
            _, ees = ops.row_sqnorms(None, ws)
            keys = ops.dist_argmin(x, ws, xx, ees, rg, code_offset=s * (k // g), keys=keys, merge=s > 0)
        assert torch.equal(keys, full)
        idx = ops.unpack_keys(full)[0]
        if roles:
            width = (rg[:, 1] - rg[:, 0])
            assert bool(((idx >= 0) & (idx < width)).all())        # slice-relative, inside the role's width
        else:
            assert bool(((idx >= 0) & (idx < k)).all())
        # the filtered search at full size gives the same indices, and both agree with the oracle (distances + the
        # reference's slice-relative argmin, vqvae/vqvae.py:110-122) on a slice the CPU finishes in seconds
        order = ops.order_by_range(rg, k) if roles else None
        filt = ops.dist_argmin_filtered(x, w, xx=xx, ee=ee, ranges=rg, order=order, exact_keys=False)
        assert torch.equal(ops.unpack_keys(filt)[0], idx)
        rows = 1024
        xs, wc = x[:rows].cpu(), w.cpu()
        rs = rg[:rows].cpu() if roles else None
        ref = vo.select_codes(vo.distances(xs, wc), rs)
        outside, inside = vo.classify_mismatches(xs, wc, idx[:rows].cpu(), ref, rs)
        assert outside == 0, f"{roles and len(roles)} roles: {outside} mismatches outside the band ({inside} inside)"


def test_stats_slice_and_two_phase_cluster_update():
    from t5vqvae_amd import ops
    n, k, d = 5000, 1024, 768
    x = synth.normal((n, d), 73, 0).cuda()
    w = synth.normal((k, d), 73, 1).cuda()
    xx, ee = ops.row_sqnorms(x, w)
    keys = ops.dist_argmin(x, w, xx, ee)
    counts = torch.empty(k, device="cuda")
    es = torch.empty(k, d, device="cuda")
    ops.gather_stats(x, w, keys, counts=counts, embed_sum=es)
    lo, cnt = 256, 512
    c2 = torch.empty(cnt, device="cuda")
    e2 = torch.empty(cnt, d, device="cuda")
    ops.gather_stats(x, w, keys, counts=c2, embed_sum=e2, stats_begin=lo, stats_count=cnt)
    assert torch.equal(c2, counts[lo:lo + cnt])
    assert rel(e2, es[lo:lo + cnt]) <= 1e-6          # fp32 atomics: same terms, order may differ
    # fused update vs two-phase update over two halves
    cs0 = torch.rand(k, device="cuda") * 3
    loss_sum = torch.tensor([123.0], device="cuda")
    args = (n, float(n * d), 0.99, 1 - 0.99, 1e-5, k * 1e-5, 0.25, False)
    cs_f = cs0.clone()
    sc_f = torch.empty(4, device="cuda")
    ops.ema_cluster(cs_f, counts, loss_sum, k, *args, 1, sc_f)
    halves, scs = [cs0[:k // 2].clone(), cs0[k // 2:].clone()], []
    for h in range(2):
        sc = torch.empty(4, device="cuda")
        ops.ema_cluster(halves[h], counts[h * k // 2:(h + 1) * k // 2].contiguous(), loss_sum, k // 2, *args, 2, sc)
        scs.append(sc)
    tot = scs[0][1:3] + scs[1][1:3]
    for h in range(2):
        scs[h][1:3] = tot
        ops.ema_cluster(halves[h], None, loss_sum, k // 2, *args, 3, scs[h])
    assert rel(torch.cat(halves), cs_f) <= 1e-6
    assert rel(scs[0][:3], sc_f[:3]) <= 1e-6 and torch.equal(scs[0], scs[1])


# ---------------------------------------------------------------------------------------------- 2 ranks, gloo
K, D, N = 512, 768, 1200


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    w0, e0 = synth.normal((K, D), 1, 0), synth.normal((K, D), 1, 1)
    st = vo.VQState(w0.clone(), e0.clone(), torch.zeros(K))
    roles = [(0, 200), (150, 400), (256, 512)] if mode == "code_roles" else None
    if mode.startswith("token"):
        from t5vqvae_amd import VectorQuantizerEMA
        from t5vqvae_amd.sharded import shard_tokens
        m = VectorQuantizerEMA(K, D, 0.25, decay=0.99).cuda()
        with torch.no_grad():
            m._embedding.weight.copy_(w0)
            m._ema_w.copy_(e0)
        shard_tokens(m)
        if mode == "token_overlap":        # statistics pass -> asynchronous all-reduce under the output pass -> finalize
            m.stats_exchange = "overlap"
    else:
        from t5vqvae_amd.sharded import CodeShardedVectorQuantizerEMA
        m = CodeShardedVectorQuantizerEMA(K, D, 0.25, decay=0.99, disentanglement=roles is not None).cuda()
        m.load_full_state_dict({"_embedding.weight": w0.cuda(), "_ema_w": e0.cuda(),
                                "_ema_cluster_size": torch.zeros(K).cuda()})
    m.train()
    m.encodings = "indices"
    errs = []
    for t in range(2):
        x = synth.normal((4, N // 4, D), 2, t)
        srl = synth.role_ranges(N, roles, 3, t).view(4, N // 4, 2) if roles else None
        w_before = st.weight.clone()
        o = vo.faithful_forward(st, x, srl, ema=True, training=True, commitment_cost=0.25,
                                disentanglement=roles is not None)
        if mode.startswith("token"):
            sl = slice(rank * 2, rank * 2 + 2)
            loss, qz, ppl, idx = m(x[sl].cuda())
            ref_idx, ref_q = o.indices.view(4, -1)[sl].reshape(-1), o.quantized[sl]
            xs = x[sl]
        else:
            loss, qz, ppl, idx = m(x.cuda(), None if srl is None else srl.cuda())
            ref_idx, ref_q, xs = o.indices, o.quantized, x
        out, _ = vo.classify_mismatches(xs.reshape(-1, D), w_before, idx.squeeze(1).cpu(), ref_idx,
                                        None if srl is None else srl)
        assert out == 0 and torch.equal(idx.squeeze(1).cpu(), ref_idx), "index mismatch"
        errs += [rel(loss, o.loss), rel(ppl, o.perplexity), rel(qz, ref_q)]
    sd = m.state_dict() if mode.startswith("token") else m.full_state_dict()
    errs += [rel(sd["_ema_cluster_size"], st.cluster_size), rel(sd["_ema_w"], st.ema_w),
             rel(sd["_embedding.weight"], st.weight)]
    q.put((rank, max(errs)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["token", "token_overlap", "code", "code_roles"])
def test_two_ranks_on_one_gpu_gloo(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        for p in procs:
            p.join(240)
            assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    finally:
        for p in procs:              # a worker stuck in a collective must not outlive the test holding the GPU
            if p.is_alive():
                p.kill()
                p.join(10)
    res = dict(q.get(timeout=5) for _ in range(2))
    assert max(res.values()) <= TOL, res
