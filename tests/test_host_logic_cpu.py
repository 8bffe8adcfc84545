"""Host logic without a GPU: the modules' launch order, autograd wiring and multi-GPU exchanges, driven with
``tests/fake_ops.py`` patched in for the kernel layer.  Multi-rank cases run world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fake_ops
import golden_cases as gc
from oracle import vq_oracle as vo
from t5vqvae_amd import synth

TOL = 1e-5


def _patch():
    import t5vqvae_amd.vqvae as v
    v.ops = fake_ops


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


@pytest.mark.parametrize("name", ["G1_ema_train3", "G3_plain_bwd", "G4_ema_roles2", "G4b_plain_roles5", "G5_ties",
                                  "G6_ema_cosine", "G6b_plain_cosine", "G7b_methods_roles", "G11_small_dim"])
def test_module_choreography_on_golden(name, monkeypatch):
    import t5vqvae_amd
    import t5vqvae_amd.vqvae as v
    monkeypatch.setattr(v, "ops", fake_ops)
    case = gc.CASES[name]
    got = gc.run_case(case, lambda cls, **kw: getattr(t5vqvae_amd, cls)(**kw), "cpu")
    exp = gc.load_case(name)
    assert set(got) == set(exp)
    for key in sorted(exp):
        if key.endswith("_idx") or key.endswith("_get_indices"):
            assert np.array_equal(got[key].numpy().ravel(), exp[key]["full"].ravel()), key
        else:
            assert gc.compare(got[key], exp[key]) <= TOL, key


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    _patch()


K, D, N = 256, 64, 600


def _reference_run(steps, roles=None, training=True):
    w, e = synth.normal((K, D), 1, 0), synth.normal((K, D), 1, 1)
    st = vo.VQState(w.clone(), e.clone(), torch.zeros(K))
    outs = []
    for t in range(steps):
        x = synth.normal((4, N // 4, D), 2, t)
        srl = synth.role_ranges(N, roles, 3, t).view(4, N // 4, 2) if roles else None
        w_before = st.weight.clone()
        o = vo.faithful_forward(st, x, srl, ema=True, training=training, commitment_cost=0.25,
                                disentanglement=roles is not None)
        o.extra["q_raw"] = w_before.index_select(0, o.indices).view(x.shape)
        outs.append(o)
    return st, outs


def _token_sharded_worker(rank, world, port, q, deterministic=False, exchange="serial"):
    _init(rank, world, port)
    from t5vqvae_amd import VectorQuantizerEMA
    from t5vqvae_amd.sharded import shard_tokens, broadcast_state
    m = VectorQuantizerEMA(K, D, 0.25, decay=0.99)
    m.deterministic = deterministic
    m.stats_exchange = exchange
    with torch.no_grad():
        if rank == 0:
            m._embedding.weight.copy_(synth.normal((K, D), 1, 0))
            m._ema_w.copy_(synth.normal((K, D), 1, 1))
    broadcast_state(m)
    shard_tokens(m)
    m.train()
    st, outs = _reference_run(2)
    errs = []
    for t in range(2):
        x = synth.normal((4, N // 4, D), 2, t)
        xs = x[rank * 2:(rank + 1) * 2].clone().requires_grad_(True)       # batch rows 0-1 / 2-3
        loss, qz, ppl, enc = m(xs)
        (loss * 1.5 + qz.sum()).backward()
        o = outs[t]
        errs += [rel(loss, o.loss), rel(ppl, o.perplexity),
                 rel(qz, o.quantized[rank * 2:(rank + 1) * 2])]
        assert torch.equal(enc.argmax(1), o.indices.view(4, -1)[rank * 2:(rank + 1) * 2].reshape(-1))
        # gradient of the GLOBAL-mean commitment loss w.r.t. the local tokens (q_raw = the step's quantized), times
        # the group size: the default convention ("mean") expects the caller to average parameter gradients over the
        # group, as DistributedDataParallel does
        xl = x[rank * 2:(rank + 1) * 2]
        g_ref = 1.0 + world * 1.5 * 0.25 * 2.0 * (xl - o.extra["q_raw"][rank * 2:(rank + 1) * 2]) / (N * D)
        errs.append(rel(xs.grad, g_ref))
    errs += [rel(m._ema_cluster_size, st.cluster_size), rel(m._ema_w.data, st.ema_w),
             rel(m._embedding.weight.data, st.weight)]
    # replicas must be bit-identical
    for tns in m.state_dict().values():
        other = [torch.empty_like(tns) for _ in range(world)]
        dist.all_gather(other, tns)
        assert all(torch.equal(o_, other[0]) for o_ in other)
    if deterministic:
        # order-independent statistics: the two-rank result is BIT-IDENTICAL to one process on the whole batch
        _patch()
        single = VectorQuantizerEMA(K, D, 0.25, decay=0.99)
        single.deterministic = True
        with torch.no_grad():
            single._embedding.weight.copy_(synth.normal((K, D), 1, 0))
            single._ema_w.copy_(synth.normal((K, D), 1, 1))
        single.train()
        for t in range(2):
            single(synth.normal((4, N // 4, D), 2, t))
        for name, tns in m.state_dict().items():
            assert torch.equal(tns, single.state_dict()[name]), name
    q.put((rank, max(errs)))
    dist.destroy_process_group()


def _plain_token_sharded_worker(rank, world, port, q, reduction):
    """Plain VectorQuantizer (codebook learned by gradient), token-sharded over 2 ranks: the gradients to the local
    inputs and to the (replicated) codebook, reduced over the group the way ``gradient_reduction`` announces, must
    equal the oracle's gradients on the concatenated batch (vqvae/vqvae.py:55-56,69,71 under autograd)."""
    _init(rank, world, port)
    from t5vqvae_amd import VectorQuantizer
    from t5vqvae_amd.sharded import shard_tokens
    m = VectorQuantizer(K, D, 0.25)
    w0 = synth.normal((K, D), 1, 0)
    with torch.no_grad():
        m._embedding.weight.copy_(w0)
    shard_tokens(m)
    m.gradient_reduction = reduction
    m.train()
    x = synth.normal((4, N // 4, D), 2, 0)
    # oracle: one process, whole batch.  The straight-through output passes its gradient to the inputs unchanged
    # (vqvae/vqvae.py:71), so the oracle only has to provide the gradient of the loss term; gq (a weight on every
    # output element, scaled to the loss gradient's magnitude so neither part drowns the other) is added by hand.
    gq = synth.normal((4, N // 4, D), 2, 5) * 1e-5
    st = vo.VQState(w0.clone().requires_grad_(True))
    xr = x.clone().requires_grad_(True)
    o = vo.faithful_forward(st, xr, ema=False, training=True, commitment_cost=0.25)
    (o.loss * 1.5).backward()
    # sharded: each rank back-propagates through its own tokens only
    xs = x[rank * 2:(rank + 1) * 2].clone().requires_grad_(True)
    loss, qz, ppl, enc = m(xs)
    (loss * 1.5 + (qz * gq[rank * 2:(rank + 1) * 2]).sum()).backward()
    errs = [rel(loss, o.loss)]
    gw = m._embedding.weight.grad.clone()
    dist.all_reduce(gw, op=dist.ReduceOp.SUM)
    if reduction == "mean":
        gw /= world                       # what DistributedDataParallel does with the codebook gradient
    errs.append(rel(gw, st.weight.grad))
    # inputs: a parameter upstream sees sum_r J_r^T gx_r and, under "mean", divides by the group size -- so the loss
    # part of gx_r carries the factor `world` against the single-process rows; the part that came in through the
    # straight-through output is this rank's own downstream gradient and passes unchanged
    scale = world if reduction == "mean" else 1
    errs.append(rel(xs.grad, gq[rank * 2:(rank + 1) * 2] + scale * xr.grad[rank * 2:(rank + 1) * 2]))
    q.put((rank, max(errs)))
    dist.destroy_process_group()


def _code_sharded_worker(rank, world, port, q, roles, training, deterministic=False):
    _init(rank, world, port)
    from t5vqvae_amd.sharded import CodeShardedVectorQuantizerEMA
    m = CodeShardedVectorQuantizerEMA(K, D, 0.25, decay=0.99, disentanglement=roles is not None)
    m.deterministic = deterministic
    m.load_full_state_dict({"_embedding.weight": synth.normal((K, D), 1, 0), "_ema_w": synth.normal((K, D), 1, 1),
                            "_ema_cluster_size": torch.zeros(K)})
    m.train(training)
    st, outs = _reference_run(2, roles, training)
    errs = []
    for t in range(2):
        x = synth.normal((4, N // 4, D), 2, t)
        srl = synth.role_ranges(N, roles, 3, t).view(4, N // 4, 2) if roles else None
        if t == 1:
            w_full = m.full_state_dict()["_embedding.weight"]
            stt = vo.VQState(w_full)
            assert rel(m.get_latent(x), vo.get_latent(stt, x)) <= TOL
            assert torch.equal(m.get_indices(x, srl), vo.get_indices(stt, x, srl, roles is not None))
            assert rel(m.get_hidden(x, srl), vo.get_hidden(stt, x, srl, roles is not None)) <= TOL
        loss, qz, ppl, enc = m(x, srl)
        o = outs[t]
        assert torch.equal(enc.argmax(1), o.indices), f"step {t}"
        errs += [rel(loss, o.loss), rel(ppl, o.perplexity), rel(qz, o.quantized)]
    full = m.full_state_dict()
    errs += [rel(full["_ema_cluster_size"], st.cluster_size), rel(full["_ema_w"], st.ema_w),
             rel(full["_embedding.weight"], st.weight)]
    if deterministic:          # fixed-point statistics: the sharded _ema_w is BIT-IDENTICAL to the unsharded one (the
        # codebook itself also divides by cluster sizes whose normaliser is reduced across ranks: equal to rounding)
        from t5vqvae_amd import VectorQuantizerEMA
        single = VectorQuantizerEMA(K, D, 0.25, decay=0.99, disentanglement=roles is not None)
        single.deterministic = True
        with torch.no_grad():
            single._embedding.weight.copy_(synth.normal((K, D), 1, 0))
            single._ema_w.copy_(synth.normal((K, D), 1, 1))
        single.train(training)
        for t in range(2):
            single(synth.normal((4, N // 4, D), 2, t),
                   synth.role_ranges(N, roles, 3, t).view(4, N // 4, 2) if roles else None)
        assert torch.equal(full["_ema_w"], single.state_dict()["_ema_w"])
        assert rel(full["_embedding.weight"], single.state_dict()["_embedding.weight"]) <= 1e-6
    q.put((rank, max(errs)))
    dist.destroy_process_group()


def _code_sharded_data_write_worker(rank, world, port, q):
    """The kept gathered codebook after a write the version counter cannot see (``weight.data``): stale rows until
    ``invalidate_code_norms()`` (called on every rank), fresh ones after; and the module stays deep-copyable."""
    import copy
    _init(rank, world, port)
    from t5vqvae_amd.sharded import CodeShardedVectorQuantizerEMA, _FULL_STATE
    m = CodeShardedVectorQuantizerEMA(K, D, 0.25, decay=0.99)
    w0, w1 = synth.normal((K, D), 1, 0), synth.normal((K, D), 7, 3)
    m.load_full_state_dict({"_embedding.weight": w0, "_ema_w": w0.clone(), "_ema_cluster_size": torch.ones(K)})
    m.eval()
    x = synth.normal((4, N // 4, D), 2, 0)
    errs = [rel(m(x)[1], vo.get_hidden(vo.VQState(w0), x))]
    assert m in _FULL_STATE and "_full" not in m.__dict__
    sl = slice(m.code_begin, m.code_begin + m.local_codes)
    m._embedding.weight.data.copy_(w1[sl])          # invisible to the Parameter's version counter
    m.invalidate_code_norms()                        # ... so the caller says so, on every rank
    assert m not in _FULL_STATE
    errs.append(rel(m(x)[1], vo.get_hidden(vo.VQState(w1), x)))
    assert torch.equal(m.get_indices(x), vo.get_indices(vo.VQState(w1), x))
    # train() / eval() switches drop the kept codebook as well (the same event on every rank)
    m.train()
    assert m not in _FULL_STATE
    m.eval()
    c = copy.deepcopy(m)                             # no c10d Work / K x D tensor hides in the module's __dict__
    assert torch.equal(c._embedding.weight, m._embedding.weight)
    q.put((rank, max(errs)))
    dist.destroy_process_group()


def _run(worker, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q) + args) for r in range(2)]
    for p in procs:
        p.start()
    try:
        for p in procs:
            p.join(180)
            assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    finally:
        for p in procs:               # a hung worker must not outlive the test
            if p.is_alive():
                p.kill()
                p.join()
    res = dict(q.get(timeout=5) for _ in range(2))
    assert set(res) == {0, 1}
    assert max(res.values()) <= TOL, res


@pytest.mark.parametrize("deterministic", [False, True])
def test_token_sharded_equals_single_process_gloo(deterministic):
    _run(_token_sharded_worker, deterministic)


def test_token_sharded_overlapped_exchange_equals_single_process_gloo():
    """stats_exchange="overlap": statistics pass, asynchronous all-reduce, output pass, wait, finalize -- the same
    results as the serial form (both are held to the oracle on the concatenated batch, replicas bit-identical)."""
    _run(_token_sharded_worker, False, "overlap")


@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_token_sharded_plain_variant_gradients_gloo(reduction):
    _run(_plain_token_sharded_worker, reduction)


@pytest.mark.parametrize("roles,training", [(None, True), (None, False),
                                            ([(0, 128), (128, 256)], True),          # ranges aligned to shards
                                            ([(0, 50), (50, 200), (120, 256), (3, 5)], True)])  # straddling
def test_code_sharded_equals_single_process_gloo(roles, training):
    _run(_code_sharded_worker, roles, training)


def test_code_sharded_kept_codebook_follows_data_writes_gloo():
    _run(_code_sharded_data_write_worker)


def test_code_sharded_deterministic_is_bit_identical_to_unsharded_gloo():
    _run(_code_sharded_worker, [(0, 50), (50, 200), (120, 256), (3, 5)], True, True)


def test_encoder_tail_host_logic_and_autograd(monkeypatch):
    """EncoderTailVQ over the CPU stand-in kernels: forward == vq(oracle tail), gradients of hidden states and of
    the norm weight == torch autograd through the oracle's restatement of the tail (modeling_t5.py:191-207,864-865)."""
    import fake_ops
    from oracle import vq_oracle as vo
    from t5vqvae_amd import vqvae as hv, encoder_tail as et
    monkeypatch.setattr(hv, "ops", fake_ops)
    monkeypatch.setattr(et, "ops", fake_ops)
    torch.manual_seed(0)
    b, s_, d, k = 3, 20, 32, 50
    vq = hv.VectorQuantizerEMA(k, d, 0.25, decay=0.99)
    vq.encodings = "indices"
    tail = et.EncoderTailVQ(vq, hidden_size=d, dropout_rate=0.1)
    with torch.no_grad():
        tail.weight.copy_(torch.rand(d) + 0.5)
    h = torch.randn(b, s_, d, requires_grad=True)
    keep = (torch.rand(b, s_, d) > 0.1).to(torch.uint8)
    w0 = vq._embedding.weight.detach().clone()
    loss, q, ppl, idx = tail(h, keep_mask=keep)
    gq = torch.randn_like(q)
    (loss * 1.3 + (q * gq).sum()).backward()

    h2 = h.detach().clone().requires_grad_(True)
    g2 = tail.weight.detach().clone().requires_grad_(True)
    x2 = vo.encoder_tail(h2, g2, 1e-6, keep, 0.1)
    st = vo.VQState(w0, w0.clone(), torch.zeros(k))
    o = vo.algorithmic_forward(st, x2.detach(), ema=True, training=True, commitment_cost=0.25)
    assert torch.equal(idx.squeeze(1), o.indices)
    qraw = w0[o.indices].view_as(x2)
    loss2 = 0.25 * ((qraw - x2) ** 2).mean()
    ste2 = x2 + (qraw - x2).detach()
    (loss2 * 1.3 + (ste2 * gq).sum()).backward()
    assert torch.allclose(loss, loss2, rtol=1e-5) and torch.allclose(q, ste2, rtol=1e-5, atol=1e-6)
    assert (h.grad - h2.grad).norm() / h2.grad.norm() < 1e-5
    assert (tail.weight.grad - g2.grad).norm() / g2.grad.norm() < 1e-5


def test_code_and_token_images_follow_the_codebook(monkeypatch):
    """The module keeps the fp16 image of its codebook between filtered searches (written by the search that finds none,
    rewritten by every EMA update, dropped by invalidate_code_norms) and hands the encoder tail's token image to the
    search.  The stand-in kernels SEARCH THE IMAGE when told it is valid, so a stale image changes results: a module
    with the cache must equal one without it through training, evaluation, a state-dict load, a tracked in-place write
    and a .data write followed by invalidate_code_norms()."""
    import copy
    import fake_ops
    from t5vqvae_amd import vqvae as hv, encoder_tail as et
    monkeypatch.setattr(hv, "ops", fake_ops)
    monkeypatch.setattr(et, "ops", fake_ops)
    torch.manual_seed(1)
    k, d = 40, 64
    a = hv.VectorQuantizerEMA(k, d, 0.25, decay=0.9)
    a.encodings, a.search, a.code_image_cache = "indices", "filter", "always"
    b = copy.deepcopy(a)
    b.code_image_cache = False
    monkeypatch.setattr(et.EncoderTailVQ, "token_image_min_elements", 0)
    seen = []
    real = fake_ops.fused_step

    def spy(*args, **kw):
        seen.append((kw.get("code_image") is not None, bool(kw.get("code_image_valid")), kw.get("token_image") is not None))
        return real(*args, **kw)
    monkeypatch.setattr(fake_ops, "fused_step", spy)

    def both(x, train):
        outs = []
        for m in (a, b):
            m.train(train)
            outs.append(m(x))
        for u, v in zip(*outs):
            assert torch.equal(u, v)
        assert torch.equal(a._embedding.weight, b._embedding.weight)

    xs = [torch.randn(2, 30, d) for _ in range(8)]
    both(xs[0], True)                                   # no image yet: the search makes it, the update rewrites it
    both(xs[1], True)                                   # valid now
    both(xs[2], False)
    assert [s_ for s_ in seen if s_[0]] == [(True, False, False), (True, True, False), (True, True, False)]
    sd = {key: v.clone() + 0.01 for key, v in a.state_dict().items()}
    a.load_state_dict(sd), b.load_state_dict(sd)        # version bump: invalid, made again
    both(xs[3], False)
    assert seen[-2] == (True, False, False)
    both(xs[3], False)
    assert seen[-2] == (True, True, False)
    with torch.no_grad():
        for m in (a, b):
            m._embedding.weight.mul_(1.5)               # tracked in-place write
    both(xs[4], True)
    assert seen[-2] == (True, False, False)
    for m in (a, b):
        m._embedding.weight.data.add_(0.25)             # invisible to the version counter ...
        m.invalidate_code_norms()                       # ... so the caller says so
    both(xs[5], True)
    assert seen[-2] == (True, False, False)
    a.search = b.search = "scan"                        # an update under the scan keeps the image current
    both(xs[6], True)
    assert seen[-2] == (True, False, False)
    a.search = b.search = "filter"
    both(xs[7], False)
    assert seen[-2] == (True, True, False)

    # the encoder tail hands its token image to the filtered search (and none to the scan)
    tail_a, tail_b = et.EncoderTailVQ(a, hidden_size=d, dropout_rate=0.0), et.EncoderTailVQ(b, hidden_size=d, dropout_rate=0.0)
    h = torch.randn(2, 30, d)
    for m in (a, b):
        m.eval()
    ra, rb = tail_a(h), tail_b(h)
    assert seen[-2] == (True, True, True) and seen[-1] == (False, False, True)
    for u, v in zip(ra, rb):
        assert torch.equal(u, v)
    a.search = "scan"
    tail_a(h)
    assert seen[-1] == (False, False, False)

    # mode "eval": evaluation calls keep the image, an update drops it and writes none
    c = copy.deepcopy(b)
    c.code_image_cache, c.search = "eval", "filter"
    for train, want in ((False, (True, False, False)), (False, (True, True, False)), (True, (False, False, False)),
                        (False, (True, False, False)), (False, (True, True, False))):
        c.train(train)
        c(xs[0])
        assert seen[-1] == want, (train, seen[-1])


def test_role_helpers_follow_the_loader():
    """t5/load_data.py:68-86 and t5/utils.py:22-44 on a hand-worked example."""
    from t5vqvae_amd import roles
    dd = {"V": (0, 1000), "ARG0": (1000, 2000), "residual": (2000, 10000)}
    r = roles.expand_role_ranges(["ARG0", "V", "ARGM-TMP", "ARG0"], [1, 2, 1, 3], dd, 10)
    assert r.dtype == torch.int64 and r.shape == (10, 2)
    assert r.tolist() == [[1000, 2000], [0, 1000], [0, 1000], [2000, 10000], [1000, 2000], [1000, 2000], [1000, 2000],
                          [2000, 10000], [2000, 10000], [2000, 10000]]          # unknown label and padding -> residual
    with pytest.raises(ValueError):
        roles.expand_role_ranges(["V"], [1, 2], dd, 10)
    with pytest.raises(ValueError):
        roles.expand_role_ranges(["V"], [11], dd, 10)
    v = roles.role_vocab([["V", "ARG0", "V"], ["ARG1", "V", "ARG0"]])
    assert v == {"V": 0, "ARG0": 1, "ARG1": 2, "FIX": 3, "PAD": 4}               # most frequent first
    order, rs = roles.bucket_by_role(r)
    assert rs[:, 0].tolist() == sorted(r[:, 0].tolist()) and order[:2].tolist() == [1, 2]   # stable


def test_index_mode_absolute_host_logic(monkeypatch):
    """index_mode="absolute": start + slice argmin used as the row, against the oracle with the same convention;
    the default stays the reference's slice-relative convention."""
    import fake_ops
    from oracle import vq_oracle as vo
    from t5vqvae_amd import vqvae as hv
    monkeypatch.setattr(hv, "ops", fake_ops)
    torch.manual_seed(1)
    n, d, k = 1500, 16, 60
    m = hv.VectorQuantizerEMA(k, d, 0.25, decay=0.99, disentanglement=True)
    m.encodings = "indices"
    m.eval()
    x = torch.randn(1, n, d)
    table = torch.tensor([[0, 10], [10, 35], [35, 60]])
    srl = table[torch.randint(0, 3, (n,))].view(1, n, 2)
    w = m._embedding.weight.detach().clone()
    dist = vo.distances(x.view(-1, d), w)
    rel_idx = vo.select_codes(dist, srl)
    abs_idx = vo.select_codes(dist, srl, absolute=True)
    assert torch.equal(abs_idx, rel_idx + srl.view(-1, 2)[:, 0])
    assert torch.equal(m(x, srl)[3].squeeze(1), rel_idx)
    m.index_mode = "absolute"
    lo, q, ppl, idx = m(x, srl)
    assert torch.equal(idx.squeeze(1), abs_idx)
    assert torch.equal(m.get_indices(x, srl).squeeze(1), abs_idx)
    ref = vo.algorithmic_forward(vo.VQState(w, w.clone(), torch.zeros(k)), x, srl, ema=True, training=False,
                                 commitment_cost=0.25, disentanglement=True, indices=abs_idx)
    assert torch.allclose(q, ref.quantized, rtol=1e-5, atol=1e-6) and torch.allclose(lo, ref.loss, rtol=1e-5)
    m.index_mode = "bogus"
    with pytest.raises(ValueError):
        m(x, srl)
