"""The module's host route (torch.ops.t5vq, csrc/t5vq_torch.cpp) against the ctypes route of the same C ABI, the
codebook norms kept between calls, the loss terms summed by the finalize kernel, and norms handed to the filtered
search.  All comparisons are between two ways of running the SAME kernels, so the bar is bit-identity (embed_sum and
what derives from it: 1e-6, the order of fp32 atomics differs from launch to launch)."""
import pytest
import torch

from t5vqvae_amd import VectorQuantizer, VectorQuantizerEMA, ops, synth, vqvae

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm())


def _state(k, d, seed):
    w = synth.normal((k, d), seed, 0, device=DEV)
    ema_w = w * 3.0 + synth.normal((k, d), seed, 1, device=DEV) * 0.1
    cs = torch.full((k,), 3.0, device=DEV)
    return w, ema_w, cs


@pytest.mark.parametrize("n,k,d,search", [(2880, 10000, 768, "scan"), (40000, 1000, 768, "scan"),
                                          (4096, 512, 768, "scan"), (2880, 10000, 768, "filter"),
                                          (777, 130, 64, "scan")])
def test_fused_step_equals_ctypes_composites(n, k, d, search):
    """t5vq::vq_step (one op) == t5vq_vq_forward_f32 + t5vq_vq_ema_finalize_f32 through ctypes."""
    w, ema_w, cs = _state(k, d, 3)
    x, _ = synth.clustered(n, w, 0.7, 4)
    hyper = dict(loss_denom=float(n * d), decay=0.99, one_minus_decay=1 - 0.99, epsilon=1e-5, k_epsilon=k * 1e-5,
                 commitment_cost=0.25, plain_variant=False)
    w_a, e_a, c_a = w.clone(), ema_w.clone(), cs.clone()
    ee_a = ops.code_sqnorms(w_a)
    ste, idx, diff, q, scalars, ovf = ops.fused_step(x, w_a, None, "mse", ee=ee_a, cluster_size=c_a, ema_w=e_a,
                                                     filtered=search == "filter", want_diff=True, update=True,
                                                     ee_out=ee_a, **hyper)
    assert q is None and (ovf is not None) == (search == "filter")
    assert scalars.shape == (4,)
    # ctypes route, caller-allocated everything, no norms given, loss reduced by its own launch
    w_b, e_b, c_b = w.clone(), ema_w.clone(), cs.clone()
    b = dict(ste=torch.empty_like(x), diff=torch.empty_like(x), idx=torch.empty(n, dtype=torch.int64, device=DEV),
             counts=torch.empty(k, device=DEV), embed_sum=torch.empty(k, d, device=DEV),
             loss_sum=torch.empty(1, device=DEV))
    (ops.vq_forward_filtered if search == "filter" else ops.vq_forward)(x, w_b, None, "mse", **b)
    sc_b = torch.empty(4, device=DEV)
    ops.ema_finalize(c_b, e_b, w_b, b["counts"], b["embed_sum"], b["loss_sum"], n, hyper["loss_denom"], 0.99, 1 - 0.99,
                     1e-5, k * 1e-5, 0.25, False, True, sc_b)
    assert torch.equal(idx, b["idx"]) and torch.equal(ste, b["ste"]) and torch.equal(diff, b["diff"])
    assert torch.equal(scalars, sc_b) and torch.equal(c_a, c_b)
    assert rel(e_a, e_b) <= 1e-6 and rel(w_a, w_b) <= 1e-6
    # the norms the finalize kernel left behind are those of the codebook it wrote, bit for bit
    assert torch.equal(ee_a, ops.row_sqnorms(None, w_a)[1])
    assert torch.isinf(ee_a[k:]).all()


def test_loss_terms_summed_by_the_finalize_kernel():
    """Unreduced loss terms handed to t5vq_ema_cluster_f32 == t5vq_reduce_partials_f32 first (same order, same bits),
    per-workgroup terms (small N) and per-token terms (ordered walk, N >= 32768)."""
    for n, k, d, with_es in ((2880, 1000, 768, True), (40000, 1000, 768, True), (40000, 1000, 768, False)):
        w, ema_w, cs = _state(k, d, 5)
        x, _ = synth.clustered(n, w, 0.7, 6)
        n_terms = ops.forward_loss_terms(n, with_es)
        assert n_terms == (n if with_es else ops.gather_blocks(n))
        terms = torch.empty(n_terms, device=DEV)
        loss_sum = torch.empty(1, device=DEV)
        counts = torch.empty(k, device=DEV)
        es = torch.empty(k, d, device=DEV) if with_es else None
        ops.vq_forward(x, w, None, "mse", ste=torch.empty_like(x), counts=counts, embed_sum=es, loss_terms=terms,
                       loss_sum=loss_sum)
        args = (k, n, float(n * d), 0.99, 1 - 0.99, 1e-5, k * 1e-5, 0.25, False, 0)
        s1, s2 = torch.empty(4, device=DEV), torch.empty(4, device=DEV)
        ops.ema_cluster(None, counts, loss_sum, *args, s1)
        ops.ema_cluster(None, counts, terms, *args, s2)
        assert torch.equal(s1, s2)
        assert rel(s1[3], (terms.double().sum() / (n * d)).float()) <= 1e-6


def test_fixed_point_finalize_emits_the_same_norms():
    k, d, n = 1000, 768, 3000
    w, ema_w, cs = _state(k, d, 7)
    x, _ = synth.clustered(n, w, 0.7, 8)
    xx, ee = ops.row_sqnorms(x, w)
    keys = ops.dist_argmin(x, w, xx, ee)
    bits = ops.max_norm(xx)
    esum = ops.embed_sum_fixed(x, keys, bits, k)
    ee_out = torch.full_like(ee, float("inf"))
    ops.ema_embed_fixed(ema_w, w, esum, bits, cs, 0.99, 1 - 0.99, ee_out=ee_out)
    assert torch.equal(ee_out, ops.row_sqnorms(None, w)[1])


def _fresh(k, d, seed, cls=VectorQuantizerEMA, **kw):
    m = cls(k, d, 0.25, **({"decay": 0.99} if cls is VectorQuantizerEMA else {}), **kw).to(DEV)
    m.encodings = "indices"
    with torch.no_grad():
        m._embedding.weight.copy_(synth.normal((k, d), seed, 0))
        if cls is VectorQuantizerEMA:
            m._ema_w.copy_(synth.normal((k, d), seed, 1))
    return m


def test_code_norm_cache_follows_the_codebook():
    """Cached |e|^2: written by the finalize kernel in training, dropped by any tracked in-place change of the
    Parameter, and never the reason for a different result (twin module with the cache off)."""
    k, d = 1000, 768
    a, b = _fresh(k, d, 9), _fresh(k, d, 9)
    b.code_norm_cache = False
    for m in (a, b):
        m.search = "scan"
        m.train()
    xs = [synth.normal((4, 300, d), 10, t).to(DEV) for t in range(4)]
    for t in range(2):
        oa, ob = a(xs[t]), b(xs[t])
        # two modules trained side by side: their codebooks agree up to the order of the statistics' fp32 atomics
        assert torch.equal(oa[3], ob[3]) and rel(oa[0], ob[0]) <= 1e-6 and rel(oa[1], ob[1]) <= 1e-6
        ent = vqvae._EE_STATE[a]
        assert torch.equal(ent[0], ops.row_sqnorms(None, a._embedding.weight.detach())[1])   # norms of the NEW codebook
    assert b not in vqvae._EE_STATE
    # a tracked in-place change (load_state_dict, copy_, optimizer step) invalidates the entry
    with torch.no_grad():
        new_w = synth.normal((k, d), 11, 0).to(DEV)
        a._embedding.weight.copy_(new_w)
        b._embedding.weight.copy_(new_w)
    assert a._code_norms(a._embedding.weight.detach()) is None
    a.eval(), b.eval()
    oa, ob = a(xs[2]), b(xs[2])
    assert torch.equal(oa[3], ob[3]) and torch.equal(oa[0], ob[0])
    assert torch.equal(a.get_indices(xs[3]), b.get_indices(xs[3]))      # eval: entry made once, reused
    sd = {key: v.clone() for key, v in b.state_dict().items()}
    sd["_embedding.weight"] = synth.normal((k, d), 12, 0).to(DEV)
    a.load_state_dict(sd), b.load_state_dict(sd)
    assert torch.equal(a.get_indices(xs[3]), b.get_indices(xs[3]))
    # a write through .data is invisible to the version counter: documented, with an explicit way out
    a._embedding.weight.data.copy_(new_w)
    a.invalidate_code_norms()
    b._embedding.weight.data.copy_(new_w)
    assert torch.equal(a.get_indices(xs[3]), b.get_indices(xs[3]))


def test_plain_variant_does_not_keep_stale_norms_across_optimizer_steps():
    k, d = 512, 64
    a, b = _fresh(k, d, 13, VectorQuantizer), _fresh(k, d, 13, VectorQuantizer)
    b.code_norm_cache = False
    opts = [torch.optim.SGD(m.parameters(), lr=5.0) for m in (a, b)]
    for t in range(3):
        x = synth.normal((2, 200, d), 14, t).to(DEV)
        outs = []
        for m, o in zip((a, b), opts):
            m.train()
            xg = x.clone().requires_grad_(True)
            loss, q, _, idx = m(xg)
            o.zero_grad()
            (loss + q.sum()).backward()
            o.step()
            outs.append((loss.detach(), idx, xg.grad))
        # the codebook gradient is a scatter-add with fp32 atomics: the twins agree to rounding, not to the bit
        assert torch.equal(outs[0][1], outs[1][1])
        assert rel(outs[0][0], outs[1][0]) <= 1e-6 and rel(outs[0][2], outs[1][2]) <= 1e-6
    assert rel(a._embedding.weight, b._embedding.weight) <= 1e-6
    assert float((a._embedding.weight - synth.normal((k, d), 13, 0).to(DEV)).abs().max()) > 1e-4   # it did move


@pytest.mark.parametrize("search", ["scan", "filter"])
def test_given_row_norms_are_consumed_by_both_searches(search):
    """`_xx` (the encoder tail's |x|^2) is an input of the search in either mode: a deliberately perturbed xx must
    steer both searches the same way -- the filtered forward no longer recomputes the norms it was given."""
    k, d, n = 2048, 768, 4096
    m = _fresh(k, d, 15)
    m.eval()
    x = synth.normal((1, n, d), 16, 0).to(DEV)
    flat = x.view(-1, d)
    xx = ops.row_sqnorms(flat, None)[0]
    m.search = search
    base = m(x, _xx=xx)
    plain = m(x)
    assert torch.equal(base[3], plain[3]) and torch.equal(base[1], plain[1])
    # a perturbation that changes the rounding of fl(xx + ee) for every code: indices may move only on near-ties,
    # but the two searches must agree with each other bit for bit under the SAME perturbed norms
    xx_p = xx * 1.5 + 7.0
    m.search = "scan"
    want = m(x, _xx=xx_p)
    m.search = search
    got = m(x, _xx=xx_p)
    assert torch.equal(got[3], want[3]) and torch.equal(got[1], want[1]) and torch.equal(got[0], want[0])
    keys_s = ops.dist_argmin(flat, m._embedding.weight.detach(), xx_p, ops.code_sqnorms(m._embedding.weight.detach()))
    assert torch.equal(ops.unpack_keys(keys_s)[0], got[3].view(-1))


@pytest.mark.parametrize("n,k,d,given", [(4096, 512, 768, "none"), (4096, 512, 768, "ee"), (4096, 512, 768, "both"),
                                         (1000, 640, 256, "none"), (130, 128, 512, "ee"), (129, 2048, 1024, "xx"),
                                         (3000, 1000, 768, "none"), (64, 4, 256, "none")])
def test_self_contained_small_search_equals_the_chain(n, k, d, given):
    """Small problems (fewer than 256 tiles of 128 x 128, D % 256 == 0, plain search) take the self-contained distance
    kernel inside the composite forward: |x|^2 from the staged X tile, per-split key slots merged by the gather
    kernel, statistics zeroed on the way -- no launch in front of it.  Everything it produces must equal the chain of
    fine-grained entry points (row norms -> distance + argmin with atomicMin merge -> gather) bit for bit."""
    w = synth.normal((k, d), 21, 0, device=DEV)
    x, _ = synth.clustered(n, w, 0.7, 22)
    xx, ee = ops.row_sqnorms(x, w)

    def bufs():
        return dict(ste=torch.full_like(x, float("nan")), diff=torch.full_like(x, float("nan")),
                    idx=torch.full((n,), -1, dtype=torch.int64, device=DEV),
                    counts=torch.full((k,), float("nan"), device=DEV),
                    embed_sum=torch.full((k, d), float("nan"), device=DEV))
    a, b = bufs(), bufs()
    ls_a, ls_b = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    ops.vq_forward(x, w, None, "mse", xx=xx if given in ("xx", "both") else None,
                   ee=ee if given in ("ee", "both") else None, loss_sum=ls_a, **a)
    keys = ops.dist_argmin(x, w, xx, ee)
    part = torch.empty(ops.gather_blocks(n), device=DEV)
    ops.gather_stats(x, w, keys, "mse", loss_partial=part, **b)
    ops.reduce_partials(part, ls_b)
    for name in ("idx", "ste", "diff", "counts"):
        assert torch.equal(a[name], b[name]), name
    assert torch.equal(ls_a, ls_b)
    assert rel(a["embed_sum"], b["embed_sum"]) <= 1e-6
    # and twice in a row on the same buffers (nothing is carried over in the key slots)
    ops.vq_forward(x, w, None, "mse", loss_sum=ls_a, **a)
    assert torch.equal(a["idx"], b["idx"]) and torch.equal(ls_a, ls_b)


def test_self_contained_search_ties_and_eval_module():
    """Exact duplicates across code splits resolve to the lowest index when the splits' key slots are merged; the
    module's evaluation path at C2's shape goes through the same kernel and agrees with get_indices (fine-grained)."""
    k, d, n = 512, 768, 4096
    m = _fresh(k, d, 23)
    with torch.no_grad():
        for dup in (70, 200, 511):                       # other 64-code tiles = other splits
            m._embedding.weight[dup] = m._embedding.weight[5]
    x = synth.normal((1, n, d), 24, 0).to(DEV)
    with torch.no_grad():
        x.view(-1, d)[:8] = m._embedding.weight[5]
    m.eval()
    m.search = "scan"
    out = m(x)
    assert (out[3].view(-1)[:8] == 5).all()
    assert torch.equal(out[3], m.get_indices(x))


def test_ops_reject_arguments_that_would_read_out_of_bounds():
    """torch.ops.t5vq.* check sizes (and devices) before a pointer reaches a kernel: a short gradient, an index
    tensor of the wrong length, shapes that disagree, a CPU tensor."""
    from t5vqvae_amd import _lib
    t = _lib.load_torch_ops()
    dev = "cuda"
    diff = torch.zeros(8, 64, device=dev)
    one = torch.ones((), device=dev)
    with pytest.raises(RuntimeError, match="element count"):
        t.ste_backward(torch.zeros(4, 64, device=dev), diff, one, 1.0)
    with pytest.raises(RuntimeError, match="scalar"):
        t.ste_backward(None, diff, torch.ones(2, device=dev), 1.0)
    with pytest.raises(RuntimeError, match="one code per row"):
        t.scatter_rows(diff, torch.zeros(7, dtype=torch.int64, device=dev), None, 1.0, 16)
    with pytest.raises(RuntimeError, match="same shape"):
        t.cosine_backward(diff, torch.zeros(8, 32, device=dev), None, one, 1.0, 1.0, False)
    with pytest.raises(RuntimeError, match="disagree on D"):
        t.vq_forward(diff, torch.zeros(16, 32, device=dev), None, None, None, 0, False, False, False, False, False)
    with pytest.raises(RuntimeError, match="row norms"):
        t.vq_forward(diff, torch.zeros(16, 64, device=dev), None, torch.zeros(7, device=dev), None, 0, False, False,
                     False, False, False)
    with pytest.raises(RuntimeError, match="no CPU path|ROCm device"):
        t.code_sqnorms(torch.zeros(16, 64))
    assert t.ste_backward(None, diff, one, 1.0).shape == diff.shape          # the valid forms still run
    assert t.scatter_rows(diff, torch.zeros(8, dtype=torch.int64, device=dev), None, 1.0, 16).shape == (16, 64)
