"""The single-launch small-shape forward (t5vq_vq_forward_small_f32, csrc/t5vq_small.hip) against the three-launch
chain (t5vq_vq_forward_f32 + t5vq_vq_ema_finalize_f32) and against the oracle: every output bit-identical to the chain
(embed_sum up to the order of its atomic adds), the state block left zeroed, exact duplicates resolved to the lowest
index across lane halves / waves / code splits, the module's evaluation and plain-variant calls routed through it."""
import pytest
import torch

from oracle import vq_oracle as vo
from t5vqvae_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
TOL = 1e-5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


def _chain(x, w, ee, stats):
    n, d = x.shape
    k = w.shape[0]
    o = dict(ste=torch.empty_like(x), diff=torch.empty_like(x), q=torch.empty_like(x),
             idx=torch.empty(n, dtype=torch.int64, device=DEV), counts=torch.empty(k, device=DEV))
    if stats:
        o["embed_sum"] = torch.empty(k, d, device=DEV)
    terms = torch.empty(ops.forward_loss_terms(n, stats), device=DEV)
    ops.vq_forward(x, w, None, "mse", ee=ee, loss_terms=terms, **o)
    sc = torch.empty(4, device=DEV)
    ops.ema_finalize(None, None, None, o["counts"], None, terms, n, float(n * d), 0.99, 0.01, 1e-5, k * 1e-5, 0.25,
                     False, False, sc)
    o["terms"], o["scalars"] = terms, sc
    return o


def _small(x, w, ee, state, stats, scalars, xx=None):
    n, d = x.shape
    k = w.shape[0]
    o = dict(ste=torch.full_like(x, float("nan")), diff=torch.full_like(x, float("nan")), q=torch.full_like(x, float("nan")),
             idx=torch.full((n,), -1, dtype=torch.int64, device=DEV), counts=torch.full((k,), float("nan"), device=DEV))
    if stats:
        o["embed_sum"] = torch.full((k, d), float("nan"), device=DEV)
    terms = torch.full((n,), float("nan"), device=DEV)
    sc = torch.full((4,), float("nan"), device=DEV) if scalars else None
    ops.vq_forward_small(x, w, state, xx=xx, ee=ee, loss_terms=terms, scalars=sc, loss_denom=float(n * d),
                         commitment_cost=0.25, **o)
    o["terms"], o["scalars"] = terms, sc
    return o


@pytest.mark.parametrize("n,k,d", [(4096, 512, 768),      # C2: 32 x 256 tiles, two code splits that share the gather
                                   (1000, 300, 768),      # 32 x 128 tiles, three splits, ragged N and K
                                   (8000, 512, 768),      # one workgroup per token tile walks two code tiles: no hand-off
                                   (104, 512, 768),       # the golden cases' size: four splits
                                   (33, 8, 256),          # a second token tile with one token; fewer codes than a sub-tile
                                   (2048, 1024, 256),     # D = 256: one float4 per lane per row
                                   (700, 2000, 1024)])    # widest rows the path takes
def test_small_forward_equals_the_chain(n, k, d):
    assert ops.small_supported(n, k, d)
    w = synth.normal((k, d), 61, 0, device=DEV)
    x, _ = synth.clustered(n, w, 0.7, 62)
    xx, ee = ops.row_sqnorms(x, w)
    state = ops.small_state(k, DEV)
    for stats, scalars in ((False, True), (True, False), (False, False)):
        a = _chain(x, w, ee, stats)
        for rep in range(2):                            # twice: the state block must come back zeroed
            b = _small(x, w, ee, state, stats, scalars, xx=xx if rep else None)
            for name in ("idx", "ste", "diff", "q", "counts", "terms"):
                assert torch.equal(a[name], b[name]), (name, stats, scalars, rep)
            if scalars:
                assert torch.equal(a["scalars"], b["scalars"])
            if stats:
                assert rel(b["embed_sum"], a["embed_sum"]) <= 1e-6
            assert float(state.float().abs().sum()) == 0.0
    # against the oracle itself (indices by the band rule, the rest at 1e-5)
    st = vo.VQState(w.cpu().clone(), w.cpu().clone(), torch.zeros(k))
    ref = vo.algorithmic_forward(st.clone(), x.cpu().view(1, n, d), ema=True, training=False, commitment_cost=0.25)
    outside, inside = vo.classify_mismatches(x.cpu(), w.cpu(), b["idx"].cpu(), ref.indices)
    assert outside == 0, f"{outside} index mismatches outside the band ({inside} inside)"
    o = vo.algorithmic_forward(st, x.cpu().view(1, n, d), ema=True, training=False, commitment_cost=0.25,
                               indices=b["idx"].cpu())
    sc = _small(x, w, ee, state, False, True)["scalars"]
    assert rel(b["ste"], o.quantized.view(n, d)) <= TOL and rel(sc[0], o.loss) <= TOL and rel(sc[1], o.perplexity) <= TOL


def test_small_forward_ties_pick_the_lowest_index():
    """Exact duplicates in other registers, lane halves, waves, code tiles and code splits of the single launch."""
    k, d = 512, 768
    w = synth.normal((k, d), 63, 0, device=DEV)
    for dup in (3 + 4, 3 + 32, 3 + 64, 3 + 128, 3 + 256, 3 + 300):
        w[dup] = w[3]
    w[200] = w[199]
    x = w[torch.tensor([3, 7, 35, 67, 131, 259, 303, 199, 200, 100], device=DEV)].clone()
    x = x.repeat(40, 1)[:4096 // 16]
    ee = ops.code_sqnorms(w)
    state = ops.small_state(k, DEV)
    b = _small(x, w, ee, state, False, True)
    want = torch.tensor([3, 3, 3, 3, 3, 3, 3, 199, 199, 100], device=DEV).repeat(40)[:x.shape[0]]
    assert torch.equal(b["idx"], want)
    assert torch.equal(b["idx"], _chain(x, w, ee, False)["idx"])


def test_module_takes_the_single_launch_and_matches_the_chain():
    """Evaluation and the plain variant's training call (no EMA update) go through the single launch; results, gradients
    included, equal the chain's bit for bit; EMA training keeps the chain."""
    from t5vqvae_amd import VectorQuantizer, VectorQuantizerEMA
    from t5vqvae_amd import vqvae as hv
    n, k, d = 4096, 512, 768
    w = synth.normal((k, d), 64, 0, device=DEV)
    x, _ = synth.clustered(n, w, 0.7, 65)
    x = x.view(64, 64, d)
    m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).to(DEV)
    with torch.no_grad():
        m._embedding.weight.copy_(w)
    m.encodings = "indices"
    m.eval()
    m.small_forward = True
    a = m(x)
    assert m in hv._SMALL_STATE                           # the single launch was taken (its state block exists)
    m.small_forward = False
    b = m(x)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # plain variant, training, gradients to the inputs and the codebook
    p = VectorQuantizer(k, d, 0.25).to(DEV)
    with torch.no_grad():
        p._embedding.weight.copy_(w)
    p.encodings = "indices"
    grads = []
    for small in (True, False):
        p.small_forward = small
        p.zero_grad()
        xg = x.clone().requires_grad_(True)
        loss, q, ppl, idx = p(xg)
        (loss * 1.3 + (q * q).sum() * 1e-3).backward()
        grads.append((loss.detach().clone(), q.detach().clone(), idx.clone(), xg.grad.clone(),
                      p._embedding.weight.grad.clone()))
    assert p in hv._SMALL_STATE
    for u, v in list(zip(*grads))[:4]:
        assert torch.equal(u, v)
    assert rel(grads[0][4], grads[1][4]) <= 1e-6          # the codebook gradient is a scatter-add by fp32 atomics
    # graph capture: the state block exists already, the captured call replays
    g = torch.cuda.CUDAGraph()
    m.small_forward = True
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m(x)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        out = m(x)
    g.replay()
    torch.cuda.synchronize()
    for u, v in zip(out, a):
        assert torch.equal(u, v)
