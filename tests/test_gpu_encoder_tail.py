"""Encoder tail fused into the VQ prologue (t5vq_rmsnorm_sqnorms_f32 / _backward_f32, t5vqvae_amd.encoder_tail)
against the oracle's restatement of T5LayerNorm + dropout (modeling_t5.py:191-207,864-865) -- parity UNPINNED
against the reference for this piece (python_transformers does not import here; see oracle/vq_oracle.py).
Tolerances: forward 1e-6 norm-wise (one reduction + one division differ in rounding), gradients 1e-5."""
import pytest
import torch

from oracle import vq_oracle as vo
from t5vqvae_amd import synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("n,d,masked", [(2880, 768, True), (45, 768, False), (257, 64, True), (130, 2048, True),
                                        (1000, 100, False)])
def test_tail_forward_and_backward_vs_oracle(n, d, masked):
    from t5vqvae_amd import ops
    h = synth.normal((n, d), 31, 0) * 3.0
    gamma = synth.normal((d,), 31, 1).abs() + 0.5
    keep = (synth.uniform_int(n * d, 10, 31, 2).view(n, d) > 0).to(torch.uint8) if masked else None
    p = 0.1
    scale = 1.0 / (1.0 - p)
    x, xx, rms = ops.rmsnorm_sqnorms(h.cuda(), gamma.cuda(), None if keep is None else keep.cuda(), scale, 1e-6)
    hr = h.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    ref = vo.encoder_tail(hr, gr, 1e-6, keep, p)
    assert rel(x, ref) <= 1e-6
    assert rel(xx, (ref.double() ** 2).sum(1)) <= 1e-6
    assert rel(rms, torch.sqrt((h.double() ** 2).mean(1) + 1e-6)) <= 1e-6
    if masked:
        assert bool((x.cpu()[keep == 0] == 0).all())
    # the row norms are bit-identical to what t5vq_row_sqnorms_f32 computes from x (same lane layout, same order)
    xx2, _ = ops.row_sqnorms(x, None)
    assert torch.equal(xx, xx2)
    g = synth.normal((n, d), 31, 3)
    (ref * g).sum().backward()
    gh, gg = ops.rmsnorm_backward(g.cuda(), h.cuda(), gamma.cuda(), None if keep is None else keep.cuda(), scale, rms)
    assert rel(gh, hr.grad) <= 1e-5
    assert rel(gg, gr.grad) <= 1e-5
    gh2, gg2 = ops.rmsnorm_backward(g.cuda(), h.cuda(), gamma.cuda(), None if keep is None else keep.cuda(), scale, rms)
    assert torch.equal(gh, gh2) and torch.equal(gg, gg2)          # no atomics: run-to-run identical


def test_encoder_tail_vq_equals_unfused_chain():
    """EncoderTailVQ(hidden) == VectorQuantizerEMA(tail(hidden)) with the tail done by the oracle on the CPU: same
    indices, loss, straight-through output, EMA state; gradients of hidden states and norm weight vs torch autograd."""
    from t5vqvae_amd import VectorQuantizerEMA
    from t5vqvae_amd.encoder_tail import EncoderTailVQ
    b, s, d, k = 64, 45, 768, 1000
    w = synth.normal((k, d), 41, 0)
    ew = synth.normal((k, d), 41, 1)

    def make():
        m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).cuda()
        with torch.no_grad():
            m._embedding.weight.copy_(w)
            m._ema_w.copy_(ew)
        m.encodings = "indices"
        return m
    fused_vq, plain_vq = make(), make()
    tail = EncoderTailVQ(fused_vq, hidden_size=d, dropout_rate=0.1).cuda()
    gamma = synth.normal((d,), 41, 2).abs() + 0.5
    with torch.no_grad():
        tail.weight.copy_(gamma)
    h = synth.normal((b, s, d), 42, 0) * 2.0
    keep = (synth.uniform_int(b * s * d, 10, 42, 1).view(b, s, d) > 0).to(torch.uint8)
    gq = synth.normal((b, s, d), 42, 2)

    hc = h.cuda().requires_grad_(True)
    loss, q, ppl, idx = tail(hc, keep_mask=keep.cuda())
    (loss * 1.3 + (q * gq.cuda()).sum()).backward()

    hr = h.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    xr = vo.encoder_tail(hr, gr, 1e-6, keep, 0.1)
    xg = xr.detach().cuda().requires_grad_(True)
    loss2, q2, ppl2, idx2 = plain_vq(xg)
    (loss2 * 1.3 + (q2 * gq.cuda()).sum()).backward()
    xr.backward(xg.grad.cpu())

    mism = int((idx != idx2).sum())
    assert mism <= 2, mism                      # x differs by rounding between the two tails: near-ties only
    assert rel(loss, loss2) <= 1e-5 and rel(ppl, ppl2) <= 1e-4
    if mism == 0:
        assert rel(q, q2) <= 1e-6
        assert rel(fused_vq._embedding.weight.data, plain_vq._embedding.weight.data) <= 1e-5
        assert rel(hc.grad, hr.grad) <= 1e-5
        assert rel(tail.weight.grad, gr.grad) <= 1e-5


def test_encoder_tail_modes():
    """eval: no mask, no EMA update; train without a given mask: a Bernoulli(1 - p) mask is drawn and survivors scaled."""
    from t5vqvae_amd import VectorQuantizerEMA
    from t5vqvae_amd.encoder_tail import EncoderTailVQ
    d, k = 768, 64
    vq = VectorQuantizerEMA(k, d, 0.25, decay=0.99).cuda()
    tail = EncoderTailVQ(vq, hidden_size=d, dropout_rate=0.25).cuda()
    h = synth.normal((4, 33, d), 51, 0).cuda()
    tail.eval()
    w_before = vq._embedding.weight.detach().clone()
    out = tail(h)
    assert torch.equal(vq._embedding.weight.detach(), w_before)
    assert out[1].shape == h.shape
    tail.train()
    vq.eval()                                   # keep the codebook fixed; only the tail's own mode matters here
    torch.manual_seed(0)
    from t5vqvae_amd import ops
    x_eval, _, _ = ops.rmsnorm_sqnorms(h.view(-1, d), tail.weight.detach(), None, 1.0, 1e-6)
    # reproduce the drawn mask through the generator state
    torch.manual_seed(0)
    keep = torch.empty(h.numel() // d, d, dtype=torch.uint8, device="cuda").bernoulli_(0.75)
    torch.manual_seed(0)
    _ = tail(h)
    frac = float(keep.float().mean())
    assert 0.70 < frac < 0.80
    x_train, _, _ = ops.rmsnorm_sqnorms(h.view(-1, d), tail.weight.detach(), keep, 1.0 / 0.75, 1e-6)
    kept = keep.bool()
    assert rel(x_train[kept], x_eval[kept] / 0.75) <= 1e-6 and bool((x_train[~kept] == 0).all())
