"""HIP path vs the CPU oracle on seeded inputs (sizes the oracle finishes in seconds), edge shapes, and
size-independent properties at the BASELINE sizes.  Module-level tests call the nn.Modules (one ``torch.ops.t5vq``
op per call, csrc/t5vq_torch.cpp, over the C ABI); the tests that name ``ops.*`` bind the C ABI directly with ctypes."""
import pytest
import torch

from oracle import vq_oracle as vo
from t5vqvae_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


def make_ema(k, d, seed, dev="cuda", **kw):
    from t5vqvae_amd import VectorQuantizerEMA
    m = VectorQuantizerEMA(k, d, 0.25, decay=0.99, **kw).to(dev)
    w, e = synth.normal((k, d), seed, 0), synth.normal((k, d), seed, 1)
    with torch.no_grad():
        m._embedding.weight.copy_(w)
        m._ema_w.copy_(e)
    m.encodings = "indices"
    return m, vo.VQState(w.clone(), e.clone(), torch.zeros(k))


def check_step(m, st, x, srl=None, training=True, disent=False, loss="mse"):
    """One forward on both sides; index band check; downstream tensors against the oracle evaluated with the
    kernel's own indices when some rows fall inside the band."""
    d = x.shape[-1]
    m.train(training)
    w_before = st.weight.clone()
    lo, q, ppl, idx = m(x.cuda(), None if srl is None else srl.cuda())
    idx = idx.squeeze(1).cpu()
    ref = vo.algorithmic_forward(st.clone(), x, srl, ema=True, training=training, commitment_cost=0.25,
                                 disentanglement=disent, loss=loss)
    outside, inside = vo.classify_mismatches(x.reshape(-1, d), w_before, idx, ref.indices,
                                             srl if disent else None)
    assert outside == 0, f"{outside} index mismatches outside the band ({inside} inside)"
    o = vo.algorithmic_forward(st, x, srl, ema=True, training=training, commitment_cost=0.25,
                               disentanglement=disent, loss=loss, indices=idx)   # mutates st like the module
    assert rel(q, o.quantized) <= TOL
    assert rel(lo, o.loss) <= TOL
    assert rel(ppl, o.perplexity) <= TOL
    if training:
        assert rel(m._ema_cluster_size, st.cluster_size) <= TOL
        assert rel(m._ema_w.data, st.ema_w) <= TOL
        assert rel(m._embedding.weight.data, st.weight) <= TOL
    return inside


@pytest.mark.parametrize("n,k,d", [(4096, 512, 768),      # C2
                                   (2880, 10000, 768),    # C1: K not a multiple of any tile
                                   (1, 7, 768),           # single token, tiny codebook
                                   (130, 129, 64),        # tile edges + small D
                                   (257, 1000, 100),      # D % 32 != 0
                                   (63, 300, 2048)])      # widest supported row
def test_ema_two_steps_vs_oracle(n, k, d):
    m, st = make_ema(k, d, 7)
    for t in range(2):
        x = synth.normal((n, d), 8, t).view(1, n, d)
        check_step(m, st, x)


def test_eval_mode_leaves_buffers_untouched():
    m, st = make_ema(512, 768, 3)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    check_step(m, st, synth.normal((4, 26, 768), 4, 0), training=False)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_disentangled_straddling_ranges():
    k, d, n = 1000, 768, 515
    roles = [(0, 130), (130, 131), (131, 700), (700, 1000), (5, 995)]
    m, st = make_ema(k, d, 5, disentanglement=True)
    for t in range(2):
        x = synth.normal((1, n, d), 6, t)
        srl = synth.role_ranges(n, roles, 6, 40 + t).view(1, n, 2)
        check_step(m, st, x, srl, disent=True)


def test_disentangled_bucketed_equals_unbucketed_and_oracle():
    """N >= 1024 takes the role-ordered route (tokens visited role by role on the device + code-tile skipping);
    bucket_roles=False takes the fine-grained entry points in token order."""
    k, d, n = 2048, 768, 3000
    roles = [(0, 256), (256, 512), (512, 1024), (1024, 1100), (1100, 2048), (100, 1500)]
    x = synth.normal((1, n, d), 16, 0)
    srl = synth.role_ranges(n, roles, 16, 41).view(1, n, 2)
    m, st = make_ema(k, d, 15, disentanglement=True)
    m.eval()
    m.bucket_roles = False
    a = m(x.cuda(), srl.cuda())
    m.bucket_roles = True
    b = m(x.cuda(), srl.cuda())
    assert torch.equal(a[3], b[3]) and torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])
    assert torch.equal(m.get_indices(x.cuda(), srl.cuda()), b[3])
    for t in range(2):
        xt = synth.normal((1, n, d), 17, t)
        srl_t = synth.role_ranges(n, roles, 17, 40 + t).view(1, n, 2)
        check_step(m, st, xt, srl_t, disent=True)


@pytest.mark.parametrize("k,offset", [(5, 0), (100, 0), (129, 7), (5000, 0), (65536, 0), (40000, 65536)])
def test_role_order_is_a_permutation_for_any_codebook_size(k, offset):
    """t5vq_order_by_range over codebook sizes on both sides of every bucket-granularity step (K < 128: one bucket
    per axis; K = 65536: 512-code buckets), random ranges incl. empty ones, ones beyond the codebook and ones before
    its first code (a code shard): always a permutation of the token ids, tokens of one (start bucket, end bucket)
    pair neighbours; the ranged search gives the same keys with it as without."""
    from t5vqvae_amd import ops
    n, d = 3000, 64
    g = torch.Generator().manual_seed(k)
    a = torch.randint(-50, k + 50, (n,), generator=g)
    b = a + torch.randint(-3, max(k // 2, 4), (n,), generator=g)
    table = torch.tensor([[0, k], [0, max(k // 3, 1)], [k // 2, k], [k + 1, k + 9]])
    pick = torch.randint(0, 4, (n,), generator=g)
    rg = torch.where((torch.arange(n) % 3 == 0).unsqueeze(1), torch.stack([a, b], 1), table[pick]) + offset
    rg = rg.clamp_min(0).contiguous().cuda()
    order = ops.order_by_range(rg, k, offset)
    assert torch.equal(torch.sort(order.long()).values, torch.arange(n, device="cuda"))
    shift = 7
    while ((k + (1 << shift) - 1) >> shift) * (((k + (1 << shift) - 1) >> shift) + 1) + 3 > k + 8:
        shift += 1
    t = (k + (1 << shift) - 1) >> shift
    lo, hi = (rg[:, 0] - offset).clamp(0, k), (rg[:, 1] - offset).clamp(0, k)
    bucket = torch.where(hi > lo, (lo >> shift) * (t + 1) + ((hi + (1 << shift) - 1) >> shift),
                         torch.full_like(lo, t * (t + 1)))
    assert bool((bucket[order.long()].diff() >= 0).all())
    if k <= 5000:
        for dd in (d, 100):          # 100: not a multiple of the 32-float chunk -> the first-generation kernel, which
            w = synth.normal((k, dd), 31, 0, device="cuda")     # takes the tokens in their own order whatever `order` says
            x = synth.normal((n, dd), 32, 0, device="cuda")
            xx, ee = ops.row_sqnorms(x, w)
            plain = ops.dist_argmin(x, w, xx, ee, rg, code_offset=offset)
            assert torch.equal(ops.dist_argmin(x, w, xx, ee, rg, code_offset=offset, order=order), plain)


def test_cosine_loss_vs_oracle():
    m, st = make_ema(512, 768, 9, loss="cosine")
    check_step(m, st, synth.normal((4, 26, 768), 10, 0), loss="cosine")


def test_exact_duplicates_pick_lowest_index():
    k, d = 640, 768
    m, st = make_ema(k, d, 11)
    with torch.no_grad():
        for dup in (130, 257, 639):          # duplicates in other tiles / lane halves / registers
            m._embedding.weight[dup] = m._embedding.weight[3]
            st.weight[dup] = st.weight[3]
        m._embedding.weight[36] = m._embedding.weight[35]
        st.weight[36] = st.weight[35]
    x = st.weight[torch.tensor([3, 130, 257, 639, 35, 36, 100])].clone().view(1, 7, d)
    m.eval()
    idx = m(x.cuda())[3].squeeze(1).cpu()
    assert idx.tolist() == [3, 3, 3, 3, 35, 35, 100]


def test_determinism_same_input_twice():
    m, _ = make_ema(2048, 768, 12)
    m.eval()
    x = synth.normal((1, 3000, 768), 13, 0).cuda()
    a = m(x)
    b = m(x)
    assert torch.equal(a[3], b[3]) and torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])


@pytest.mark.parametrize("pattern", ["uniform", "hot21", "hot1", "runs"])
@pytest.mark.parametrize("loss", ["mse", "cosine"])
def test_ordered_tokens_give_the_same_statistics(pattern, loss):
    """t5vq_order_by_code + gather_stats(perm=) against the plain walk: perm is a permutation grouped by code, every
    per-token output and the counts are equal, embed_sum within fp32 summation-order noise, the loss (summed in
    token order from per-token terms) reproducible bit for bit and within tolerance of the per-workgroup form."""
    from t5vqvae_amd import ops
    n, k, d = 40000, 1000, 768          # n not a multiple of 64, above ORDER_MIN_TOKENS
    dev = torch.device("cuda", 0)
    x = synth.normal((n, d), 41, 0, device=dev)
    w = synth.normal((k, d), 41, 1, device=dev)
    codes = {"uniform": synth.uniform_int(n, k, 41, 2, dev),
             "hot21": synth.uniform_int(n, 21, 41, 3, dev) * 40,
             "hot1": torch.full((n,), 7, dtype=torch.int64, device=dev),
             "runs": synth.uniform_int(n, k, 41, 4, dev)[torch.arange(n, device=dev) // 5]}[pattern]
    keys = codes.to(torch.int64) | (torch.arange(n, device=dev) << 32)       # some distance word; the index decides

    def run(perm):
        out = dict(ste=torch.empty_like(x), q=torch.empty_like(x), diff=torch.empty_like(x),
                   idx=torch.empty(n, dtype=torch.int64, device=dev), counts=torch.empty(k, device=dev),
                   embed_sum=torch.empty(k, d, device=dev))
        part = torch.empty(n if perm is not None else ops.gather_blocks(n), device=dev)
        ops.gather_stats(x, w, keys, loss, loss_partial=part, perm=perm, **out)
        ls = torch.empty(1, device=dev)
        ops.reduce_partials(part, ls)
        return out, ls

    perm = ops.order_by_code(keys, k)
    assert torch.equal(torch.sort(perm.long())[0], torch.arange(n, device=dev))
    pc = codes[perm.long()]
    grouped = bool((pc[1:] >= pc[:-1]).all())
    identity = torch.equal(perm.long(), torch.arange(n, device=dev))
    # spread codes are grouped; batches whose neighbours already share codes are left in place (device-side decision)
    assert (grouped and not identity) if pattern in ("uniform", "hot21") else identity
    plain, ls_plain = run(None)
    ordered, ls_ord = run(perm)
    for name in ("ste", "q", "diff", "idx", "counts"):
        assert torch.equal(plain[name], ordered[name]), name
    assert rel(ordered["embed_sum"], plain["embed_sum"]) < 1e-6
    assert rel(ls_ord, ls_plain) < 1e-6
    perm2 = ops.order_by_code(keys, k)
    _, ls_again = run(perm2)
    assert torch.equal(ls_ord, ls_again)


def oracle_slice_check(x, w, idx, ranges=None, rows=4096):
    """Indices of the first ``rows`` tokens against the oracle's distances + argmin on that slice (the band rule)."""
    xs, ws = x[:rows].float().cpu(), w.float().cpu()
    rs = None if ranges is None else ranges[:rows].cpu()
    ref = vo.select_codes(vo.distances(xs, ws), rs)
    outside, inside = vo.classify_mismatches(xs, ws, idx[:rows].cpu(), ref, rs)
    assert outside == 0, f"{outside} index mismatches outside the band ({inside} inside)"
    return inside


@pytest.mark.parametrize("search", ["scan", "filter"])
def test_full_size_c3_vs_oracle(search):
    """BASELINE config 3 (N=65536, K=8192) in the 'uniform' regime, whole step against the oracle -- once with the
    fp32 scan (the benchmark's headline search), once with the filtered search (what search="auto" takes here)."""
    n, k, d = 65536, 8192, 768
    from t5vqvae_amd import VectorQuantizerEMA
    w = synth.normal((k, d), 21, 0)
    cs = torch.full((k,), n / k)
    m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).cuda()
    with torch.no_grad():
        m._embedding.weight.copy_(w)
        m._ema_w.copy_(w * cs.unsqueeze(1))
        m._ema_cluster_size.copy_(cs)
    m.encodings = "indices"
    m.search = search
    st = vo.VQState(w.clone(), w * cs.unsqueeze(1), cs.clone())
    x, c = synth.clustered(n, w, 0.5, 22)
    inside = check_step(m, st, x.view(64, 1024, d))
    print(f"C3 ({search}) in-band mismatches:", inside)


def test_full_size_properties_c3():
    """Size-independent properties at N=65536, K=8192: exact codewords round-trip, histogram mass,
    embed-sum mass, idempotence."""
    n, k, d = 65536, 8192, 768
    m, st = make_ema(k, d, 31)
    w = st.weight
    c = synth.uniform_int(n, k, 32)
    x = w.index_select(0, c).view(1, n, d).cuda()
    m.train(True)
    loss, q, ppl, idx = m(x)
    assert torch.equal(idx.squeeze(1).cpu(), c)              # every token sits exactly on its codeword
    assert float(loss) == 0.0 and torch.equal(q, x)
    counts = torch.bincount(c, minlength=k).float()
    cs_expected = counts * (1 - 0.99)
    nsum = cs_expected.sum()
    cs_expected = (cs_expected + 1e-5) / (nsum + k * 1e-5) * nsum
    assert rel(m._ema_cluster_size, cs_expected) <= TOL
    # idempotence on the updated codebook: quantising codewords returns them
    m.eval()
    x2 = m._embedding.weight.data[:4096].clone().view(1, 4096, d)
    finite = torch.isfinite(x2).all(-1).squeeze(0)
    _, q2, _, idx2 = m(x2)
    assert torch.equal(q2[0][finite], x2[0][finite])


def test_full_size_properties_c4_share():
    """C4's per-GPU share (N = 2^20 / 8, K = 16384): exact codewords round-trip, histogram mass is conserved,
    embed sums equal counts x codeword (all members of a code are identical rows)."""
    n, k, d = 131072, 16384, 768
    from t5vqvae_amd import ops
    w = synth.normal((k, d), 91, 0, device="cuda")
    c = synth.uniform_int(n, k, 92, device="cuda")
    x = w.index_select(0, c)
    xx, ee = ops.row_sqnorms(x, w)
    keys = ops.dist_argmin(x, w, xx, ee)
    assert torch.equal(ops.unpack_keys(keys)[0], c)
    counts = torch.empty(k, device="cuda")
    es = torch.empty(k, d, device="cuda")
    q = torch.empty_like(x)
    part = torch.empty(ops.gather_blocks(n), device="cuda")
    ops.gather_stats(x, w, keys, q=q, counts=counts, embed_sum=es, loss_partial=part)
    assert torch.equal(q, x) and float(part.sum()) == 0.0
    ref_counts = torch.bincount(c, minlength=k).float()
    assert torch.equal(counts, ref_counts) and float(counts.sum()) == n
    assert rel(es, w * ref_counts.unsqueeze(1)) <= 1e-6
    # against the oracle, on a slice the CPU finishes in seconds: noisy tokens (no exact hits), both searches at full size
    xn, _ = synth.clustered(n, w.cpu(), 0.5, 93)
    xn = xn.cuda()
    xxn, _ = ops.row_sqnorms(xn, None)
    scan = ops.dist_argmin(xn, w, xxn, ee)
    filt = ops.dist_argmin_filtered(xn, w, xx=xxn, ee=ee)
    assert torch.equal(scan, filt)
    inside = oracle_slice_check(xn, w, ops.unpack_keys(scan)[0])
    print("C4 share in-band mismatches on the oracle slice:", inside)


def test_rejects_unsupported_inputs():
    from t5vqvae_amd import VectorQuantizerEMA
    m = VectorQuantizerEMA(64, 6, 0.25, decay=0.99).cuda()           # D % 4 != 0
    with pytest.raises(RuntimeError, match="multiple of 4"):
        m(torch.zeros(2, 3, 6, device="cuda"))
    m = VectorQuantizerEMA(64, 8, 0.25, decay=0.99).cuda()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 8, device="cuda", dtype=torch.float64))   # the reference path is fp32
    with pytest.raises(RuntimeError):
        m(torch.zeros(0, 3, 8, device="cuda"))                        # empty batch: nothing to quantise
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 8, device="cuda").transpose(0, 2))        # not viewable as [-1, D] (vqvae.py:103)


def test_methods_and_encodings_modes():
    m, st = make_ema(512, 768, 41)
    x = synth.normal((1, 45, 768), 42, 0)
    lat = m.get_latent(x.cuda())
    assert lat.shape == (45, 512)
    assert rel(lat, vo.get_latent(st, x)) <= TOL
    gi = m.get_indices(x.cuda())
    assert gi.shape == (45, 1) and gi.dtype == torch.int64
    assert torch.equal(gi.cpu(), vo.get_indices(st, x))
    assert torch.equal(m.get_hidden(x.cuda()).cpu(), vo.get_hidden(st, x))
    m.encodings = "dense"
    m.eval()
    enc = m(x.cuda())[3]
    assert enc.shape == (45, 512) and enc.dtype == torch.float32
    assert torch.equal(enc.argmax(1).cpu(), gi.squeeze(1).cpu()) and float(enc.sum()) == 45.0


def test_autograd_plain_and_ema():
    from t5vqvae_amd import VectorQuantizer
    k, d = 300, 768
    w = synth.normal((k, d), 51, 0) * 0.05
    x = synth.normal((2, 70, d), 52, 0)
    gq = synth.normal((2, 70, d), 53, 0)
    m = VectorQuantizer(k, d, 0.25, decay=0.99).cuda()
    with torch.no_grad():
        m._embedding.weight.copy_(w)
    xg = x.cuda().requires_grad_(True)
    loss, q, _, _ = m(xg)
    (loss * 1.3 + (q * gq.cuda()).sum()).backward()
    st = vo.VQState(w.clone().requires_grad_(True))
    xc = x.clone().requires_grad_(True)
    o = vo.faithful_forward(st, xc, ema=False, training=True, commitment_cost=0.25)
    (o.loss * 1.3 + (o.quantized * gq).sum()).backward()
    assert rel(loss, o.loss) <= TOL and rel(xg.grad, xc.grad) <= TOL
    assert rel(m._embedding.weight.grad, st.weight.grad) <= TOL
    # `loss += ...` style in-place use by a caller must work
    loss2, _, _, _ = m(xg)
    loss2 += 1.0


def test_state_dict_round_trip_and_inplace_identity():
    m, _ = make_ema(256, 768, 61)
    ids = {k: v.data_ptr() for k, v in m.state_dict().items()}
    m.train()
    m(synth.normal((1, 100, 768), 62, 0).cuda())
    sd = m.state_dict()
    assert set(sd) == {"_embedding.weight", "_ema_w", "_ema_cluster_size"}
    assert all(sd[k].data_ptr() == ids[k] for k in sd)       # updated in place, not re-bound
    from t5vqvae_amd import VectorQuantizerEMA
    m2 = VectorQuantizerEMA(256, 768, 0.25, decay=0.99).cuda()
    m2.load_state_dict({k: v.cpu() for k, v in sd.items()})
    x = synth.normal((1, 50, 768), 63, 0).cuda()
    m.eval(), m2.eval()
    assert torch.equal(m(x)[1], m2(x)[1])


@pytest.mark.parametrize("n,k,d,ranged,loss", [(2880, 10000, 768, False, "mse"), (777, 130, 64, True, "mse"),
                                               (4096, 512, 768, False, "cosine")])
def test_composite_calls_equal_fine_grained(n, k, d, ranged, loss):
    """t5vq_vq_forward_f32 / t5vq_vq_ema_finalize_f32 (one host call each) == the chain of fine-grained entry
    points, bit for bit (atomics aside: embed_sum to 1e-6)."""
    from t5vqvae_amd import ops
    dev = "cuda"
    w = synth.normal((k, d), 5, 0, device=dev)
    x, _ = synth.clustered(n, w, 0.7, 6)
    rg = synth.role_ranges(n, [(0, k // 3), (k // 3, k)], 7, 0, device=dev) if ranged else None

    def bufs():
        return dict(ste=torch.empty_like(x), diff=torch.empty_like(x), idx=torch.empty(n, dtype=torch.int64, device=dev),
                    counts=torch.empty(k, device=dev), embed_sum=torch.empty(k, d, device=dev))
    a, b = bufs(), bufs()
    ls_a, ls_b = torch.empty(1, device=dev), torch.empty(1, device=dev)
    ops.vq_forward(x, w, rg, loss, loss_sum=ls_a, **a)
    xx, ee = ops.row_sqnorms(x, w)
    keys = ops.dist_argmin(x, w, xx, ee, rg)
    part = torch.empty(ops.gather_blocks(n), device=dev)
    ops.gather_stats(x, w, keys, loss, loss_partial=part, **b)
    ops.reduce_partials(part, ls_b)
    for name in ("ste", "diff", "idx", "counts"):
        assert torch.equal(a[name], b[name]), name
    assert torch.equal(ls_a, ls_b)
    assert rel(a["embed_sum"], b["embed_sum"]) < 1e-6

    cs0 = torch.rand(k, device=dev) + 0.5
    ew0 = synth.normal((k, d), 8, 0, device=dev)
    args = (n, float(n * d), 0.99, 1 - 0.99, 1e-5, k * 1e-5, 0.25)
    cs_a, ew_a, w_a, sc_a = cs0.clone(), ew0.clone(), w.clone(), torch.empty(4, device=dev)
    ops.ema_finalize(cs_a, ew_a, w_a, b["counts"], b["embed_sum"], ls_b, *args, False, True, sc_a)
    cs_b, ew_b, w_b, sc_b = cs0.clone(), ew0.clone(), w.clone(), torch.empty(4, device=dev)
    ops.ema_cluster(cs_b, b["counts"], ls_b, k, *args, False, 1, sc_b)
    ops.ema_embed(ew_b, w_b, b["embed_sum"], cs_b, 0.99, 1 - 0.99)
    for u, v in ((cs_a, cs_b), (ew_a, ew_b), (w_a, w_b), (sc_a, sc_b)):
        assert torch.equal(u, v)
    # eval form: no update, buffers untouched
    sc_e = torch.empty(4, device=dev)
    ops.ema_finalize(None, None, None, b["counts"], None, ls_b, *args, False, False, sc_e)
    assert torch.equal(sc_e[:2], sc_b[:2]) and float(sc_e[2]) == 0.0


@pytest.mark.parametrize("disent", [False, True])
def test_module_step_replays_from_a_hip_graph(disent):
    """The C ABI only enqueues on the stream it is given and allocates nothing, so a whole training-mode call can be
    captured once into a HIP graph and replayed: the replayed steps equal eager steps on a twin module.  The
    disentangled form (N >= 1024) also orders the tokens by role inside the captured call."""
    n, k, d = 2880, 1000, 768
    m1, st = make_ema(k, d, 11, disentanglement=disent)
    m2, _ = make_ema(k, d, 11, disentanglement=disent)
    m1.encodings = m2.encodings = "indices"
    x = synth.normal((n, d), 12, 0, device="cuda").view(64, 45, d)
    srl = None
    if disent:
        srl = synth.role_ranges(n, [(0, 300), (300, 1000), (100, 600)], 12, 1, device="cuda").view(64, 45, 2)
    xs = x.clone()
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m1(xs, srl)                             # warm-up call outside the capture (first-launch set-up)
        torch.cuda.current_stream().wait_stream(side)
        m2(x, srl)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m1(xs, srl)
        for step in range(3):
            xs.copy_(x * (1.0 + 0.1 * step))
            g.replay()
            ref = m2(x * (1.0 + 0.1 * step), srl)
            torch.cuda.synchronize()
            assert torch.equal(out[3], ref[3])
            # embed_sum is accumulated with float atomics, so the two codebooks agree to rounding, not bitwise
            assert rel(out[1], ref[1]) < 1e-6 and rel(out[0], ref[0]) < 1e-6 and rel(out[2], ref[2]) < 1e-6
            assert rel(m1._embedding.weight.data, m2._embedding.weight.data) < 1e-6
            assert rel(m1._ema_cluster_size, m2._ema_cluster_size) < 1e-7


@pytest.mark.parametrize("n,k,d,topk", [(45, 10000, 768, 10), (300, 257, 64, 257), (2880, 1000, 768, 3)])
def test_topk_codes_vs_oracle(n, k, d, topk):
    """top-k search without the N x K matrix == stable sort of the oracle's distance rows (ties lowest index first);
    duplicated codebook rows make exact ties certain."""
    m, st = make_ema(k, d, 21)
    with torch.no_grad():
        m._embedding.weight[k // 2] = m._embedding.weight[3]
        m._embedding.weight[k - 1] = m._embedding.weight[3]
    st.weight[k // 2] = st.weight[3]
    st.weight[k - 1] = st.weight[3]
    x = synth.normal((n, d), 22, 0)
    val, idx = m.topk_codes(x.cuda(), topk)
    assert val.shape == (n, topk) and idx.shape == (n, topk)
    val, idx = val.cpu(), idx.cpu()
    d64 = vo.distances_fp64(x, st.weight)
    # order: the kernel's own distances must be ascending, ties by index; values must be the distances of idx
    assert bool((val[:, 1:] >= val[:, :-1]).all())
    tie = val[:, 1:] == val[:, :-1]
    assert bool((idx[:, 1:][tie] > idx[:, :-1][tie]).all())
    assert len(set(idx[0].tolist())) == topk
    assert rel(val, d64.gather(1, idx)) <= TOL
    # membership: every returned code is within the band of the oracle's top-k threshold
    ref_val, ref_idx = vo.topk_codes(st, x, topk)
    band = 1e-6 * ((x.double() ** 2).sum(1) + (st.weight.double() ** 2).sum(1).max())
    kth = torch.sort(d64, dim=1).values[:, topk - 1]
    assert bool((d64.gather(1, idx) <= (kth + band).unsqueeze(1)).all())
    # rank by rank: a code different from the oracle's must be a near-tie of it (fp64 distances within the band)
    gap = (d64.gather(1, idx) - d64.gather(1, ref_idx)).abs()
    assert bool((gap[idx != ref_idx] <= band.unsqueeze(1).expand_as(gap)[idx != ref_idx]).all())
    assert int((idx != ref_idx).sum()) <= 0.01 * idx.numel() + 2
    if topk == k:                                         # a full sort is a permutation of all codes
        assert bool((torch.sort(idx, dim=1).values == torch.arange(k)).all())


def test_blended_indices_vs_oracle():
    n, k, d = 45, 10000, 768
    m, st = make_ema(k, d, 23)
    a, b = synth.normal((n, d), 24, 0), synth.normal((n, d), 24, 1)
    for z in (0.0, 0.1, 0.5, 0.9, 1.0):
        got = m.blended_indices(a.cuda(), b.cuda(), 1 - z, z).cpu()
        ref = vo.blended_indices(st, a, b, 1 - z, z)
        assert got.shape == (n, 1) and got.dtype == torch.int64
        bad = (got != ref).nonzero()[:, 0]
        if len(bad):                                       # accept only near-ties of the blended distance
            dl, dr = vo.distances_fp64(a, st.weight), vo.distances_fp64(b, st.weight)
            bl = (1 - z) * dl + z * dr
            gap = (bl.gather(1, got) - bl.gather(1, ref)).abs().squeeze(1)[bad]
            band = 1e-6 * ((a.double() ** 2).sum(1) + (b.double() ** 2).sum(1) + (st.weight.double() ** 2).sum(1).max())
            assert bool((gap <= band[bad]).all())


@pytest.mark.parametrize("n", [300, 4096])      # below / above the role-bucketing threshold
def test_index_mode_absolute_vs_oracle(n):
    """Opt-in index_mode="absolute" (start + slice argmin as the row; no counterpart in the reference) against the
    oracle evaluated with that convention; the default mode on the same inputs stays slice-relative."""
    k, d = 1000, 768
    m, st = make_ema(k, d, 61, disentanglement=True)
    m.eval()
    x = synth.normal((1, n, d), 62, 0)
    srl = synth.role_ranges(n, [(0, 200), (200, 201), (201, 640), (640, 1000)], 63, 0).view(1, n, 2)
    dist = vo.distances(x.view(-1, d), st.weight)
    abs_ref = vo.select_codes(dist, srl, absolute=True)
    rel_ref = vo.select_codes(dist, srl)
    rel_got = m(x.cuda(), srl.cuda())[3].squeeze(1).cpu()
    m.index_mode = "absolute"
    lo, q, ppl, idx = m(x.cuda(), srl.cuda())
    idx = idx.squeeze(1).cpu()
    starts = srl.view(-1, 2)[:, 0]
    assert torch.equal(idx, rel_got + starts)                     # same search, two conventions
    assert int((idx != abs_ref).sum()) <= 1 and int((rel_got != rel_ref).sum()) <= 1   # near-ties only
    assert bool(((idx >= starts) & (idx < srl.view(-1, 2)[:, 1])).all())
    o = vo.algorithmic_forward(st.clone(), x, srl, ema=True, training=False, commitment_cost=0.25,
                               disentanglement=True, indices=idx)
    assert rel(q, o.quantized) <= TOL and rel(lo, o.loss) <= TOL and rel(ppl, o.perplexity) <= TOL
    assert torch.equal(m.get_indices(x.cuda(), srl.cuda()).squeeze(1).cpu(), idx)


def test_deterministic_statistics_are_order_independent_and_exact():
    """module.deterministic = True: EMA buffers bit-identical from run to run, bit-identical whether the statistics are
    formed in one piece or as the integer sum of two token shards, and within 1e-6 of the oracle (the fixed-point sum
    is the exact sum rounded once; the oracle's fp32 index_add carries accumulation error)."""
    from t5vqvae_amd import VectorQuantizerEMA, ops
    n, k, d = 6000, 300, 768
    w, ew = synth.normal((k, d), 111, 0), synth.normal((k, d), 111, 1)
    xs = [synth.normal((6, n // 6, d), 112, t) * (1.0 + 3.0 * t) for t in range(3)]

    def run():
        m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).cuda().train()
        with torch.no_grad():
            m._embedding.weight.copy_(w), m._ema_w.copy_(ew)
        m.encodings, m.deterministic = "indices", True
        for x in xs:
            m(x.cuda())
        return m
    a, b = run(), run()
    for name, t in a.state_dict().items():
        assert torch.equal(t, b.state_dict()[name]), name
    st = vo.VQState(w.clone(), ew.clone(), torch.zeros(k))
    for x in xs:
        vo.algorithmic_forward(st, x, ema=True, training=True, commitment_cost=0.25)
    assert rel(a._ema_w.data, st.ema_w) <= 1e-6 and rel(a._ema_cluster_size, st.cluster_size) <= 1e-6
    # shards add exactly
    flat = xs[0].cuda().view(-1, d)
    wq = w.cuda()
    xx, ee = ops.row_sqnorms(flat, wq)
    keys = ops.dist_argmin(flat, wq, xx, ee)
    mb = ops.max_norm(xx)
    full = ops.embed_sum_fixed(flat, keys, mb, k)
    h = n // 2 + 77
    parts = ops.embed_sum_fixed(flat[:h].contiguous(), keys[:h].contiguous(), mb, k) \
        + ops.embed_sum_fixed(flat[h:].contiguous(), keys[h:].contiguous(), mb, k)
    assert torch.equal(full, parts)
    idx = (keys & 0xFFFFFFFF).cpu()
    exact = torch.zeros(k, d, dtype=torch.float64).index_add_(0, idx, xs[0].view(-1, d).double())
    shift = 38 - ((((int(mb) >> 23) & 0xFF) - 127 + 2) >> 1)
    got = full.cpu().double() * 2.0 ** -shift
    assert float((got - exact).abs().max()) <= n * 2.0 ** -shift          # each term rounded to 2^-shift once
    assert rel(got, exact) < 1e-9
    # slices for code-sharded codebooks
    sl = ops.embed_sum_fixed(flat, keys, mb, k, stats_begin=100, stats_count=50)
    assert torch.equal(sl, full[100:150])
