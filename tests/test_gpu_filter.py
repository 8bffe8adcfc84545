"""Filtered search (fp16 filter + fp32 decision, t5vq_dist_argmin_filtered_f32) against the fp32 scan
(t5vq_dist_argmin_f32): the keys must be IDENTICAL, bit for bit -- same distances, same tie rule -- on ordinary and
on degenerate inputs; in index-only mode the index word must be identical."""
import pytest
import torch

from oracle import vq_oracle as vo
from t5vqvae_amd import synth

pytestmark = pytest.mark.gpu


def both(x, w, exact=True):
    from t5vqvae_amd import ops
    xx, ee = ops.row_sqnorms(x, w)
    scan = ops.dist_argmin(x, w, xx, ee)
    filt, ovf = ops.dist_argmin_filtered(x, w, xx, ee, want_overflow=True, exact_keys=exact)
    return scan, filt, int(ovf)


@pytest.mark.parametrize("n,k,d", [(4096, 512, 768), (2880, 10000, 768), (1, 7, 64), (300, 257, 64), (1000, 130, 128),
                                   (513, 1025, 2048), (65536, 8192, 768), (255, 256, 192), (257, 255, 4096)])
def test_filtered_keys_identical_to_scan(n, k, d):
    w = synth.normal((k, d), 71, 0, device="cuda")
    x = synth.normal((n, d), 72, 0, device="cuda")
    scan, filt, ovf = both(x, w)
    assert torch.equal(scan, filt), int((scan != filt).sum())
    scan, filt, _ = both(x, w, exact=False)
    assert torch.equal(scan & 0xFFFFFFFF, filt & 0xFFFFFFFF)
    xc, _ = synth.clustered(n, w, 0.5, 73)
    scan, filt, ovf = both(xc, w)
    assert torch.equal(scan, filt) and ovf == 0            # well-separated data: the filter settles every token


def test_filtered_search_on_degenerate_codebooks():
    """Everything the bound has to survive: identical rows (every code is a candidate -> the scan decides), exact
    duplicates (lowest index), dead codes with huge norms, zero rows, tiny and huge magnitudes, mixed scales."""
    n, k, d = 1000, 600, 768
    x = synth.normal((n, d), 74, 0, device="cuda")
    w = synth.normal((k, d), 75, 0, device="cuda")
    cases = {}
    cases["all rows equal"] = (x, w[:1].repeat(k, 1).contiguous())
    wd = w.clone(); wd[300] = wd[7]; wd[599] = wd[7]; wd[9] = wd[8]
    cases["duplicates"] = (x, wd)
    wdead = w.clone(); wdead[::3] *= 4.0e5
    cases["dead codes"] = (x, wdead)
    wz = w.clone(); wz[5] = 0; xz = x.clone(); xz[11] = 0
    cases["zero rows"] = (xz, wz)
    cases["plain-variant init"] = (x, (synth.normal((k, d), 76, 0, device="cuda") * (1.0 / k)))
    cases["tiny"] = (x * 1e-18, w * 1e-18)
    cases["huge"] = (x * 1e15, w * 1e15)
    ws = w * torch.logspace(-6, 6, k, device="cuda").unsqueeze(1)
    xs = x * torch.logspace(-3, 3, n, device="cuda").unsqueeze(1)
    cases["mixed scales"] = (xs.contiguous(), ws.contiguous())
    wi = w.clone(); wi[:, ::2] *= 1e-9                      # elements far below their row maximum (fp16 denormals)
    cases["wide rows"] = (x, wi)
    for name, (xa, wa) in cases.items():
        scan, filt, ovf = both(xa, wa)
        assert torch.equal(scan, filt), (name, int((scan != filt).sum()), ovf)
        scan, filt, _ = both(xa, wa, exact=False)
        assert torch.equal(scan & 0xFFFFFFFF, filt & 0xFFFFFFFF), name
    _, _, ovf = both(*cases["all rows equal"])
    assert ovf == n                                        # nothing can be settled by bounds: all tokens take the scan
    # and the indices are what the oracle says (ties -> lowest index)
    scan, filt, _ = both(*cases["duplicates"])
    ref = vo.select_codes(vo.distances(x.cpu(), wd.cpu()), None)
    assert int(((filt & 0xFFFFFFFF).cpu() != ref).sum()) <= 1


def test_module_filter_and_scan_give_the_same_step():
    from t5vqvae_amd import VectorQuantizerEMA
    n, k, d = 8192, 4096, 768
    w, ew = synth.normal((k, d), 77, 0), synth.normal((k, d), 77, 1)
    x = synth.normal((8, n // 8, d), 78, 0).cuda()

    def run(search):
        m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).cuda()
        with torch.no_grad():
            m._embedding.weight.copy_(w), m._ema_w.copy_(ew)
        m.encodings, m.search = "indices", search
        out = [m(x) for _ in range(2)]
        return m, out
    ms, outs = run("scan")
    mf, outf = run("filter")
    ma, outa = run("auto")
    a, b = outs[0], outf[0]                                 # first call: same codebook -> everything identical
    assert torch.equal(a[3], b[3]) and torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert torch.equal(outs[0][3], outa[0][3])
    # second call: the codebooks differ in the last bits (embed_sum is summed with float atomics in both modes)
    a, b = outs[1], outf[1]
    assert int((a[3] != b[3]).sum()) <= 2
    assert float((a[0] - b[0]).abs() / a[0].abs()) < 1e-5
    assert float((ms._ema_cluster_size - mf._ema_cluster_size).abs().max()) <= 1e-6 * float(ms._ema_cluster_size.max())
    with torch.no_grad():                                   # same codebook again -> identical indices
        mf._embedding.weight.copy_(ms._embedding.weight)
    assert torch.equal(ms.get_indices(x), mf.get_indices(x))


def test_auto_backs_off_on_a_degenerate_codebook():
    import copy
    from t5vqvae_amd import VectorQuantizerEMA, vqvae as hv
    n, k, d = 8192, 4096, 768
    m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).cuda().eval()
    with torch.no_grad():
        m._embedding.weight.copy_(synth.normal((1, d), 79, 0).repeat(k, 1))
    m.encodings = "indices"
    x = synth.normal((8, n // 8, d), 80, 0).cuda()
    ref = torch.zeros(n, 1, dtype=torch.int64)              # all rows equal: first minimum = code 0
    used = []
    for _ in range(4):
        idx = m(x)[3]
        torch.cuda.synchronize()
        assert torch.equal(idx.cpu(), ref)
        used.append(hv._filter_state(m)["skip"])
    assert used[0] == 0 and max(used) > 0                  # the first call filters, the count comes back, then scan
    m2 = copy.deepcopy(m)                                   # the back-off state does not travel with the module
    assert hv._filter_state(m2)["skip"] == 0 and torch.equal(m2(x)[3].cpu(), ref)


def test_filter_tracks_the_scan_through_codebook_collapse():
    """Cold start -> EMA updates -> dead codes with norms of 1e5..1e6 and a few live ones: at every step the filtered
    search returns the scan's indices on the module's current codebook."""
    from t5vqvae_amd import VectorQuantizerEMA, ops
    n, k, d = 8192, 4096, 768
    m = VectorQuantizerEMA(k, d, 0.25, decay=0.99).cuda().train()
    with torch.no_grad():
        m._embedding.weight.copy_(synth.normal((k, d), 81, 0))
        m._ema_w.copy_(synth.normal((k, d), 81, 1))
    m.encodings = "indices"
    unsettled = []
    for step in range(10):
        x = synth.normal((8, n // 8, d), 82, step).cuda()
        flat = x.view(-1, d)
        w = m._embedding.weight.detach()
        xx, ee = ops.row_sqnorms(flat, w)
        scan = ops.dist_argmin(flat, w, xx, ee)
        filt, ovf = ops.dist_argmin_filtered(flat, w, xx, ee, want_overflow=True)
        assert torch.equal(scan, filt), (step, int((scan != filt).sum()))
        unsettled.append(int(ovf))
        m.search = "auto"
        idx = m(x)[3]                                     # training step on the same codebook (pre-update search)
        assert torch.equal(idx.squeeze(1), scan & 0xFFFFFFFF)
    assert float(m._embedding.weight.abs().max()) > 1e4    # the codebook did collapse (dead codes blew up)
    assert max(unsettled) < n // 8


def test_filter_computes_the_same_row_norms_itself():
    from t5vqvae_amd import ops
    n, k, d = 3000, 1000, 768
    x = synth.normal((n, d), 91, 0, device="cuda")
    w = synth.normal((k, d), 92, 0, device="cuda")
    xx, ee = ops.row_sqnorms(x, w)
    k1 = ops.dist_argmin_filtered(x, w, xx, ee)
    # norms that are not given are written by the call's own first pass, bit-identical to row_sqnorms; given ones
    # are inputs and stay untouched (all four combinations of the C ABI's compute_norms)
    for gx, ge in ((None, None), (xx, None), (None, ee)):
        keep_x, keep_e = xx.clone(), ee.clone()
        k2, xx2, ee2 = ops.dist_argmin_filtered(x, w, gx, ge, return_norms=True)
        assert torch.equal(k1, k2) and torch.equal(xx, xx2) and torch.equal(ee, ee2)
        assert torch.equal(xx, keep_x) and torch.equal(ee, keep_e)


@pytest.mark.parametrize("n,k,d,offset", [(4096, 2048, 768, 0), (1000, 700, 128, 0), (3000, 1024, 768, 1024),
                                          (65536, 8192, 768, 0)])
def test_filtered_keys_identical_to_scan_with_role_ranges(n, k, d, offset):
    """Disentangled form: per-token [start, end) ranges in global code ids, a codebook slice that starts at
    `offset`; ranges that straddle the slice, miss it entirely, or are empty.  Keys (slice-relative indices, 'none'
    for tokens without an admissible code) must equal the scan's."""
    from t5vqvae_amd import ops
    w = synth.normal((k, d), 93, 0, device="cuda")
    x = synth.normal((n, d), 94, 0, device="cuda")
    hi = offset + k
    roles = [(offset, offset + k // 5), (offset + k // 5, offset + k // 5 + 1), (offset + k // 5 + 1, offset + k // 2),
             (offset + k // 2, hi), (max(0, offset - 300), offset + 40), (hi - 17, hi + 500), (hi + 10, hi + 90),
             (offset + 7, offset + 7)]
    rg = synth.role_ranges(n, roles, 95, 0, device="cuda")
    order = torch.argsort(rg[:, 0] * (1 << 32) + rg[:, 1], stable=True)
    for ranges, xs in ((rg, x), (rg.index_select(0, order), x.index_select(0, order))):   # token order, role-bucketed
        xx, ee = ops.row_sqnorms(xs, w)
        scan = ops.dist_argmin(xs, w, xx, ee, ranges, code_offset=offset)
        filt, ovf = ops.dist_argmin_filtered(xs, w, xx, ee, want_overflow=True, ranges=ranges, code_offset=offset)
        assert torch.equal(scan, filt), (int((scan != filt).sum()), int(ovf))
        assert int(ovf) < n // 20                            # tokens whose range misses the slice are not rescanned
        fidx = ops.dist_argmin_filtered(xs, w, ranges=ranges, code_offset=offset, exact_keys=False)
        none = scan == 0x7FFFFFFFFFFFFFFF
        assert torch.equal(none, fidx == 0x7FFFFFFFFFFFFFFF)
        assert torch.equal((scan & 0xFFFFFFFF)[~none], (fidx & 0xFFFFFFFF)[~none])
    assert int(none.sum()) > 0                               # the empty / missing ranges were exercised

    # the same searches told to VISIT the tokens role by role (t5vq_order_by_range) instead of being handed permuted
    # rows: x, xx, ranges and keys stay in token order, the keys must not change -- nor with any other permutation
    xx, ee = ops.row_sqnorms(x, w)
    scan = ops.dist_argmin(x, w, xx, ee, rg, code_offset=offset)
    dev_order = ops.order_by_range(rg, k, offset)
    assert dev_order.dtype == torch.int32
    assert torch.equal(torch.sort(dev_order.long()).values, torch.arange(n, device="cuda"))
    a = (rg[:, 0] - offset).clamp(0, k)
    e = (rg[:, 1] - offset).clamp(0, k)
    t = (k + 127) // 128                                     # these K keep the 128-code granularity
    bucket = torch.where(e > a, (a // 128) * (t + 1) + (e + 127) // 128, torch.full_like(a, t * (t + 1)))
    assert bool((bucket[dev_order.long()].diff() >= 0).all())   # tokens of one (start, end) bucket are neighbours
    g = torch.Generator(device="cuda").manual_seed(5)
    for order in (dev_order, torch.randperm(n, device="cuda", generator=g).to(torch.int32)):
        assert torch.equal(ops.dist_argmin(x, w, xx, ee, rg, code_offset=offset, order=order), scan)
        assert torch.equal(ops.dist_argmin_filtered(x, w, xx, ee, ranges=rg, code_offset=offset, order=order), scan)
        keys2, xx2, ee2 = ops.dist_argmin_filtered(x, w, ranges=rg, code_offset=offset, order=order, return_norms=True)
        assert torch.equal(keys2, scan) and torch.equal(xx2, xx) and torch.equal(ee2, ee)   # norms land by token
    with pytest.raises(RuntimeError):
        ops.dist_argmin(x, w, xx, ee, None, order=dev_order)        # an order without ranges is a caller's mistake


@pytest.mark.parametrize("roles", [[(i * 8192, (i + 1) * 8192) for i in range(8)],
                                   [(0, 6553), (6553, 21845), (21845, 32773), (32773, 56174), (56174, 65536)],
                                   [(0, 24576), (32768, 65536)]])
def test_filtered_ranged_search_on_a_code_shard_where_most_tokens_belong_elsewhere(roles):
    """One code shard ([24576, 32768) of 65536 codes) of a grouped search: the tokens of most roles have no code here,
    the code splits of the ranged form leave workgroups without work (they write no scanner records) -- keys equal
    the scan's, in token order and visited role by role; with no token at all in the shard every key is 'none'."""
    from t5vqvae_amd import ops
    n, kl, d, off = 16384 + 77, 8192, 768, 24576
    w = synth.normal((kl, d), 96, 0, device="cuda")
    x, _ = synth.clustered(n, w, 0.5, 97)
    rg = synth.role_ranges(n, roles, 98, 0, device="cuda").contiguous()
    xx, ee = ops.row_sqnorms(x, w)
    scan = ops.dist_argmin(x, w, xx, ee, rg, code_offset=off)
    for order in (None, ops.order_by_range(rg, kl, off)):
        for exact in (True, False):
            f = ops.dist_argmin_filtered(x, w, xx, ee, ranges=rg, code_offset=off, order=order, exact_keys=exact)
            none = scan == 0x7FFFFFFFFFFFFFFF
            assert torch.equal(none, f == 0x7FFFFFFFFFFFFFFF)
            if exact:
                assert torch.equal(scan, f)
            else:
                assert torch.equal((scan & 0xFFFFFFFF)[~none], (f & 0xFFFFFFFF)[~none])
    inside = ((rg[:, 0] < off + kl) & (rg[:, 1] > off))
    assert torch.equal(inside, scan != 0x7FFFFFFFFFFFFFFF)


def test_role_order_with_more_distinct_ranges_than_a_workgroup_table_holds():
    """t5vq_order_by_range aggregates a workgroup's tokens per (start, end) bucket in a 64-entry LDS table and sends what
    does not fit straight to memory: per-token random ranges put hundreds of distinct buckets into every workgroup.  The
    order must be a permutation with the tokens of a bucket adjacent, and the ranged searches must not care."""
    from t5vqvae_amd import ops
    n, k, d = 7000, 16384, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randint(0, k, (n,), device="cuda", generator=g)
    e = torch.randint(0, k, (n,), device="cuda", generator=g)
    lo, hi = torch.minimum(a, e), torch.maximum(a, e) + torch.randint(0, 2, (n,), device="cuda", generator=g)
    rg = torch.stack([lo, hi], 1).contiguous()                      # a few empty ranges (hi == lo) among them
    order = ops.order_by_range(rg, k)
    assert torch.equal(torch.sort(order.long()).values, torch.arange(n, device="cuda"))
    shift = 7                                                       # the library's bucket granularity: 128 codes, doubled
    while True:                                                     # until the T (T + 1) + 1 buckets fit its K + 8 words
        t = (k + (1 << shift) - 1) >> shift
        if t * (t + 1) + 1 + 2 <= k + 8:
            break
        shift += 1
    aa, ee_ = lo.clamp(0, k), hi.clamp(0, k)
    bucket = torch.where(ee_ > aa, (aa >> shift) * (t + 1) + ((ee_ + (1 << shift) - 1) >> shift),
                         torch.full_like(aa, t * (t + 1)))
    assert int(bucket.unique().numel()) > 1000
    assert bool((bucket[order.long()].diff() >= 0).all())
    w = synth.normal((k, d), 12, 0, device="cuda")
    x = synth.normal((n, d), 13, 0, device="cuda")
    xx, ee = ops.row_sqnorms(x, w)
    ref = ops.dist_argmin(x, w, xx, ee, rg)
    assert torch.equal(ops.dist_argmin(x, w, xx, ee, rg, order=order), ref)
    assert torch.equal(ops.dist_argmin_filtered(x, w, xx, ee, ranges=rg, order=order), ref)


def test_disentangled_module_filter_equals_scan():
    from t5vqvae_amd import VectorQuantizerEMA
    n, k, d = 8192, 4096, 768
    w, ew = synth.normal((k, d), 96, 0), synth.normal((k, d), 96, 1)
    x = synth.normal((8, n // 8, d), 97, 0).cuda()
    srl = synth.role_ranges(n, [(0, 1000), (1000, 1500), (1500, 4096)], 98, 0, device="cuda").view(8, n // 8, 2)
    outs = {}
    for search in ("scan", "filter", "auto"):
        for mode in ("reference", "absolute"):
            m = VectorQuantizerEMA(k, d, 0.25, decay=0.99, disentanglement=True).cuda()
            with torch.no_grad():
                m._embedding.weight.copy_(w), m._ema_w.copy_(ew)
            m.encodings, m.search, m.index_mode = "indices", search, mode
            outs[(search, mode)] = m(x, srl)
    for mode in ("reference", "absolute"):
        a = outs[("scan", mode)]
        for search in ("filter", "auto"):
            b = outs[(search, mode)]
            assert torch.equal(a[3], b[3]) and torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])


def test_filter_survives_worst_case_aligned_rounding():
    """Operands built so that every fp16 rounding error has the same sign and the maximal size (elements just below
    the midpoint between two fp16 values, all products of one sign): the fp16 distances then RANK THE WRONG CODE FIRST
    by ~80 % of the error bound.  The bound must keep the right code among the candidates: keys equal the scan's."""
    from t5vqvae_amd import ops
    n, d = 512, 768
    a = 1.0 + 0.99 * 2.0 ** -11                            # rounds to 1.0 in fp16: relative error ~2^-11
    signs = (synth.uniform_int(n * d, 2, 101, 0).view(n, d).float() * 2 - 1).cuda()
    x = (signs * a).contiguous()
    w = torch.cat([x, signs], 0).contiguous()              # code i = x_i exactly; code n + i = its fp16 image
    scan, filt, ovf = both(x, w)
    assert torch.equal(scan, filt), int((scan != filt).sum())
    idx = (scan & 0xFFFFFFFF).cpu()
    tok = torch.arange(n)
    assert bool(((idx == tok) | (idx == tok + n)).all())   # one of the two near-twins, as fp32 arithmetic decides
    assert int((idx == tok).sum()) > n // 2                # mostly the exact match, which fp16 alone ranks second
    for scale_x, scale_w in ((1e-3, 1e3), (37.0, 1.0 / 37.0)):   # row scalings must not matter
        scan, filt, _ = both((x * scale_x).contiguous(), (w * scale_w).contiguous())
        assert torch.equal(scan, filt)


@pytest.mark.parametrize("n,k,d,ranged,loss", [(2880, 10000, 768, False, "mse"), (40000, 1000, 768, False, "mse"),
                                               (900, 640, 128, True, "cosine")])
def test_composite_filtered_forward_equals_the_scan_form(n, k, d, ranged, loss):
    """t5vq_vq_forward_filtered_f32 (one host call) == t5vq_vq_forward_f32: every per-token output, the counts and
    the loss sum bit for bit; embed_sum up to the order of its atomics."""
    from t5vqvae_amd import ops, synth
    dev = "cuda"
    w = synth.normal((k, d), 61, 0, device=dev)
    x, _ = synth.clustered(n, w, 0.7, 62)
    x = x.to(dev)
    rg = synth.role_ranges(n, [(0, k // 3), (k // 3, k)], 63, 0, device=dev) if ranged else None

    def bufs():
        return dict(ste=torch.empty_like(x), q=torch.empty_like(x), diff=torch.empty_like(x),
                    idx=torch.empty(n, dtype=torch.int64, device=dev), counts=torch.empty(k, device=dev),
                    embed_sum=torch.empty(k, d, device=dev), loss_sum=torch.empty(1, device=dev))
    a, b = bufs(), bufs()
    ops.vq_forward(x, w, rg, loss, **a)
    ovf = ops.vq_forward_filtered(x, w, rg, loss, **b)
    assert int(ovf) >= 0
    for name in ("ste", "q", "diff", "idx", "counts", "loss_sum"):
        assert torch.equal(a[name], b[name]), name
    den = float(a["embed_sum"].double().norm())
    assert float((a["embed_sum"].double() - b["embed_sum"].double()).norm()) <= 1e-6 * den


def test_filtered_keys_identical_to_scan_on_random_shapes():
    """Shape fuzz (fixed seed): ragged token / code counts around the 256 x 256 tile and the split planner's
    boundaries, every supported row length class, plain and role-ranged, three data regimes; run twice (the second
    pass reuses whatever the first left in the caches and in the workspace allocator).  Full keys must equal the
    scan's every time."""
    import random
    from t5vqvae_amd import ops
    rnd = random.Random(20240511)
    dims = [64, 128, 192, 256, 512, 768, 1024]
    cases = []
    for _ in range(96):
        n = rnd.choice([1, 63, 255, 256, 257, 511, 777, 2880, 3001, 5000, 9000])
        k = rnd.choice([1, 31, 255, 256, 257, 513, 1000, 2049, 5000, 10000])
        cases.append((n, k, rnd.choice(dims), rnd.random() < 0.4, rnd.randrange(3)))
    bad = []
    for rep in range(2):
        for i, (n, k, d, ranged, regime) in enumerate(cases):
            w = synth.normal((k, d), 500 + i, 0, device="cuda")
            if regime == 0:
                x = synth.normal((n, d), 600 + i, 0, device="cuda")
            elif regime == 1:
                x = synth.clustered(n, w, 0.3, 600 + i)[0].cuda()
            else:                                           # near-ties: every token sits between two codewords
                a = synth.uniform_int(n, k, 700 + i, 0, "cuda")
                b = synth.uniform_int(n, k, 700 + i, 1, "cuda")
                x = 0.5 * (w[a] + w[b]) + 1e-4 * synth.normal((n, d), 600 + i, 0, device="cuda")
            rg, off, order = None, 0, None
            if ranged:
                # the codebook is a slice [off, off + k) of a larger one: roles inside it, across its borders, beside it
                # (workgroups of the ranged filter whose code split misses their tokens' ranges write no records)
                off = (0, 0, 300, 4096)[i % 4]
                cut, top = off + max(1, k // 3), off + k
                roles = [(off, cut), (cut, top), (off, top), (cut, cut), (max(0, off - 200), off + 1),
                         (top - 1, top + 700), (top + 5, top + 900), (0, off)]
                rg = synth.role_ranges(n, roles[:4] if off == 0 else roles, 800 + i, 0, device="cuda").contiguous()
                if i % 2:
                    order = ops.order_by_range(rg, k, off)
            xx, ee = ops.row_sqnorms(x, w)
            scan = ops.dist_argmin(x, w, xx, ee, rg, code_offset=off, order=order)
            filt = ops.dist_argmin_filtered(x, w, xx, ee, ranges=rg, code_offset=off, order=order)
            if not torch.equal(scan, filt):
                bad.append((rep, n, k, d, ranged, regime, int((scan != filt).sum())))
    assert not bad, bad


def _image_reference(rows):
    """fp16 image of fp32 rows by the rule of csrc/t5vq_internal.h, in plain torch: every row scaled by the power of two
    that puts its largest magnitude into [2^14, 2^15) (no scaling for an all-zero row), rounded to nearest even."""
    m = rows.abs().amax(1)
    _, ex = torch.frexp(m)                                   # m = mant * 2^ex, mant in [0.5, 1): floor(log2 m) = ex - 1
    sexp = torch.where(m > 0, 15 - ex, torch.zeros_like(ex)).clamp(-100, 100)
    img = (rows * torch.pow(2.0, sexp.float()).unsqueeze(1)).to(torch.float16)
    return img, torch.pow(2.0, -sexp.float())


def _code_image_parts(block, k, d):
    a256 = lambda v: (v + 255) // 256 * 256
    kpad = (k + 127) // 128 * 128
    o1 = a256(2 * k * d)
    o2 = o1 + a256(4 * kpad)
    return (block[:2 * k * d].view(torch.float16).view(k, d), block[o1:o1 + 4 * kpad].view(torch.float32),
            block[o2:o2 + 4 * kpad].view(torch.float32))


@pytest.mark.parametrize("n,k,d", [(4096, 1000, 768), (300, 130, 64), (65536, 8192, 768)])
def test_images_from_the_producers_are_the_search_s_own(n, k, d):
    """The fp16 images written by the codebook update (t5vq_ema_embed_img_f32, the fixed-point form too), by a search
    asked to fill a block, and by the fused encoder tail (t5vq_rmsnorm_sqnorms_img_f32) are bit-identical to each other
    and to the rule restated in torch; searches that read them return the keys of a search that makes its own."""
    from t5vqvae_amd import ops
    dev = "cuda"
    torch.manual_seed(n + k)
    w0 = synth.normal((k, d), 11, 0, device=dev)
    w0[3] *= 1e6                                             # a dead code's magnitude
    w0[5] = 0.0                                              # an all-zero row
    x, _ = synth.clustered(n, w0, 0.6, 12)
    cs = torch.rand(k, device=dev) + 0.5
    es = synth.normal((k, d), 13, 0, device=dev)
    kpad = (k + 127) // 128 * 128

    # (1) the update emits the image of the codebook it writes
    ema_w, w = (w0 * cs.unsqueeze(1)).clone(), w0.clone()
    ee = torch.full((kpad,), float("inf"), device=dev)
    img_u = ops.code_image(k, d, dev)
    ops.ema_embed(ema_w, w, es, cs, 0.9, 0.1, ee_out=ee, code_image=img_u)
    ema2, w2 = (w0 * cs.unsqueeze(1)).clone(), w0.clone()
    ops.ema_embed(ema2, w2, es, cs, 0.9, 0.1)
    assert torch.equal(w, w2) and torch.equal(ema_w, ema2)   # the codebook itself is untouched by the extra output
    rows, inv, se = _code_image_parts(img_u, k, d)
    ref_rows, ref_inv = _image_reference(w)
    assert torch.equal(rows, ref_rows) and torch.equal(inv[:k], ref_inv)
    assert torch.equal(se[:k], ee[:k].sqrt()) and bool(torch.isinf(se[k:]).all()) and bool((inv[k:] == 1).all())

    # (2) a search told to fill a block leaves the same bytes there; keys do not depend on where the image came from
    xx = ops.row_sqnorms(x, None)[0]
    base = ops.dist_argmin_filtered(x, w, xx, ee)
    img_f = ops.code_image(k, d, dev)
    k_fill = ops.dist_argmin_filtered(x, w, xx, ee, code_image=img_f, code_image_valid=False)
    for a, b in zip(_code_image_parts(img_f, k, d), _code_image_parts(img_u, k, d)):
        assert torch.equal(a, b)
    k_read = ops.dist_argmin_filtered(x, w, xx, ee, code_image=img_u, code_image_valid=True)
    k_scan = ops.dist_argmin(x, w, xx, ee)
    assert torch.equal(base, k_scan) and torch.equal(k_fill, k_scan) and torch.equal(k_read, k_scan)
    with pytest.raises(RuntimeError):
        ops.dist_argmin_filtered(x, w, xx, None, code_image=img_u, code_image_valid=True)   # a valid image needs ee

    # (3) the fixed-point update emits it too
    mb = ops.max_norm(xx)
    esum = ops.embed_sum_fixed(x, k_scan, mb, k)
    ema3, w3, img3 = (w0 * cs.unsqueeze(1)).clone(), w0.clone(), ops.code_image(k, d, dev)
    ee3 = torch.full((kpad,), float("inf"), device=dev)
    ops.ema_embed_fixed(ema3, w3, esum, mb, cs, 0.9, 0.1, ee_out=ee3, code_image=img3)
    r3, i3, s3 = _code_image_parts(img3, k, d)
    ref3, refi3 = _image_reference(w3)
    assert torch.equal(r3, ref3) and torch.equal(i3[:k], refi3) and torch.equal(s3[:k], ee3[:k].sqrt())

    # (4) the encoder tail emits the image of the x it produces; the search that reads it returns the same keys
    if d <= 2048:
        h = synth.normal((n, d), 14, 0, device=dev)
        gamma = torch.rand(d, device=dev) + 0.5
        keep = (torch.rand(n, d, device=dev) > 0.1).to(torch.uint8)
        timg = ops.token_image(n, d, dev)
        xt, xxt, _ = ops.rmsnorm_sqnorms(h, gamma, keep, 1 / 0.9, 1e-6, token_image=timg)
        xt2, xxt2, _ = ops.rmsnorm_sqnorms(h, gamma, keep, 1 / 0.9, 1e-6)
        assert torch.equal(xt, xt2) and torch.equal(xxt, xxt2)
        tr, ti = _image_reference(xt)
        off = (2 * n * d + 255) // 256 * 256
        assert torch.equal(timg[:2 * n * d].view(torch.float16).view(n, d), tr)
        assert torch.equal(timg[off:off + 4 * n].view(torch.float32), ti)
        want = ops.dist_argmin(xt, w, xxt, ee)
        got = ops.dist_argmin_filtered(xt, w, xxt, ee, code_image=img_u, code_image_valid=True, token_image=timg)
        assert torch.equal(got, want)
        with pytest.raises(RuntimeError):
            ops.dist_argmin_filtered(xt, w, None, ee, token_image=timg)                      # a token image needs xx


def test_module_keeps_the_code_image_between_filtered_calls():
    """Training and evaluation with search='filter': a module that keeps the codebook's image (written by every update,
    read by the next search) equals one that does not, bit for bit, through state-dict loads and in-place writes; and
    the encoder tail's token image changes nothing but the work."""
    import copy
    from t5vqvae_amd import VectorQuantizerEMA
    from t5vqvae_amd import vqvae as hv
    from t5vqvae_amd.encoder_tail import EncoderTailVQ
    n, k, d = 8192, 2048, 768
    a = VectorQuantizerEMA(k, d, 0.25, decay=0.99).cuda()
    a.encodings, a.search, a.code_image_cache = "indices", "filter", "always"
    b = copy.deepcopy(a)
    b.code_image_cache = False
    w0 = synth.normal((k, d), 21, 0, device="cuda")
    for m in (a, b):
        with torch.no_grad():
            m._embedding.weight.copy_(w0), m._ema_w.copy_(w0)

    def both(x, train, call=lambda m, x: m(x)):
        outs = []
        for m in (a, b):
            m.train(train)
            outs.append(call(m, x))
        for u, v in zip(*outs):
            assert torch.equal(u, v)
        exact = a.deterministic or not train           # embed_sum by fp32 atomics: last bits depend on arrival order
        for name in ("_embedding.weight", "_ema_w", "_ema_cluster_size"):
            u, v = a.state_dict()[name], b.state_dict()[name]
            assert torch.equal(u, v) if exact else float((u - v).norm() / v.norm()) <= 1e-6, name
        if not exact:                                  # same start for the next call: its outputs must be equal again
            b.load_state_dict(a.state_dict())

    for step in range(4):
        x, _ = synth.clustered(n, a._embedding.weight.detach(), 0.7, 30 + step)
        both(x.view(8, n // 8, d), True)
        ent = hv._IMG_STATE[a]
        assert ent[1] == a._embedding.weight.data_ptr() and ent[2] == a._embedding.weight._version   # valid for the next call
    assert b not in hv._IMG_STATE
    both(x.view(8, n // 8, d), False)
    sd = {key: v.clone() for key, v in a.state_dict().items()}
    sd["_embedding.weight"] = sd["_embedding.weight"] * 1.25
    a.load_state_dict(sd), b.load_state_dict(sd)
    both(x.view(8, n // 8, d), False)
    both(x.view(8, n // 8, d), True)
    for m in (a, b):
        m._embedding.weight.data.mul_(0.5)
        m.invalidate_code_norms()
    both(x.view(8, n // 8, d), True)
    a.deterministic = b.deterministic = True               # the fixed-point update writes the image as well
    both(x.view(8, n // 8, d), True)
    both(x.view(8, n // 8, d), False)
    a.deterministic = b.deterministic = False
    # encoder tail: token image from the norm kernel
    ta, tb = EncoderTailVQ(a, hidden_size=d, dropout_rate=0.1).cuda(), EncoderTailVQ(b, hidden_size=d, dropout_rate=0.1).cuda()
    ta.token_image_min_elements = 0
    tb._wants_image = lambda *args: False
    h = synth.normal((8, n // 8, d), 40, 0, device="cuda")
    keep = (torch.rand(8, n // 8, d, device="cuda") > 0.1)
    both(h, True, call=lambda m, x: (ta if m is a else tb)(x, keep_mask=keep))
    both(h, False, call=lambda m, x: (ta if m is a else tb).eval()(x))      # (eval: the tail draws no dropout mask)

