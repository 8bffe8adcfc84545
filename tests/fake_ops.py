"""CPU stand-in for ``t5vqvae_amd.ops`` -- TEST INFRASTRUCTURE ONLY.

The host logic (launch order, statistics-buffer layout, collectives, cluster modes, autograd wiring) is
exercised without a GPU by monkeypatching ``t5vqvae_amd.vqvae.ops`` with this module inside a test.  Each
function restates the CONTRACT of the C-ABI entry point of the same name (include/t5vq.h) with torch CPU
ops.  The product never selects this module by itself.
"""
import torch

LOSS_KINDS = {"mse": 0, "cosine": 1}
KEY_NONE = 0x7FFFFFFFFFFFFFFF


def require_device(t, name):
    pass


def padded_codes(k):
    return (k + 127) // 128 * 128


def gather_blocks(n):
    return n if n < 32768 else (n + 63) // 64


def small_supported(n, k, d):
    return False          # the stand-in has no single-launch form: host-logic tests take the chain


def small_state(k, device):
    return torch.zeros(16, dtype=torch.uint8, device=device)


def row_sqnorms(x, w):
    xx = None if x is None else (x * x).sum(1)
    ee = None
    if w is not None:
        ee = torch.full((padded_codes(w.shape[0]),), float("inf"))
        ee[:w.shape[0]] = (w * w).sum(1)
    return xx, ee


def _pack(d, idx):
    b = d.contiguous().view(torch.int32).to(torch.int64)
    s = torch.where(b >= 0, b, b ^ 0x7FFFFFFF)
    return (s << 32) | (idx.to(torch.int64) & 0xFFFFFFFF)


def order_by_range(ranges, k, code_offset=0):
    """A visiting order only: any permutation gives the same keys."""
    return torch.argsort(ranges[:, 0] * (1 << 32) + ranges[:, 1], stable=True).to(torch.int32)


def dist_argmin(x, w, xx, ee, ranges=None, code_offset=0, keys=None, merge=False, order=None):
    assert order is None or (ranges is not None and sorted(order.tolist()) == list(range(x.shape[0])))
    k = w.shape[0]
    d = (xx.unsqueeze(1) + ee[:k]) - 2 * (x @ w.t())
    gid = torch.arange(k).unsqueeze(0) + code_offset
    if ranges is not None:
        ok = (gid >= ranges[:, :1]) & (gid < ranges[:, 1:2])
        d = torch.where(ok, d, torch.full_like(d, float("inf")))
        any_ok = ok.any(1)
        # first admissible code wins ties / all-inf rows
        first_ok = torch.where(ok, gid.expand_as(ok), torch.full_like(gid.expand_as(ok), 1 << 40)).min(1).values
    loc = torch.argmin(d, dim=1)
    dmin = d.gather(1, loc.unsqueeze(1)).squeeze(1)
    gi = loc + code_offset
    if ranges is not None:
        gi = torch.where(torch.isinf(dmin) & any_ok, first_ok, gi) - ranges[:, 0]
    new = _pack(dmin, gi)
    if ranges is not None:
        new = torch.where(any_ok, new, torch.full_like(new, KEY_NONE))
    if keys is None:
        return new
    if merge:
        keys.copy_(torch.minimum(keys, new))
    else:
        keys.copy_(new)
    return keys


# fp16 image blocks, as the host logic sees them: a block that holds a COPY of the rows it stands for.  A search told that
# its code image is valid searches THE IMAGE, so a stale or never-written image changes results and a test notices.
def code_image(k, d, device=None):
    return torch.full((k * d,), float("nan"))


def token_image(n, d, device=None):
    return torch.full((n * d,), float("nan"))


def _through_images(x, w, code_image, code_image_valid, token_image):
    if code_image is not None:
        if code_image_valid:
            w = code_image.view(w.shape).clone()
        else:
            code_image.copy_(w.reshape(-1))
    if token_image is not None:
        x = token_image.view(x.shape).clone()
    return x, w


def dist_argmin_filtered(x, w, xx=None, ee=None, keys=None, want_overflow=False, exact_keys=True, ranges=None,
                         code_offset=0, order=None, code_image=None, code_image_valid=False, token_image=None,
                         range_sparse=False):
    assert not code_image_valid or ee is not None
    assert token_image is None or xx is not None
    x, w = _through_images(x, w, code_image, code_image_valid, None if order is not None else token_image)
    xx_, ee_ = row_sqnorms(x, w)
    xx, ee = (xx_ if xx is None else xx), (ee_ if ee is None else ee)
    out = dist_argmin(x, w, xx, ee, ranges, code_offset=code_offset, keys=keys, order=order)
    return (out, torch.zeros(1, dtype=torch.int32)) if want_overflow else out


def unpack_keys(keys, want_idx=True, want_dmin=False):
    idx = (keys & 0xFFFFFFFF) if want_idx else None
    dmin = None
    if want_dmin:
        s = (keys >> 32).to(torch.int32)
        dmin = torch.where(s >= 0, s, s ^ 0x7FFFFFFF).view(torch.float32)
    return idx, dmin


def distances(x, w, xx=None, ee=None):
    if xx is None or ee is None:
        xx, ee = row_sqnorms(x, w)
    return (xx.unsqueeze(1) + ee[:w.shape[0]]) - 2 * (x @ w.t())


def gather_stats(x, w, keys, loss="mse", *, ste=None, q=None, diff=None, idx=None, counts=None, embed_sum=None,
                 loss_partial=None, stats_begin=0, stats_count=None, perm=None):
    n, d = x.shape
    k = w.shape[0]
    if stats_count is None:
        stats_count = k - stats_begin
    i = (keys & 0xFFFFFFFF).clamp(0, k - 1)
    qq = w.index_select(0, i)
    df = qq - x
    if ste is not None:
        ste.copy_(x + df)
    if q is not None:
        q.copy_(qq)
    if diff is not None:
        diff.copy_(df)
    if idx is not None:
        idx.copy_(i)
    local = i - stats_begin
    inside = (local >= 0) & (local < stats_count)
    if counts is not None:
        counts.copy_(torch.bincount(local[inside], minlength=stats_count).float())
    if embed_sum is not None:
        embed_sum.zero_().index_add_(0, local[inside], x[inside])
    if loss_partial is not None:
        if loss == "mse":
            per = (df * df).sum(1)
        else:
            per = 1 - (qq * x).sum(1) / (((qq * qq).sum(1) + 1e-12) * ((x * x).sum(1) + 1e-12)).sqrt()
        if perm is not None or n < 32768:
            loss_partial.copy_(per)
        else:
            pad = torch.zeros(gather_blocks(n) * 64)
            pad[:n] = per
            loss_partial.copy_(pad.view(-1, 64).sum(1))


ORDER_MIN_TOKENS = 32768


def order_by_code(keys, k):
    return torch.argsort((keys & 0xFFFFFFFF).clamp(0, k - 1), stable=True).to(torch.int32)


def forward_loss_terms(n, with_embed_sum):
    return n if with_embed_sum else gather_blocks(n)


def vq_forward(x, w, ranges, loss, *, xx=None, ee=None, ste=None, q=None, diff=None, idx=None, counts=None,
               embed_sum=None, loss_terms=None, loss_sum=None):
    xx_, ee_ = row_sqnorms(x, w)
    xx, ee = (xx_ if xx is None else xx), (ee_ if ee is None else ee)
    keys = dist_argmin(x, w, xx, ee, ranges)
    n = x.shape[0]
    perm = order_by_code(keys, w.shape[0]) if (embed_sum is not None and n >= ORDER_MIN_TOKENS) else None
    partial = loss_terms
    if partial is None and loss_sum is not None:
        partial = torch.empty(forward_loss_terms(n, embed_sum is not None))
    gather_stats(x, w, keys, loss, ste=ste, q=q, diff=diff, idx=idx, counts=counts, embed_sum=embed_sum,
                 loss_partial=partial, perm=perm)
    if loss_sum is not None:
        reduce_partials(partial, loss_sum)


def vq_forward_filtered(x, w, ranges, loss, **kw):
    vq_forward(x, w, ranges, loss, **kw)
    return torch.zeros(1, dtype=torch.int32)


def ema_finalize(cluster_size, ema_w, w, counts, embed_sum, loss_sum, n_tokens, loss_denom, decay, one_minus_decay,
                 epsilon, k_epsilon, commitment_cost, plain_variant, update, scalars, ee_out=None):
    ema_cluster(cluster_size if update else None, counts, loss_sum, counts.numel(), n_tokens, loss_denom, decay,
                one_minus_decay, epsilon, k_epsilon, commitment_cost, plain_variant, 1 if update else 0, scalars)
    if update:
        ema_embed(ema_w, w, embed_sum, cluster_size, decay, one_minus_decay, ee_out)


# ---- the module's route (t5vqvae_amd.ops.fused_*: torch.ops.t5vq in the product) ------------------------------
def stats_layout(n, k, d, with_embed_sum):
    kd = k * d if with_embed_sum else 0
    loss_off = kd + k
    terms_off = (loss_off + 1 + 3) // 4 * 4
    return kd, kd, loss_off, terms_off, forward_loss_terms(n, with_embed_sum)


def code_sqnorms(w):
    return row_sqnorms(None, w)[1]


def fused_forward(x, w, ranges, loss, *, xx=None, ee=None, filtered=False, want_diff=False, want_q=False,
                  want_embed_sum=False, reduce_loss=False, code_image=None, code_image_valid=False, token_image=None):
    if filtered:
        assert not code_image_valid or ee is not None
        assert token_image is None or xx is not None
        x, w = _through_images(x, w, code_image, code_image_valid, token_image)
    else:
        assert not code_image_valid and token_image is None
    n, d = x.shape
    k = w.shape[0]
    kd, counts_off, loss_off, terms_off, n_terms = stats_layout(n, k, d, want_embed_sum)
    stats = torch.full((terms_off + n_terms,), float("nan"))
    ste, idx = torch.empty_like(x), torch.empty(n, dtype=torch.int64)
    diff = torch.empty_like(x) if want_diff else None
    q = torch.empty_like(x) if want_q else None
    vq_forward(x, w, ranges, loss, xx=xx, ee=ee, ste=ste, q=q, diff=diff, idx=idx,
               counts=stats[counts_off:counts_off + k], embed_sum=stats[:kd].view(k, d) if kd else None,
               loss_terms=stats[terms_off:terms_off + n_terms],
               loss_sum=stats[loss_off:loss_off + 1] if reduce_loss else None)
    return ste, idx, diff, q, stats, (torch.zeros(1, dtype=torch.int32) if filtered else None)


def fused_finalize(cluster_size, ema_w, w, stats, n_local, k, d, with_embed_sum, reduced_loss, n_tokens, loss_denom,
                   decay, one_minus_decay, epsilon, k_epsilon, commitment_cost, plain_variant, update, ee_out=None,
                   code_image=None):
    kd, counts_off, loss_off, terms_off, n_terms = stats_layout(n_local, k, d, with_embed_sum)
    assert stats.numel() == terms_off + n_terms
    scalars = torch.empty(4)
    loss_in = stats[loss_off:loss_off + 1] if reduced_loss else stats[terms_off:terms_off + n_terms]
    ema_finalize(cluster_size, ema_w, w, stats[counts_off:counts_off + k], stats[:kd].view(k, d) if kd else None,
                 loss_in, n_tokens, loss_denom, decay, one_minus_decay, epsilon, k_epsilon, commitment_cost,
                 plain_variant, update, scalars, ee_out)
    if update and code_image is not None:
        code_image.copy_(w.reshape(-1))
    return scalars


def fused_step(x, w, ranges, loss, *, xx=None, ee=None, cluster_size=None, ema_w=None, filtered=False, want_diff=False,
               want_q=False, loss_denom, decay, one_minus_decay, epsilon, k_epsilon, commitment_cost, plain_variant,
               update, ee_out=None, small_state=None, code_image=None, code_image_valid=False, token_image=None):
    ste, idx, diff, q, stats, ovf = fused_forward(x, w, ranges, loss, xx=xx, ee=ee, filtered=filtered,
                                                  want_diff=want_diff, want_q=want_q, want_embed_sum=update,
                                                  code_image=code_image if filtered else None,
                                                  code_image_valid=code_image_valid, token_image=token_image)
    scalars = fused_finalize(cluster_size, ema_w, w if update else None, stats, x.shape[0], w.shape[0], x.shape[1],
                             update, False, x.shape[0], loss_denom, decay, one_minus_decay, epsilon, k_epsilon,
                             commitment_cost, plain_variant, update, ee_out, code_image)
    return ste, idx, diff, q, scalars, ovf


def rmsnorm_sqnorms(hidden, gamma, keep, scale, eps, want_rms=True, token_image=None):
    r = torch.sqrt((hidden * hidden).mean(-1, keepdim=True) + eps)
    x = gamma * (hidden / r)
    if keep is not None:
        x = torch.where(keep.bool(), x * torch.tensor(scale, dtype=torch.float32), torch.zeros_like(x))
    if token_image is not None:
        token_image.copy_(x.reshape(-1))
    return x, (x * x).sum(1), (r.squeeze(1) if want_rms else None)


def rmsnorm_backward(grad_x, hidden, gamma, keep, scale, rms, want_grad_gamma=True):
    gy = grad_x if keep is None else torch.where(keep.bool(), grad_x * scale, torch.zeros_like(grad_x))
    r = rms.unsqueeze(1)
    a = gy * gamma
    gh = a / r - hidden * ((a * hidden).sum(1, keepdim=True) / (hidden.shape[1] * r ** 3))
    gg = (gy * hidden / r).sum(0) if want_grad_gamma else None
    return gh, gg


def reduce_partials(partials, out):
    out[0] = partials.double().sum().float()


def ema_cluster(cluster_size, counts, loss_sum, k, n_tokens, loss_denom, decay, one_minus_decay, epsilon, k_epsilon,
                commitment_cost, plain_variant, mode, scalars):
    f = lambda v: torch.tensor(v, dtype=torch.float32)
    if mode != 3:
        p = counts / f(float(n_tokens))
        ent = (p * torch.log(p + 1e-10)).double().sum()
        n = f(0.0)
        if mode != 0 and cluster_size is not None:
            cluster_size.copy_(cluster_size * f(decay) + f(one_minus_decay) * counts)
            n = cluster_size.double().sum().float()
    else:
        ent = scalars[1].double().clone()
        n = scalars[2].clone()
    if mode in (1, 3) and cluster_size is not None:
        cluster_size.copy_((cluster_size + f(epsilon)) / (n + f(k_epsilon)) * n)
    m = loss_sum.double().sum().float() / f(loss_denom) if loss_sum is not None else f(0.0)
    cm = f(commitment_cost) * m
    scalars[0] = m + cm if plain_variant else cm
    scalars[1] = ent.float() if mode == 2 else torch.exp(-ent.float())
    scalars[2] = n
    scalars[3] = m


def ema_embed(ema_w, w, embed_sum, cluster_size, decay, one_minus_decay, ee_out=None, code_image=None):
    f = lambda v: torch.tensor(v, dtype=torch.float32)
    ema_w.copy_(ema_w * f(decay) + f(one_minus_decay) * embed_sum)
    w.copy_(ema_w / cluster_size.unsqueeze(1))
    if ee_out is not None:
        ee_out[:w.shape[0]] = (w * w).sum(1)
    if code_image is not None:
        code_image.copy_(w.reshape(-1))


def _fixed_shift(max_bits):
    e2 = ((int(max_bits[0]) >> 23) & 0xFF) - 127
    return 38 - ((e2 + 2) >> 1)


def max_norm(xx):
    return xx.max().reshape(1).view(torch.int32).clone()


def embed_sum_fixed(x, keys, max_bits, k, stats_begin=0, stats_count=None):
    if stats_count is None:
        stats_count = k - stats_begin
    i = (keys & 0xFFFFFFFF).clamp(0, k - 1) - stats_begin
    inside = (i >= 0) & (i < stats_count)
    q = torch.round(x.double() * 2.0 ** _fixed_shift(max_bits)).to(torch.int64)
    return torch.zeros(stats_count, x.shape[1], dtype=torch.int64).index_add_(0, i[inside], q[inside])


def ema_embed_fixed(ema_w, w, embed_sum_i64, max_bits, cluster_size, decay, one_minus_decay, ee_out=None,
                    code_image=None):
    dw = (embed_sum_i64.double() * 2.0 ** -_fixed_shift(max_bits)).float()
    ema_embed(ema_w, w, dw, cluster_size, decay, one_minus_decay, ee_out, code_image)


def ste_backward(g_out, diff, g_loss, coef):
    s = (g_loss * coef) if g_loss is not None else 0.0
    g = -diff * s
    return g if g_out is None else g + g_out


def scatter_rows(rows, idx, k, scale_dev=None, scale_host=1.0):
    out = torch.zeros(k, rows.shape[1]).index_add_(0, idx.clamp(0, k - 1), rows)
    return out * (scale_host * (float(scale_dev) if scale_dev is not None else 1.0))


def cosine_backward(x, q, g_out, g_loss, coef_x, coef_q, want_grad_q):
    gl = float(g_loss) if g_loss is not None else 0.0
    pd = (q * x).sum(1, keepdim=True)
    m1 = (q * q).sum(1, keepdim=True) + 1e-12
    m2 = (x * x).sum(1, keepdim=True) + 1e-12
    inv = 1.0 / (m1 * m2).sqrt()
    gx = (x * (pd * inv / m2) - q * inv) * (gl * coef_x)
    if g_out is not None:
        gx = gx + g_out
    gq = (q * (pd * inv / m1) - x * inv) * (gl * coef_q) if want_grad_q else None
    return gx, gq


def one_hot(idx, k):
    return torch.zeros(idx.numel(), k).scatter_(1, idx.view(-1, 1), 1)
