"""CPU oracle for the VQ bottleneck -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the CPU baseline.  Nothing under
``t5vqvae_amd/`` imports it; the product path has no CPU fallback.

It restates, in plain PyTorch fp32 on the CPU, the algorithm of the reference's
``vqvae/vqvae.py`` (all ``file:line`` citations are relative to /root/reference):

* ``faithful_*``  -- op-for-op: dense distance matrix, dense one-hot, three matmuls, the
  same elementwise order (vqvae/vqvae.py:20-76, 99-165).  This is the "reference CPU
  path" that ``bench.py`` times beside the GPU numbers.
* ``algorithmic_*`` -- the form the HIP kernels mirror: gather instead of one-hot @ W,
  bincount instead of sum(one-hot), index_add instead of one-hot^T @ x.
* fp64 helpers to classify an index mismatch as inside / outside the near-tie band.

Pinned against outputs of the reference itself: ``tests/golden/make_golden.py`` imports
/root/reference/vqvae/vqvae.py in the build container and records G1..G10;
``tests/test_oracle_golden.py`` checks both restatements against those vectors.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F

COS_EPS = 1e-12  # ATen cosine_embedding_loss EPSILON (aten/src/ATen/native/Loss.cpp)


@dataclass
class VQState:
    """Mutable state of one quantizer.  EMA variant: vqvae/vqvae.py:86-92; plain: :13-14."""
    weight: torch.Tensor                       # _embedding.weight [K, D]
    ema_w: Optional[torch.Tensor] = None       # _ema_w [K, D]            (EMA only)
    cluster_size: Optional[torch.Tensor] = None  # _ema_cluster_size [K]  (EMA only)

    def clone(self) -> "VQState":
        c = lambda t: None if t is None else t.detach().clone()
        return VQState(c(self.weight), c(self.ema_w), c(self.cluster_size))


@dataclass
class VQOut:
    loss: torch.Tensor
    quantized: torch.Tensor
    perplexity: torch.Tensor
    indices: torch.Tensor                      # [N] int64 (slice-relative when ranges are given)
    encodings: Optional[torch.Tensor] = None   # [N, K] fp32 one-hot (faithful path only)
    counts: Optional[torch.Tensor] = None      # [K] fp32
    embed_sum: Optional[torch.Tensor] = None   # [K, D] fp32 (training EMA only)
    n: Optional[torch.Tensor] = None           # Laplace normaliser sum(cluster_size) (training EMA only)
    extra: dict = field(default_factory=dict)


# --------------------------------------------------------------------------- distances / argmin

def distances(flat: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """d = (|x|^2 + |e|^2) - 2 x.e^T, rounding order of vqvae/vqvae.py:106-108 (= :27-29, :174-176)."""
    return (torch.sum(flat ** 2, dim=1, keepdim=True)
            + torch.sum(weight ** 2, dim=1)
            - 2 * torch.matmul(flat, weight.t()))


def select_codes(d: torch.Tensor, ranges: Optional[torch.Tensor], absolute: bool = False) -> torch.Tensor:
    """First-minimum argmin per row (vqvae/vqvae.py:122).  With per-row [start, end) ranges the
    result is the position INSIDE the slice and is used unchanged as an absolute codebook row --
    the reference's behaviour at vqvae/vqvae.py:110-119 (no ``+ start``).  ``absolute=True`` is the
    product's opt-in ``index_mode="absolute"`` (start + argmin), which has no counterpart in the reference."""
    if ranges is None:
        return torch.argmin(d, dim=1)
    r = ranges.reshape(-1, 2)
    k = d.shape[1]
    col = torch.arange(k).unsqueeze(0)
    masked = torch.where((col >= r[:, :1]) & (col < r[:, 1:2]), d, torch.full_like(d, float("inf")))
    return torch.argmin(masked, dim=1) - (0 if absolute else r[:, 0])


def distances_fp64(flat: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    x = flat.double()
    w = weight.double()
    return (x * x).sum(1, keepdim=True) + (w * w).sum(1) - 2.0 * (x @ w.t())


def band_eps(flat: torch.Tensor, weight: torch.Tensor, rows: torch.Tensor, a: torch.Tensor, b: torch.Tensor,
             rel: float = 1e-6) -> tuple[torch.Tensor, torch.Tensor]:
    """For candidate code pairs (a_i, b_i) of the given rows return (|d64(a)-d64(b)|, eps_i) with
    eps_i = rel * (|x_i|^2 + max(|e_a|^2, |e_b|^2)) -- the stated near-tie band (hard part 2)."""
    x = flat[rows].double()
    ea = weight[a].double()
    eb = weight[b].double()
    da = ((x - ea) ** 2).sum(1)
    db = ((x - eb) ** 2).sum(1)
    eps = rel * ((x * x).sum(1) + torch.maximum((ea * ea).sum(1), (eb * eb).sum(1)))
    return (da - db).abs(), eps


def classify_mismatches(flat, weight, idx_test, idx_ref, ranges=None, rel: float = 1e-6):
    """Returns (n_outside_band, n_inside_band).  Indices are slice-relative when ranges are given, so they
    are shifted back to absolute rows before the fp64 distances are taken."""
    idx_test = idx_test.reshape(-1).long().cpu()
    idx_ref = idx_ref.reshape(-1).long().cpu()
    rows = torch.nonzero(idx_test != idx_ref).reshape(-1)
    if rows.numel() == 0:
        return 0, 0
    a, b = idx_test[rows], idx_ref[rows]
    if ranges is not None:
        st = ranges.reshape(-1, 2)[rows, 0].cpu()
        a, b = a + st, b + st
    gap, eps = band_eps(flat.cpu(), weight.cpu(), rows, a, b, rel)
    outside = int((gap > eps).sum())
    return outside, int(rows.numel()) - outside


# --------------------------------------------------------------------------- losses

def _cosine_sum(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """sum_i (1 - cos(a_i, b_i)) as ATen's cosine_embedding_loss(target=1, reduction='sum') forms it."""
    prod = (a * b).sum(1)
    m1 = (a * a).sum(1) + COS_EPS
    m2 = (b * b).sum(1) + COS_EPS
    return (1 - prod / (m1 * m2).sqrt()).sum()


def latent_loss(kind: str, quantized: torch.Tensor, inputs: torch.Tensor, dim: int) -> torch.Tensor:
    """One latent loss term (vqvae/vqvae.py:147-155 / :54-67).  The cosine variant divides by the
    batch size input_shape[0], not by the token count (:152)."""
    if kind == "mse":
        return F.mse_loss(quantized, inputs)
    if kind == "cosine":
        return _cosine_sum(quantized.reshape(-1, dim), inputs.reshape(-1, dim)) / torch.tensor(inputs.shape[0])
    raise SystemExit("Error: wrong loss name")  # reference: exit(...) at :67 / :155


def perplexity_from_counts(counts: torch.Tensor, n_tokens: int) -> torch.Tensor:
    """exp(-sum p log(p + 1e-10)), p = counts / N (vqvae/vqvae.py:161-162)."""
    p = counts / n_tokens
    return torch.exp(-torch.sum(p * torch.log(p + 1e-10)))


# --------------------------------------------------------------------------- EMA update

def ema_update(state: VQState, counts: torch.Tensor, embed_sum: torch.Tensor, decay: float, epsilon: float):
    """In-place restatement of vqvae/vqvae.py:132-144.  Returns n = sum(cluster_size) before smoothing."""
    k = state.weight.shape[0]
    cs = state.cluster_size * decay + (1 - decay) * counts
    n = torch.sum(cs)
    cs = (cs + epsilon) / (n + k * epsilon) * n
    state.cluster_size = cs
    state.ema_w = state.ema_w * decay + (1 - decay) * embed_sum
    state.weight = state.ema_w / cs.unsqueeze(1)
    return n


# --------------------------------------------------------------------------- forward passes

def faithful_forward(state: VQState, inputs: torch.Tensor, srl: Optional[torch.Tensor] = None, *,
                     ema: bool, training: bool, commitment_cost: float, decay: float = 0.99,
                     epsilon: float = 1e-5, disentanglement: bool = False, loss: str = "mse") -> VQOut:
    """Op-for-op CPU path.  ``state.weight`` may require grad (plain variant, vqvae/vqvae.py:55-69);
    ``inputs`` may require grad.  Mutates ``state`` when ``ema and training`` (pre-update codebook is
    used for this call's outputs, :128 precedes :131-144)."""
    dim = state.weight.shape[1]
    k = state.weight.shape[0]
    flat = inputs.view(-1, dim)
    d = distances(flat, state.weight)
    idx = select_codes(d.detach(), srl if disentanglement else None)
    enc = torch.zeros(idx.shape[0], k)
    enc.scatter_(1, idx.unsqueeze(1), 1)
    weight_used = state.weight
    quantized = torch.matmul(enc, weight_used).view(inputs.shape)
    counts = torch.sum(enc, 0)
    embed_sum = n = None
    if ema:
        if training:
            with torch.no_grad():
                embed_sum = torch.matmul(enc.t(), flat.detach())
                n = ema_update(state, counts, embed_sum, decay, epsilon)
        total = commitment_cost * latent_loss(loss, quantized.detach(), inputs, dim)
    else:
        e_lat = latent_loss(loss, quantized.detach(), inputs, dim)
        q_lat = latent_loss(loss, quantized, inputs.detach(), dim)
        total = q_lat + commitment_cost * e_lat
    ste = inputs + (quantized - inputs).detach()
    avg = torch.mean(enc, dim=0)                                    # :161
    ppl = torch.exp(-torch.sum(avg * torch.log(avg + 1e-10)))       # :162
    return VQOut(total, ste, ppl, idx, enc, counts, embed_sum, n)


def algorithmic_forward(state: VQState, inputs: torch.Tensor, srl: Optional[torch.Tensor] = None, *,
                        ema: bool, training: bool, commitment_cost: float, decay: float = 0.99,
                        epsilon: float = 1e-5, disentanglement: bool = False, loss: str = "mse",
                        indices: Optional[torch.Tensor] = None) -> VQOut:
    """Same results through gather / bincount / index_add (the structure of the HIP kernels).
    ``indices`` overrides the argmin (to evaluate everything downstream of a given assignment, e.g. the
    kernel's own indices when some rows sit inside the near-tie band)."""
    dim = state.weight.shape[1]
    k = state.weight.shape[0]
    flat = inputs.reshape(-1, dim)
    if indices is None:
        idx = select_codes(distances(flat.detach(), state.weight.detach()), srl if disentanglement else None)
    else:
        idx = indices.reshape(-1).long()
    quantized = state.weight.index_select(0, idx).view(inputs.shape)
    counts = torch.bincount(idx, minlength=k).to(torch.float32)
    embed_sum = n = None
    if ema:
        if training:
            with torch.no_grad():
                embed_sum = torch.zeros(k, dim).index_add_(0, idx, flat.detach())
                n = ema_update(state, counts, embed_sum, decay, epsilon)
        total = commitment_cost * latent_loss(loss, quantized.detach(), inputs, dim)
    else:
        e_lat = latent_loss(loss, quantized.detach(), inputs, dim)
        q_lat = latent_loss(loss, quantized, inputs.detach(), dim)
        total = q_lat + commitment_cost * e_lat
    ste = inputs + (quantized - inputs).detach()
    ppl = perplexity_from_counts(counts, idx.shape[0])
    return VQOut(total, ste, ppl, idx, None, counts, embed_sum, n)


def get_latent(state: VQState, inputs: torch.Tensor) -> torch.Tensor:
    """vqvae/vqvae.py:167-178."""
    return distances(inputs.view(-1, state.weight.shape[1]), state.weight)


def get_indices(state: VQState, inputs: torch.Tensor, srl=None, disentanglement=False) -> torch.Tensor:
    """vqvae/vqvae.py:180-205 -> [N, 1] int64."""
    d = distances(inputs.view(-1, state.weight.shape[1]), state.weight)
    return select_codes(d, srl if disentanglement else None).unsqueeze(1)


def get_hidden(state: VQState, inputs: torch.Tensor, srl=None, disentanglement=False) -> torch.Tensor:
    """vqvae/vqvae.py:207-238 -> quantized (no straight-through, no update)."""
    idx = get_indices(state, inputs, srl, disentanglement).squeeze(1)
    return state.weight.index_select(0, idx).view(inputs.shape)


# --------------------------------------------------------------------------- closed-form gradients

def topk_codes(state: VQState, inputs: torch.Tensor, top_k: int):
    """(values, indices) [N, top_k] of the top_k smallest distances per row, ascending, ties lowest index first:
    ``torch.sort(distances[dim])[:top_k]`` of the traversal (evaluate_vqvae.py:129-135) for every row."""
    d = get_latent(state, inputs)
    val, idx = torch.sort(d, dim=1, descending=False, stable=True)
    return val[:, :top_k], idx[:, :top_k]


def blended_indices(state: VQState, source_code: torch.Tensor, target_code: torch.Tensor, source_t: float,
                    target_t: float) -> torch.Tensor:
    """One step of the latent-space interpolation (evaluate_vqvae.py:186-200): argmin over codes of
    ``source_t * d(source_code, W) + target_t * d(target_code, W)``, [N, 1] int64."""
    w = state.weight
    left = distances(source_code.reshape(-1, w.shape[1]), w)
    right = distances(target_code.reshape(-1, w.shape[1]), w)
    return torch.argmin(source_t * left + target_t * right, dim=1).unsqueeze(1)


def t5_layer_norm(hidden: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """T5LayerNorm.forward, python_transformers/modeling_t5.py:200-207 (fp32 branch): RMS norm without mean
    subtraction, learned scale.  PARITY UNPINNED against the reference: python_transformers does not import here
    (ModuleNotFoundError: sacremoses), so this restatement is checked against the cited lines, against
    the upstream `transformers` T5LayerNorm the fork was cut from (to a few ulp: upstream multiplies by rsqrt) and
    against torch's dropout for the mask convention (tests/test_oracle_golden.py) -- not against the fork itself."""
    variance = hidden.to(torch.float32).pow(2).mean(-1, keepdim=True)       # :202
    x = hidden / torch.sqrt(variance + eps)                                 # :203
    return weight * x                                                       # :207


def encoder_tail(hidden: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, keep: Optional[torch.Tensor] = None,
                 p: float = 0.1) -> torch.Tensor:
    """``dropout(final_layer_norm(hidden_states))`` (modeling_t5.py:864-865; :853-854 at the mid-encoder call site)
    with the dropout mask made explicit: ``keep`` is the 0/1 mask torch would draw, survivors are scaled by
    ``1 / (1 - p)`` (torch.nn.functional.dropout).  keep=None: evaluation mode (identity)."""
    y = t5_layer_norm(hidden, weight, eps)
    if keep is None:
        return y
    return y * (keep.to(y.dtype) * torch.tensor(1.0 / (1.0 - p), dtype=y.dtype))


def grad_inputs_mse(g_q, g_loss, inputs, quantized_raw, commitment_cost):
    """d/dx of [STE output . g_q + g_loss * cc * mse(q.detach(), x)]."""
    return g_q + g_loss * commitment_cost * 2.0 * (inputs - quantized_raw) / inputs.numel()


def grad_weight_mse(g_loss, inputs, quantized_raw, idx, k):
    """Plain variant: d/dW of g_loss * mse(q, x.detach()) = scatter-add of 2 (q - x) / (N D) by idx."""
    dim = inputs.shape[-1]
    g = g_loss * 2.0 * (quantized_raw - inputs).reshape(-1, dim) / inputs.numel()
    return torch.zeros(k, dim).index_add_(0, idx.reshape(-1), g)


# --------------------------------------------------------------------------- multi-GPU restatements

def orderable_key(dmin: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """Packed signed-int64 key whose ordering is (distance, index): float bits mapped to a signed
    order-preserving int32 in the high word, index in the low word.  min() over shards of these keys
    is torch.argmin's first-minimum rule across the concatenated codebook."""
    b = dmin.contiguous().view(torch.int32).to(torch.int64)
    s = torch.where(b >= 0, b, b ^ 0x7FFFFFFF)
    return (s << 32) | (idx.to(torch.int64) & 0xFFFFFFFF)


def key_index(key: torch.Tensor) -> torch.Tensor:
    return key & 0xFFFFFFFF
