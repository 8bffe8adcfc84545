"""Golden cases G1..G10: one definition of inputs + one runner, shared by

* ``tests/golden/make_golden.py`` -- drives the REFERENCE modules (imported from /root/reference in the
  build container only) and writes ``tests/golden/*.npz``;
* ``tests/test_oracle_golden.py`` -- drives the CPU oracle through the same runner (not gpu);
* ``tests/test_gpu_golden.py``    -- drives the HIP drop-in modules through the same runner (gpu).

Inputs and initial state come from ``t5vqvae_amd.synth`` (bit-reproducible everywhere), never from a
library RNG, so only outputs are stored.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np
import torch

from t5vqvae_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_LIMIT = 1 << 18          # tensors up to 256k elements are stored whole; larger ones as checksums


@dataclass
class Case:
    name: str
    cls: str                              # "VectorQuantizerEMA" | "VectorQuantizer"
    K: int
    D: int
    shape: tuple                          # leading shape of inputs, e.g. (4, 26)
    steps: int = 1
    training: bool = True
    backward: bool = False
    loss: str = "mse"
    roles: Optional[list] = None          # list of (start, end) -> disentanglement=True
    init: str = "cold"                    # cold | clustered | ties | small
    data: str = "normal"                  # normal | clustered | c1 | ties
    methods: bool = False                 # also record get_latent / get_indices / get_hidden
    searches: bool = False                # also record the evaluation-side searches (top-k, blended argmin)
    seed: int = 0
    cc: float = 0.25
    decay: float = 0.99
    sigma: float = 0.3
    ctor_extra: dict = field(default_factory=dict)


CASES = {c.name: c for c in [
    Case("G1_ema_train3", "VectorQuantizerEMA", 512, 768, (4, 26), steps=3, seed=101),
    Case("G2_ema_eval", "VectorQuantizerEMA", 512, 768, (4, 26), training=False, seed=101),
    Case("G3_plain_bwd", "VectorQuantizer", 512, 768, (4, 26), backward=True, init="small", seed=103),
    Case("G3b_ema_bwd", "VectorQuantizerEMA", 512, 768, (4, 26), backward=True, seed=113),
    Case("G4_ema_roles2", "VectorQuantizerEMA", 512, 768, (4, 26), steps=2, roles=[(0, 100), (100, 512)],
         seed=104, ctor_extra={"V": (0, 100), "residual": (100, 512)}),
    Case("G4b_plain_roles5", "VectorQuantizer", 500, 768, (4, 26), backward=True, init="small",
         roles=[(0, 100), (100, 200), (200, 300), (300, 400), (400, 500)], seed=114),
    Case("G5_ties", "VectorQuantizerEMA", 512, 768, (4, 26), steps=2, init="ties", data="ties", seed=105),
    Case("G6_ema_cosine", "VectorQuantizerEMA", 512, 768, (4, 26), backward=True, loss="cosine", seed=106),
    Case("G6b_plain_cosine", "VectorQuantizer", 512, 768, (4, 26), backward=True, loss="cosine", init="small",
         seed=116),
    Case("G7_methods", "VectorQuantizerEMA", 512, 768, (1, 45), training=False, methods=True, seed=107,
         roles=None),
    Case("G7b_methods_roles", "VectorQuantizerEMA", 512, 768, (1, 45), training=False, methods=True, seed=117,
         roles=[(0, 100), (100, 512)]),
    Case("G8_published_shape", "VectorQuantizerEMA", 10000, 768, (64, 45), steps=2, data="c1", seed=108),
    Case("G9_clustered", "VectorQuantizerEMA", 256, 768, (16, 64), steps=4, init="clustered", data="clustered",
         seed=109),
    Case("G10_shape_1x45", "VectorQuantizerEMA", 512, 768, (1, 45), seed=110),
    Case("G11_small_dim", "VectorQuantizerEMA", 100, 64, (3, 50), steps=2, seed=111),
    Case("G12_eval_searches", "VectorQuantizerEMA", 1000, 768, (1, 45), training=False, searches=True, seed=112),
]}


# --------------------------------------------------------------------------- inputs

def initial_state(c: Case) -> dict:
    """Explicit module state (never the module's own normal_()/uniform_() draws)."""
    W = synth.normal((c.K, c.D), c.seed, 0)
    st = {}
    if c.init == "small":                               # plain-variant scale, cf. uniform_(-1/K, 1/K) at :14
        W = W * torch.tensor(0.05)
    if c.init == "ties":                                # exact duplicate rows: lowest index must win (:122)
        W[300] = W[7]
        W[450] = W[7]
        W[9] = W[8]
    st["_embedding.weight"] = W
    if c.cls == "VectorQuantizerEMA":
        if c.init == "clustered":                       # steady state: ema_w = W * cs, cs = N / K
            n = int(np.prod(c.shape))
            cs = torch.full((c.K,), n / c.K)
            st["_ema_cluster_size"] = cs
            st["_ema_w"] = W * cs.unsqueeze(1)
        else:                                           # reference cold start: independent ema_w (:87-92)
            st["_ema_cluster_size"] = torch.zeros(c.K)
            st["_ema_w"] = synth.normal((c.K, c.D), c.seed, 1)
    return st


def step_inputs(c: Case, t: int, W0: torch.Tensor):
    """(inputs [*, D] fp32, srl int64 [*, 2] | None, g_q | None, g_loss | None) for step t."""
    n = int(np.prod(c.shape))
    if c.data == "normal":
        x = synth.normal((n, c.D), c.seed, 10 + t)
    elif c.data == "c1":
        x = synth.c1_activations(n, c.D, 0.1, c.seed, 10 + t)
    elif c.data == "clustered":
        x, _ = synth.clustered(n, W0, c.sigma, c.seed, 10 + 2 * t)
    elif c.data == "ties":
        which = synth.uniform_int(n, 4, c.seed, 30 + t)
        rows = torch.tensor([7, 300, 8, 450]).index_select(0, which)
        x = W0.index_select(0, rows) + synth.normal((n, c.D), c.seed, 10 + t) * torch.tensor(0.01)
        x[::3] = W0.index_select(0, rows)[::3]          # some tokens sit exactly on a duplicated codeword
    else:
        raise ValueError(c.data)
    x = x.view(*c.shape, c.D)
    srl = None
    if c.roles is not None:
        srl = synth.role_ranges(n, c.roles, c.seed, 40 + t).view(*c.shape, 2)
    gq = gl = None
    if c.backward:
        gq = synth.normal((*c.shape, c.D), c.seed, 50 + t)
        gl = torch.tensor(1.7)
    return x, srl, gq, gl


def ctor_kwargs(c: Case) -> dict:
    kw = dict(num_embeddings=c.K, embedding_dim=c.D, commitment_cost=c.cc,
              disentanglement=c.roles is not None, loss=c.loss)
    kw["decay"] = c.decay          # also passed to the plain class, as train_t5.py:322 does
    kw.update(c.ctor_extra)        # role dict splatted into **kwargs (train_t5.py:309,320)
    return kw


# --------------------------------------------------------------------------- runner

def run_case(c: Case, factory: Callable, device: str = "cpu") -> dict:
    """Drive any implementation exposing the reference's nn.Module interface.  Returns name -> cpu tensor."""
    mod = factory(c.cls, **ctor_kwargs(c))
    st = initial_state(c)
    W0 = st["_embedding.weight"].clone()
    mod = mod.to(device)
    with torch.no_grad():
        mod._embedding.weight.data.copy_(st["_embedding.weight"])
        if c.cls == "VectorQuantizerEMA":
            mod._ema_w.data.copy_(st["_ema_w"])
            mod._ema_cluster_size.data.copy_(st["_ema_cluster_size"])
    mod.train(c.training)
    out = {}
    for t in range(c.steps):
        x, srl, gq, gl = step_inputs(c, t, W0)
        x = x.to(device)
        srl = None if srl is None else srl.to(device)
        if c.backward:
            x.requires_grad_(True)
        if c.methods:
            out[f"s{t}_latent"] = mod.get_latent(x).detach().cpu()
            out[f"s{t}_get_indices"] = mod.get_indices(x, srl).detach().cpu()
            out[f"s{t}_hidden"] = mod.get_hidden(x, srl).detach().cpu()
        if c.searches:
            out.update(run_searches(mod, x, synth.normal((*c.shape, c.D), c.seed, 70 + t).to(device), t))
        loss, q, ppl, enc = mod(x, srl)
        out[f"s{t}_loss"] = loss.detach().cpu().reshape(1)
        out[f"s{t}_perplexity"] = ppl.detach().cpu().reshape(1)
        out[f"s{t}_quantized"] = q.detach().cpu()
        assert enc.shape == (x.numel() // c.D, c.K)
        out[f"s{t}_idx"] = enc.detach().argmax(1).to(torch.int32).cpu()
        assert float(enc.detach().sum()) == enc.shape[0]
        if c.backward:
            w = mod._embedding.weight
            if w.grad is not None:
                w.grad = None
            (loss * gl.to(device) + (q * gq.to(device)).sum()).backward()
            out[f"s{t}_grad_x"] = x.grad.detach().cpu()
            if c.cls == "VectorQuantizer":
                out[f"s{t}_grad_w"] = mod._embedding.weight.grad.detach().cpu()
        if c.cls == "VectorQuantizerEMA":
            out[f"s{t}_cluster_size"] = mod._ema_cluster_size.detach().cpu().clone()
    sd = mod.state_dict()
    for k, v in sd.items():
        out["final." + k] = v.detach().cpu().clone()
    return out


SEARCH_TOP_K = 10
# not 0.5: both inputs are codebook rows there, so d(a,b) = d(b,a) makes the two candidates an exact symmetric tie
# that only rounding noise decides (in the reference as well)
BLEND_STEPS = (0.0, 0.3, 0.6, 1.0)


def run_searches(mod, x, x2, t) -> dict:
    """The two searches evaluate_vqvae.py runs on top of the module.  An implementation that offers them as
    methods (``topk_codes``, ``blended_indices``) is driven through those; the reference module is driven through
    the caller-side torch expressions the script uses (restated here from evaluate_vqvae.py:129-135 and :182-200)."""
    out = {}
    with torch.no_grad():
        if hasattr(mod, "topk_codes"):
            val, idx = mod.topk_codes(x, SEARCH_TOP_K)
        else:
            dist = mod.get_latent(x)                                   # :129
            val, idx = torch.sort(dist, dim=1, descending=False)       # :134, for every row `dim` at once
            val, idx = val[:, :SEARCH_TOP_K], idx[:, :SEARCH_TOP_K]    # :135
        out[f"s{t}_topk_val"] = val.detach().cpu()
        out[f"s{t}_topk_idx"] = idx.detach().to(torch.int32).cpu()
        src = mod(x)[1].squeeze(0)                                     # :182
        tgt = mod(x2)[1].squeeze(0)                                    # :183
        for z in BLEND_STEPS:
            source_t, target_t = 1 - z, z                              # :186-187
            if hasattr(mod, "blended_indices"):
                bi = mod.blended_indices(src, tgt, source_t, target_t)
            else:
                w = mod._embedding.weight
                left = (src ** 2).sum(1, keepdim=True) + (w ** 2).sum(1) - 2 * torch.matmul(src, w.t())    # :189
                right = (tgt ** 2).sum(1, keepdim=True) + (w ** 2).sum(1) - 2 * torch.matmul(tgt, w.t())   # :190
                bi = torch.argmin(source_t * left + target_t * right, dim=1).unsqueeze(1)                  # :191,200
            out[f"s{t}_blend{int(z * 10):02d}_idx"] = bi.detach().to(torch.int32).cpu()
    return out


# --------------------------------------------------------------------------- storage / comparison

def summarize(t: torch.Tensor) -> dict:
    """Whole tensor if small, else fp64 checksums along both axes plus a strided sample of rows."""
    a = t.detach().cpu()
    if a.numel() <= FULL_LIMIT or a.dim() < 2:
        return {"full": a.numpy()}
    m = a.reshape(-1, a.shape[-1])
    d = m.double()
    step = max(1, m.shape[0] // 64)
    return {"rowsum": d.sum(1).numpy(), "colsum": d.sum(0).numpy(), "abssum": np.array([d.abs().sum().item()]),
            "sample": m[::step].numpy(), "sample_step": np.array([step])}


def save_case(name: str, results: dict):
    flat = {}
    for k, v in results.items():
        for kk, vv in summarize(v).items():
            flat[f"{k}|{kk}"] = vv
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **flat)


def load_case(name: str) -> dict:
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    out: dict = {}
    for key in z.files:
        k, kk = key.split("|")
        out.setdefault(k, {})[kk] = z[key]
    return out


def rel_err(got: np.ndarray, exp: np.ndarray) -> float:
    """Norm-wise relative error (SURVEY 8d parity report)."""
    g = np.asarray(got, dtype=np.float64)
    e = np.asarray(exp, dtype=np.float64)
    den = np.linalg.norm(e.ravel())
    num = np.linalg.norm((g - e).ravel())
    return float(num / den) if den > 0 else float(num)


def compare(got: torch.Tensor, exp: dict, rows: Optional[np.ndarray] = None) -> float:
    """Worst norm-wise relative error of ``got`` against a stored entry.  ``rows`` (bool mask over the first
    flattened axis) restricts a full comparison to rows whose code assignment agreed."""
    g = got.detach().cpu()
    if "full" in exp:
        e = exp["full"]
        gn = g.numpy().reshape(e.shape)
        if rows is not None and e.ndim >= 2:
            gm = gn.reshape(-1, e.shape[-1])[rows]
            em = e.reshape(-1, e.shape[-1])[rows]
            return rel_err(gm, em)
        return rel_err(gn, e)
    m = g.reshape(-1, g.shape[-1])
    d = m.double()
    step = int(exp["sample_step"][0])
    # checksum deviations are scaled by the mean absolute row/column mass, not by the (cancelling) sums
    errs = [np.abs(d.sum(0).numpy() - exp["colsum"]).max() / max(exp["abssum"][0] / d.shape[1], 1e-300),
            np.abs(d.sum(1).numpy() - exp["rowsum"]).max() / max(exp["abssum"][0] / d.shape[0], 1e-300),
            rel_err(m[::step].numpy(), exp["sample"])]
    return float(max(errs))
