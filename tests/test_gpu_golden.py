"""HIP drop-in modules vs the golden vectors recorded from the reference (tests/golden/*.npz).

Bar (north star): codeword indices exact, except rows whose two candidates lie within the stated band
eps_i = 1e-6 * (|x_i|^2 + max |e|^2) in fp64 (count reported, must be 0 outside); every tensor / scalar
within 1e-5 norm-wise relative error.
"""
import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import vq_oracle as vo
from oracle_module import factory as oracle_factory

pytestmark = pytest.mark.gpu
TOL = 1e-5


def hip_factory(cls, **kw):
    import t5vqvae_amd
    return getattr(t5vqvae_amd, cls)(**kw)


def _state_before_step(case, t):
    """Codebook the reference used at step t: replay the (golden-pinned) oracle for t steps."""
    mod = oracle_factory("faithful")(case.cls, **gc.ctor_kwargs(case))
    st = gc.initial_state(case)
    w0 = st["_embedding.weight"].clone()
    with torch.no_grad():
        mod._embedding.weight.data.copy_(st["_embedding.weight"])
        if case.cls == "VectorQuantizerEMA":
            mod._ema_w.data.copy_(st["_ema_w"])
            mod._ema_cluster_size.data.copy_(st["_ema_cluster_size"])
    mod.train(case.training)
    for s in range(t):
        x, srl, _, _ = gc.step_inputs(case, s, w0)
        with torch.no_grad():
            mod(x, srl)
    return mod._embedding.weight.detach().clone(), w0


@pytest.mark.parametrize("name", sorted(gc.CASES))
def test_hip_matches_reference_golden(name):
    case = gc.CASES[name]
    got = gc.run_case(case, hip_factory, "cuda")
    exp = gc.load_case(name)
    assert set(got) == set(exp)
    diverged = False
    report = []
    for t in range(case.steps):
        key = f"s{t}_idx"
        g, e = got[key].numpy().ravel(), exp[key]["full"].ravel()
        nmis = int((g != e).sum())
        if nmis and not diverged:
            w_t, w0 = _state_before_step(case, t)
            x, srl, _, _ = gc.step_inputs(case, t, w0)
            outside, inside = vo.classify_mismatches(x.reshape(-1, case.D), w_t, torch.from_numpy(g),
                                                     torch.from_numpy(e), srl)
            report.append((t, inside, outside))
            assert outside == 0, f"{name} step {t}: {outside} index mismatches OUTSIDE the near-tie band"
            diverged = True   # later steps start from a (legitimately) different codebook
    if diverged:
        pytest.skip(f"{name}: in-band index mismatches {report}; downstream tensors checked in test_gpu_parity")
    for key in sorted(exp):
        if key.endswith("_topk_idx") or "_blend" in key:      # evaluation-side searches: exact
            assert np.array_equal(got[key].numpy().ravel(), exp[key]["full"].ravel()), key
            continue
        if key.endswith("_idx"):
            continue
        if key.endswith("_get_indices"):
            assert np.array_equal(got[key].numpy().ravel(), exp[key]["full"].ravel()), key
            continue
        err = gc.compare(got[key], exp[key])
        assert err <= TOL, (name, key, err)
