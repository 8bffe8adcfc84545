"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares; argument
validation that returns before any device work; the modules keep the reference's Python contract."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "t5vq.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t5vq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from t5vqvae_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/t5vq.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "ctypes signature table and header disagree"
    assert lib.t5vq_version() >= 100
    assert lib.t5vq_padded_codes(10000) == 10112 and lib.t5vq_padded_codes(128) == 128
    assert lib.t5vq_gather_blocks(65) == 65 and lib.t5vq_gather_blocks(10000) == 10000 \
        and lib.t5vq_gather_blocks(65536) == 1024                 # one loss term per token / per 64-token workgroup


def test_invalid_arguments_are_rejected_before_any_launch():
    from t5vqvae_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.t5vq_row_sqnorms_f32(p, 4, p, p, 4, p, 6, None) == 1          # D % 4 != 0
    assert b"multiple of 4" in lib.t5vq_last_error()
    assert lib.t5vq_dist_argmin_f32(None, p, p, p, None, None, 4, 4, 8, 0, 0, p, None) == 1
    assert lib.t5vq_gather_stats_f32(p, p, p, 4, 4, 8, 7, None, None, None, None, 0, 4, None, None, None, None,
                                     None) == 1
    assert lib.t5vq_gather_stats_f32(p, p, p, 4, 4, 4096, 0, None, None, None, None, 0, 4, None, None, None,
                                     None, None) == 1                               # D > 2048
    assert lib.t5vq_order_by_code(p, 4, 4, None, p, None) == 1
    assert lib.t5vq_ema_cluster_f32(None, p, None, 0, 4, 4, 1.0, .99, .01, 1e-5, 4e-5, .25, 0, 9, p, None) == 1
    # composite forwards: workspace checks come before any launch
    assert lib.t5vq_vq_forward_f32(p, p, None, None, None, 4, 4, 8, 0, None, 0, None, None, None, None, None, None, None,
                                   None, None) == 1
    assert lib.t5vq_forward_filtered_workspace_bytes(4096, 512, 768) > lib.t5vq_forward_workspace_bytes(4096, 512)
    assert lib.t5vq_vq_forward_filtered_f32(p, p, None, None, None, 4, 4, 64, 0, None, 0, None, None, None, None, None,
                                            None, None, None, None, None) == 1
    assert b"workspace" in lib.t5vq_last_error()


def test_torch_extension_loads_and_registers_every_op():
    """The PyTorch-ROCm extension (csrc/t5vq_torch.cpp) loads without a GPU, registers the ops the modules call, and
    refuses CPU tensors with a clear message (no compute here)."""
    import torch
    from t5vqvae_amd import _lib
    ops = _lib.load_torch_ops()
    for name in ("vq_forward", "vq_finalize", "vq_step", "ste_backward", "scatter_rows", "cosine_backward",
                 "distances", "one_hot", "code_sqnorms"):
        assert hasattr(ops, name), name
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.code_sqnorms(torch.zeros(4, 8))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.distances(torch.zeros(4, 8), torch.zeros(3, 8))


def test_missing_library_fails_loudly(monkeypatch):
    from t5vqvae_amd import _lib
    # the loaders remember a failure (next test): every cache is patched, so that undo() restores this process's state
    for name in ("_lib", "_lib_error", "_torch_ops", "_torch_ops_error"):
        monkeypatch.setattr(_lib, name, None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libt5vq.so")
    with pytest.raises(_lib.T5VQError, match="no CPU fallback"):
        _lib.load()
    monkeypatch.undo()
    for name in ("_torch_ops", "_torch_ops_error"):
        monkeypatch.setattr(_lib, name, None)
    monkeypatch.setattr(_lib, "TORCH_LIB_PATH", "/nonexistent/libt5vq_torch.so")
    with pytest.raises(_lib.T5VQError, match="no CPU fallback"):
        _lib.load_torch_ops()


def test_failed_library_load_is_sticky_and_prompt(tmp_path):
    """A library that fails to LOAD (here: a file that is no shared object) is tried once: the second call re-raises the
    first error at once instead of calling dlopen / torch.ops.load_library again on a half-registered library, and the
    modules load both libraries when they are constructed -- never first on the autograd worker thread.  (Round 2: such
    a retry, made by a backward-only flow's worker thread, hung inside run_backward.)  Own process: the failure is
    meant to stay with the process that saw it."""
    import subprocess
    import sys
    bad = tmp_path / "libt5vq_torch.so"
    bad.write_bytes(b"not a shared object")
    code = r"""
import time, sys
from unittest import mock
import torch
from t5vqvae_amd import _lib
calls = []
real = torch.ops.load_library
def counting(path):
    calls.append(path)
    return real(path)
with mock.patch.object(torch.ops, "load_library", counting):
    errs = []
    for _ in range(2):
        t0 = time.time()
        try:
            _lib.load_torch_ops()
            sys.exit("load_torch_ops() did not raise")
        except Exception as e:
            errs.append(e)
        assert time.time() - t0 < 5.0, "not prompt"
    assert errs[0] is errs[1], "the second call must re-raise the first error"
    assert len(calls) == 1, f"load_library called {len(calls)} times"
    # the modules load eagerly: construction fails here, on this thread, with the same error
    from t5vqvae_amd import VectorQuantizerEMA
    try:
        VectorQuantizerEMA(16, 8, 0.25, decay=0.99)
        sys.exit("constructor did not raise")
    except Exception as e:
        assert e is errs[0]
    assert len(calls) == 1
print("ok")
"""
    env = dict(os.environ, T5VQ_TORCH_LIB=str(bad), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_module_contract_matches_reference_signatures():
    from t5vqvae_amd import VectorQuantizer, VectorQuantizerEMA
    # vqvae/vqvae.py:7 and :80
    assert list(inspect.signature(VectorQuantizer.__init__).parameters) == [
        "self", "num_embeddings", "embedding_dim", "commitment_cost", "disentanglement", "loss", "kwargs"]
    assert list(inspect.signature(VectorQuantizerEMA.__init__).parameters) == [
        "self", "num_embeddings", "embedding_dim", "commitment_cost", "decay", "epsilon", "disentanglement", "loss",
        "kwargs"]
    torch.manual_seed(0)
    m = VectorQuantizerEMA(num_embeddings=32, embedding_dim=8, commitment_cost=0.25, decay=0.99,
                           disentanglement=False, loss="mse", V=(0, 10), residual=(10, 32))
    assert list(m.state_dict()) == ["_ema_w", "_ema_cluster_size", "_embedding.weight"]
    assert [n for n, _ in m.named_parameters()] == ["_ema_w", "_embedding.weight"]
    assert isinstance(m._ema_w, torch.nn.Parameter) and isinstance(m._embedding, torch.nn.Embedding)
    assert m._num_embeddings == 32 and m._embedding_dim == 8 and m._epsilon == 1e-5 and m._decay == 0.99
    assert float(m._ema_cluster_size.abs().sum()) == 0.0
    assert not torch.equal(m._ema_w.data, m._embedding.weight.data)      # independent draws (:87, :92)
    p = VectorQuantizer(32, 8, 0.25, decay=0.99, loss="mse", V=(0, 10))  # decay= swallowed (train_t5.py:322)
    assert list(p.state_dict()) == ["_embedding.weight"]
    assert float(p._embedding.weight.abs().max()) <= 1 / 32
    assert m._embedding(torch.tensor([1, 2])).shape == (2, 8)            # callable nn.Embedding (t5/load_t5.py:488)


def test_no_cpu_path_and_wrong_loss_name():
    from t5vqvae_amd import VectorQuantizerEMA
    m = VectorQuantizerEMA(16, 8, 0.25, decay=0.99)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 3, 8))
    m.loss = "huber"
    with pytest.raises(SystemExit):                                      # vqvae/vqvae.py:155
        m(torch.zeros(2, 3, 8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "t5vqvae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"
