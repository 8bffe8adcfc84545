"""t5vq_vq_forward_noalloc_f32 (include/t5vq.h): the forward for callers that cannot allocate, against
t5vq_vq_forward_f32 with a caller-brought workspace.  Both run the same kernels, so indices, straight-through
output, q - x, the loss sum and the counts must be bit-identical; embed_sum is accumulated with fp32 atomics whose
arrival order differs from launch to launch, so it is compared at 1e-6 norm-wise (the bar of every EMA-buffer test
here is 1e-5).  Two calls in a row, the second with a smaller input (the static scratch keeps the first call's
contents); a shape beyond the scratch is refused with a message.  Runs last (file name): the round's end runs
pytest -x."""
import ctypes

import pytest
import torch

from t5vqvae_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _outputs(n, k, d, dev):
    return {"ste": torch.full((n, d), float("nan"), device=dev), "diff": torch.full((n, d), float("nan"), device=dev),
            "idx": torch.full((n,), -1, dtype=torch.int64, device=dev),
            "counts": torch.full((k,), float("nan"), device=dev),
            "embed_sum": torch.full((k, d), float("nan"), device=dev),
            "loss_sum": torch.full((1,), float("nan"), device=dev)}


def _run(lib, x, w, ranges, noalloc):
    n, d = x.shape
    k = w.shape[0]
    o = _outputs(n, k, d, x.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    tail = (_p(o["ste"]), None, _p(o["diff"]), _p(o["idx"]), _p(o["counts"]), _p(o["embed_sum"]), None,
            _p(o["loss_sum"]), stream)
    if noalloc:
        rc = lib.t5vq_vq_forward_noalloc_f32(_p(x), _p(w), _p(ranges), None, None, n, k, d, 0, *tail)
    else:
        nbytes = int(lib.t5vq_forward_workspace_bytes(n, k))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = lib.t5vq_vq_forward_f32(_p(x), _p(w), _p(ranges), None, None, n, k, d, 0, _p(ws), nbytes, *tail)
    _lib.check(rc, "noalloc" if noalloc else "workspace")
    torch.cuda.synchronize()
    return o


def _compare(a, b):
    for name in ("idx", "ste", "diff", "loss_sum", "counts"):
        assert torch.equal(a[name], b[name]), name
    err = float((a["embed_sum"].double() - b["embed_sum"].double()).norm() / b["embed_sum"].double().norm())
    assert err <= 1e-6, f"embed_sum differs by {err}"
    assert float(a["counts"].sum()) == a["idx"].numel()


def test_noalloc_forward_equals_workspace_forward_twice():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    d = 768
    # 1: enough tokens for the ordered statistics walk (every region of the scratch is used);
    # 2: smaller N AND smaller K, the ranged search with the tokens visited role by role (N >= 1024), no ordering by
    #    code -- over the first call's leftovers;
    # 3: ranged AND ordered walk -- the token-order region holds the role order during the search and is rewritten
    #    with the by-code order for the statistics kernel;
    # 4: ranged, fewer than 1024 tokens: no role order at all, over what case 3 left in that region;
    # 5: a small plain problem (C2's shape): the single-launch small-shape forward and its own use of the scratch;
    # 6: the same with statistics for a training step
    cases = [(40000, 1000, None), (1500, 300, [(0, 100), (100, 300), (20, 250)]),
             (36000, 900, [(0, 300), (300, 900), (100, 500)]), (700, 300, [(0, 100), (100, 300), (20, 250)]),
             (4096, 512, None), (1000, 256, None)]
    for i, (n, k, roles) in enumerate(cases):
        x = synth.normal((n, d), 70 + i, 0).to(dev)
        w = synth.normal((k, d), 70 + i, 1).to(dev)
        ranges = None if roles is None else synth.role_ranges(n, roles, 70 + i, 2).to(dev).contiguous()
        got = _run(lib, x, w, ranges, noalloc=True)
        want = _run(lib, x, w, ranges, noalloc=False)
        _compare(got, want)
    # and the first shape once more after the small one
    n, k, _ = cases[0]
    x = synth.normal((n, d), 70, 0).to(dev)
    w = synth.normal((k, d), 70, 1).to(dev)
    _compare(_run(lib, x, w, None, noalloc=True), _run(lib, x, w, None, noalloc=False))


def test_noalloc_forward_is_capturable_after_a_warm_up_call():
    """Inside HIP graph capture a hipMalloc (or any other forbidden call) fails the capture: after ONE warm-up call per
    device outside capture (include/t5vq.h: the first call looks up the scratch symbol and sets kernel attributes) the
    entry point is captured, replayed on new inputs, and equals the workspace form."""
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    d = 768
    for n, k in ((3000, 1000), (4096, 512)):
        x = synth.normal((n, d), 90, 0).to(dev)
        w = synth.normal((k, d), 90, 1).to(dev)
        _run(lib, x, w, None, noalloc=True)                      # the warm-up call, outside capture
        o = _outputs(n, k, d, dev)
        xs = torch.zeros_like(x)                                 # static input of the graph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                rc = lib.t5vq_vq_forward_noalloc_f32(_p(xs), _p(w), None, None, None, n, k, d, 0, _p(o["ste"]), None,
                                                     _p(o["diff"]), _p(o["idx"]), _p(o["counts"]), _p(o["embed_sum"]),
                                                     None, _p(o["loss_sum"]), stream)
                _lib.check(rc, "noalloc under capture")
        torch.cuda.current_stream().wait_stream(side)
        for rep in range(2):                                     # replay twice, on different inputs
            xr = synth.normal((n, d), 91 + rep, 0).to(dev)
            xs.copy_(xr)
            g.replay()
            torch.cuda.synchronize()
            _compare(o, _run(lib, xr, w, None, noalloc=False))


def test_noalloc_forward_refuses_shapes_beyond_the_scratch():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n, k, d = 1 << 20, 16384, 768          # 20 N + 8 K = 21 MB of temporaries > 4 MiB; refused before any launch
    assert int(lib.t5vq_forward_workspace_bytes(n, k)) > (4 << 20)
    x = torch.empty(16, d, device=dev)     # never dereferenced: the size check comes first
    rc = lib.t5vq_vq_forward_noalloc_f32(_p(x), _p(x), None, None, None, n, k, d, 0, None, None, None, None, None, None,
                                         None, None, None)
    assert rc == 1
    msg = lib.t5vq_last_error().decode()
    assert "T5VQ_NOALLOC_SCRATCH_BYTES" in msg and "t5vq_vq_forward_f32" in msg
