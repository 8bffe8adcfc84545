"""bench.py's own launcher on CPU: `python bench.py --gpus 2 --dry-run` must start two fresh rank processes itself,
rendezvous (gloo here, RCCL on the GPU box), run the token-sharded step with its statistics all-reduce (and the
code-sharded form with its key / codebook exchanges) and print ONE JSON line with the contract's keys.  The kernels are
replaced by the test stand-in (tests/fake_ops.py): nothing printed here is a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config"}


def _run(*extra, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1",
                                            *extra]
    e = dict(os.environ)
    e.pop("RANK", None), e.pop("WORLD_SIZE", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks_token_sharded():
    r = _run("--gpus", "2")
    assert CONTRACT_KEYS <= set(r)
    assert r["n_gpus"] == 2 and r["rccl_world_size"] == 2 and r["dry_run"] is True
    assert r["scaling"] == "strong" and "C4" in r["config"]["workload"]        # the default for N > 1
    assert r["config"]["global_tokens"] == 2 * r["config"]["tokens_per_gpu"]
    assert r["allreduce_ms_per_step"] > 0 and r["allreduce_bytes"] == 4.0 * (128 * 64 + 128 + 1)
    assert r["value"] > 0 and r["unit"] == "vectors/s"


def test_bench_weak_scaling_and_code_sharded_workloads():
    r = _run("--gpus", "2", "--workload", "c3")
    assert r["scaling"] == "weak" and r["n_gpus"] == 2
    r = _run("--gpus", "2", "--workload", "c5")
    assert r["scaling"] == "strong" and r["config"]["codes_per_gpu"] * 2 == r["config"]["K"]
    assert set(r["grouped"]) == {"8_roles_aligned_to_shards", "5_unequal_roles_straddling_shards"}
    assert {"prefetch_ms_per_step", "sync_allgather_ms_per_step"} <= set(r["codebook_exchange"])


def test_bench_under_torch_distributed_run_does_not_spawn():
    r = _run("--gpus", "2", launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                      "--master-addr", "127.0.0.1", "--master-port", "29617"])
    assert r["n_gpus"] == 2 and r["rccl_world_size"] == 2


def test_forced_single_rank_rehearsal():
    r = _run("--gpus", "1", "--workload", "c3", env={"T5VQ_BENCH_FORCE_DIST": "1"})
    assert r["n_gpus"] == 1 and r["rccl_world_size"] == 1 and r["scaling"] == "weak"


def test_single_gpu_only_workloads_refuse_several_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "2", "--workload", "c2",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0
