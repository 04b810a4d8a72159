"""bench.py --gpus N must stand alone: without a launcher around it the parent starts the N ranks itself, stays
off the GPU, relays rank 0's single JSON line and fails when a rank fails.  Runs with --stub (gloo, no kernels) on CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_two_ranks_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    assert abs(out["value"] - 32 * 1024 * 2 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1", "--stub"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_failing_rank_fails_the_launcher():
    r = _run(["--gpus", "2", "--steps", "1", "--stub"], {"MLSP_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")] or r.returncode != 0
