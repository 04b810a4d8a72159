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


# ---- the evidence chain of the bench line: the counter profile is chosen BY WORKLOAD and must hold the priced kernel ----
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_headline_traffic_comes_from_split_kernel_rows_of_the_headline_profile():
    """roofline.traffic of the headline is read from profiles/pmc_r<N>/summary[_vK].csv (the configs[1] step), from gemm_split_kernel rows,
    whatever other summaries lie in the same directory (round 5: summary_config4.csv sorted last and was read instead)."""
    import csv
    b = _bench_module()
    f = b.pmc_summary_file("configs1")
    assert f is not None and "config4" not in os.path.basename(f)
    rows = [r for r in csv.DictReader(open(f)) if "gemm_split_kernel" in r["kernel"]]
    assert rows, "the committed headline profile must hold gemm_split_kernel rows"
    traffic, src, n = b.pmc_gemm_traffic("gemm_split_kernel", "configs1")
    assert src == os.path.relpath(f, ROOT) and n == sum(int(r["launches"]) for r in rows)
    want = sum(int(r["launches"]) * (2 * float(r["fetch_KB_per_launch_raw"]) + float(r["write_KB_per_launch"])) * 1024 for r in rows) / n
    assert abs(traffic - want) <= 1e-9 * want
    assert 20e6 < traffic < 2e9            # tens to hundreds of MB per launch; a 20 MB figure was the round-5 defect
    kinds, ksrc = b.pmc_split_traffic_by_kind("configs1")
    assert ksrc == src and set(kinds) == {"fwd", "dgrad", "wgrad"}
    f4 = b.pmc_summary_file("configs4")
    assert f4 is not None and "config4" in os.path.basename(f4) and f4 != f


def test_profile_without_the_priced_kernel_is_an_error(tmp_path):
    b = _bench_module()
    d = tmp_path / "profiles" / "pmc_r9"
    d.mkdir(parents=True)
    hdr = "kernel,launches,fetch_KB_per_launch_raw,write_KB_per_launch\n"
    (d / "summary.csv").write_text(hdr + "void gemm_f32_kernel<false; true>,10,100,50\n")
    (d / "summary_config4.csv").write_text(hdr + "void gemm_split_kernel<false; true; 2>,10,100,50\n")      # sorts last: must NOT be picked
    import pytest
    with pytest.raises(RuntimeError, match="no row of gemm_split_kernel"):
        b.pmc_gemm_traffic("gemm_split_kernel", "configs1", root=str(tmp_path))
    t, src, n = b.pmc_gemm_traffic("gemm_f32_kernel", "configs1", root=str(tmp_path))
    assert n == 10 and src.endswith("summary.csv") and abs(t - 250 * 1024) < 1e-6
    # a newer version of the headline summary wins over the unversioned one, never a configs[4] file
    (d / "summary_v2.csv").write_text(hdr + "void gemm_split_kernel<true; false; 2>,4,10,5\n")
    t, src, n = b.pmc_gemm_traffic("gemm_split_kernel", "configs1", root=str(tmp_path))
    assert src.endswith("summary_v2.csv") and n == 4
    assert b.pmc_gemm_traffic("gemm_split_kernel", "configs1", root=str(tmp_path / "nowhere")) == (None, None, 0)
