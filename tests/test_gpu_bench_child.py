"""bench.py's north-star block as a supervised child job (round 6), end to end on the GPU: two ranks sharing the one GPU (the
dry-run transport: gloo, host-staged halos; flagged "oversubscribed"), tiny shape (BENCH_NORTH_STAR_TEST=1).  (1) the child
runs and its line is merged into `north_star`; (2) the child is killed by the wall-clock limit and the headline still
arrives.  The unit-level cases (exit code, no nesting, environment hygiene) are CPU tests in tests/test_host_logic.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, timeout=600):
    env = dict(os.environ, BENCH_NORTH_STAR_TEST="1", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TOMO_BENCH_ARGV", "TOMO_BENCH_CHILD"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--north-star", "--n", "192", "--nz", "24", "--angles", "96",
           "--os", "4", "--inner", "6", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-pmc"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_north_star_child_runs_and_is_merged():
    line, err = _run({"BENCH_NORTH_STAR_TIMEOUT": "500"})
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["oversubscribed"] is True
    ns = line["north_star"]
    assert "skipped" not in ns, ns
    assert ns["scaling"] == "strong" and ns["n_gpus"] == 2 and ns["value"] > 0 and ns["per_gpu_slices"] == 12
    assert "configs[4]" in ns["workload"] and ns["config"]["n"] == 192
    assert "[bench] headline (kept whatever the north-star child job does)" in err


def test_north_star_child_killed_headline_survives():
    line, err = _run({"BENCH_NORTH_STAR_TIMEOUT": "0.5"})
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["ms_per_step"] > 0
    ns = line["north_star"]
    assert "killed after the" in ns["skipped"] and "value" not in ns, ns
