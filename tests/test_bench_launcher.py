"""bench.py's own launcher (`python bench.py --gpus N` without a torchrun environment), exercised without a GPU:
DQP_BENCH_DRY_RUN=1 stops every rank after the rendezvous (gloo) and before any GPU call."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(DQP_BENCH_DRY_RUN="1", **env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"


def test_config_4_is_strong_scaling_and_single_rank_needs_no_launcher():
    r = run(["--config", "4"])
    assert r.returncode == 0, r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"] == 4


def test_child_failure_fails_the_parent():
    r = run(["--gpus", "2"], DQP_BENCH_DRY_RUN_FAIL_RANK="1")
    assert r.returncode != 0


def test_world_size_mismatch_is_an_error():
    r = run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
