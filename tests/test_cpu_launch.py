"""`script --gpus N` without an outer torchrun (slam_amd/launch.py, used by bench.py and tools/run_streams.py): the
parent starts N rank processes with distinct RANK / LOCAL_RANK, relays exactly rank 0's JSON line, and reports failure."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "rank_echo.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_gpus_n_spawns_n_distinct_ranks_and_relays_one_line():
    for n in (2, 3):
        out = subprocess.run([sys.executable, HELPER, "--gpus", str(n)], capture_output=True, text=True, timeout=180, env=_clean_env())
        assert out.returncode == 0, out.stderr
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, out.stdout  # rank 0's line only
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["ranks"] == list(range(n)) and d["local_rank"] == 0


def test_single_gpu_does_not_spawn():
    out = subprocess.run([sys.executable, HELPER, "--gpus", "1"], capture_output=True, text=True, timeout=60, env=_clean_env())
    assert out.returncode == 0 and json.loads(out.stdout)["n_gpus"] == 1


def test_a_failing_rank_fails_the_launch():
    out = subprocess.run([sys.executable, HELPER, "--gpus", "2", "--fail-rank", "1"], capture_output=True, text=True, timeout=180, env=_clean_env())
    assert out.returncode == 7


def test_under_an_outer_launcher_the_script_is_a_rank():
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, HELPER, "--gpus", "4"], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode == 0 and json.loads(out.stdout)["n_gpus"] == 1  # WORLD_SIZE from the launcher wins, nothing is spawned


def test_bench_and_run_streams_use_the_launcher():
    for f in ("bench.py", os.path.join("tools", "run_streams.py")):
        src = open(os.path.join(ROOT, f)).read()
        assert "spawn_ranks" in src and "launched_by_parent" in src
