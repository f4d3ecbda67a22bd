"""bench.py's launch paths, rehearsed on the CPU (gloo, world_size 2, --dry-run): `python bench.py --gpus N` as the driver invokes
it for N = 1 (no launcher: the script starts its own ranks) and under `python -m torch.distributed.run` as the driver invokes
it for N > 1.  One JSON line on stdout, exit codes propagated, nobody hangs when a rank dies."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_without_launcher_spawns_its_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--res", "64x36"],
                       capture_output=True, text=True, timeout=180, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and p.stdout.strip().count("\n") == 0      # the JSON line is ALL of stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    assert d["config"]["local_rows_sum"] == 40          # 36 rows in 8-row blocks over 2 ranks: 24 + 16 (padding rows included)


def test_gpus_2_under_torch_distributed_run():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "2", "--warmup", "0", "--res", "64x36"],
                       capture_output=True, text=True, timeout=240, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [d for d in _json_lines(p.stdout) if d.get("metric")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_a_failing_rank_fails_the_launch_and_nobody_hangs():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "2", "--fail-rank", "1", "--res", "64x36"],
                       capture_output=True, text=True, timeout=180, env=_env())
    assert p.returncode != 0
    assert not [d for d in _json_lines(p.stdout) if d.get("metric")]
    assert "injected failure" in p.stderr


def test_world_size_mismatch_is_an_error():
    env = _env(); env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--backend", "gloo"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_single_rank_dry_run():
    p = subprocess.run([sys.executable, BENCH, "--dry-run", "--steps", "2", "--res", "64x36"], capture_output=True, text=True, timeout=120, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    (d,) = _json_lines(p.stdout)
    assert d["n_gpus"] == 1 and d["config"]["local_rows_sum"] == 36
