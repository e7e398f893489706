"""bench.py's multi-rank plumbing on the CPU (LARVA_BENCH_DRY=1: rendezvous, collectives and the
JSON relay with no kernels): `python bench.py --gpus N` must start its own ranks -- the driver
invokes it without a launcher -- and the torch.distributed.run entry must keep working."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = dict(os.environ, LARVA_BENCH_DRY="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 8])
def test_plain_python_invocation_launches_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "3", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout          # stdout carries exactly one line: rank 0's JSON
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["max_rank_plus_1"] == float(n) and line["backend"] == "gloo"
    assert line["steps"] == 3 and line["warmup"] == 1


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(LARVA_BENCH_DRY_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode != 0 and not r.stdout.strip()
    assert "rank 1 exited with code 7" in r.stderr


@pytest.mark.timeout(300)
def test_torch_distributed_run_entry_still_works():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2"],
                       env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True
