"""bench.py's own launcher for --gpus N: started plain, it must start N rank processes before making any GPU
call, let them meet, print ONE JSON line from rank 0, and fail loudly -- non-zero exit code, the failing rank's stderr -- when a rank
fails.  No GPU here: --dry-launch stops behind the rendezvous; the failure case uses the ranks' own "no MI355X" exit."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + list(argv), capture_output=True, text=True, timeout=300, env=e)


def test_plain_start_launches_its_own_ranks_and_they_meet():
    out = run("--gpus", "2", "--dry-launch")
    assert out.returncode == 0, out.stderr
    lines = [w for w in out.stdout.splitlines() if w.startswith("{")]
    assert len(lines) == 1, out.stdout  # one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d == {"dry_launch": True, "n_gpus": 2, "ranks_met": [0, 1], "distinct_processes": 2}


def test_three_ranks_meet_under_torch_distributed_run_too():
    e = dict(os.environ)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", BENCH, "--gpus", "3", "--dry-launch"], capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr
    d = json.loads([w for w in out.stdout.splitlines() if w.startswith("{")][-1])
    assert d["ranks_met"] == [0, 1, 2] and d["distinct_processes"] == 3


def test_a_failing_rank_fails_the_launch_with_its_stderr():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU: the ranks' failure here is their own 'no MI355X' exit")
    out = run("--gpus", "2", "--steps", "2", "--warmup", "0")
    assert out.returncode != 0
    assert "rank" in out.stderr and "exited with code" in out.stderr and "needs an MI355X" in out.stderr, out.stderr
    assert not [w for w in out.stdout.splitlines() if w.startswith("{")]  # no half-measured line


def test_world_size_mismatch_is_refused():
    out = run("--gpus", "2", "--dry-launch", env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534"})
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr
