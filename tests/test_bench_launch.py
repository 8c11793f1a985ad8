"""`python bench.py --gpus N` from a plain shell must start its N ranks itself (the driver calls it that way for N = 1 and
may do so for N > 1).  --dry-run exercises the launcher, the rendezvous on 127.0.0.1, the collectives of the timed region
(barrier, MAX all-reduce, gather to rank 0) over gloo and the relay of rank 0's JSON line and of the exit status -- no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=280)


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] and out["n_gpus"] == 2 and out["requested_gpus"] == 2
    assert out["steps"] == 3 and out["warmup"] == 1  # the arguments reached the ranks
    assert out["gathered_ranks"] == [0.0, 1.0] and out["max_over_ranks"] == 2.0
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_command():
    p = _run(["--gpus", "2", "--dry-run"], {"MIL_BENCH_DRY_RUN_FAIL": "1"})
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_single_process_dry_run_needs_no_launcher():
    p = _run(["--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and "torch.distributed.run" not in p.stderr
