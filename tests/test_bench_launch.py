"""`python bench.py --gpus N` from a plain shell must start its N ranks itself (the driver calls it that way for N = 1 and
may do so for N > 1).  --dry-run exercises the launcher, the rendezvous on 127.0.0.1, the collectives of the timed region
(barrier, MAX all-reduce, gather to rank 0) over gloo and the relay of rank 0's JSON line and of the exit status -- no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


BATCH = ["--batch-time-points-per-gpu", "3"]


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
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"] + BATCH)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] and out["n_gpus"] == 2 and out["requested_gpus"] == 2
    assert out["steps"] == 3 and out["warmup"] == 1  # the arguments reached the ranks
    assert out["gathered_ranks"] == [0.0, 1.0] and out["max_over_ranks"] == 2.0
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr
    # the config-5 driver ran over gloo with the stand-in pipeline: regMode 3 (two lanes per rank) and the regMode-2 token
    # ring, every time point's five products gathered to rank 0 in time order
    figs = out["extra"]["batch_config5_by_mode"]
    assert out["extra"]["errors"] == [] and out["extra"]["other"] == "ran"
    for mode in ("regmode3", "regmode2"):
        f = figs[mode]
        assert f["time_points"] == 6 and f["mip_stacks"]["mip3d_y"] == [6, 3, 6, 9]
        assert f["mip_gather"]["products"] == ["xy", "yz", "zx", "mip3d_x", "mip3d_y"]
        assert f["mip_gather"]["bytes_received_rank0"] == 3 * f["mip_gather"]["bytes_per_time_point"]  # rank 1's three
        assert f["registration_evaluations_per_time_point_rank0"]["registrations"] == 3


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


def _line(p):
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_a_failing_extra_still_prints_the_line():
    """An exception inside a supplementary figure is recorded in extra.errors; the line is printed and the status is 0."""
    p = _run(["--gpus", "2", "--dry-run"] + BATCH, {"MIL_BENCH_FAIL_EXTRA": "batch_config5"})
    assert p.returncode == 0, p.stderr[-2000:]
    out = _line(p)
    assert [e["extra"] for e in out["extra"]["errors"]] == ["batch_config5"]
    assert "deliberate failure" in out["extra"]["errors"][0]["error"]
    assert out["extra"]["other"] == "ran" and "batch_config5_by_mode" not in out["extra"]  # the later extras still ran


@pytest.mark.timeout(300)
def test_one_rank_failing_inside_the_batch_is_agreed_on_not_waited_for():
    """Rank 1 cannot load its time points: the agreement handshake before the gather (batch._agree) makes rank 0 raise too
    instead of waiting in the gather for ever; the failure lands in extra.errors, the line is printed, status 0."""
    p = _run(["--gpus", "2", "--dry-run", "--dist-timeout", "60"] + BATCH, {"MIL_BENCH_DRY_RUN_BATCH_FAIL": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    out = _line(p)
    errs = out["extra"]["errors"]
    assert len(errs) == 1 and errs[0]["extra"] == "batch_config5" and "another rank failed" in errs[0]["error"]
    assert out["extra"]["other"] == "ran"


@pytest.mark.timeout(300)
def test_a_hanging_extra_is_ended_by_the_watchdog():
    """A rank stuck for ever inside an extra (a collective whose peer is gone): at --extras-deadline the watchdog prints the
    line as it stands and ends the process with status 0."""
    p = _run(["--dry-run", "--extras-deadline", "5"] + BATCH, {"MIL_BENCH_DRY_RUN_HANG": "0"})
    assert p.returncode == 0, p.stderr[-2000:]
    out = _line(p)
    assert out["extra"]["errors"][-1]["extra"] == "hang" and "deadline" in out["extra"]["errors"][-1]["error"]
    assert "regmode3" in out["extra"]["batch_config5_by_mode"]  # what had finished is in the line
    assert out["extras_truncated"] is True  # top level: the reader of the line sees the run was cut short
    # a CI that wants a truncated run to FAIL asks for it: same line, non-zero status
    p = _run(["--dry-run", "--extras-deadline", "3"] + BATCH, {"MIL_BENCH_DRY_RUN_HANG": "0", "MIL_BENCH_DEADLINE_RC": "3"})
    assert p.returncode == 3, p.stderr[-2000:]
    assert _line(p)["extras_truncated"] is True
