"""Multi-worker orchestration of the spimFusionBatch tool (microimagelib_amd/cli/spim_fusion_batch_main.cpp) WITHOUT a
GPU and under ThreadSanitizer: the tool is compiled against host-only stand-ins for the compute entry points
(tests/stubs/libapi_host_stub.cpp; the TIFF functions are the real csrc/tiff_io.cpp) and must, with 1, 2 and 3 workers,
write byte-identical stacks, matrix files and projections, a log in time-point order, and trip no data-race report.
Mirrors the time-point loop and the regMode-2 chain of the reference, src/spim_fusion_batch.cpp:613-966, 786-808.
The GPU run of the same comparison (two workers on one device) is tests/test_gpu_cli.py."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build", "tsan")


def _write_tiff32(path, vol):
    """Minimal classic little-endian TIFF stack, one float32 strip per slice (what writetifstack emits)."""
    vol = np.ascontiguousarray(vol, dtype="<f4")
    nz, ny, nx = vol.shape
    n_tags = 9
    ifd_len = 2 + 12 * n_tags + 4
    slice_bytes = nx * ny * 4
    out = bytearray(b"II*\x00" + struct.pack("<I", 8))
    for z in range(nz):
        ifd_off = 8 + z * (ifd_len + slice_bytes)
        data_off = ifd_off + ifd_len
        tags = [(256, 4, 1, nx), (257, 4, 1, ny), (258, 3, 1, 32), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, data_off),
                (277, 3, 1, 1), (279, 4, 1, slice_bytes), (339, 3, 1, 3)]
        out += struct.pack("<H", n_tags)
        for tag, typ, cnt, val in tags:
            out += struct.pack("<HHII", tag, typ, cnt, val)
        out += struct.pack("<I", 0 if z == nz - 1 else ifd_off + ifd_len + slice_bytes)
        out += vol[z].tobytes()
    with open(path, "wb") as f:
        f.write(out)


@pytest.fixture(scope="module")
def tool():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "spimFusionBatch_tsan")
    srcs = [os.path.join(ROOT, "microimagelib_amd", "cli", "spim_fusion_batch_main.cpp"),
            os.path.join(ROOT, "tests", "stubs", "libapi_host_stub.cpp"),
            os.path.join(ROOT, "microimagelib_amd", "csrc", "tiff_io.cpp")]
    newest = max(os.path.getmtime(s) for s in srcs + [os.path.join(ROOT, "microimagelib_amd", "cli", "cli_common.h")])
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-o", exe] + srcs + ["-lz"], check=True,
                       timeout=600)
    return exe


def _series(tmp_path, n_points):
    rng = np.random.default_rng(7)
    os.makedirs(tmp_path / "SPIMA", exist_ok=True)
    os.makedirs(tmp_path / "SPIMB", exist_ok=True)
    _write_tiff32(tmp_path / "pa.tif", rng.uniform(0, 1, (3, 3, 3)))
    _write_tiff32(tmp_path / "pb.tif", rng.uniform(0, 1, (3, 3, 3)))
    for t in range(n_points):
        _write_tiff32(tmp_path / "SPIMA" / f"SPIMA_{t}.tif", rng.uniform(10, 500, (6, 10, 12)))
        _write_tiff32(tmp_path / "SPIMB" / f"SPIMB_{t}.tif", rng.uniform(10, 500, (6, 10, 12)))


def _args(tmp_path, out, reg_mode, devices, n_points):
    return [str(out) + "/", str(tmp_path / "SPIMA") + "/", str(tmp_path / "SPIMB") + "/", "SPIMA_", "SPIMB_", "0",
            str(n_points - 1), "1", "2", "1", "1", "1", "1", "1", "1", str(reg_mode), "0", "0", "none", "0.001", "150", "0", "1",
            str(tmp_path / "pa.tif"), str(tmp_path / "pb.tif"), "3", "0", "0", "1", "0", "0", "32", "0", devices]


def _tree(root):
    files = {}
    for d, _, names in os.walk(root):
        for n in names:
            if n != "ProcessingLog.txt":
                p = os.path.join(d, n)
                files[os.path.relpath(p, root)] = open(p, "rb").read()
    return files


@pytest.mark.parametrize("reg_mode", [2, 3, 1, 0])
def test_workers_reproduce_the_single_worker_run(tmp_path, tool, reg_mode):
    n_points = 7
    _series(tmp_path, n_points)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    trees = {}
    # the time point resident on the "device" (the default; "0" alone = two workers on one device) and the host-pointer path
    for devices, path in (("0", "device"), ("0,1", "device"), ("2,0,1", "device"), ("0", "host"), ("1,0", "host")):
        out = tmp_path / f"out_{path}_{devices.replace(',', '_')}"
        r = subprocess.run([tool] + _args(tmp_path, out, reg_mode, devices, n_points), capture_output=True, text=True,
                           timeout=600, env=dict(env, MIL_BATCH_PATH=path))
        assert r.returncode == 0, (devices, path, r.stdout[-1500:], r.stderr[-3000:])
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        trees[devices + " " + path] = _tree(out)
        log = open(out / "ProcessingLog.txt").read()
        seen = [int(line.split(":")[1]) for line in log.splitlines() if line.startswith("*** Image time point number")]
        want = list(range(n_points))
        assert seen == ([2] + want if reg_mode == 1 else want), (devices, seen)  # ordered whichever worker finished first
        assert "Batch processing completed" in log
    one = trees["0 host"]
    assert len(one) == 4 * n_points  # Decon_, Matrix_, MP_XY_ and RegB per time point
    for devices, t in trees.items():
        assert set(t) == set(one), devices
        for name in one:
            assert t[name] == one[name], (devices, name)
    if reg_mode == 2:  # the chain really chains: matrices differ from time point to time point
        mats = [one[os.path.join("TMX", f"Matrix_{t}.tmx")] for t in range(n_points)]
        assert len(set(mats)) == n_points


def test_device_list_from_the_environment(tmp_path, tool):
    _series(tmp_path, 4)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", MIL_DEVICES="0,0")
    out = tmp_path / "out_env"
    r = subprocess.run([tool] + _args(tmp_path, out, 3, "0", 4), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "2 workers on GPU devices 0,0" in open(out / "ProcessingLog.txt").read()


def test_a_single_device_gets_two_workers_only_when_two_time_points_fit(tmp_path, tool):
    """One device named once: two workers when the device reports room for two time points (28 float volumes each), one --
    the reference's sequential order -- when it does not, instead of a second worker whose allocation ends the batch.
    MIL_BATCH_WORKERS_PER_DEVICE forces the number either way.  Same files in every case."""
    _series(tmp_path, 4)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    per_worker = 28 * 4 * 6 * 10 * 12
    trees = {}
    for name, extra, want in (("roomy", {}, "2 workers on GPU devices 0,0"),
                              ("tight", {"STUB_DEVICE_FREE_BYTES": str(2 * per_worker - 1)}, "1 worker on GPU device 0"),
                              ("exact", {"STUB_DEVICE_FREE_BYTES": str(2 * per_worker)}, "2 workers on GPU devices 0,0"),
                              ("forced1", {"MIL_BATCH_WORKERS_PER_DEVICE": "1"}, "1 worker on GPU device 0"),
                              ("forced3", {"MIL_BATCH_WORKERS_PER_DEVICE": "3", "STUB_DEVICE_FREE_BYTES": "1"},
                               "3 workers on GPU devices 0,0,0")):
        out = tmp_path / f"out_{name}"
        r = subprocess.run([tool] + _args(tmp_path, out, 3, "0", 4), capture_output=True, text=True, timeout=600,
                           env=dict(env, **extra))
        assert r.returncode == 0, (name, r.stderr[-2000:])
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        assert want in open(out / "ProcessingLog.txt").read(), name
        trees[name] = _tree(out)
    for name, t in trees.items():
        assert t == trees["roomy"], name


@pytest.mark.parametrize("devices", ["0", "0,0"])
def test_a_failed_library_call_stops_the_batch(tmp_path, tool, devices):
    """A registration call that fails (the library prints its error and returns non-zero without touching the outputs) must not
    be papered over with the previous time point's buffers: the tool stops, exits non-zero and writes no result for that
    time point.  (Found on the GPU: four workers on one device, a failed cost evaluation, a Matrix_k.tmx holding the identity.)"""
    _series(tmp_path, 6)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", MIL_STUB_FAIL_REG_CALL="3")
    out = tmp_path / "out_fail"
    r = subprocess.run([tool] + _args(tmp_path, out, 3, devices, 6), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert "registration failed (status -1), processing stopped" in r.stdout
    written = [n for n in _tree(out) if n.startswith("TMX")]
    assert len(written) < 6  # the failed time point (and whatever would have followed it on that worker) has no matrix file
    log = open(out / "ProcessingLog.txt").read()
    assert "registration failed" in log and "Batch processing completed" not in log


@pytest.mark.parametrize("devices", ["0", "0,1,2"])
def test_a_missing_stack_stops_the_batch_in_order(tmp_path, tool, devices):
    """A time point whose input stack is missing: the library's TIFF readers exit() the process like the reference's, which
    from one of several worker threads would cut the other workers' time points off and drop the buffered log.  The worker
    checks first and routes the failure through the batch's stop path: exit status 1, every earlier time point complete
    (stack, matrix, projection -- files whole, not truncated), log flushed in order up to the failure."""
    n_points = 7
    _series(tmp_path, n_points)
    os.remove(tmp_path / "SPIMB" / "SPIMB_4.tif")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    out = tmp_path / "out_missing"
    r = subprocess.run([tool] + _args(tmp_path, out, 3, devices, n_points), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 1, (r.returncode, r.stdout[-1500:], r.stderr[-2000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert "SPIMB_4.tif does not exist" in r.stderr
    tree = _tree(out)
    full = tmp_path / "out_full"  # what a complete time point looks like
    _write_tiff32(tmp_path / "SPIMB" / "SPIMB_4.tif", np.random.default_rng(1).uniform(10, 500, (6, 10, 12)))
    r2 = subprocess.run([tool] + _args(tmp_path, full, 3, "0", n_points), capture_output=True, text=True, timeout=600, env=env)
    assert r2.returncode == 0
    want = _tree(full)
    for t in range(4):  # time points before the missing one: present and byte-identical with the complete run
        for name in (os.path.join("Decon", f"Decon_{t}.tif"), os.path.join("TMX", f"Matrix_{t}.tmx")):
            assert tree.get(name) == want[name], (devices, name)
    assert os.path.join("TMX", "Matrix_4.tmx") not in tree
    for name, data in tree.items():  # nothing half-written under a valid name
        assert data == want[name], name
    log = open(out / "ProcessingLog.txt").read()
    assert "does not exist, processing stopped" in log and "Batch processing completed" not in log
    seen = [int(line.split(":")[1]) for line in log.splitlines() if line.startswith("*** Image time point number")]
    assert seen == sorted(seen) and seen[:5] == [0, 1, 2, 3, 4]


def test_a_failed_chained_registration_does_not_advance_the_chain(tmp_path, tool):
    """regMode 2, three workers: the registration of time point 2 fails.  Its successors must NOT register from the state it
    left behind (the stop sentinel used to be overwritten by `next = i + 1`): no matrix file for time point 2 or later."""
    _series(tmp_path, 7)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", MIL_STUB_FAIL_REG_CALL="3")
    out = tmp_path / "out_chain_fail"
    r = subprocess.run([tool] + _args(tmp_path, out, 2, "0,1,2", 7), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    mats = sorted(n for n in _tree(out) if n.startswith("TMX"))
    assert mats == [os.path.join("TMX", "Matrix_0.tmx"), os.path.join("TMX", "Matrix_1.tmx")], mats
