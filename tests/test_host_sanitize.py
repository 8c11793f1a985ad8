"""Host-only code of the library (TIFF codec, Powell) under AddressSanitizer + UBSan on the CPU build."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_tiff_and_powell_clean_under_asan_ubsan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "host_sanitize"
    csrc = os.path.join(ROOT, "microimagelib_amd", "csrc")
    build = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-fno-omit-frame-pointer", "-ffp-contract=off", os.path.join(ROOT, "tests", "host_sanitize", "main.cpp"),
                            os.path.join(csrc, "tiff_io.cpp"), os.path.join(csrc, "powell.cpp"), "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert "powell:" in run.stdout
