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
                            os.path.join(csrc, "tiff_io.cpp"), os.path.join(csrc, "powell.cpp"), "-o", str(exe), "-lz"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert "powell:" in run.stdout

    # malformed stacks: a second directory that is larger / deeper than the first, and a tag whose count exceeds the file
    import struct

    def tiff(slices, bogus_count=False):
        """Little-endian classic TIFF, one strip per slice; slices = [(width, height, bits)]."""
        blob = bytearray(b"II*\x00\x00\x00\x00\x00")
        link = 4
        for (w, h, bits) in slices:
            data_off = len(blob)
            blob += bytes(w * h * bits // 8)
            if len(blob) & 1:
                blob += b"\x00"
            ifd = len(blob)
            blob[link:link + 4] = struct.pack("<I", ifd)
            ents = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (273, 4, 1, data_off),
                    (277, 3, 1, 1), (279, 4, 0x7FFFFFFF if bogus_count else 1, w * h * bits // 8)]
            if bits == 32:
                ents.append((339, 3, 1, 3))
            blob += struct.pack("<H", len(ents))
            for e in ents:
                blob += struct.pack("<HHII", *e)
            link = len(blob)
            blob += b"\x00\x00\x00\x00"
        return bytes(blob)

    cases = {"bigger_second_slice.tif": tiff([(4, 4, 16), (64, 64, 16)]),
             "deeper_second_slice.tif": tiff([(8, 8, 16), (8, 8, 32)]),
             "huge_tag_count.tif": tiff([(8, 8, 16)], bogus_count=True)}
    for name, blob in cases.items():
        path = tmp_path / name
        path.write_bytes(blob)
        bad = subprocess.run([str(exe), str(tmp_path), "read", str(path)], capture_output=True, text=True,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0"))
        assert bad.returncode == 1, (name, bad.returncode, bad.stderr[-1500:])
        assert "TIFF read error" in bad.stderr and "AddressSanitizer" not in bad.stderr, (name, bad.stderr[-1500:])
    # compressed strips / BigTIFF (round 3): valid files decode, randomly damaged ones end in the error exit or decode to
    # something -- never in a sanitizer report (the decoders clip every write to the slice buffer)
    import numpy as np
    from tests.test_tiff_io import _write_stack
    rng = np.random.default_rng(4)
    vol = (rng.integers(0, 12, (2, 24, 40)) * 9).astype(np.uint16)
    for comp, big in ((5, False), (8, False), (32773, False), (5, True), (1, True)):
        src = tmp_path / f"c{comp}_{int(big)}.tif"
        _write_stack(str(src), vol, compression=comp, rows_per_strip=8, big=big, predictor=2 if comp in (5, 8) else 1)
        ok = subprocess.run([str(exe), str(tmp_path), "read", str(src)], capture_output=True, text=True)
        assert ok.returncode == 0 and "read 40 x 24 x 2" in ok.stdout, (comp, big, ok.stderr[-1500:])
        raw = src.read_bytes()
        for trial in range(25):
            blob = bytearray(raw)
            for _ in range(int(rng.integers(1, 6))):
                at = int(rng.integers(0, len(blob)))
                blob[at] = int(rng.integers(0, 256))
            if trial % 5 == 0:
                blob = blob[:int(rng.integers(16, len(blob)))]
            bad_path = tmp_path / "fuzz.tif"
            bad_path.write_bytes(bytes(blob))
            r = subprocess.run([str(exe), str(tmp_path), "read", str(bad_path)], capture_output=True, text=True,
                               env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0"), timeout=60)
            assert r.returncode in (0, 1), (comp, big, trial, r.returncode, r.stderr[-1500:])
            assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (comp, big, trial, r.stderr[-1500:])
    good = tmp_path / "good.tif"
    good.write_bytes(tiff([(8, 8, 16), (8, 8, 16)]))
    ok = subprocess.run([str(exe), str(tmp_path), "read", str(good)], capture_output=True, text=True)
    assert ok.returncode == 0 and "read 8 x 8 x 2" in ok.stdout, ok.stderr[-1500:]


@pytest.mark.timeout(600)
def test_cpu_decon_path_clean_under_asan_ubsan(tmp_path):
    """The gpuMemMode 0 host path (bundled FFT + RL loop, threaded) under ASan + UBSan on ragged / odd-radix boxes."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    rocm_inc = "/opt/rocm/include"
    if not os.path.isdir(rocm_inc):
        pytest.skip("HIP headers not available (float2 / Dims come from mil_common.h)")
    exe = tmp_path / "cpu_decon_sanitize"
    csrc = os.path.join(ROOT, "microimagelib_amd", "csrc")
    build = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-pthread", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-ffp-contract=off",
                            "-D__HIP_PLATFORM_AMD__", "-I" + rocm_inc,
                            os.path.join(ROOT, "tests", "host_sanitize", "cpu_decon_main.cpp"),
                            os.path.join(csrc, "cpu_fft.cpp"), os.path.join(csrc, "cpu_decon.cpp"), "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert "cpu decon clean" in run.stdout
