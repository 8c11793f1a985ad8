"""Command-line front ends (microimagelib_amd/cli): flag tables and help paths, no GPU needed."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "microimagelib_amd", "bin")

# flags of the reference tools: src/decon_sv.cpp:25-39, src/decon_dv.cpp:27-45, src/reg3D.cpp:24-45
FLAGS = {
    "deconSingleView": ["-i", "-fp", "-o", "-bp", "-it", "-gm", "-dev", "-cON", "-cOFF", "-bit", "-verbON", "-verbOFF", "-log"],
    "deconDualView": ["-i1", "-i2", "-fp1", "-fp2", "-o", "-bp1", "-bp2", "-it", "-cON", "-cOFF", "-gm", "-dev", "-bit",
                      "-verbON", "-verbOFF", "-log"],
    # src/spim_fusion.cpp:22-79
    "spimFusion": ["-i1", "-i2", "-fp1", "-fp2", "-o", "-pxx1", "-pxy1", "-pxz1", "-pxx2", "-pxy2", "-pxz2", "-imgrot",
                   "-oreg1", "-oreg2", "-itmx", "-otmx", "-regc", "-affm", "-ftol", "-itreg", "-bp1", "-bp2", "-it", "-cON",
                   "-cOFF", "-gm", "-dev", "-bit", "-verbON", "-verbOFF", "-log"],
    "reg3D": ["-t", "-s", "-o", "-itmx", "-otmx", "-regc", "-affm", "-ftol", "-it", "-gm", "-dev", "-bit", "-verbON",
              "-verbOFF", "-log"],
}


@pytest.mark.parametrize("tool", sorted(FLAGS))
def test_help_lists_every_reference_flag(tool):
    exe = os.path.join(BIN, tool)
    assert os.path.exists(exe), "run make -C microimagelib_amd/csrc"
    brief = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert brief.returncode == 0 and "Usage" in brief.stdout  # no arguments: usage, exit 0 (reference behaviour)
    full = subprocess.run([exe, "-h"], capture_output=True, text=True, timeout=60)
    assert full.returncode == 0
    for flag in FLAGS[tool]:
        assert f"\t{flag} " in full.stdout or f"\t{flag}\t" in full.stdout, flag


def test_bad_memory_mode_is_rejected_before_any_io():
    r = subprocess.run([os.path.join(BIN, "deconSingleView"), "-gm", "7"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Wrong GPU mode" in r.stdout


def test_missing_value_is_an_error():
    r = subprocess.run([os.path.join(BIN, "reg3D"), "-it"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0


def test_batch_tool_argument_count():
    exe = os.path.join(BIN, "spimFusionBatch")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "34:" in r.stdout and "35, 36" in r.stdout
    r = subprocess.run([exe, "a", "b", "c"], capture_output=True, text=True, timeout=60)
    assert "Arguments do NOT match" in r.stdout
