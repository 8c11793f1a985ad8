"""GPU end-to-end drop-in check: the REFERENCE's own command-line programs (mains compiled from
/root/reference/src/*.cpp by oracle/Makefile, linked against our libapi.so through our
include/libapi.h) run on TIFF files and must reproduce the oracle.  Skipped when the prebuilt
binaries did not travel (they are built in the container that has the reference tree)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import decon_oracle as do
from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "oracle", "_ref", "cli")


def _need(name):
    path = os.path.join(CLI, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (needs the reference tree at build time)")
    return path


def rel_err(got, want):
    floor = float(np.mean(np.abs(want)))
    return float(np.max(np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), floor)))


def test_reference_deconSingleView_cli(tmp_path):
    from microimagelib_amd import api
    exe = _need("deconSingleView")
    shape = (20, 36, 50)  # box 64x64x32: pad + crop
    psf = do.gaussian_psf((9, 9, 9), (2.0, 1.2, 1.2))
    img = do.synthetic_view(shape, psf, seed=0, noise_seed=1, n_blobs=12)
    api.writetifstack(str(tmp_path / "img.tif"), img, 32)
    api.writetifstack(str(tmp_path / "psf.tif"), psf, 32)
    out = tmp_path / "decon.tif"
    r = subprocess.run([exe, "-i", str(tmp_path / "img.tif"), "-fp", str(tmp_path / "psf.tif"), "-o", str(out),
                        "-it", "6", "-bit", "32", "-verbOFF"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runStatus: 0" in r.stdout
    got = api.readtifstack(str(out))
    assert got.shape == shape
    assert rel_err(got, do.decon_singleview(img, psf, 6)) < 1e-5


def test_reference_deconDualView_cli(tmp_path):
    from microimagelib_amd import api
    exe = _need("deconDualView")
    shape = (32, 64, 64)
    psf_a = do.gaussian_psf((9, 9, 9), (2.0, 1.2, 1.2))
    psf_b = do.gaussian_psf((9, 9, 9), (1.2, 1.2, 2.0))
    a = do.synthetic_view(shape, psf_a, seed=0, noise_seed=1, n_blobs=12)
    b = do.synthetic_view(shape, psf_b, seed=0, noise_seed=2, n_blobs=12)
    for name, arr in (("a", a), ("b", b), ("pa", psf_a), ("pb", psf_b)):
        api.writetifstack(str(tmp_path / f"{name}.tif"), arr, 32)
    out = tmp_path / "decon.tif"
    r = subprocess.run([exe, "-i1", str(tmp_path / "a.tif"), "-i2", str(tmp_path / "b.tif"), "-fp1",
                        str(tmp_path / "pa.tif"), "-fp2", str(tmp_path / "pb.tif"), "-o", str(out), "-it", "4", "-bit",
                        "32"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = api.readtifstack(str(out))
    assert rel_err(got, do.decon_dualview(a, b, psf_a, psf_b, 4)) < 1e-5


def test_reference_reg3D_cli(tmp_path):
    """reg3D with -regc 0 and an input matrix: the resample path of reg3d through the reference's main."""
    from microimagelib_amd import api
    exe = _need("reg3D")
    tgt = do.blob_volume((16, 24, 32), n_blobs=8, seed=3)
    m = ro.dof9tomatrix(np.array([0, 0.7, -0.4, 0.3, 2.0, -1.0, 1.5, 1.01, 0.99, 1.0], dtype=np.float32), 9)
    api.writetifstack(str(tmp_path / "t.tif"), tgt, 32)
    api.writetifstack(str(tmp_path / "s.tif"), tgt, 32)
    with open(tmp_path / "in.tmx", "w") as f:  # .tmx: 3 rows x 4 + "0 0 0 1" (reference src/reg3D.cpp:316-326)
        for r in range(3):
            f.write("\t".join(f"{v:f}" for v in m[4 * r:4 * r + 4]) + "\n")
        f.write("0.000000\t0.000000\t0.000000\t1.000000\n")
    out = tmp_path / "reg.tif"
    r = subprocess.run([exe, "-t", str(tmp_path / "t.tif"), "-s", str(tmp_path / "s.tif"), "-o", str(out), "-itmx",
                        str(tmp_path / "in.tmx"), "-otmx", str(tmp_path / "out.tmx"), "-regc", "0", "-bit", "32"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = api.readtifstack(str(out))
    m_file = np.array([float(v) for v in open(tmp_path / "in.tmx").read().split()[:12]], dtype=np.float32)
    want = ro.affine_transform(tgt, m_file, tgt.shape)
    assert np.allclose(got, want, rtol=0, atol=1e-4 * float(np.max(np.abs(want))))
    assert os.path.exists(tmp_path / "out.tmx")
