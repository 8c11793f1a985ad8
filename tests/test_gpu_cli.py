"""GPU end-to-end drop-in check on TIFF files, for two sets of command-line programs:
  reference  the REFERENCE's own mains (compiled from /root/reference/src/*.cpp by oracle/Makefile, linked against
             our libapi.so through our include/libapi.h); skipped when the prebuilt binaries did not travel
             (they are built in the container that has the reference tree);
  ours       microimagelib_amd/bin/* (microimagelib_amd/cli/*.cpp, same flags).
Both must reproduce the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import decon_oracle as do
from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIRS = {"reference": os.path.join(ROOT, "oracle", "_ref", "cli"), "ours": os.path.join(ROOT, "microimagelib_amd", "bin")}


@pytest.fixture(params=["reference", "ours"])
def which(request):
    return request.param


def _need(name, which):
    path = os.path.join(DIRS[which], name)
    if not os.path.exists(path):
        if which == "ours":
            raise AssertionError(f"{path} missing: run make -C microimagelib_amd/csrc")
        pytest.skip(f"{path} not built (needs the reference tree at build time)")
    return path


def rel_err(got, want):
    floor = float(np.mean(np.abs(want)))
    return float(np.max(np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), floor)))


def test_deconSingleView_cli(tmp_path, which):
    from microimagelib_amd import api
    exe = _need("deconSingleView", which)
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


def test_deconDualView_cli(tmp_path, which):
    from microimagelib_amd import api
    exe = _need("deconDualView", which)
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


def test_reg3D_cli(tmp_path, which):
    """reg3D with -regc 0 and an input matrix: the resample path of reg3d through the reference's main."""
    from microimagelib_amd import api
    exe = _need("reg3D", which)
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


def test_16bit_output_and_const_init(tmp_path, which):
    """-bit 16 truncates like the reference writer ((uint16)f, src/apifunc.cpp:255); -cON uses the constant start."""
    from microimagelib_amd import api
    exe = _need("deconSingleView", which)
    shape = (16, 32, 32)
    psf = do.gaussian_psf((7, 7, 7), (1.5, 1.0, 1.0))
    img = do.synthetic_view(shape, psf, seed=2, noise_seed=3, n_blobs=6)
    api.writetifstack(str(tmp_path / "img.tif"), img, 16)  # 16-bit input: the tool reads what the writer truncated
    api.writetifstack(str(tmp_path / "psf.tif"), psf, 32)
    img16 = api.readtifstack(str(tmp_path / "img.tif"))
    assert np.array_equal(img16, np.floor(img).astype(np.float32))
    out = tmp_path / "decon.tif"
    r = subprocess.run([exe, "-i", str(tmp_path / "img.tif"), "-fp", str(tmp_path / "psf.tif"), "-o", str(out),
                        "-it", "3", "-cON", "-verbOFF"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert api.gettifinfo(str(out))[0] == 16  # same bit depth as the input image by default
    want = do.decon_singleview(img16, psf, 3, const_init=True)
    got = api.readtifstack(str(out))
    assert np.max(np.abs(got - np.floor(want))) <= 1.0  # truncation of values that differ by rounding


def test_spimFusion_cli(tmp_path, which):
    """spimFusion end to end: anisotropic view A (z resampled x2), view B rotated -90 deg about Y, 3-DOF
    registration, joint RL.  Checked against the same chain through the Python API (libapi.so calls in the same
    order, src/spim_fusion.cpp:563-672)."""
    from microimagelib_amd import api
    exe = _need("spimFusion", which)
    psf_a = do.gaussian_psf((7, 7, 7), (1.5, 1.0, 1.0))
    psf_b = do.gaussian_psf((7, 7, 7), (1.0, 1.0, 1.5))
    a_in = do.synthetic_view((16, 32, 32), psf_a, seed=1, noise_seed=2, n_blobs=10)   # z sampled 2x coarser
    b_in = do.synthetic_view((32, 32, 16), psf_b, seed=1, noise_seed=3, n_blobs=10)   # x sampled 2x coarser
    for name, arr in (("a", a_in), ("b", b_in), ("pa", psf_a), ("pb", psf_b)):
        api.writetifstack(str(tmp_path / f"{name}.tif"), arr, 32)
    out, otmx = tmp_path / "decon.tif", tmp_path / "out.tmx"
    r = subprocess.run([exe, "-i1", str(tmp_path / "a.tif"), "-i2", str(tmp_path / "b.tif"), "-fp1",
                        str(tmp_path / "pa.tif"), "-fp2", str(tmp_path / "pb.tif"), "-o", str(out), "-pxx1", "1", "-pxy1",
                        "1", "-pxz1", "2", "-pxx2", "2", "-pxy2", "1", "-pxz2", "1", "-imgrot", "-1", "-regc", "2", "-affm",
                        "1", "-itreg", "200", "-otmx", str(otmx), "-it", "3", "-bit", "32", "-verbOFF"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = api.readtifstack(str(out))
    # same chain through the API
    a = api.imresize3d(a_in, (32, 32, 32))
    b = api.imresize3d(api.imoperation3D(b_in, 2), (32, 32, 32))
    b_reg, tmx, _ = api.reg3d(a, b, reg_choice=2, aff_method=1, it_limit=200)
    want, _ = api.decon_dualview(a, b_reg, psf_a, psf_b, 3)
    assert got.shape == want.shape == (32, 32, 32)
    assert np.array_equal(got, want)
    m_file = np.array([float(v) for v in open(otmx).read().split()[:12]], dtype=np.float32)
    assert np.allclose(m_file, tmx, atol=1e-5)


def _batch_args(tmp_path, out, reg_mode, initial="0", itmx="none"):
    return [str(out) + "/", str(tmp_path / "SPIMA") + "/", str(tmp_path / "SPIMB") + "/", "SPIMA_", "SPIMB_", "0", "2", "1", "1",
            "1", "1", "1", "1", "1", "1", str(reg_mode), "0", initial, itmx, "0.001", "150", "0", "1", str(tmp_path / "pa.tif"),
            str(tmp_path / "pb.tif"), "3", "0", "0", "1", "0", "0", "32", "0", "0"]


def _make_series(tmp_path):
    from microimagelib_amd import api
    psf_a = do.gaussian_psf((7, 7, 7), (1.5, 1.0, 1.0))
    psf_b = do.gaussian_psf((7, 7, 7), (1.0, 1.0, 1.5))
    os.makedirs(tmp_path / "SPIMA", exist_ok=True)
    os.makedirs(tmp_path / "SPIMB", exist_ok=True)
    api.writetifstack(str(tmp_path / "pa.tif"), psf_a, 32)
    api.writetifstack(str(tmp_path / "pb.tif"), psf_b, 32)
    for t in range(3):
        api.writetifstack(str(tmp_path / "SPIMA" / f"SPIMA_{t}.tif"),
                          do.synthetic_view((16, 32, 32), psf_a, seed=t, noise_seed=10 + t, n_blobs=10), 32)
        api.writetifstack(str(tmp_path / "SPIMB" / f"SPIMB_{t}.tif"),
                          do.synthetic_view((16, 32, 32), psf_b, seed=t, noise_seed=20 + t, n_blobs=10), 32)


@pytest.mark.parametrize("reg_mode", [3, 1, 2])
def test_spimFusionBatch_ours_equals_reference_main(tmp_path, reg_mode):
    """Our spimFusionBatch main and the reference's main, both on our libapi.so, produce identical outputs
    (Decon stacks, matrix files, Z projections) for the independent, register-once and chained registration modes."""
    from microimagelib_amd import api
    ours, ref = _need("spimFusionBatch", "ours"), _need("spimFusionBatch", "reference")
    _make_series(tmp_path)
    outs = {}
    for name, exe in (("ours", ours), ("ref", ref)):
        out = tmp_path / f"out_{name}"
        r = subprocess.run([exe] + _batch_args(tmp_path, out, reg_mode), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = out
    for t in range(3):
        a = api.readtifstack(str(outs["ours"] / "Decon" / f"Decon_{t}.tif"))
        b = api.readtifstack(str(outs["ref"] / "Decon" / f"Decon_{t}.tif"))
        assert np.array_equal(a, b), t
        ma = open(outs["ours"] / "TMX" / f"Matrix_{t}.tmx").read().split()
        mb = open(outs["ref"] / "TMX" / f"Matrix_{t}.tmx").read().split()
        assert [float(v) for v in ma] == [float(v) for v in mb]
        mip = api.readtifstack(str(outs["ours"] / "Decon" / "MP_ZProj" / f"MP_XY_{t}.tif"))
        assert np.array_equal(mip[0], np.maximum(a.max(axis=0), 0))
        assert os.path.exists(outs["ours"] / "RegB" / f"SPIMB_reg_{t}.tif")
    assert "Image time point number" in open(outs["ours"] / "ProcessingLog.txt").read()


@pytest.mark.parametrize("reg_mode", [2, 3])
def test_spimFusionBatch_two_workers_on_one_device(tmp_path, reg_mode):
    """one host worker thread per listed device.  Two workers sharing device 0 (the only way to
    run the multi-device code on a one-GPU box: separate threads, engines, OTFs, streams and buffers inside libapi.so)
    reproduce the single-worker stacks, matrix files and projections bit for bit -- in the chained mode too, where the
    chain state is handed from worker to worker (src/spim_fusion_batch.cpp:786-808)."""
    exe = _need("spimFusionBatch", "ours")
    _make_series(tmp_path)
    outs = {}
    for name, devices in (("one", "0"), ("two", "0,0")):
        out = tmp_path / f"out_{name}"
        args = _batch_args(tmp_path, out, reg_mode)
        args[-1] = devices
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, MIL_BATCH_WORKERS_PER_DEVICE="1"))  # "0" is ONE worker here (the default is two)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = out
    assert "2 workers on GPU devices 0,0" in open(outs["two"] / "ProcessingLog.txt").read()
    assert "workers on GPU devices" not in open(outs["one"] / "ProcessingLog.txt").read()
    for sub, names in (("Decon", [f"Decon_{t}.tif" for t in range(3)]), ("TMX", [f"Matrix_{t}.tmx" for t in range(3)]),
                       (os.path.join("Decon", "MP_ZProj"), [f"MP_XY_{t}.tif" for t in range(3)]),
                       ("RegB", [f"SPIMB_reg_{t}.tif" for t in range(3)])):
        for n in names:
            a = open(outs["one"] / sub / n, "rb").read()
            b = open(outs["two"] / sub / n, "rb").read()
            assert a == b, (sub, n)


def test_spimFusionBatch_device_path_all_stages_equal_reference_main(tmp_path):
    """Our main keeps a time point in HBM from the TIFF reader to the TIFF writer (include/mil_device.h stage entries:
    mil_imresize3d, mil_rot_y90, mil_reg3d, mil_decon_dualview, mil_max_projection, mil_mip3d, 16-bit conversions on the
    device).  With every stage switched on -- 16-bit input stacks, view A resized along z, view B rotated about y AND resized,
    registration, both registered views saved, deconvolution, the three 2-D and the two 36-angle 3-D projections, 16-bit
    output -- every file it writes is, byte for byte, the file the reference's own main writes through the host-pointer
    calls of the same library, and the file our main writes on its host-pointer path (MIL_BATCH_PATH=host)."""
    from microimagelib_amd import api
    ours, ref = _need("spimFusionBatch", "ours"), _need("spimFusionBatch", "reference")
    psf_a = do.gaussian_psf((7, 7, 7), (1.5, 1.0, 1.0))
    psf_b = do.gaussian_psf((7, 7, 7), (1.0, 1.0, 1.5))
    os.makedirs(tmp_path / "SPIMA")
    os.makedirs(tmp_path / "SPIMB")
    api.writetifstack(str(tmp_path / "pa.tif"), psf_a, 32)
    api.writetifstack(str(tmp_path / "pb.tif"), psf_b, 32)
    for t in range(3):  # A: z sampled 2x coarser; B: acquired rotated, its x (-> z after the rotation) 2x coarser
        a = do.synthetic_view((12, 40, 48), psf_a, seed=t, noise_seed=10 + t, n_blobs=10)
        b = do.synthetic_view((48, 40, 12), psf_b, seed=t, noise_seed=20 + t, n_blobs=10)
        api.writetifstack(str(tmp_path / "SPIMA" / f"SPIMA_{t}.tif"), a, 16)
        api.writetifstack(str(tmp_path / "SPIMB" / f"SPIMB_{t}.tif"), b, 16)

    def args(out):
        return [str(out) + "/", str(tmp_path / "SPIMA") + "/", str(tmp_path / "SPIMB") + "/", "SPIMA_", "SPIMB_", "0", "2", "1", "1",
                "1", "1", "2", "2", "1", "1", "3", "-1", "0", "none", "0.001", "120", "1", "1", str(tmp_path / "pa.tif"),
                str(tmp_path / "pb.tif"), "3", "1", "1", "1", "1", "1", "16", "0", "0"]
    outs = {}
    for name, exe, env in (("device", ours, {}), ("host", ours, {"MIL_BATCH_PATH": "host"}), ("ref", ref, {})):
        out = tmp_path / f"out_{name}"
        r = subprocess.run([exe] + args(out), capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, name + r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = out
    files = []
    for root, _, names in os.walk(outs["ref"]):
        for n in names:
            if n != "ProcessingLog.txt":
                files.append(os.path.relpath(os.path.join(root, n), outs["ref"]))
    assert len(files) == 3 * (1 + 1 + 2 + 3 + 2), sorted(files)  # per time point: Decon, matrix, RegA + RegB, XY / YZ / ZX, 3-D x / y
    for f in sorted(files):
        want = open(outs["ref"] / f, "rb").read()
        assert open(outs["device"] / f, "rb").read() == want, f"device path: {f}"
        assert open(outs["host"] / f, "rb").read() == want, f"host path: {f}"
    d0 = api.readtifstack(str(outs["device"] / "Decon" / "Decon_0.tif"))
    assert d0.shape == (24, 40, 48) and d0.max() > 0
