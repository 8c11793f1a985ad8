"""gpuMemMode 0 -- the host Richardson-Lucy path (BASELINE.json configs[0]; reference decon_*_OTF0,
src/api_subfunc.cu:3310-3359, 3519-3585) -- runs WITHOUT a GPU: these tests go through the same C ABI as the GPU tests
(libapi.h entry points, host pointers) and compare with the oracle and the committed golden vectors at 1e-5."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import decon_oracle as do

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32
RTOL = 1e-5


def rel_err(got, want):
    want = want.astype(np.float64)
    floor = float(np.mean(np.abs(want)))
    return float(np.max(np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), floor)))


@pytest.fixture(scope="module")
def api():
    from microimagelib_amd import api
    return api


# ---------------------------------------------------------------- the bundled transform
@pytest.mark.parametrize("shape", [(16, 16, 16), (32, 16, 64), (8, 24, 36), (20, 28, 44), (64, 192, 320), (16, 64, 2),
                                   (1, 1, 16), (3, 5, 14), (26, 22, 10), (128, 128, 128)])
@pytest.mark.parametrize("threads", [1, 3])
def test_cpu_rfft3d_matches_numpy(lib, shape, threads):
    rng = np.random.default_rng(sum(shape))
    nz, ny, nx = shape
    a = rng.uniform(-1, 1, shape).astype(f32)
    spec = np.zeros((nz, ny, nx // 2 + 1), dtype=np.complex64)
    fp = C.POINTER(C.c_float)
    assert lib.mil_cpu_rfft3d(a.ctypes.data_as(fp), spec.ctypes.data_as(C.c_void_p), nx, ny, nz, 0, threads) == 0
    want = np.fft.rfftn(a.astype(np.float64))
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(spec - want))) / scale < 2e-6
    back = np.zeros(shape, dtype=f32)
    assert lib.mil_cpu_rfft3d(back.ctypes.data_as(fp), spec.ctypes.data_as(C.c_void_p), nx, ny, nz, 1, threads) == 0
    assert float(np.max(np.abs(back / a.size - a))) < 2e-6  # unnormalised round trip = N * input


# ---------------------------------------------------------------- golden vectors through libapi.h, gpuMemMode 0
def test_golden_singleview_small(api):
    g = np.load(os.path.join(GOLD, "decon_sv_small.npz"))
    out, rec = api.decon_singleview(g["img"], g["psf"], int(g["iters"]), gpu_mem_mode=0)
    assert rec[0] == 0  # deconRecords[0] = the mode that ran
    assert rel_err(out, g["expected"]) < RTOL
    out, _ = api.decon_singleview(g["img"], g["psf"], int(g["iters"]), const_init=True, gpu_mem_mode=0)
    assert rel_err(out, g["expected_const"]) < RTOL


def test_golden_dualview_small(api):
    g = np.load(os.path.join(GOLD, "decon_dv_small.npz"))
    out, rec = api.decon_dualview(g["img1"], g["img2"], g["psf1"], g["psf2"], int(g["iters"]), gpu_mem_mode=0)
    assert rec[0] == 0
    assert rel_err(out, g["expected"]) < RTOL
    out, _ = api.decon_dualview(g["img1"], g["img2"], g["psf1"], g["psf2"], int(g["iters"]), const_init=True,
                                gpu_mem_mode=0)
    assert rel_err(out, g["expected_const"]) < RTOL


def test_golden_psf_equals_fft_box(api):
    g = np.load(os.path.join(GOLD, "decon_sv_psf_eq_fft.npz"))
    out, _ = api.decon_singleview(g["img"], g["psf"], int(g["iters"]), gpu_mem_mode=0)
    assert rel_err(out, g["expected"]) < RTOL


def test_golden_unmatched_back_projector(api):
    g = np.load(os.path.join(GOLD, "decon_sv_unmatched.npz"))
    out, _ = api.decon_singleview(g["img"], g["psf"], int(g["iters"]), unmatch=True, psf_bp=g["psf_bp"],
                                  gpu_mem_mode=0)
    assert rel_err(out, g["expected"]) < RTOL


def test_golden_psf_larger_than_box(api):
    g = np.load(os.path.join(GOLD, "decon_sv_bigpsf.npz"))
    out, _ = api.decon_singleview(g["img"], g["psf"], int(g["iters"]), gpu_mem_mode=0)
    assert rel_err(out, g["expected"]) < RTOL


# ---------------------------------------------------------------- BASELINE.json configs[0]
@pytest.mark.parametrize("psf_n", [32, 128])
def test_config1_128_cubed_10_iterations(api, psf_n):
    """deconSingleView 128x128x128 float32, 10 RL iterations on the CPU path, with the 32^3 and the 128^3 PSF of
    SURVEY.md 8(d) (the reference's own padPSFcpu is only right for the second)."""
    shape = (128, 128, 128)
    psf = do.gaussian_psf((psf_n,) * 3, (3.0, 1.5, 1.5))
    img = do.synthetic_view(shape, do.gaussian_psf((32, 32, 32), (3.0, 1.5, 1.5)), seed=0, noise_seed=1)
    out, rec = api.decon_singleview(img, psf, 10, gpu_mem_mode=0)
    assert rec[0] == 0 and rec[8] > 0
    assert rel_err(out, do.decon_singleview(img, psf, 10)) < RTOL


def test_config1_dual_view_and_threads(api, monkeypatch):
    shape = (64, 64, 128)
    psf_a = do.gaussian_psf((17, 17, 17), (3.0, 1.5, 1.5))
    psf_b = do.gaussian_psf((17, 17, 17), (1.5, 1.5, 3.0))
    img_a = do.synthetic_view(shape, psf_a, seed=0, noise_seed=1, n_blobs=20)
    img_b = do.synthetic_view(shape, psf_b, seed=0, noise_seed=2, n_blobs=20)
    want = do.decon_dualview(img_a, img_b, psf_a, psf_b, 6)
    one, _ = api.decon_dualview(img_a, img_b, psf_a, psf_b, 6, gpu_mem_mode=0)
    assert rel_err(one, want) < RTOL
    monkeypatch.setenv("MIL_CPU_THREADS", "4")
    four, _ = api.decon_dualview(img_a, img_b, psf_a, psf_b, 6, gpu_mem_mode=0)
    assert np.array_equal(one, four)  # every pass is element-wise or per-column: threads do not change a bit


def test_dual_view_unmatched_back_projectors(api):
    shape = (24, 40, 44)
    psf_a = do.gaussian_psf((9, 9, 9), (2.0, 1.2, 1.2))
    psf_b = do.gaussian_psf((9, 9, 9), (1.2, 1.2, 2.0))
    bp_a = do.gaussian_psf((9, 9, 9), (1.2, 1.0, 1.0))
    bp_b = do.gaussian_psf((9, 9, 9), (1.0, 1.0, 1.2))
    img_a = do.synthetic_view(shape, psf_a, seed=5, noise_seed=6, n_blobs=10)
    img_b = do.synthetic_view(shape, psf_b, seed=5, noise_seed=7, n_blobs=10)
    out, rec = api.decon_dualview(img_a, img_b, psf_a, psf_b, 4, gpu_mem_mode=0, unmatch=True, psf_bp1=bp_a, psf_bp2=bp_b)
    assert rec[0] == 0
    want = do.decon_dualview(img_a, img_b, psf_a, psf_b, 4, unmatch=True, psf_bp1=bp_a, psf_bp2=bp_b)
    assert rel_err(out, want) < RTOL


@pytest.mark.parametrize("shape", [(20, 36, 50), (33, 130, 300)])
def test_ragged_sizes_pad_and_crop(api, shape):
    psf = do.gaussian_psf((11, 11, 11), (2.0, 1.3, 1.3))
    img = do.synthetic_view(shape, psf, seed=11, noise_seed=12, n_blobs=15)
    out, _ = api.decon_singleview(img, psf, 4, gpu_mem_mode=0)
    assert rel_err(out, do.decon_singleview(img, psf, 4)) < RTOL


def test_zero_iterations_and_bad_mode(api):
    rng = np.random.default_rng(0)
    img = rng.uniform(-1, 5, (16, 32, 32)).astype(f32)
    psf = do.gaussian_psf((5, 5, 5), (1, 1, 1))
    out, _ = api.decon_singleview(img, psf, 0, gpu_mem_mode=0)
    assert np.array_equal(out, do.clamp_small(img))
    with pytest.raises(api.LibApiError) as e:
        api.decon_singleview(img, psf, 1, gpu_mem_mode=7)  # reference src/api_decon.cpp:316-318
    assert e.value.code == 1
    with pytest.raises(api.LibApiError) as e:
        api.decon_dualview(img, img, psf, psf, 1, gpu_mem_mode=-2)  # :685-687
    assert e.value.code == -1
