"""The oracle itself (CPU): known answers that follow from the reference's code
 and agreement with the committed golden vectors."""
import os

import numpy as np
import pytest

from oracle import decon_oracle as do
from oracle import reg_oracle as ro

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
f32 = np.float32


def test_snap_transform_size_known_values():
    for n, want in ((20, 32), (36, 64), (50, 64), (100, 128), (128, 128), (130, 192), (256, 256), (300, 320),
                    (500, 512), (512, 512)):
        assert do.snap_transform_size(n) == want


def test_delta_psf_identity():
    img = do.synthetic_view((20, 36, 50), do.gaussian_psf((9, 9, 9), (2, 1.5, 1.5)), n_blobs=6)
    psf = np.zeros((5, 5, 5), dtype=f32)
    psf[2, 2, 2] = 3
    out = do.decon_singleview(img, psf, 4)
    assert np.max(np.abs(out - do.clamp_small(img))) <= 1e-6 * img.max()


def test_unnormalised_round_trip_scale():
    """Forward pass yields N*(E conv psf); the two factors N cancel."""
    F = (8, 16, 16)
    rng = np.random.default_rng(0)
    E = rng.uniform(1, 2, F).astype(f32)
    otf = do.gen_otf(do.gaussian_psf((5, 5, 5), (1, 1, 1)), F)
    t = do.c2r(do.cmul(do.r2c(E), otf), F)
    n = np.prod(F)
    assert abs(t.mean() / n - E.mean()) < 1e-4


def test_pad_psf_puts_centre_at_origin():
    psf = np.zeros((4, 5, 6), dtype=f32)
    psf[2, 2, 3] = 1  # index s//2 on every axis
    padded = do.pad_psf(psf, (8, 8, 8))
    assert padded[0, 0, 0] == 1 and padded.sum() == 1


def test_align_size_c_truncation():
    a = np.arange(5, dtype=f32).reshape(1, 1, 5)
    # (4 - 5)/2 == 0 in C (truncation), -1 with floor division
    assert np.array_equal(do.align_size(a, (1, 1, 4))[0, 0], [0, 1, 2, 3])
    assert np.array_equal(do.align_size(a, (1, 1, 2))[0, 0], [1, 2])  # (2-5)/2 = -1
    assert np.array_equal(do.align_size(a, (1, 1, 8))[0, 0], [0, 0, 1, 2, 3, 4, 0, 0])  # (8-5)/2 = 1


def test_golden_decon_reproducible():
    g = np.load(os.path.join(GOLD, "decon_sv_psf_eq_fft.npz"))
    out = do.decon_singleview(g["img"], g["psf"], int(g["iters"]))
    assert np.max(np.abs(out - g["expected"])) <= 1e-5 * np.abs(g["expected"]).max()


def test_identity_resample_is_exact():
    src = np.random.default_rng(0).standard_normal((5, 7, 9)).astype(f32)
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32)
    assert np.array_equal(ro.affine_transform(src, ident, src.shape), src)


def test_window_rules_differ_between_cost_and_resample():
    """cost: 0 < t < s (strict); resample: 0 <= t < s."""
    src = np.ones((4, 4, 4), dtype=f32)
    m = np.array([1, 0, 0, -0.5, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32)  # t_x = x - 0.5 + 0.5 = x -> 0 at x = 0
    out = ro.affine_transform(src, m, src.shape)
    assert out[0, 0, 0] == 1  # inside for the resample
    ss, _ = ro.corr_sums(src, src, m)
    assert ss == 4 * 4 * 3  # the x = 0 column is outside for the cost


def test_self_registration_cost_is_minus_one():
    vol = do.blob_volume((12, 16, 20), n_blobs=6, seed=1)
    ms, _ = ro.mean_subtract(vol)
    sd = f32(np.sqrt(float(np.sum((ms * ms).astype(np.float64)))))
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32)
    assert abs(float(ro.corrfunc(ms, sd, ms, ident)) - 1.0) < 1e-6


def test_degenerate_cost_is_minus_two():
    vol = do.blob_volume((8, 8, 8), n_blobs=2, seed=1)
    m = np.array([1, 0, 0, 100, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32)
    assert ro.corrfunc(vol, f32(1), vol, m) == f32(-2.0)
