"""ORACLE (test infrastructure, not product): NumPy restatement of the reference
Richardson-Lucy deconvolution path, GPU-path semantics (``gpuMemMode 1``).

PARITY STATUS: **partly pinned**.  eguomin/microImageLib ships no tests / golden vectors for
this path.  What IS pinned, bit for bit, to reference code compiled by gcc
(oracle/Makefile target `subfunc`: the plain-C++ host helpers of src/api_subfunc.cu, cut out by
line range; tests/test_oracle_pinned.py): snap_transform_size, flip3, pad_psf (equal-size case),
pad_stack, crop_stack, align_size, clamp_small, sum_f64, cmul, rl_ratio, rl_update.  What is NOT:
the order of operations of the RL loop and of gen_otf (restated from decon_*_OTF1 / genOTFgpu
below; those functions need cuda_runtime.h / cufft.h / fftw3.h and cannot be built here without
stand-ins, so they are not built) and the FFT itself (pocketfft here; any unnormalised DFT is a
valid stand-in for cuFFT / FFTW up to rounding).  Additionally checked against closed-form known
answers (tests/test_oracle.py).

Array convention: every volume is a C-ordered ``[z][y][x]`` float32 ndarray that
views the TIFF-ordered flat buffer ``idx = x + sx*(y + sy*z)`` the C API takes
(``imSize = {sx, sy, sz}``).  The reference renames axes so that its "x" is the
slowest one (src/api_decon.cpp:68,72) -- in ndarray terms nothing moves.

All FFTs are unnormalised in both directions, exactly like cuFFT / FFTW
(src/api_subfunc.cu:3404-3416 applies no scaling; the two factors N cancel).
"""
from __future__ import annotations

import numpy as np

SMALLVALUE = np.float32(0.01)  # src/api_subfunc.cu:24


# --------------------------------------------------------------------------
# src/api_subfunc.cu:57-87  snapTransformSize
# --------------------------------------------------------------------------
def snap_transform_size(n: int) -> int:
    n = int(n)
    n = n if n % 16 == 0 else n - n % 16 + 16  # iAlignUp(dataSize, 16)  :62
    hi_bit = n.bit_length() - 1
    low_pot = 1 << hi_bit
    if low_pot == n:  # :72
        return n
    hi_pot = 1 << (hi_bit + 1)
    if hi_pot <= 128:  # :79
        return hi_pot
    return n if n % 64 == 0 else n - n % 64 + 64  # iAlignUp(dataSize, 64)  :85


def fft_shape(shape):
    """FFT box for an image of ndarray shape (sz, sy, sx) -- src/api_decon.cpp:77-79."""
    return tuple(snap_transform_size(s) for s in shape)


def _cdiv2(a: int) -> int:
    """C integer division by 2 (truncates toward zero), as in (sx - sx2) / 2."""
    return int(a / 2) if a < 0 else a // 2


# --------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------
def flip3(a: np.ndarray) -> np.ndarray:
    """flipgpukernel, include/cukernel.cuh:668-677: reversal on all three axes."""
    return np.ascontiguousarray(a[::-1, ::-1, ::-1])


def align_size(a: np.ndarray, out_shape) -> np.ndarray:
    """alignsize3Dgpu, src/api_subfunc.cu:1779-1790 + cukernel.cuh:755-769:
    centred crop-or-zero-pad; offset = (S - s) / 2 with C truncation."""
    out = np.zeros(out_shape, dtype=a.dtype)
    src_sl, dst_sl = [], []
    for S, s in zip(out_shape, a.shape):
        so = _cdiv2(S - s)
        # out[d] = in[d - so] when 0 <= d - so < s
        d0 = max(0, so)
        d1 = min(S, so + s)
        dst_sl.append(slice(d0, d1))
        src_sl.append(slice(d0 - so, d1 - so))
    out[tuple(dst_sl)] = a[tuple(src_sl)]
    return out


def pad_psf(p: np.ndarray, F) -> np.ndarray:
    """padPSFgpu, src/api_subfunc.cu:1691-1701 + cukernel.cuh:680-697:
    zero box, then out[(i - s//2) mod F] = in[i] on every axis (PSF centre -> origin)."""
    out = np.zeros(F, dtype=p.dtype)
    idx = []
    for S, s in zip(F, p.shape):
        i = np.arange(s) - s // 2
        i[i < 0] += S
        assert (i >= 0).all() and (i < S).all()
        idx.append(i)
    out[np.ix_(*idx)] = p
    return out


def pad_stack(a: np.ndarray, F) -> np.ndarray:
    """padstackgpu, src/api_subfunc.cu:1713-1724 + cukernel.cuh:700-737:
    centred placement, offset (F - s)/2, edge replication outside."""
    idx = []
    for S, s in zip(F, a.shape):
        so = (S - s) // 2
        idx.append(np.clip(np.arange(S) - so, 0, s - 1))
    return np.ascontiguousarray(a[np.ix_(*idx)])


def crop_stack(a: np.ndarray, shape) -> np.ndarray:
    """cropgpu, src/api_subfunc.cu:1736-1747 + cukernel.cuh:740-753: centred crop."""
    sl = []
    for S, s in zip(a.shape, shape):
        so = (S - s) // 2
        sl.append(slice(so, so + s))
    return np.ascontiguousarray(a[tuple(sl)])


def clamp_small(a: np.ndarray) -> np.ndarray:
    """maxvalue3Dgpukernel, include/cukernel.cuh:383-392: ``(v > eps) ? v : eps`` -- a NaN
    collapses to eps (np.maximum would propagate it)."""
    return np.where(a > SMALLVALUE, a, SMALLVALUE).astype(np.float32)


def sum_f64(a: np.ndarray) -> float:
    """sum3Dgpu, src/api_subfunc.cu:386-402: float64 accumulation of a float32 volume."""
    return float(np.sum(a, dtype=np.float64))


# --------------------------------------------------------------------------
# FFT (unnormalised, float32 in / complex64 out)
# --------------------------------------------------------------------------
def r2c(a: np.ndarray) -> np.ndarray:
    out = np.fft.rfftn(a.astype(np.float32, copy=False))
    return out.astype(np.complex64, copy=False)


def c2r(s: np.ndarray, F) -> np.ndarray:
    out = np.fft.irfftn(s, s=F, axes=(0, 1, 2), norm="forward")  # norm="forward": inverse is unscaled
    return out.astype(np.float32, copy=False)


def cmul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """multicomplex3Dkernel, include/cukernel.cuh:140-152: plain complex product in
    float32, each of the four products / two sums rounded separately (no FMA)."""
    ar, ai = a.real, a.imag
    br, bi = b.real, b.imag
    out = np.empty(a.shape, dtype=np.complex64)
    out.real = ar * br - ai * bi
    out.imag = ar * bi + ai * br
    return out


# --------------------------------------------------------------------------
# src/api_subfunc.cu:3269-3307  genOTFgpu
# --------------------------------------------------------------------------
def gen_otf(psf: np.ndarray, F, normalise: bool = True) -> np.ndarray:
    psf = np.asarray(psf, dtype=np.float32)
    if normalise:
        s = sum_f64(psf)  # :3283
        p = psf * np.float32(1.0 / s)  # (float)(1 / sumValue)  :3284
    else:
        p = psf.copy()
    if any(Fd < sd for Fd, sd in zip(F, psf.shape)):  # :3288
        p = align_size(p, F)  # :3289
        padded = pad_psf(p, F)  # :3290 (PSF box == FFT box, origin F//2)
    else:
        padded = pad_psf(p, F)  # :3293
    return r2c(padded)  # :3298-3299


# --------------------------------------------------------------------------
# RL loops
# --------------------------------------------------------------------------
def rl_view_update(E, M, otf, otf_bp, F):
    """One view update, src/api_subfunc.cu:3406-3415 (== :3636-3646)."""
    T = c2r(cmul(r2c(E), otf), F)  # R2C, *OTF, C2R
    T = M / T  # div3Dgpu, unguarded  :3409
    T = c2r(cmul(r2c(T), otf_bp), F)
    E = E * T  # multi3Dgpu  :3414
    return clamp_small(E)  # maxvalue3Dgpu  :3415


def decon_singleview_otf(A, otf, otf_bp, iters: int, const_init: bool):
    """decon_singleview_OTF1, src/api_subfunc.cu:3361-3430, on an FFT-sized volume."""
    F = A.shape
    A = clamp_small(A.astype(np.float32))  # :3380
    if const_init:
        mean_value = np.float32(sum_f64(A))  # (float)sum3Dgpu -- the SUM  :3382
        E = np.full(F, mean_value, dtype=np.float32)  # :3383-3384
    else:
        E = A.copy()  # :3387
    for _ in range(iters):
        E = rl_view_update(E, A, otf, otf_bp, F)
    return E


def decon_dualview_otf(A, B, otf1, otf2, otf_bp1, otf_bp2, iters: int, const_init: bool):
    """decon_dualview_OTF1, src/api_subfunc.cu:3587-3674."""
    F = A.shape
    A = clamp_small(A.astype(np.float32))  # :3606
    B = clamp_small(B.astype(np.float32))  # :3607
    if const_init:
        m1 = np.float32(sum_f64(A))  # :3610
        m2 = np.float32(sum_f64(B))  # :3611
        E = np.full(F, (m1 + m2) / np.float32(2), dtype=np.float32)  # :3613
    else:
        E = (A + B) * np.float32(0.5)  # :3616-3617
    for _ in range(iters):
        E = rl_view_update(E, A, otf1, otf_bp1, F)  # :3636-3646
        E = rl_view_update(E, B, otf2, otf_bp2, F)  # :3649-3659
    return E


# --------------------------------------------------------------------------
# API level (src/api_decon.cpp:53-331 case 1, :333-704 case 1)
# --------------------------------------------------------------------------
def _otf_pair(psf, psf_bp, F, unmatch):
    otf = gen_otf(psf, F)  # src/api_decon.cpp:214
    if unmatch:
        otf_bp = gen_otf(psf_bp, F)  # :218
    else:
        otf_bp = gen_otf(flip3(np.asarray(psf, dtype=np.float32)), F)  # :221-222
    return otf, otf_bp


def decon_singleview(img, psf, iters, const_init=False, unmatch=False, psf_bp=None):
    img = np.asarray(img, dtype=np.float32)
    F = fft_shape(img.shape)
    otf, otf_bp = _otf_pair(psf, psf_bp, F, unmatch)
    A = pad_stack(img, F) if F != img.shape else img  # src/api_decon.cpp:225-231
    E = decon_singleview_otf(A, otf, otf_bp, iters, const_init)  # :234
    return crop_stack(E, img.shape) if F != img.shape else E  # :237-243


def decon_dualview(img1, img2, psf1, psf2, iters, const_init=False, unmatch=False,
                   psf_bp1=None, psf_bp2=None):
    img1 = np.asarray(img1, dtype=np.float32)
    img2 = np.asarray(img2, dtype=np.float32)
    F = fft_shape(img1.shape)
    otf1, otf_bp1 = _otf_pair(psf1, psf_bp1, F, unmatch)  # src/api_decon.cpp:524-534
    otf2, otf_bp2 = _otf_pair(psf2, psf_bp2, F, unmatch)  # :537-547
    if F != img1.shape:
        A, B = pad_stack(img1, F), pad_stack(img2, F)  # :551-565
    else:
        A, B = img1, img2
    E = decon_dualview_otf(A, B, otf1, otf2, otf_bp1, otf_bp2, iters, const_init)  # :569
    return crop_stack(E, img1.shape) if F != img1.shape else E  # :572-578


# --------------------------------------------------------------------------
# Element-wise kernels in isolation (bit-exact targets for the HIP kernels)
# --------------------------------------------------------------------------
def rl_ratio(A, T):
    """div3Dkernel, include/cukernel.cuh:196-206."""
    return (A / T).astype(np.float32)


def rl_update(E, T):
    """multi3Dkernel + maxvalue3Dgpukernel, include/cukernel.cuh:115-126, 383-392."""
    return clamp_small((E * T).astype(np.float32))


# --------------------------------------------------------------------------
# synthetic inputs -- shared by tests and bench
# --------------------------------------------------------------------------
def gaussian_psf(shape, sigma_zyx) -> np.ndarray:
    """Unnormalised anisotropic Gaussian PSF, centre at shape//2 (the index the
    reference treats as the PSF origin, cukernel.cuh:687 with sox = s/2)."""
    g = []
    for s, sg in zip(shape, sigma_zyx):
        c = np.arange(s, dtype=np.float64) - s // 2
        g.append(np.exp(-0.5 * (c / sg) ** 2))
    return (g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]).astype(np.float32)


def blob_volume(shape, n_blobs=40, seed=0, background=100.0) -> np.ndarray:
    """Ground-truth volume: Gaussian blobs on a constant background."""
    rng = np.random.default_rng(seed)
    sz, sy, sx = shape
    vol = np.full(shape, background, dtype=np.float64)
    zz, yy, xx = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij", sparse=True)
    for _ in range(n_blobs):
        c = rng.uniform(0, 1, 3) * np.array(shape)
        sg = rng.uniform(2.0, 5.0)
        amp = rng.uniform(200.0, 2000.0)
        r = int(4 * sg) + 1
        z0, z1 = max(0, int(c[0]) - r), min(sz, int(c[0]) + r + 1)
        y0, y1 = max(0, int(c[1]) - r), min(sy, int(c[1]) + r + 1)
        x0, x1 = max(0, int(c[2]) - r), min(sx, int(c[2]) + r + 1)
        d2 = ((zz[z0:z1] - c[0]) ** 2 + (yy[:, y0:y1] - c[1]) ** 2 + (xx[:, :, x0:x1] - c[2]) ** 2)
        vol[z0:z1, y0:y1, x0:x1] += amp * np.exp(-0.5 * d2 / sg ** 2)
    return vol.astype(np.float32)


def blur(vol: np.ndarray, psf: np.ndarray) -> np.ndarray:
    """Circular convolution of vol with the normalised PSF (float64 FFT)."""
    p = psf.astype(np.float64)
    p /= p.sum()
    padded = pad_psf(align_size(p, vol.shape) if any(a < b for a, b in zip(vol.shape, p.shape)) else p,
                     vol.shape)
    out = np.fft.irfftn(np.fft.rfftn(vol.astype(np.float64)) * np.fft.rfftn(padded), s=vol.shape, axes=(0, 1, 2))
    return out.astype(np.float32)


def synthetic_view(shape, psf, seed=0, noise_seed=1, n_blobs=40) -> np.ndarray:
    truth = blob_volume(shape, n_blobs=n_blobs, seed=seed)
    img = np.maximum(blur(truth, psf), 0)
    rng = np.random.default_rng(noise_seed)
    return rng.poisson(img).astype(np.float32)
