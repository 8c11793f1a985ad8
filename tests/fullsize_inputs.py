"""Closed-form inputs for the benchmark-size (512 x 512 x 256) parity fixtures.

Every value is an integer function of the voxel index computed in uint32 arithmetic, small enough to be exact in
float32, so the GPU box rebuilds exactly the bytes the build container fed to the oracle -- no RNG state, no libm, no
FFT.  Shared by tests/golden/make_golden_fullsize.py (which runs the oracle on them) and the ``-m gpu`` tests (which
run the HIP path on them); nothing here touches oracle/ or the product.
"""
import numpy as np

FULL_SHAPE = (256, 512, 512)  # ndarray [z][y][x] of imSize {512, 512, 256}: BASELINE.json configs 2-4
N_SAMPLES = 65536
SAMPLE_STRIDE = 1000003  # odd -> a permutation of 2^26 voxel indices


def _mix(h):
    h = h.copy()
    h ^= h >> np.uint32(13)
    h *= np.uint32(0x5BD1E995)
    h ^= h >> np.uint32(15)
    h *= np.uint32(0x27D4EB2F)
    h ^= h >> np.uint32(16)
    return h


def _lattice(shape, sx, sy, sz, salt):
    nz, ny, nx = shape
    z = (np.arange(nz, dtype=np.uint32) >> np.uint32(sz))[:, None, None]
    y = (np.arange(ny, dtype=np.uint32) >> np.uint32(sy))[None, :, None]
    x = (np.arange(nx, dtype=np.uint32) >> np.uint32(sx))[None, None, :]
    with np.errstate(over="ignore"):
        h = (x * np.uint32(73856093)) ^ (y * np.uint32(19349663)) ^ (z * np.uint32(83492791)) ^ np.uint32(salt)
        return _mix(h)


def hash_volume(shape, salt, structure_salt=12345):
    """float32 [z][y][x]: background 100 + per-voxel noise 0..63 (salted) + bright 16x16x8 bricks (one in eight,
    amplitude 200..1223, shared by every salt so two views show the same structure).  Integers <= 1386."""
    noise = _lattice(shape, 0, 0, 0, salt) & np.uint32(63)
    hb = _lattice(shape, 4, 4, 3, structure_salt)
    amp = np.where((hb & np.uint32(7)) == 0, np.uint32(200) + ((hb >> np.uint32(3)) & np.uint32(1023)), np.uint32(0))
    return (np.uint32(100) + noise + amp).astype(np.float32)


def displaced_source(shape, noise_salt=3):
    """The registration source of the end-to-end fixture (config 4): view 3's noise on the shared brick structure, with
    the structure sampled at SHEARED and shifted integer coordinates

        x' = x + (y >> 5) - 5,   y' = y - (z >> 4) + 3,   z' = z + (x >> 6) - 2      (clamped to the volume)

    -- a piecewise-integer stand-in for a rotation of a degree or two per axis pair plus a shift, in uint32 arithmetic like
    everything else here.  The optimum of the target -> source matrix is therefore a genuine 12-parameter matrix
    (translation about (5, -3, 2), shears 1/32, 1/16, 1/64), which the 3 -> 6 -> 9 -> 12 DOF cascade approaches in stages."""
    nz, ny, nx = shape
    z = np.arange(nz, dtype=np.int64)[:, None, None]
    y = np.arange(ny, dtype=np.int64)[None, :, None]
    x = np.arange(nx, dtype=np.int64)[None, None, :]
    xs = np.clip(x + (y >> 5) - 5 + 0 * z, 0, nx - 1).astype(np.uint32)
    ys = np.clip(y - (z >> 4) + 3 + 0 * x, 0, ny - 1).astype(np.uint32)
    zs = np.clip(z + (x >> 6) - 2 + 0 * y, 0, nz - 1).astype(np.uint32)
    with np.errstate(over="ignore"):
        hb = _mix(((xs >> np.uint32(4)) * np.uint32(73856093)) ^ ((ys >> np.uint32(4)) * np.uint32(19349663)) ^
                  ((zs >> np.uint32(3)) * np.uint32(83492791)) ^ np.uint32(12345))
    del xs, ys, zs
    amp = np.where((hb & np.uint32(7)) == 0, np.uint32(200) + ((hb >> np.uint32(3)) & np.uint32(1023)), np.uint32(0))
    noise = _lattice(shape, 0, 0, 0, noise_salt) & np.uint32(63)
    return (np.uint32(100) + noise + amp).astype(np.float32)


def binomial_psf(nz, ny, nx):
    """Unnormalised separable PSF of binomial rows C(n-1, k) (n odd -> symmetric about n//2): an integer-valued
    stand-in for a Gaussian of sigma = sqrt(n-1)/2 per axis."""
    def row(n):
        r = np.ones(1, dtype=np.int64)
        for _ in range(n - 1):
            r = np.convolve(r, np.array([1, 1], dtype=np.int64))
        return r
    p = row(nz)[:, None, None] * row(ny)[None, :, None] * row(nx)[None, None, :]
    return p.astype(np.float32)


def psf_a():
    return binomial_psf(33, 9, 9)  # sigma (z, y, x) = (2.83, 1.41, 1.41)


def psf_b():
    return binomial_psf(9, 9, 33)


def psf_bench_like():
    """64 x 64 x 64, even-sized and anisotropic like bench.py's Gaussian (sigma (z, y, x) ~ (6, 2.5, 2.5)): rows of a
    binomial of width 4 sigma^2 + 1 centred at index 32 and zero-padded to 64 samples, scaled by exact powers of two --
    additions and multiplications of float64 numbers only, cast to float32 at the end."""
    def row(n_taps):
        r = np.ones(1, dtype=np.float64)
        for _ in range(n_taps - 1):
            r = np.convolve(r, np.array([0.5, 0.5]))
        out = np.zeros(64, dtype=np.float64)
        lo = 32 - n_taps // 2
        out[lo:lo + n_taps] = r
        return out
    p = row(63)[:, None, None] * row(25)[None, :, None] * row(25)[None, None, :]  # sigma = sqrt(n - 1) / 2 = 3.9, 2.4, 2.4
    return (p * 2.0 ** 20).astype(np.float32)


def sample_indices(n_voxels, count=N_SAMPLES):
    return (np.arange(count, dtype=np.int64) * SAMPLE_STRIDE + 7) % n_voxels


def registration_matrices():
    """Row-major 3x4 float32 matrices (target voxel -> source coordinate) for the full-size cost / resample checks:
    identity, the benchmark's 3 deg about y + 2 % + shift, 10 deg about z, 45 deg about x (about the volume centre),
    an anisotropic scale with shift, and a general 12-DOF matrix.  cos / sin are given as literal decimals."""
    nz, ny, nx = FULL_SHAPE
    cx, cy, cz = (nx - 1) / 2.0, (ny - 1) / 2.0, (nz - 1) / 2.0

    def about_centre(r, shift=(0.0, 0.0, 0.0)):
        r = np.asarray(r, dtype=np.float64).reshape(3, 3)
        c = np.array([cx, cy, cz])
        t = c - r @ c + np.asarray(shift)
        return np.concatenate([r, t[:, None]], axis=1).reshape(12).astype(np.float32)

    c3, s3 = 0.9986295347545738, 0.05233595624294383
    c10, s10 = 0.984807753012208, 0.17364817766693033
    c45 = 0.7071067811865476
    mats = [
        np.eye(3, 4).reshape(12).astype(np.float32),
        about_centre(1.02 * np.array([[c3, 0, -s3], [0, 1, 0], [s3, 0, c3]]), (5.0, -3.0, 2.0)),
        about_centre([[c10, s10, 0], [-s10, c10, 0], [0, 0, 1]]),
        about_centre([[1, 0, 0], [0, c45, c45], [0, -c45, c45]]),
        about_centre(np.diag([0.9, 1.1, 1.05]), (-7.25, 4.5, 1.75)),
        about_centre([[1.01, 0.02, -0.03], [-0.015, 0.98, 0.025], [0.04, -0.01, 1.03]], (3.3, -2.2, 1.1)),
    ]
    return np.stack(mats)
