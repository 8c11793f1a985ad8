"""Pins the oracle's geometry / matrix / element-wise / trilinear pieces to REFERENCE CODE: the plain-C++ host helpers of
/root/reference/src/api_subfunc.cu, compiled where they lie by `make -C oracle subfunc` into
oracle/_ref/libsubfunc_ref.so (line ranges in oracle/Makefile; wrappers in oracle/ref_subfunc_shim.cpp).

What this pins (bit-exact unless stated): snapTransformSize; sum / add-value / multiply / divide / max-value / complex
multiply loops (the arithmetic of the RL ratio, update, clamp, OTF product and mean subtraction); p2matrix, matrix2p,
matrixmultiply, rot2matrix, dof9tomatrix; flip, PSF origin shift (equal-size case, the only one padPSFcpu gets right:
SURVEY.md 8a row a10), edge-replicating pad, centred crop, align-size; and the trilinear interpolation + validity window
+ correlation sums through affinetransformcpu / corrfunccpu2 (interior voxels, where the CPU helper's bounds rule and
the CUDA kernel's agree).  What stays unpinned: the FFT itself (cuFFT / FFTW are absent; any correct DFT is a valid
stand-in) and the order of operations of the RL loop, restated from src/api_subfunc.cu:3404-3416.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import decon_oracle as do
from oracle import reg_oracle as ro

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libsubfunc_ref.so")
f32 = np.float32
fp = C.POINTER(C.c_float)
LL = C.c_longlong


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(LIB):
        if not os.path.isdir("/root/reference"):
            pytest.skip("oracle/_ref/libsubfunc_ref.so needs the reference tree at build time")
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "subfunc"])
    lib = C.CDLL(LIB)
    lib.ref_sumcpu.restype = C.c_double
    lib.ref_sumcpu.argtypes = [fp, C.c_size_t]
    lib.ref_addvaluecpu.argtypes = [fp, fp, C.c_float, C.c_size_t]
    lib.ref_maxvaluecpu.argtypes = [fp, fp, C.c_float, C.c_size_t]
    for name in ("ref_multicpu", "ref_divcpu", "ref_multicomplexcpu"):
        getattr(lib, name).argtypes = [fp, fp, fp, C.c_size_t]
    lib.ref_rot2matrix.argtypes = [fp, C.c_float, LL, LL, LL, C.c_int]
    lib.ref_dof9tomatrix.argtypes = [fp, fp, C.c_int]
    lib.ref_itrilerp2.restype = C.c_float
    lib.ref_itrilerp2.argtypes = [C.c_float] * 6 + [fp]
    lib.ref_affinetransformcpu.argtypes = [fp, fp, fp] + [C.c_int] * 6
    lib.ref_corrfunccpu2.restype = C.c_double
    lib.ref_corrfunccpu2.argtypes = [fp, fp, fp] + [C.c_int] * 6
    lib.ref_flipcpu.argtypes = [fp, fp, LL, LL, LL]
    for name in ("ref_padPSFcpu", "ref_padstackcpu", "ref_cropcpu", "ref_alignsize3Dcpu"):
        getattr(lib, name).argtypes = [fp, fp] + [LL] * 6
    return lib


def P(a):
    assert a.dtype == f32 and a.flags.c_contiguous
    return a.ctypes.data_as(fp)


def bits(a):
    return np.ascontiguousarray(a, dtype=f32).view(np.uint32)


# ------------------------------------------------------------------ a1
def test_snap_transform_size(ref):
    for n in range(1, 4200):
        assert do.snap_transform_size(n) == ref.ref_snapTransformSize(n), n


# ------------------------------------------------------------------ a10 / a11 element-wise arithmetic
def test_elementwise_loops(ref):
    rng = np.random.default_rng(0)
    n = 100_003
    a = rng.uniform(-3, 900, n).astype(f32)
    b = rng.uniform(-2, 5, n).astype(f32)
    a[:8] = [0, -0.0, np.inf, -np.inf, np.nan, 0.01, 0.0100001, 1e-30]
    b[:8] = [0, 1, 2, np.inf, 3, 0, np.nan, 1e30]
    out = np.empty(n, dtype=f32)
    with np.errstate(all="ignore"):
        ref.ref_divcpu(P(out), P(a), P(b), n)  # T = A / T   src/api_subfunc.cu:142-146
        assert np.array_equal(bits(do.rl_ratio(a, b)), bits(out))
        ref.ref_multicpu(P(out), P(a), P(b), n)  # E * T, then max(., 0.01)   :133-137, :200-208
        ref.ref_maxvaluecpu(P(out), P(out), f32(0.01), n)
        assert np.array_equal(bits(do.rl_update(a, b)), bits(out))
        ref.ref_maxvaluecpu(P(out), P(a), f32(0.01), n)
        assert np.array_equal(bits(do.clamp_small(a)), bits(out))
    fin = rng.uniform(0, 2000, n).astype(f32)
    s = ref.ref_sumcpu(P(fin), n)  # sequential float64 sum  :93-99
    assert abs(do.sum_f64(fin) - s) <= 1e-12 * s
    ms, total = ro.mean_subtract(fin)  # -float(sum) / float(N), then add  (src/api_subfunc.cu:2840 + addvaluecpu :114)
    assert abs(total - s) <= 1e-12 * s
    ref.ref_addvaluecpu(P(out), P(fin), f32(-(f32(total) / f32(n))), n)
    assert np.array_equal(bits(ms), bits(out))


def test_complex_multiply(ref):
    rng = np.random.default_rng(1)
    n = 50_000
    a = (rng.normal(0, 100, n) + 1j * rng.normal(0, 100, n)).astype(np.complex64)
    b = (rng.normal(0, 1, n) + 1j * rng.normal(0, 1, n)).astype(np.complex64)
    out = np.empty(n, dtype=np.complex64)
    ref.ref_multicomplexcpu(out.ctypes.data_as(fp), a.ctypes.data_as(fp), b.ctypes.data_as(fp), n)  # :160-169
    assert np.array_equal(do.cmul(a, b).view(np.uint32), out.view(np.uint32))


# ------------------------------------------------------------------ a16 matrices
def test_matrix_helpers(ref):
    rng = np.random.default_rng(2)
    for _ in range(200):
        x = rng.uniform(-5, 5, 13).astype(f32)
        m = np.zeros(12, dtype=f32)
        ref.ref_p2matrix(P(m), P(x))
        assert np.array_equal(bits(ro.p2matrix(x)), bits(m))
        back = np.full(13, 7, dtype=f32)
        ref.ref_matrix2p(P(m), P(back))
        assert np.array_equal(bits(ro.matrix2p(m)), bits(back))
        m1 = np.zeros(16, dtype=f32)
        m2 = np.zeros(16, dtype=f32)
        m1[:12] = rng.uniform(-2, 2, 12)
        m2[:12] = rng.uniform(-300, 300, 12)
        out = np.zeros(16, dtype=f32)
        ref.ref_matrixmultiply(P(out), P(m1), P(m2))
        assert np.array_equal(bits(ro.matrixmultiply(m1[:12], m2[:12])), bits(out[:12]))


def test_dof9tomatrix(ref):
    """Bit-exact: the oracle takes cosf / sinf from the C library like the reference does (src/api_subfunc.cu:715-824)."""
    rng = np.random.default_rng(3)
    for dof in (3, 6, 7, 9):
        for _ in range(300):
            p = np.zeros(10, dtype=f32)
            p[1:4] = rng.uniform(-40, 40, 3)
            p[4:7] = rng.uniform(-30, 30, 3)
            p[7:10] = rng.uniform(0.8, 1.25, 3)
            out = np.zeros(12, dtype=f32)
            ref.ref_dof9tomatrix(P(out), P(p), dof)
            assert np.array_equal(bits(ro.dof9tomatrix(p, dof)), bits(out)), (dof, p)


def test_rot2matrix(ref):
    """src/api_subfunc.cu:626-713, every angle mip3dgpu uses (src/apifunc.cpp:576-644) about every axis."""
    for axis in (1, 2, 3):
        for k in range(36):
            theta = f32(2 * 3.14159 / 36 * k)
            for (sx, sy, sz) in ((512, 512, 256), (97, 33, 50)):
                out = np.zeros(12, dtype=f32)
                ref.ref_rot2matrix(P(out), theta, sx, sy, sz, axis)
                assert np.array_equal(bits(ro.rot2matrix(theta, sx, sy, sz, axis)), bits(out)), (axis, k)


# ------------------------------------------------------------------ a5 / a6 / a10 geometry
GEOM_CASES = [((5, 6, 7), (8, 8, 8)), ((20, 36, 50), (32, 64, 64)), ((7, 9, 4), (7, 16, 4)), ((3, 3, 3), (3, 3, 3))]


def test_flip(ref):
    rng = np.random.default_rng(4)
    for shape in ((5, 6, 7), (4, 4, 4), (1, 9, 2)):
        a = rng.uniform(0, 1, shape).astype(f32)
        out = np.empty_like(a)
        ref.ref_flipcpu(P(out), P(a), *shape)  # :1467-1477
        assert np.array_equal(do.flip3(a), out)


@pytest.mark.parametrize("small,big", GEOM_CASES)
def test_pad_crop_align(ref, small, big):
    rng = np.random.default_rng(5)
    a = rng.uniform(0, 1, small).astype(f32)
    out = np.full(big, -1, dtype=f32)
    ref.ref_padstackcpu(P(out), P(a), *big, *small)  # edge-replicating centred pad  :1514-1556
    assert np.array_equal(do.pad_stack(a, big), out)
    back = np.full(small, -1, dtype=f32)
    ref.ref_cropcpu(P(back), P(out), *small, *big)  # centred crop  :1568-1586
    assert np.array_equal(do.crop_stack(out, small), back)
    assert np.array_equal(back, a)
    al = np.full(big, -1, dtype=f32)
    ref.ref_alignsize3Dcpu(P(al), P(a), *big, *small)  # zero pad  :1622-1641
    assert np.array_equal(do.align_size(a, big), al)
    b = rng.uniform(0, 1, big).astype(f32)
    cr = np.full(small, -1, dtype=f32)
    ref.ref_alignsize3Dcpu(P(cr), P(b), *small, *big)  # centred crop through the same helper
    assert np.array_equal(do.align_size(b, small), cr)


@pytest.mark.parametrize("shape", [(8, 8, 8), (5, 6, 7), (16, 32, 32), (1, 4, 9)])
def test_psf_origin_shift_equal_sizes(ref, shape):
    """padPSFcpu is only correct when PSF box == FFT box (its source stride bug, :1487-1497); that case pins the origin
    convention: the PSF voxel at index s/2 lands on index 0."""
    rng = np.random.default_rng(6)
    p = rng.uniform(0, 1, shape).astype(f32)
    out = np.full(shape, -1, dtype=f32)
    ref.ref_padPSFcpu(P(out), P(p), *shape, *shape)
    assert np.array_equal(do.pad_psf(p, shape), out)


# ------------------------------------------------------------------ a17 / a18 trilinear, window, sums
def _to_ref_layout(a):
    """Oracle volume [z][y][x] (x fastest, the registration path's layout: src/api_reg.cpp:304) -> the CPU helpers'
    layout [x][y][z] (their first index `i` = x is the SLOWEST: `h_t[x*sy*sz + y*sz + z]`, src/api_subfunc.cu:1437).
    Axis NAMES stay the same, so the matrix passes through unchanged and itrilerp2 interpolates x first, then y, then z
    -- the same order by name as the oracle."""
    return np.ascontiguousarray(np.asarray(a, dtype=f32).transpose(2, 1, 0))


def _ref_affine(ref, src, m, shape):
    sz, sy, sx = shape
    s2z, s2y, s2x = src.shape
    out = np.full((sx, sy, sz), -1, dtype=f32)
    ref.ref_affinetransformcpu(P(out), P(_to_ref_layout(src)), P(np.ascontiguousarray(m, dtype=f32)), sx, sy, sz,
                               s2x, s2y, s2z)
    return np.ascontiguousarray(out.transpose(2, 1, 0))


def _interior(m, shape, src_shape):
    """Target voxels where the CPU helper's rule (0 <= floor(t), floor(t) + 1 < s) holds on every axis -- there the
    CUDA kernel's window and clamping are inert, so both must return the same value."""
    tx, ty, tz = ro._coords(m, shape)
    h = f32(0.5)
    ok = np.ones(shape, dtype=bool)
    for t, n in ((tx, src_shape[2]), (ty, src_shape[1]), (tz, src_shape[0])):
        b = t - h
        ok &= (b >= 0) & (np.floor(b) + 1 < n)
    return ok


DYADIC = [
    [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0],
    [1.0078125, 0.03125, -0.015625, 1.3125, -0.0234375, 0.984375, 0.0625, 2.75, 0.046875, -0.03125, 1.015625, -1.5],
    [0.75, 0, 0.25, 3.5, 0, 1.25, 0, -2.125, -0.25, 0, 0.875, 4.0625],
    [0.5, 0.5, 0, 1.03125, -0.5, 0.5, 0, 9.5, 0, 0, 1, 0.5],
]


@pytest.mark.parametrize("m", DYADIC)
def test_trilinear_bit_exact_on_dyadic_matrices(ref, m):
    """Matrix entries that are multiples of 2^-7 and volumes < 64 voxels wide: every coordinate is exact in float32,
    so the CUDA kernel's (t + 0.5) - 0.5 equals the CPU helper's t and the two interpolations must agree BIT FOR BIT
    wherever the bounds rules agree (affinetransformcpu :1409-1463, itrilerp2 :1075-1093 vs cukernel.cuh:501-524)."""
    rng = np.random.default_rng(7)
    shape, src_shape = (20, 28, 36), (22, 30, 33)
    src = rng.uniform(-50, 400, src_shape).astype(f32)
    m = np.array(m, dtype=f32)
    want = _ref_affine(ref, src, m, shape)
    got = ro.affine_transform(src, m, shape)
    ok = _interior(m, shape, src_shape)
    assert ok.sum() > 0.3 * ok.size
    assert np.array_equal(bits(got[ok]), bits(want[ok]))
    # correlation sums over the interior only: zero the target outside it so that both sides sum the same voxels
    tgt = rng.uniform(-100, 100, shape).astype(f32)
    tgt_in = np.where(ok, tgt, f32(0)).astype(f32)
    zncc_ref = ref.ref_corrfunccpu2(P(_to_ref_layout(tgt_in)), P(_to_ref_layout(src)), P(m), *shape[::-1],
                                    *src_shape[::-1])  # st / sqrt(ss)
    s = np.where(ok, got, f32(0)).astype(np.float64)
    # corrfunccpu2's ss runs over every voxel ITS rule accepts, i.e. the interior plus the band -1 < t < 0 where
    # (int) truncation extrapolates; recompute its ss from `want` so that only st is compared through the oracle
    ss_ref = float(np.sum(want.astype(np.float64) ** 2))
    st = float(np.sum(s * tgt_in.astype(np.float64)))
    assert abs(st / np.sqrt(ss_ref) - zncc_ref) <= 1e-12 * abs(zncc_ref)


def test_trilinear_general_matrices_close(ref):
    """General matrices: coordinates round differently ((t + 0.5) - 0.5 vs t), so closeness instead of bits."""
    rng = np.random.default_rng(8)
    shape = (24, 40, 48)
    src = rng.uniform(0, 400, shape).astype(f32)
    for _ in range(5):
        p = np.zeros(10, dtype=f32)
        p[1:4] = rng.uniform(-3, 3, 3)
        p[4:7] = rng.uniform(-8, 8, 3)
        p[7:10] = rng.uniform(0.95, 1.05, 3)
        m = ro.dof9tomatrix(p, 9)
        want = _ref_affine(ref, src, m, shape)
        got = ro.affine_transform(src, m, shape)
        ok = _interior(m, shape, shape)
        assert np.max(np.abs(got[ok] - want[ok])) < 2e-3  # |grad| * 1 ulp of a coordinate ~ 400 * 48 * 6e-8
        assert np.mean(bits(got[ok]) == bits(want[ok])) > 0.5


def test_itrilerp2_weights(ref):
    """The order of the three lerps and which weight multiplies which corner (x first, then y, then z)."""
    rng = np.random.default_rng(9)
    for _ in range(200):
        q = rng.uniform(-10, 10, 8).astype(f32)  # q111 q112 q121 q122 q211 q212 q221 q222 = (z, y, x) corners
        w = rng.uniform(0, 1, 3).astype(f32)
        dx1, dy1, dz1 = w
        dx2, dy2, dz2 = f32(1) - w
        want = ref.ref_itrilerp2(dx1, dx2, dy1, dy2, dz1, dz2, P(q))
        src = q.reshape(2, 2, 2)
        h = f32(0.5)
        got = ro._trilinear(src, np.array([dx1 + h], dtype=f32), np.array([dy1 + h], dtype=f32),
                            np.array([dz1 + h], dtype=f32))[0]
        # (w + 0.5) - 0.5 may round w; feed the reference the weights the oracle actually used
        bx, by, bz = f32(f32(dx1 + h) - h), f32(f32(dy1 + h) - h), f32(f32(dz1 + h) - h)
        want = ref.ref_itrilerp2(bx, f32(1) - bx, by, f32(1) - by, bz, f32(1) - bz, P(q))
        assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32)


# ------------------------------------------------------------------ a20: checkmatrix, compiled from src/api_reg.cpp:246-262
def test_checkmatrix_oracle_and_product_against_the_compiled_reference():
    """checkmatrix decides the batch's fall-back ladder (src/spim_fusion_batch.cpp:730-746).  The reference's own function
    (oracle/_ref/libregcheck_ref.so: src/api_reg.cpp:246-262 cut out by line range, `make -C oracle regcheck`) against the
    oracle's restatement AND libapi.so's checkmatrix on 10 000 random matrices plus every threshold and its float32
    neighbours on both sides (scale bounds 0.5 / 1.4, scale-sum bounds 2 / 4, shift bound 0.8 * extent), NaN and inf."""
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libregcheck_ref.so")
    if not os.path.exists(lib_path):
        if not os.path.isdir("/root/reference"):
            pytest.skip("oracle/_ref/libregcheck_ref.so needs the reference tree at build time")
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "regcheck"])
    ref = C.CDLL(lib_path)
    ref.ref_checkmatrix.argtypes = [fp, LL, LL, LL]
    from microimagelib_amd import api
    rng = np.random.default_rng(20)
    dims = [(512, 512, 256), (100, 37, 5), (1, 1, 1), (3000, 2, 700)]
    mats = []
    for _ in range(10000):
        m = rng.normal(0, 0.3, 12).astype(f32)
        m[[0, 5, 10]] = rng.uniform(0.3, 1.7, 3).astype(f32)
        m[[3, 7, 11]] = (rng.uniform(-1.2, 1.2, 3) * np.array(dims[len(mats) % 4])).astype(f32)
        mats.append(m)
    ident = np.eye(3, 4, dtype=f32).reshape(12)

    def near(v):
        v = f32(v)
        return [np.nextafter(v, f32(-np.inf)), v, np.nextafter(v, f32(np.inf))]
    for idx in (0, 5, 10):
        for bound in (0.5, 1.4):
            for v in near(bound):
                m = ident.copy()
                m[idx] = v
                mats.append(m)
    for total in (2.0, 4.0):  # the float32 sum (m0 + m5) + m10 at the bounds: 2 = 0.5 + 0.5 + 1, 4 = 1.4 + 1.3 + 1.3 ...
        for v in near(total - 1.4 - 0.6 if total == 2.0 else total - 1.4 - 1.4):
            m = ident.copy()
            m[[0, 5]] = (1.4, 0.6) if total == 2.0 else (1.4, 1.4)
            m[10] = v
            mats.append(m)
    for idx, axis in ((3, 0), (7, 1), (11, 2)):
        for d in dims:
            for sign in (1.0, -1.0):
                for v in near(f32(0.8) * f32(d[axis])):
                    m = ident.copy()
                    m[idx] = f32(sign) * v
                    mats.append(m)
    for bad in (np.nan, np.inf, -np.inf):
        for idx in (0, 3, 10, 11):
            m = ident.copy()
            m[idx] = bad
            mats.append(m)
    n_true = 0
    for i, m in enumerate(mats):
        m = np.ascontiguousarray(m, dtype=f32)
        for sx, sy, sz in (dims if i >= 10000 else [dims[i % 4]]):
            want = bool(ref.ref_checkmatrix(P(m), sx, sy, sz))
            assert ro.checkmatrix(m, sx, sy, sz) == want, (i, m, (sx, sy, sz), "oracle")
            assert api.checkmatrix(m, sx, sy, sz) == want, (i, m, (sx, sy, sz), "libapi.so")
            n_true += want
    assert 500 < n_true < len(mats) * 4  # both outcomes are exercised
