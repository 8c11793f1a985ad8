"""ORACLE (test infrastructure, not product): NumPy restatement of the reference
3-D affine registration path -- matrix parameterisations, trilinear resample,
correlation cost and the ``reg3d_affine1`` driver.

PARITY STATUS: **partly pinned**.  The reference has no tests / vectors for this path and
``reg3d`` has no CPU path (src/api_reg.cpp:390-393).  Pinned bit for bit to reference code compiled
by gcc (oracle/Makefile target `subfunc`; tests/test_oracle_pinned.py): p2matrix, matrix2p,
matrixmultiply, dof9tomatrix, rot2matrix, mean_subtract's arithmetic, and -- through the reference's
own itrilerp2 / affinetransformcpu / corrfunccpu2 -- the trilinear interpolation (order of the three
lerps, weight definitions) and the correlation sums wherever the CPU helper's bounds rule and the
CUDA kernel's window agree.  Restated from the CUDA source without a runnable reference: the
validity windows, the clamp addressing at the rim, the +0.5 texel-centre convention and the
``reg3d_affine1`` driver.  Interpolation weights are *exact float32* by default (the CUDA texture
unit quantises them to 8 fractional bits); ``WEIGHT_BITS = 8`` mirrors the
product's emulation switch.  The optimiser itself IS pinned: ``oracle/powell_oracle.py`` is checked
bit-for-bit against ``oracle/_ref/libpowell_ref.so``, built from the reference's own
``src/api_powell.c`` (oracle/Makefile).

Registration layout: API names used directly, ``x`` fastest
(src/api_reg.cpp:304, include/cukernel.cuh:549).  Volumes are ``[z][y][x]``
float32 ndarrays; matrices are row-major 3x4 ``m[12]`` mapping a target voxel
``(x, y, z)`` to a source coordinate.
"""
from __future__ import annotations

import numpy as np

import ctypes as _C

f32 = np.float32
NDIM = 12

# The reference's `cos(theta)` / `sin(theta)` on a float argument resolve to the float overloads, i.e. the C library's
# cosf / sinf (src/api_subfunc.cu:661-662, 795-796, 805-806, 815-816).  NumPy's float32 cos / sin are its own SIMD
# routines and differ from glibc's in the last bit for ~9 % of arguments, so the oracle calls libm -- checked bit-exact
# against the compiled reference helpers in tests/test_oracle_pinned.py.
_libm = _C.CDLL("libm.so.6")
_libm.cosf.restype = _libm.sinf.restype = _C.c_float
_libm.cosf.argtypes = _libm.sinf.argtypes = [_C.c_float]


def _cosf(a):
    return f32(_libm.cosf(float(f32(a))))


def _sinf(a):
    return f32(_libm.sinf(float(f32(a))))


# ---------------------------------------------------------------- matrices
def p2matrix(x):
    """src/api_subfunc.cu:557-562 -- 1-based 12-vector -> row-major 3x4."""
    m = np.zeros(12, dtype=f32)
    m[0], m[1], m[2], m[3] = x[4], x[5], x[6], x[1]
    m[4], m[5], m[6], m[7] = x[7], x[8], x[9], x[2]
    m[8], m[9], m[10], m[11] = x[10], x[11], x[12], x[3]
    return m


def matrix2p(m):
    """src/api_subfunc.cu:569-574."""
    x = np.zeros(13, dtype=f32)
    x[1], x[2], x[3], x[4] = m[3], m[7], m[11], m[0]
    x[5], x[6], x[7], x[8] = m[1], m[2], m[4], m[5]
    x[9], x[10], x[11], x[12] = m[6], m[8], m[9], m[10]
    return x


def matrixmultiply(m1, m2):
    """src/api_subfunc.cu:584-598 -- 3x4 affine product, float32, left-to-right."""
    m1 = np.asarray(m1, dtype=f32)
    m2 = np.asarray(m2, dtype=f32)
    m = np.zeros(12, dtype=f32)
    for r in range(3):
        a, b, c, d = m1[4 * r], m1[4 * r + 1], m1[4 * r + 2], m1[4 * r + 3]
        for col in range(4):
            v = f32(f32(f32(a * m2[col]) + f32(b * m2[4 + col])) + f32(c * m2[8 + col]))
            if col == 3:
                v = f32(v + d)
            m[4 * r + col] = v
    return m


def dof9tomatrix(p_dof, dof_num):
    """src/api_subfunc.cu:715-824: M = [diag(a,b,c)|t] * Rz(alpha) * Rx(beta) * Ry(theta);
    angles are degrees / 57.3 (double division, stored as float, :744-746)."""
    x, y, z = f32(p_dof[1]), f32(p_dof[2]), f32(p_dof[3])
    alpha = beta = theta = f32(0)
    a = b = c = f32(1)
    if dof_num >= 6:
        alpha = f32(float(p_dof[4]) / 57.3)
        beta = f32(float(p_dof[5]) / 57.3)
        theta = f32(float(p_dof[6]) / 57.3)
    if dof_num == 7:
        a = b = c = f32(p_dof[7])
    if dof_num == 9:
        a, b, c = f32(p_dof[7]), f32(p_dof[8]), f32(p_dof[9])
    t2 = np.zeros(12, dtype=f32)
    t2[3], t2[7], t2[11] = x, y, z
    t2[0], t2[5], t2[10] = a, b, c
    ca, sa = _cosf(alpha), _sinf(alpha)
    t3 = np.array([ca, sa, 0, 0, -sa, ca, 0, 0, 0, 0, 1, 0], dtype=f32)  # Rz  :795-797
    t1 = matrixmultiply(t2, t3)
    cb, sb = _cosf(beta), _sinf(beta)
    t3 = np.array([1, 0, 0, 0, 0, cb, sb, 0, 0, -sb, cb, 0], dtype=f32)  # Rx  :805-807
    t2 = matrixmultiply(t1, t3)
    ct, st = _cosf(theta), _sinf(theta)
    t3 = np.array([ct, 0, -st, 0, 0, 1, 0, 0, st, 0, ct, 0], dtype=f32)  # Ry  :815-817
    return matrixmultiply(t2, t3)


def rot2matrix(theta, sx, sy, sz, rot_axis):
    """src/api_subfunc.cu:626-713 (used by mip3dgpu).  Integer halves ``s / 2``."""
    theta = f32(theta)
    t1 = np.zeros(12, dtype=f32)
    t2 = np.zeros(12, dtype=f32)
    t3 = np.zeros(12, dtype=f32)
    t1[0] = t1[5] = t1[10] = 1
    t3[0] = t3[5] = t3[10] = 1
    c, s = _cosf(theta), _sinf(theta)
    if rot_axis == 1:
        t1[7], t1[11] = sy // 2, sz // 2
        t2[:] = [1, 0, 0, 0, 0, c, s, 0, 0, -s, c, 0]
        s_new = int(np.round(np.sqrt(float(sy * sy + sz * sz))))
        t3[7] = t3[11] = int(-s_new / 2)
    elif rot_axis == 2:
        t1[3], t1[11] = sx // 2, sz // 2
        t2[:] = [c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0]
        s_new = int(np.round(np.sqrt(float(sx * sx + sz * sz))))
        t3[3] = t3[11] = int(-s_new / 2)
    elif rot_axis == 3:
        t1[3], t1[7] = sx // 2, sy // 2
        t2[:] = [c, s, 0, 0, -s, c, 0, 0, 0, 0, 1, 0]
        s_new = int(np.round(np.sqrt(float(sx * sx + sy * sy))))
        t3[3] = t3[7] = int(-s_new / 2)
    return matrixmultiply(matrixmultiply(t1, t2), t3)


def checkmatrix(m, sx, sy, sz):
    """src/api_reg.cpp:247-262."""
    m = np.asarray(m, dtype=f32)
    ok = True
    lo, up, slo, sup, ratio = f32(0.5), f32(1.4), f32(2), f32(4), f32(0.8)
    if m[0] < lo or m[0] > up or m[5] < lo or m[5] > up or m[10] < lo or m[10] > up:
        ok = False
    ssum = f32(f32(m[0] + m[5]) + m[10])
    if ssum < slo or ssum > sup:
        ok = False
    if abs(m[3]) > f32(ratio * f32(sx)) or abs(m[7]) > f32(ratio * f32(sy)) or abs(m[11]) > f32(ratio * f32(sz)):
        ok = False
    return ok


# ---------------------------------------------------------------- sampling
def _coords(m, shape, z_range=None):
    """Texture coordinates of every target voxel, float32, evaluated exactly as
    ``d_aff[0]*ix + d_aff[1]*iy + d_aff[2]*iz + d_aff[3] + 0.5`` left to right without
    FMA contraction (include/cukernel.cuh:510-512, 539-541).  ``z_range = (z0, z1)`` restricts the
    result to that slab of target planes (full-size fixtures are computed slab by slab)."""
    sz, sy, sx = shape
    z0, z1 = (0, sz) if z_range is None else z_range
    shape = (z1 - z0, sy, sx)
    m = np.asarray(m, dtype=f32)
    ix = np.arange(sx, dtype=f32)[None, None, :]
    iy = np.arange(sy, dtype=f32)[None, :, None]
    iz = np.arange(z0, z1, dtype=f32)[:, None, None]
    half = f32(0.5)
    out = []
    for r in range(3):
        t = (m[4 * r] * ix + m[4 * r + 1] * iy)
        t = t + m[4 * r + 2] * iz
        t = t + m[4 * r + 3]
        t = t + half
        out.append(np.broadcast_to(t, shape).astype(f32))
    return out  # tx, ty, tz


WEIGHT_BITS = 0  # 8: emulate the CUDA texture unit's 8 fractional weight bits (round to nearest assumed); test aid


def _trilinear(src, tx, ty, tz):
    """Linear texture fetch with unnormalised coordinates, texel centres at +0.5,
    neighbour indices clamped to the edge (BindTexture, src/api_subfunc.cu:885-895;
    wrap degrades to clamp for unnormalised coordinates).  Exact float32 weights:
    v = (1-w)*q0 + w*q1 along x, then y, then z (itrilerp2, :1075-1091)."""
    s2z, s2y, s2x = src.shape
    half = f32(0.5)
    one = f32(1)

    def split(t, n):
        b = t - half
        i0f = np.floor(b)
        w1 = (b - i0f).astype(f32)
        if WEIGHT_BITS == 8:  # CUDA guide: 9-bit fixed point with 8 bits of fractional value, 1.0 exactly represented
            w1 = (np.floor(w1 * f32(256) + f32(0.5)) / f32(256)).astype(f32)
        w0 = (one - w1).astype(f32)
        i0 = i0f.astype(np.int64)
        i1 = i0 + 1
        return np.clip(i0, 0, n - 1), np.clip(i1, 0, n - 1), w0, w1

    x0, x1, wx0, wx1 = split(tx, s2x)
    y0, y1, wy0, wy1 = split(ty, s2y)
    z0, z1, wz0, wz1 = split(tz, s2z)

    def g(zz, yy, xx):
        return src[zz, yy, xx]

    c00 = wx0 * g(z0, y0, x0) + wx1 * g(z0, y0, x1)
    c01 = wx0 * g(z0, y1, x0) + wx1 * g(z0, y1, x1)
    c10 = wx0 * g(z1, y0, x0) + wx1 * g(z1, y0, x1)
    c11 = wx0 * g(z1, y1, x0) + wx1 * g(z1, y1, x1)
    c0 = wy0 * c00 + wy1 * c01
    c1 = wy0 * c10 + wy1 * c11
    return (wz0 * c0 + wz1 * c1).astype(f32)


def affine_transform(src, m, out_shape, z_range=None):
    """affinetransformkernel, include/cukernel.cuh:501-524: non-strict lower bound
    ``0 <= t < s2`` on the +0.5 coordinates, 0 outside."""
    src = np.asarray(src, dtype=f32)
    tx, ty, tz = _coords(m, out_shape, z_range)
    s2z, s2y, s2x = src.shape
    valid = ((tx >= 0) & (tx < f32(s2x)) & (ty >= 0) & (ty < f32(s2y)) & (tz >= 0) & (tz < f32(s2z)))
    v = _trilinear(src, tx, ty, tz)
    return np.where(valid, v, f32(0)).astype(f32)


def corr_sums(target, src, m, z_range=None):
    """corrkernel, include/cukernel.cuh:526-556: strict ``0 < t < s2``; returns
    (sum s*s, sum s*t) accumulated in float64.  With ``z_range`` only that slab of target planes."""
    target = np.asarray(target, dtype=f32)
    src = np.asarray(src, dtype=f32)
    tx, ty, tz = _coords(m, target.shape, z_range)
    if z_range is not None:
        target = target[z_range[0]:z_range[1]]
    s2z, s2y, s2x = src.shape
    valid = ((tx > 0) & (tx < f32(s2x)) & (ty > 0) & (ty < f32(s2y)) & (tz > 0) & (tz < f32(s2z)))
    s = np.where(valid, _trilinear(src, tx, ty, tz), f32(0)).astype(np.float64)
    ss = float(np.sum(s * s))
    st = float(np.sum(s * target.astype(np.float64)))
    return ss, st


def corrfunc(target, sd_t, src, m):
    """corrfunc, src/api_subfunc.cu:954-988."""
    ss, st = corr_sums(target, src, m)
    if np.sqrt(ss) == 0:
        return f32(-2.0)
    return f32(f32(st / np.sqrt(ss)) / f32(sd_t))


def mean_subtract(a):
    """addvaluegpu(-float(sum)/float(N)), src/api_subfunc.cu:2840,2844,2860."""
    a = np.asarray(a, dtype=f32)
    s = float(np.sum(a, dtype=np.float64))
    mean = f32(f32(s) / f32(a.size))
    return (a + (-mean)).astype(f32), s


# ---------------------------------------------------------------- driver
class Reg3dAffine:
    """reg3d_affine1, src/api_subfunc.cu:2733-2994, with Powell supplied by
    ``powell_fn(p, xi, n, ftol, func, counter, it_limit) -> (iter, fret)`` operating on
    1-based float32 arrays (oracle/powell_oracle.py or the `_ref` build)."""

    def __init__(self, powell_fn):
        self.powell = powell_fn

    def run(self, target, source, aff_method, flag_tmx=False, i_tmx=None, ftol=1e-4, it_limit=3000):
        target = np.asarray(target, dtype=f32)
        source = np.asarray(source, dtype=f32)
        shape = target.shape
        records = np.zeros(11, dtype=f32)
        i_tmx = None if i_tmx is None else np.asarray(i_tmx, dtype=f32).copy()
        if aff_method == 0:  # :2767-2781
            if flag_tmx:
                return affine_transform(source, i_tmx, shape), i_tmx, records
            ident = np.zeros(12, dtype=f32)
            ident[0] = ident[5] = ident[10] = 1
            return source.copy(), ident, records

        aff_init = np.zeros(12, dtype=f32)
        aff_init[0] = aff_init[5] = aff_init[10] = 1
        xi = np.zeros((NDIM + 1, NDIM + 1), dtype=f32)
        for i in range(1, NDIM + 1):
            xi[i, i] = 1
        xi9 = np.zeros((10, 10), dtype=f32)
        p9 = np.zeros(10, dtype=f32)
        src_work = source
        if flag_tmx:
            if aff_method == 5:
                aff_init = i_tmx.copy()  # :2820
            else:
                src_work = affine_transform(source, i_tmx, shape)  # :2824
        if aff_method != 5:
            for i in range(1, 10):
                xi9[i, i] = 1
            p9[7] = p9[8] = p9[9] = 1  # :2832-2835
        src_ms, _ = mean_subtract(src_work)  # :2838-2845
        if np.sqrt(float(np.sum(src_ms.astype(np.float64) ** 2))) == 0:
            raise ValueError("SD of image 2 is zero")  # :2853-2856
        tgt_ms, _ = mean_subtract(target)  # :2859-2860
        sd_t = f32(np.sqrt(float(np.sum((tgt_ms * tgt_ms).astype(np.float64)))))  # :2861-2863
        if sd_t == 0:
            raise ValueError("SD of image 1 is zero")

        state = {"aff": None, "dof9": False, "dof": 12}
        counter = [0]

        def costfunc(x):  # :2377-2388
            if state["dof9"]:
                state["aff"] = dof9tomatrix(x, state["dof"])
            else:
                state["aff"] = p2matrix(x)
            c = corrfunc(tgt_ms, sd_t, src_ms, state["aff"])
            counter[0] += 1
            return f32(-c)

        p = matrix2p(aff_init)  # :2879
        records[1] = -costfunc(p)  # :2881
        counter[0] = 0  # :2888
        fret = f32(0)

        def run9(dof, tol):
            state["dof9"], state["dof"] = True, dof
            return self.powell(p9, xi9, dof, f32(tol), costfunc, counter, it_limit)

        def run12(tol):
            state["dof9"], state["dof"] = False, 12
            return self.powell(p, xi, 12, f32(tol), costfunc, counter, it_limit)

        if aff_method in (1, 2, 3, 4):  # :2890-2909
            _, fret = run9({1: 3, 2: 6, 3: 7, 4: 9}[aff_method], ftol)
        elif aff_method == 5:  # :2910-2914
            _, fret = run12(ftol)
        elif aff_method == 6:  # :2915-2929
            _, fret = run9(6, 0.01)
            records[2] = -fret
            p = matrix2p(state["aff"])  # the LAST evaluated matrix (App. C quirk)
            _, fret = run12(ftol)
        elif aff_method == 7:  # :2930-2954
            _, fret = run9(3, 0.01)
            _, fret = run9(6, 0.01)
            _, fret = run9(9, 0.005)
            records[2] = -fret
            p = matrix2p(state["aff"])
            _, fret = run12(ftol)
        else:
            raise ValueError("wrong affMethod")
        aff = state["aff"]
        if flag_tmx and aff_method != 5:
            aff = matrixmultiply(i_tmx, aff)  # :2958-2961
        records[3] = -fret
        records[5] = counter[0]
        reg = affine_transform(source, aff, shape)  # raw source  :2974-2977
        return reg, aff, records


# ---------------------------------------------------------------- 2-D MIP shift search (regChoice 4)
def max_projection(vol, direction):
    """maxprojectionkernel, include/cukernel.cuh:396-418 (+ wrapper src/api_subfunc.cu:490-510): the
    accumulator starts at 0, so negative data projects to 0.  direction 1: along z -> [sy][sx];
    direction 2: along y -> image of width sz, height sx ([sx][sz]); direction 3: along x -> [sz][sy]."""
    vol = np.asarray(vol, dtype=f32)
    if direction == 1:
        p = vol.max(axis=0)                 # [y][x]
    elif direction == 2:
        p = vol.max(axis=1).T               # [x][z]
    elif direction == 3:
        p = vol.max(axis=2)                 # [z][y]
    else:
        raise ValueError(direction)
    return np.maximum(p, f32(0)).astype(f32)


def _matrix_2d(a6):
    """2-D matrix (a0 a1 a2; a3 a4 a5) as the 3-D matrix of a one-plane volume: identical float32
    arithmetic, since the z terms add exact zeros (include/cukernel.cuh:585-586 vs :539-541)."""
    a = np.asarray(a6, dtype=f32)
    return np.array([a[0], a[1], 0, a[2], a[3], a[4], 0, a[5], 0, 0, 1, 0], dtype=f32)


def corrfunc2d(target_ms, sd_t, source_ms, a6):
    """corrfunc2D + corr2Dkernel, src/api_subfunc.cu:1014-1036, include/cukernel.cuh:575-597: the products
    t*t and s*t are formed in float32 and summed in float64 on the host."""
    tgt = np.asarray(target_ms, dtype=f32)[None]
    src = np.asarray(source_ms, dtype=f32)[None]
    tx, ty, tz = _coords(_matrix_2d(a6), tgt.shape)
    valid = (tx > 0) & (tx < f32(src.shape[2])) & (ty > 0) & (ty < f32(src.shape[1]))
    t = np.where(valid, _trilinear(src, tx, ty, tz), f32(0)).astype(f32)
    sqr = float(np.sum((t * t).astype(f32), dtype=np.float64))
    cor = float(np.sum((tgt * t).astype(f32), dtype=np.float64))
    if np.sqrt(sqr) == 0:
        return f32(-2.0)
    return f32(f32(cor / np.sqrt(sqr)) / f32(sd_t))


def shift_align_2d(img1, img2, tmx6=None, x_only=False, shift_region=0.3, total_step=30):
    """reg2d_shiftalign1 / reg2d_shiftalignX1, src/api_subfunc.cu:1860-1990, 2000-2120: exhaustive search of
    the translation on a (2*total_step)^2 (or 2*total_step) grid, best ZNCC wins (strictly greater than 0).
    Images are [h][w]; returns the 6-vector (a0 a1 tx; a3 a4 ty)."""
    img1 = np.asarray(img1, dtype=f32)
    img2 = np.asarray(img2, dtype=f32)
    h1, w1 = img1.shape
    h2, w2 = img2.shape
    if tmx6 is not None:
        aff = np.asarray(tmx6, dtype=f32).copy()
    else:
        aff = np.array([1, 0, (w2 - w1) // 2, 0, 1, (h2 - h1) // 2], dtype=f32)
    t_ms, _ = mean_subtract(img1)
    sq = float(np.sum((t_ms * t_ms).astype(f32), dtype=np.float64))
    sd_t = f32(np.sqrt(sq))
    if sd_t == 0:
        raise ValueError("SD of image 1 is zero")
    s_ms, _ = mean_subtract(img2)
    p = aff.copy()
    off_x, off_y = p[2], p[5]
    best, sx_best, sy_best = f32(0), f32(0), f32(0)
    step_x = f32(f32(f32(w2) * f32(shift_region)) / f32(total_step))
    step_y = f32(f32(f32(h2) * f32(shift_region)) / f32(total_step))
    n = int(total_step)
    for i in range(-n, n):
        p[2] = f32(off_x + f32(step_x * f32(i)))
        js = [None] if x_only else range(-n, n)
        for j in js:
            if j is not None:
                p[5] = f32(off_y + f32(step_y * f32(j)))
            c = corrfunc2d(t_ms, sd_t, s_ms, p)
            if c > best:
                best, sx_best = c, p[2]
                if j is not None:
                    sy_best = p[5]
    p[2] = sx_best
    if not x_only:
        p[5] = sy_best
    return p.astype(f32), f32(corrfunc2d(t_ms, sd_t, s_ms, p))


def mip_shift_search(target, source):
    """regChoice 4 pre-registration, src/api_reg.cpp:462-492: XY shift from the z-projections, then the
    z shift from the y-projections with the x shift held.  Returns (shift_x, shift_y, shift_z)."""
    t1, _ = shift_align_2d(max_projection(target, 1), max_projection(source, 1))
    t2, _ = shift_align_2d(max_projection(target, 2), max_projection(source, 2),
                           tmx6=np.array([1, 0, 0, 0, 1, t1[2]], dtype=f32), x_only=True)
    return np.array([t1[2], t1[5], t2[2]], dtype=f32)


# ---------------------------------------------------------------- 3-D phasor registration (regChoice 1 / 3)
def zncc1(a, b):
    """zncc1, src/api_subfunc.cu:2409-2432: means removed with float(sum)/float(N), products formed in float32
    (multi3Dgpu), sums in float64; -2 when a variance vanishes."""
    a, _ = mean_subtract(a)
    b, _ = mean_subtract(b)
    st = float(np.sum((a * b).astype(f32), dtype=np.float64))
    tt = float(np.sum((a * a).astype(f32), dtype=np.float64))
    ss = float(np.sum((b * b).astype(f32), dtype=np.float64))
    den = f32(np.sqrt(tt * ss))
    return f32(st / den) if den != 0 else f32(-2.0)


def circshift(a, dx, dy, dz):
    """circshiftgpukernel, include/cukernel.cuh:457-473: out[z][y][x] = in[z-dz][y-dy][x-dx] (circular)."""
    return np.roll(a, shift=(dz, dy, dx), axis=(0, 1, 2))


def imshift(a, dx, dy, dz):
    """imshiftgpukernel, include/cukernel.cuh:476-489: as circshift but zero-filled."""
    out = np.zeros_like(a)
    sz, sy, sx = a.shape

    def rng(d, n):  # destination slice, source slice
        lo, hi = max(d, 0), min(n + d, n)
        return slice(lo, hi), slice(lo - d, hi - d)
    (zd, zs), (yd, ys), (xd, xs) = rng(dz, sz), rng(dy, sy), rng(dx, sx)
    if zd.start < zd.stop and yd.start < yd.stop and xd.start < xd.stop:
        out[zd, yd, xd] = a[zs, ys, xs]
    return out


def reg3d_phasor(target, source):
    """reg3d_phasor1, src/api_subfunc.cu:2465-2590: peak of the inverse transform of the normalised cross-power
    spectrum conj(F(target)) * F(source) / |.|, centred by a half-size circular shift; when a shift exceeds a
    quarter of the extent, each axis' wrap-around alternative is judged by the ZNCC of the overlapping crops.
    Returns integer (shift_x, shift_y, shift_z)."""
    t = np.asarray(target, dtype=f32)
    s = np.asarray(source, dtype=f32)
    sz, sy, sx = t.shape
    a = np.conj(np.fft.rfftn(t.astype(np.float64)))
    b = np.fft.rfftn(s.astype(np.float64))
    c = a * b
    mag = np.abs(c)
    c = np.where(mag != 0, c / np.where(mag != 0, mag, 1), 0)
    ph = np.fft.irfftn(c, s=t.shape, axes=(0, 1, 2))
    ph = circshift(ph, sx // 2, sy // 2, sz // 2)
    # max3Dgpu :437-471: first maximum along z per column, then columns scanned x-outer / y-inner, strict <
    col_arg = np.argmax(ph, axis=0)                      # first occurrence along z
    col_max = np.take_along_axis(ph, col_arg[None], 0)[0]  # [y][x]
    order = col_max.T.reshape(-1)                         # x outer, y inner
    k = int(np.argmax(order))
    cx, cy = k // sy, k % sy
    cz = int(col_arg[cy, cx])
    shift = [cx - sx // 2, cy - sy // 2, cz - sz // 2]
    ab = [abs(v) for v in shift]
    beta = 4
    if ab[0] > sx // beta or ab[1] > sy // beta or ab[2] > sz // beta:
        moved = circshift(s, -shift[0], -shift[1], -shift[2])
        dims = (sx, sy, sz)
        size = [[dims[d] - ab[d], ab[d]] for d in range(3)]
        org = [([0, dims[d] - ab[d]] if shift[d] > 0 else [ab[d], 0]) for d in range(3)]
        best, ind = f32(-3), [0, 0, 0]
        for i in range(2):
            if size[0][i] > sx // beta:
                for j in range(2):
                    if size[1][j] > sy // beta:
                        for kk in range(2):
                            if size[2][kk] > sz // beta:
                                sl = (slice(org[2][kk], org[2][kk] + size[2][kk]),
                                      slice(org[1][j], org[1][j] + size[1][j]),
                                      slice(org[0][i], org[0][i] + size[0][i]))
                                cc = zncc1(t[sl], moved[sl])
                                if best < cc:
                                    best, ind = cc, [i, j, kk]
        for d in range(3):
            if ind[d] == 1:
                shift[d] = shift[d] - dims[d] if shift[d] > 0 else shift[d] + dims[d]
    return np.array(shift, dtype=np.int64)


# ---------------------------------------------------------------- 2-D registration (reg2d)
def affine_transform_2d(src, a6, out_shape):
    """affineTransform2Dkernel, include/cukernel.cuh:558-573: strict window 0 < t < s2 (unlike the 3-D kernel)."""
    src = np.asarray(src, dtype=f32)[None]
    tx, ty, tz = _coords(_matrix_2d(a6), (1,) + tuple(out_shape))
    valid = (tx > 0) & (tx < f32(src.shape[2])) & (ty > 0) & (ty < f32(src.shape[1]))
    return np.where(valid, _trilinear(src, tx, ty, tz), f32(0)).astype(f32)[0]


def reg2d(img1, img2, reg_choice, powell_fn=None, flag_tmx=False, i_tmx=None, ftol=1e-4, it_limit=3000):
    """reg2d, src/api_reg.cpp:115-244 (GPU mode): 0 apply matrix | 1 translation grid (0.4, 40 steps) |
    2 6-parameter Powell (reg2d_affine1, src/api_subfunc.cu:2233-2336) | 3 phasor shift.
    Returns (registered, a6, records[11])."""
    img1 = np.asarray(img1, dtype=f32)
    img2 = np.asarray(img2, dtype=f32)
    h1, w1 = img1.shape
    h2, w2 = img2.shape
    rec = np.zeros(11, dtype=f32)
    if flag_tmx:
        a = np.asarray(i_tmx, dtype=f32).copy()
    else:
        a = np.array([1, 0, (w2 - w1) // 2, 0, 1, (h2 - h1) // 2], dtype=f32)
    if reg_choice == 0:
        return (affine_transform_2d(img2, a, img1.shape) if flag_tmx else None), a, rec
    if reg_choice == 1:
        a, _ = shift_align_2d(img1, img2, tmx6=a, shift_region=0.4, total_step=40)
        return affine_transform_2d(img2, a, img1.shape), a, rec
    if reg_choice == 3:
        sh = reg3d_phasor(img1[None], img2[None])
        a = np.array([1, 0, sh[0], 0, 1, sh[1]], dtype=f32)
        return imshift(img2[None], -int(sh[0]), -int(sh[1]), 0)[0], a, rec
    if reg_choice != 2:
        raise ValueError(reg_choice)
    t_ms, _ = mean_subtract(img1)
    sd_t = f32(np.sqrt(float(np.sum((t_ms * t_ms).astype(f32), dtype=np.float64))))
    s_ms, _ = mean_subtract(img2)
    last = a.copy()
    counter = [0]

    def cost(x):  # costfunc2D, 1-based vector
        last[:] = x[1:7]
        counter[0] += 1
        return f32(-corrfunc2d(t_ms, sd_t, s_ms, last))

    p = np.zeros(7, dtype=f32)
    p[1:] = a
    xi = np.zeros((7, 7), dtype=f32)
    for i in range(1, 7):
        xi[i, i] = 1
    rec[1] = -cost(p)
    counter[0] = 0
    _, fret = powell_fn(p, xi, 6, f32(ftol), cost, counter, it_limit)
    rec[3] = -fret
    rec[5] = counter[0]
    return affine_transform_2d(img2, last, img1.shape), last.copy(), rec
