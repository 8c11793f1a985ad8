"""Host-side registration helpers of the product (no GPU) vs the oracle and golden tables."""
import os

import numpy as np
import pytest

from oracle import reg_oracle as ro

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def fp(a):
    from microimagelib_amd import _lib
    return a.ctypes.data_as(_lib.c_float_p)


@pytest.fixture(scope="module")
def tables():
    return np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))


def test_p2matrix_roundtrip(lib):
    rng = np.random.default_rng(0)
    p = rng.normal(0, 1, 12).astype(f32)
    m = np.zeros(12, dtype=f32)
    lib.mil_p2matrix(fp(m), fp(p))
    want = ro.p2matrix(np.concatenate([[0], p]).astype(f32))
    assert np.array_equal(m, want)
    back = np.zeros(12, dtype=f32)
    lib.mil_matrix2p(fp(back), fp(m))
    assert np.array_equal(back, p)
    assert np.array_equal(ro.matrix2p(m)[1:], p)


def test_matrixmultiply_table(lib, tables):
    for a, b, want in zip(tables["mm_a"], tables["mm_b"], tables["mm_out"]):
        out = np.zeros(12, dtype=f32)
        lib.mil_matrixmultiply(fp(out), fp(np.ascontiguousarray(a)), fp(np.ascontiguousarray(b)))
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_matrixmultiply_allows_aliasing(lib):
    a = np.arange(12, dtype=f32) * f32(0.1) + f32(1)
    want = ro.matrixmultiply(a, a)
    out = a.copy()
    lib.mil_matrixmultiply(fp(out), fp(out), fp(out))
    assert np.array_equal(out, want)


def test_dof9tomatrix_table(lib, tables):
    for p, dof, want in zip(tables["p9"], tables["dof"], tables["m9"]):
        out = np.zeros(12, dtype=f32)
        lib.mil_dof9tomatrix(fp(out), fp(np.ascontiguousarray(p[1:])), int(dof))
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))  # cosf / sinf from libm on both sides


def test_dof9_unused_parameters_take_defaults(lib):
    p = np.array([1, 2, 3, 40, 50, 60, 7, 8, 9], dtype=f32)
    out = np.zeros(12, dtype=f32)
    lib.mil_dof9tomatrix(fp(out), fp(p), 3)
    assert np.array_equal(out, np.array([1, 0, 0, 1, 0, 1, 0, 2, 0, 0, 1, 3], dtype=f32))
    lib.mil_dof9tomatrix(fp(out), fp(p), 7)  # one common scale
    q = p.copy()
    q[7] = q[8] = 7
    ref = np.zeros(12, dtype=f32)
    lib.mil_dof9tomatrix(fp(ref), fp(q), 9)
    assert np.array_equal(out, ref)


def test_rot2matrix_table(lib, tables):
    for axis, want in zip((1, 2, 3), tables["rot"]):
        out = np.zeros(12, dtype=f32)
        lib.mil_rot2matrix(fp(out), float(f32(0.7)), 50, 36, 20, axis)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_checkmatrix_table(lib, tables):
    for m, want in zip(tables["check_in"], tables["check_out"]):
        m = np.ascontiguousarray(m, dtype=f32)
        assert bool(lib.checkmatrix(fp(m), 50, 36, 20)) == bool(want)


def test_fft_digit_reversal_tables():
    """Host tables of the fused transform backend: freq_of_pos for the in-place mixed-radix
    (8,..,8,{4,2},{3,5,7,11,13}) decimation-in-frequency plan, checked against a NumPy model of that plan."""
    import ctypes as C
    from microimagelib_amd import _lib
    lib = _lib.load()

    def model(n):
        if n == 1:
            return np.zeros(1, dtype=np.int64)
        r = next(r for r in (8, 4, 2, 3, 5, 7, 11, 13) if n % r == 0)
        m = n // r
        sub = model(m)
        p = np.arange(n)
        return r * sub[p % m] + p // m

    for n in (8, 16, 32, 64, 128, 256, 512, 1024, 96, 160, 192, 320, 352, 416, 448, 576, 704, 832, 896, 960):
        buf = (C.c_ushort * n)()
        assert lib.mil_fft_digit_reversal(n, buf) == 0
        got = np.frombuffer(buf, dtype=np.uint16).astype(np.int64)
        assert np.array_equal(got, model(n))
        assert np.array_equal(np.sort(got), np.arange(n))
    assert lib.mil_fft_digit_reversal(0, (C.c_ushort * 4)()) == -1


def test_cost_kernel_choice_per_matrix(lib):
    """pick_kernel (csrc/corr_padded.hip): which padded-source cost kernel a matrix gets -- host logic, no GPU.
    0 plain layout + plane reuse, 2 row pairs + plane reuse, 1 row pairs without reuse."""
    def rot_y(deg, scale=1.0):
        c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
        m = np.array([c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0], dtype=np.float64) * scale
        m[[3, 7, 11]] = (4.0, -2.0, 1.0)
        return m.astype(f32)

    def pick(m):
        return lib.mil_corr_padded_pick_kernel(fp(np.ascontiguousarray(m, dtype=f32)))

    ident = np.eye(3, 4, dtype=f32).reshape(12)
    assert pick(ident) == 0                      # every lane reuses its plane at every step
    assert pick(rot_y(0.0, 1.05)) == 0           # pure scale: all lanes of a wave cross planes together
    assert pick(rot_y(1.0)) == 0 and pick(rot_y(2.0)) == 0
    assert pick(rot_y(3.0, 1.02)) == 2           # a few degrees AND a few per cent: lanes drift out of step
    assert pick(rot_y(5.0)) == 2 and pick(rot_y(10.0)) == 2
    assert pick(rot_y(20.0)) == 1 and pick(rot_y(45.0)) == 1   # re-gather rate beyond 2: both planes gathered every step
    assert pick(np.full(12, np.nan, dtype=f32)) == 0   # NaN never satisfies a threshold; any kernel rejects every voxel

    # the kernel that really runs on a target whose width is a multiple of 128 (two voxels per lane; csrc/corr_padded.hip
    # pick_kernel): kernel 6 marches along y where THAT axis re-gathers rarely -- rate below 0.08 and below half of z's
    def pick2(m, wide=1):
        return lib.mil_corr_padded_pick_kernel2(fp(np.ascontiguousarray(m, dtype=f32)), wide)

    def rot_x(deg):
        c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
        return np.array([1, 0, 0, 2.0, 0, c, -s, -1.0, 0, s, c, 0.5], dtype=f32)

    assert pick2(ident) == 3 and pick2(rot_y(0.0, 1.05)) == 3   # plain layout, two voxels per lane, along z
    assert pick2(rot_y(1.0)) == 6 and pick2(rot_y(2.0)) == 6    # rotations about y leave the y axis quiet ...
    assert pick2(rot_y(3.0, 1.02)) == 6                         # bench.py's matrix: 83 % re-gathers along z, 2 % along y
    assert pick2(rot_y(5.0)) == 6                               # ... while a wave's lanes stay within ~6 source planes (5.6 here)
    assert pick2(rot_y(10.0)) == 5                              # 11 planes per wave: row pairs + reuse, two voxels per lane
    assert pick2(rot_x(5.0)) == 5 and pick2(rot_x(30.0)) == 5   # about x both axes drift; the reuse pays up to a rate of 2
    assert pick2(rot_y(20.0)) == 1 and pick2(rot_y(45.0)) == 1  # large: row pairs, both planes, one voxel per lane
    for m in (ident, rot_y(3.0, 1.02), rot_x(5.0), rot_y(20.0)):
        assert pick2(m, wide=0) == pick(m)                      # narrow targets: the one-voxel kernel of that layout


def test_resample_kernel_gate():
    """Which kernel affine_resample takes (csrc/affine.hip small_source, csrc/resample_march.hip) -- host logic, no GPU.
    32-bit offsets with UNSIGNED 24-bit multiplies hold for planes below 2^24 voxels and sources below 2 GiB; the marching
    kernel additionally needs the plane pitch in BYTES below 2^24."""
    from microimagelib_amd import _lib
    lib = _lib.load()
    path = lambda o, s: lib.mil_affine_resample_path(o[0], o[1], o[2], s[0], s[1], s[2])
    assert path((512, 512, 256), (512, 512, 256)) == 2
    assert path((572, 512, 572), (512, 512, 256)) == 2           # a rotated box larger than the source
    assert path((64, 64, 2), (2048, 2047, 2)) == 2               # plane pitch just below 2^24 bytes
    assert path((64, 64, 2), (2048, 2048, 2)) == 1               # at 2^24 bytes: one voxel per thread, 32-bit offsets
    assert path((64, 64, 2), (4096, 2048, 2)) == 1               # 2^23 voxels per plane (the signed multiply's trap)
    assert path((64, 64, 2), (4096, 4095, 2)) == 1               # just below 2^24 voxels per plane
    assert path((64, 64, 2), (4096, 4096, 2)) == 0               # 2^24 voxels per plane: 64-bit arithmetic
    assert path((64, 64, 64), (1024, 1024, 512)) == 0            # 2 GiB
    assert path((64, 64, 64), (1, 64, 64)) == 0                  # no x pair to gather
    assert path((1024, 1024, 1024), (64, 64, 64)) == 1           # output offsets beyond 32 bits: not the marching kernel
