"""GPU parity of the hand-written kernels, through the mil_device.h C ABI.
Integer / index / single-rounding float work is compared BIT-EXACT with the oracle."""
import os

import numpy as np
import pytest

from oracle import decon_oracle as do
from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32


@pytest.fixture(scope="module")
def dev():
    import torch
    from microimagelib_amd import device
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return device


def T(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    return np.array_equal(a.view(np.uint8), b.view(np.uint8))


SIZES = [0, 1, 3, 4, 5, 63, 64, 1023, 1024, 4099, 1 << 20, (1 << 20) + 7]


@pytest.mark.parametrize("n", SIZES)
def test_rl_ratio_bit_exact(dev, n):
    rng = np.random.default_rng(n)
    A = rng.uniform(0.01, 3000, n).astype(f32)
    Tv = rng.uniform(-5, 4000, n).astype(f32)
    if n > 4:
        Tv[3] = 0.0  # the reference divide is unguarded: inf must propagate identically
    with np.errstate(divide="ignore", invalid="ignore"):
        want = do.rl_ratio(A, Tv)
    got = dev.rl_ratio_(T(Tv), T(A)).cpu().numpy()
    assert bits_equal(got, want)


@pytest.mark.parametrize("n", SIZES)
def test_rl_update_bit_exact(dev, n):
    rng = np.random.default_rng(n + 1)
    E = rng.uniform(0.01, 3000, n).astype(f32)
    Tv = rng.uniform(-0.5, 2.0, n).astype(f32)
    want = do.rl_update(E, Tv)
    got = dev.rl_update_(T(E), T(Tv)).cpu().numpy()
    assert bits_equal(got, want)
    assert n == 0 or got.min() >= f32(0.01)


def test_rl_update_nan_goes_to_eps(dev):
    """(v > eps) ? v : eps -- a NaN product collapses to eps like the reference's
    maxvalue3Dgpukernel (include/cukernel.cuh:389), unlike np.maximum."""
    E = np.array([1.0, np.nan, 2.0, np.inf, 5.0], dtype=f32)
    Tv = np.array([np.nan, 1.0, 0.0, 0.0, 2.0], dtype=f32)
    got = dev.rl_update_(T(E), T(Tv)).cpu().numpy()
    assert bits_equal(got, np.array([0.01, 0.01, 0.01, 0.01, 10.0], dtype=f32))


@pytest.mark.parametrize("n", SIZES)
def test_cmul_bit_exact(dev, n):
    rng = np.random.default_rng(n + 2)
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * f32(1e3)
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    want = do.cmul(a, b)
    got = dev.cmul_(T(a), T(b)).cpu().numpy()
    assert bits_equal(got, want)


def test_elementwise_misaligned_views(dev):
    """Sub-buffers that are not 16-byte aligned take the scalar path; same bits."""
    rng = np.random.default_rng(5)
    n = 10007
    E = rng.uniform(0.01, 100, n + 3).astype(f32)
    Tv = rng.uniform(0.5, 2.0, n + 3).astype(f32)
    dE, dT = T(E), T(Tv)
    dev.rl_update_(dE[1:n + 1], dT[2:n + 2])
    want = E.copy()
    want[1:n + 1] = do.rl_update(E[1:n + 1], Tv[2:n + 2])
    assert bits_equal(dE.cpu().numpy(), want)
    dA = T(E)
    dT2 = T(Tv)
    dev.rl_ratio_(dT2[3:n + 3], dA[1:n + 1])
    want = Tv.copy()
    want[3:n + 3] = do.rl_ratio(E[1:n + 1], Tv[3:n + 3])
    assert bits_equal(dT2.cpu().numpy(), want)


@pytest.mark.parametrize("n", [1, 5, 4096, 1000003])
def test_sum_f64(dev, n):
    rng = np.random.default_rng(n)
    a = rng.uniform(0, 2000, n).astype(f32)
    want = float(np.sum(a, dtype=np.float64))
    got = dev.sum_f64(T(a))
    assert abs(got - want) <= 1e-12 * abs(want)
    assert dev.sum_f64(T(a)) == got  # deterministic two-stage reduction


def test_sum_f64_empty(dev):
    import torch
    assert dev.sum_f64(torch.empty(0, dtype=torch.float32, device="cuda")) == 0.0


def test_geometry_golden(dev):
    g = np.load(os.path.join(GOLD, "geometry.npz"))
    a = g["a"]
    da = T(a)
    assert bits_equal(dev.pad_stack(da, (16, 16, 16)).cpu().numpy(), g["pad_stack"])
    assert bits_equal(dev.crop_stack(T(g["pad_stack"]), a.shape).cpu().numpy(), g["crop"])
    assert bits_equal(dev.align_size(da, (9, 12, 16)).cpu().numpy(), g["align_grow"])
    assert bits_equal(dev.align_size(da, (4, 7, 8)).cpu().numpy(), g["align_shrink"])
    assert bits_equal(dev.align_size(da, (12, 5, 13)).cpu().numpy(), g["align_mixed"])
    assert bits_equal(dev.pad_psf(da, (16, 16, 16)).cpu().numpy(), g["pad_psf"])
    assert bits_equal(dev.flip3(da).cpu().numpy(), g["flip"])


@pytest.mark.parametrize("shape,box", [((5, 6, 7), (8, 8, 8)), ((8, 8, 8), (8, 8, 8)), ((1, 1, 1), (16, 32, 64)),
                                       ((9, 16, 31), (16, 16, 32)), ((64, 64, 64), (64, 128, 320))])
def test_geometry_vs_oracle(dev, shape, box):
    rng = np.random.default_rng(sum(shape))
    a = rng.standard_normal(shape).astype(f32)
    da = T(a)
    assert bits_equal(dev.pad_stack(da, box).cpu().numpy(), do.pad_stack(a, box))
    assert bits_equal(dev.pad_psf(da, box).cpu().numpy(), do.pad_psf(a, box))
    assert bits_equal(dev.pad_psf(da, box, scale=0.37, flip=True).cpu().numpy(),
                      do.pad_psf(do.flip3(a) * f32(0.37), box))
    big = rng.standard_normal(box).astype(f32)
    assert bits_equal(dev.crop_stack(T(big), shape).cpu().numpy(), do.crop_stack(big, shape))
    assert bits_equal(dev.align_size(da, box).cpu().numpy(), do.align_size(a, box))
    assert bits_equal(dev.align_size(T(big), shape).cpu().numpy(), do.align_size(big, shape))
    assert bits_equal(dev.flip3(da).cpu().numpy(), do.flip3(a))


def test_pad_crop_roundtrip_full_size(dev):
    """Size-independent property at a production-like size: crop(pad(x)) == x."""
    import torch
    x = torch.rand((200, 300, 500), device="cuda")
    box = tuple(do.snap_transform_size(s) for s in x.shape)
    assert box == (256, 320, 512)
    y = dev.crop_stack(dev.pad_stack(x, box), x.shape)
    assert torch.equal(x, y)


def test_rot_y90_and_projection(dev):
    rng = np.random.default_rng(2)
    a = rng.standard_normal((6, 9, 13)).astype(f32)  # [z][y][x]
    da = T(a)
    # reference rotbyyaxiskernel (include/cukernel.cuh:438-453):
    #  +1: out[(sx-1-i), j, k] (x-fastest index k) = in(i,j,k)  -> out[z'][y][x'] with x' = k, z' = sx-1-i
    plus = np.transpose(a, (2, 1, 0))[::-1, :, :]
    minus = np.transpose(a, (2, 1, 0))[:, :, ::-1]
    assert bits_equal(dev.rot_y90(da, 1).cpu().numpy(), np.ascontiguousarray(plus))
    assert bits_equal(dev.rot_y90(da, -1).cpu().numpy(), np.ascontiguousarray(minus))
    # +90 then -90 is the identity
    assert bits_equal(dev.rot_y90(dev.rot_y90(da, 1), -1).cpu().numpy(), a)
    # maxprojectionkernel (:396-418): accumulator starts at 0
    zero = f32(0)
    assert bits_equal(dev.max_projection(da, 1).cpu().numpy(), np.maximum(a.max(axis=0), zero))           # [y][x]
    assert bits_equal(dev.max_projection(da, 2).cpu().numpy(), np.maximum(a.max(axis=1), zero).T.copy())  # [x][z]
    assert bits_equal(dev.max_projection(da, 3).cpu().numpy(), np.maximum(a.max(axis=2), zero))           # [z][y]


def test_affine_golden(dev):
    g = np.load(os.path.join(GOLD, "affine.npz"))
    src, tgt = T(g["src"]), T(g["tgt"])
    for i, m in enumerate(g["mats"]):
        got = dev.affine_resample(src, m, g["tgt"].shape).cpu().numpy()
        assert bits_equal(got, g["resampled"][i]), f"matrix {i}"
        ss, st = dev.corr_sums(tgt, src, m)
        want_ss, want_st = g["corr_sums"][i]
        assert abs(ss - want_ss) <= 1e-12 * max(abs(want_ss), 1e-30), f"matrix {i}"
        assert abs(st - want_st) <= 1e-11 * max(abs(want_ss), 1e-30), f"matrix {i}"
    assert g["corr_sums"][2][0] == 0.0  # the all-out-of-bounds matrix
    got = dev.affine_resample(src, g["resize_matrix"], g["resized"].shape).cpu().numpy()
    assert bits_equal(got, g["resized"])


def _bits(v):
    return np.float64(v).view(np.uint64)


def _same_sums(got, want, kern):
    """Kernels 0-5 form their float64 partial sums in ONE order (the 64 x 4 x 16 decomposition): same bits.  Kernel 6 marches
    along y (16 y steps per lane) and the per-matrix choice (-1) may pick it: same voxels and float32 values, another order of
    the float64 additions -- equal to 1e-13 of the sum of squares."""
    if kern in (0, 1, 2, 3, 4, 5):
        return (np.float64(got[0]).view(np.uint64), np.float64(got[1]).view(np.uint64)) == \
               (np.float64(want[0]).view(np.uint64), np.float64(want[1]).view(np.uint64))
    tol = 1e-13 * max(abs(want[0]), 1e-300)
    return abs(got[0] - want[0]) <= tol and abs(got[1] - want[1]) <= tol



def test_padded_cost_kernel_is_bit_identical(dev):
    """The registration's cost kernels read the source through border-padded copies (csrc/corr_padded.hip: plain and
    row-pair layouts, with and without plane reuse): same two sums, bit for bit, as the general kernel -- golden matrices (texel centres, edges, all-out-of-window), random
    affine matrices, and matrices that put many voxels on the replicated border."""
    g = np.load(os.path.join(GOLD, "affine.npz"))
    src, tgt = T(g["src"]), T(g["tgt"])
    pad = dev.PaddedSource(src)
    for i, m in enumerate(g["mats"]):
        a = dev.corr_sums(tgt, src, m)
        for kern in (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):  # per-matrix choice, then each padded-source kernel (3-6: two voxels per lane)
            dev.corr_padded_force_kernel(kern)
            b = dev.corr_sums(tgt, pad, m)
            assert _same_sums(b, a, kern), f"golden matrix {i}, kernel {kern}: {b} vs {a}"
        dev.corr_padded_force_kernel(-1)
        want_ss, want_st = g["corr_sums"][i]
        assert abs(b[0] - want_ss) <= 1e-12 * max(abs(want_ss), 1e-30)
        assert abs(b[1] - want_st) <= 1e-11 * max(abs(want_ss), 1e-30)
    pad.close()
    rng = np.random.default_rng(7)
    for shape_s, shape_t in (((17, 23, 70), (19, 21, 66)), ((5, 3, 2), (6, 4, 3)), ((1, 1, 1), (3, 3, 3)),
                             ((33, 40, 130), (33, 40, 130)),
                             # target widths that are multiples of 128: the two-voxels-per-lane kernels (3-5), with
                             # ragged y / z blocks and a source of another size
                             ((17, 23, 70), (19, 21, 128)), ((9, 40, 300), (18, 10, 256)),
                             # ragged 128-wide columns (round 6): widths within 15 % below a multiple of 128 run the chunked
                             # two-voxel kernels with the lanes beyond the row kept out of the sums; 130 stays with the one-voxel ones
                             ((9, 40, 300), (18, 10, 250)), ((12, 30, 240), (7, 9, 500)), ((17, 23, 70), (19, 21, 130))):
        src = rng.standard_normal(shape_s).astype(f32)
        tgt = rng.standard_normal(shape_t).astype(f32)
        ds, dt = T(src), T(tgt)
        pad = dev.PaddedSource(ds)
        mats = []
        for k in range(6):
            m = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32) + rng.normal(0, 0.05 * k, 12).astype(f32)
            m[[3, 7, 11]] = rng.uniform(-3, 3, 3).astype(f32)
            mats.append(m)
        ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32)
        for d in (-0.5, -0.25, 0.25, 0.5, -1.0):  # half-voxel shifts: taps on the replicated border / exact centres
            m = ident.copy()
            m[[3, 7, 11]] = d
            mats.append(m)
        sx_, sy_, sz_ = shape_s[2], shape_s[1], shape_s[0]
        mats.append(np.array([-1, 0, 0, sx_ - 1.25, 0, 1, 0, 0.5, 0, 0, 1, -0.75], dtype=f32))   # mirror in x: t falls with x
        mats.append(np.array([1, 0, 0, 0, 0, -1, 0, sy_ - 1, 0, 0, -1, sz_ - 1], dtype=f32))      # t falls with y and z
        mats.append(np.array([0, 1, 0, 0.3, 1, 0, 0, -0.2, 0, 0, 1, 0], dtype=f32))               # x and y swapped
        mats.append(np.array([0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0], dtype=f32))                    # x and z swapped
        mats.append(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 2.5], dtype=f32))                  # every z on one source plane
        mats.append(ident * f32(0.0))                       # everything maps to t = 0.5
        mats.append(np.full(12, 1e30, dtype=f32))           # overflows to inf / nan: all out of the window
        mats.append(np.full(12, np.nan, dtype=f32))
        for i, m in enumerate(mats):
            a = dev.corr_sums(dt, ds, m)
            for kern in (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
                dev.corr_padded_force_kernel(kern)
                b = dev.corr_sums(dt, pad, m)
                assert _same_sums(b, a, kern), f"{shape_s} matrix {i}, kernel {kern}: {b} vs {a}"
            dev.corr_padded_force_kernel(-1)
            if i < 16:
                wss, wst = ro.corr_sums(tgt, src, m)
                assert abs(b[0] - wss) <= 1e-12 * max(wss, 1e-30) and abs(b[1] - wst) <= 1e-11 * max(wss, 1e-30)
        pad.close()


def test_padded_source_eligibility(dev):
    assert dev.PaddedSource.eligible((256, 512, 512), (256, 512, 512))
    assert dev.PaddedSource.eligible((64, 1024, 1024), (64, 1024, 1024))
    assert not dev.PaddedSource.eligible((64, 2048, 2048), (64, 2048, 2048))   # plane pitch beyond 2^23 bytes
    assert not dev.PaddedSource.eligible((1024, 1024, 1024), (1024, 1024, 1024))  # beyond 2 GiB


def test_affine_identity_is_exact_copy(dev):
    """Sample points are texel centres: the identity matrix reproduces the input."""
    import torch
    x = torch.rand((40, 70, 130), device="cuda")
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32)
    assert torch.equal(dev.affine_resample(x, ident, x.shape), x)


def test_marching_resample_equals_the_one_voxel_kernel(dev, monkeypatch):
    """affine_resample runs as a marching kernel (csrc/resample_march.hip: two voxels per lane, plane / row reuse, clamps
    only on the rim); MIL_RESAMPLE_MARCH=0 keeps the one-voxel kernel of csrc/affine.hip.  Same bits -- interior, rim, out of
    the window, ragged widths (lanes beyond the row), both marching axes, outputs smaller and larger than the source, a
    one-plane source (never off the rim), large rotations."""
    import torch
    rng = np.random.default_rng(5)
    ident = np.eye(3, 4, dtype=f32).reshape(12)

    def rot(axis, deg, scale=1.0, shift=(0, 0, 0), shape=(40, 70, 130)):
        c, s_ = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
        r = {"y": [[c, 0, -s_], [0, 1, 0], [s_, 0, c]], "x": [[1, 0, 0], [0, c, -s_], [0, s_, c]], "z": [[c, -s_, 0], [s_, c, 0], [0, 0, 1]]}[axis]
        lin = scale * np.array(r)
        ctr = np.array([shape[2] / 2, shape[1] / 2, shape[0] / 2])
        return np.concatenate([lin, (ctr - lin @ ctr + np.array(shift))[:, None]], axis=1).astype(f32).reshape(12)
    cases = []
    for s_shape, o_shape in (((40, 70, 130), (40, 70, 130)), ((40, 70, 130), (37, 45, 300)), ((17, 23, 70), (19, 21, 66)),
                             ((33, 40, 256), (33, 40, 256)), ((1, 50, 90), (4, 50, 90)), ((9, 2, 2), (9, 5, 7)),
                             ((48, 64, 128), (64, 16, 128))):
        mats = [ident, rot("y", 3, 1.02, (5, -3, 2), s_shape), rot("y", 2, 1.0, (0.5, 0, -0.5), s_shape),
                rot("x", 2, 0.99, (3, -2, 1), s_shape), rot("z", 7, 1.0, (0, 0, 0), s_shape), rot("y", 45, 1.0, (0, 0, 0), s_shape),
                rot("x", 90, 1.0, (0, 0, 0), s_shape)]
        for d in (-0.5, 0.5, -1.0, 0.25):
            m = ident.copy()
            m[[3, 7, 11]] = d
            mats.append(m)
        for k in range(3):
            m = ident + rng.normal(0, 0.05 * (k + 1), 12).astype(f32)
            m[[3, 7, 11]] = rng.uniform(-4, 4, 3).astype(f32)
            mats.append(m)
        mats.append(np.full(12, np.nan, dtype=f32))
        mats.append(ident * f32(0))
        src = T(rng.standard_normal(s_shape).astype(f32))
        for i, m in enumerate(mats):
            monkeypatch.setenv("MIL_RESAMPLE_MARCH", "0")
            want = dev.affine_resample(src, m, o_shape)
            monkeypatch.delenv("MIL_RESAMPLE_MARCH")
            got = dev.affine_resample(src, m, o_shape)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (s_shape, o_shape, i)
            cases.append(1)
    assert len(cases) > 80


def test_marching_resample_ticket_counters_across_streams_and_graph_replays(dev):
    """The marching resample hands its pieces out through a pair of counters per stream (csrc/resample_march.hip: launch n
    draws from word n % 2 and zeroes the other one for its successor; memset nodes around a launch that is captured into a
    graph).  Back-to-back launches with different grids on two streams, a captured launch replayed between eager ones: every
    output -- written into buffers full of NaN -- equals the first eager result bit for bit."""
    import torch
    rng = np.random.default_rng(11)
    shape = (48, 64, 256)
    src = T(rng.standard_normal(shape).astype(f32))
    m1 = np.array([1, 0, 0, 0.5, 0, 1, 0, -0.25, 0, 0, 1, 0.75], dtype=f32)
    c, s_ = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
    m2 = np.array([c, 0, -s_, 4, 0, 1, 0, 0, s_, 0, c, -3], dtype=f32)
    small = (16, 32, 128)
    want1, want2, want3 = dev.affine_resample(src, m1, shape), dev.affine_resample(src, m2, shape), dev.affine_resample(src, m1, small)
    torch.cuda.synchronize()

    def same(a, b):
        return torch.equal(a.view(torch.int32), b.view(torch.int32))
    lib = dev._lib.load()

    def into(out, m, stream):  # resample into a caller's buffer (pre-filled with NaN) on `stream`
        mm = np.ascontiguousarray(m, dtype=f32)
        rc = lib.mil_affine_resample(out.data_ptr(), src.data_ptr(), out.shape[2], out.shape[1], out.shape[0], shape[2], shape[1],
                                     shape[0], mm.ctypes.data_as(dev._lib.c_float_p), stream.cuda_stream)
        assert rc == 0
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for k in range(7):  # alternating grids on two streams, no host sync in between
        for st, m, w, shp in ((s1, m1, want1, shape), (s2, m2, want2, shape), (s1, m1, want3, small), (s2, m1, want1, shape)):
            o = torch.full(shp, float("nan"), dtype=torch.float32, device="cuda")
            torch.cuda.current_stream().synchronize()
            into(o, m, st)
            outs.append((o, w))
    torch.cuda.synchronize()
    assert all(same(o, w) for o, w in outs)
    # a captured launch, replayed between eager launches of either parity on the same stream
    g = torch.cuda.CUDAGraph()
    gout = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        into(gout, m2, s1)
    torch.cuda.synchronize()
    for k in range(5):
        gout.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            g.replay()
            for j in range(k % 3):  # 0, 1 or 2 eager launches between replays: both parities meet a replayed word
                o = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
                into(o, m1, s1)
                outs.append((o, want1))
        torch.cuda.synchronize()
        assert same(gout, want2), k
    assert all(same(o, w) for o, w in outs)


def test_resample_of_planes_beyond_2_to_23_voxels(dev, monkeypatch):
    """fetch_linear_small forms its offsets with UNSIGNED 24-bit multiplies: a source whose planes hold 2^23 .. 2^24 voxels
    (4096 x 2048) must resample exactly like the 64-bit address path (MIL_RESAMPLE_SMALL=0) -- the signed multiply read such
    a plane pitch as negative and every tap above plane 0 fell outside the buffer (zeros)."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(3)
    src = torch.rand((3, 2048, 4096), device="cuda", generator=g) + 1.0
    m = np.eye(3, 4, dtype=f32).reshape(12)
    m[[3, 7, 11]] = (0.25, -0.5, 0.75)
    m[1], m[4] = 0.001, -0.001
    out_shape = (3, 256, 4096)
    monkeypatch.setenv("MIL_RESAMPLE_MARCH", "0")
    small = dev.affine_resample(src, m, out_shape)
    monkeypatch.setenv("MIL_RESAMPLE_SMALL", "0")
    wide = dev.affine_resample(src, m, out_shape)
    assert torch.equal(small.view(torch.int32), wide.view(torch.int32))
    assert float(small[1, 8:-8, 8:-8].min()) > 0.5  # plane 1 (taps in source planes 1 and 2) holds data, not the zeros of an
    #                                                 out-of-range load; plane 2 maps beyond the source (tz = 3.25)
    monkeypatch.delenv("MIL_RESAMPLE_SMALL")
    monkeypatch.delenv("MIL_RESAMPLE_MARCH")
    mip_a = dev.mip3d(src[:, :64, :256].contiguous(), 2, 4)
    assert torch.isfinite(mip_a).all()


@pytest.mark.parametrize("seed", range(4))
def test_affine_random_matrices_vs_oracle(dev, seed):
    rng = np.random.default_rng(100 + seed)
    src = rng.standard_normal((17, 23, 70)).astype(f32)
    tgt = rng.standard_normal((19, 21, 66)).astype(f32)
    m = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32) + rng.normal(0, 0.08, 12).astype(f32)
    m[[3, 7, 11]] = rng.uniform(-4, 4, 3).astype(f32)
    got = dev.affine_resample(T(src), m, tgt.shape).cpu().numpy()
    assert bits_equal(got, ro.affine_transform(src, m, tgt.shape))
    ss, st = dev.corr_sums(T(tgt), T(src), m)
    wss, wst = ro.corr_sums(tgt, src, m)
    assert abs(ss - wss) <= 1e-12 * wss and abs(st - wst) <= 1e-11 * wss
