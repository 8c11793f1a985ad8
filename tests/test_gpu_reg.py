"""GPU parity of the registration path (reg3d / atrans3dgpu / mil_reg3d_affine)."""
import os

import numpy as np
import pytest

from oracle import decon_oracle as do
from oracle import powell_oracle as po
from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from microimagelib_amd import api
    return api


@pytest.fixture(scope="module")
def dev():
    from microimagelib_amd import device
    return device


def T(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _same_sums(got, want, kern):
    """Kernels 0-5 form their float64 partial sums in ONE order (the 64 x 4 x 16 decomposition): same bits.  Kernel 6 marches
    along y (16 y steps per lane) and the per-matrix choice (-1) may pick it: same voxels and float32 values, another order of
    the float64 additions -- equal to 1e-13 of the sum of squares."""
    if kern in (0, 1, 2, 3, 4, 5):
        return (np.float64(got[0]).view(np.uint64), np.float64(got[1]).view(np.uint64)) == \
               (np.float64(want[0]).view(np.uint64), np.float64(want[1]).view(np.uint64))
    tol = 1e-13 * max(abs(want[0]), 1e-300)
    return abs(got[0] - want[0]) <= tol and abs(got[1] - want[1]) <= tol



def test_atrans3dgpu_golden(api):
    g = np.load(os.path.join(GOLD, "affine.npz"))
    for i, m in enumerate(g["mats"]):
        out = api.atrans3dgpu(g["src"], m, g["tgt"].shape)
        assert np.array_equal(out.view(np.uint32), g["resampled"][i].view(np.uint32)), f"matrix {i}"
    out = api.atrans3dgpu(g["src"], g["resize_matrix"], g["resized"].shape)
    assert np.array_equal(out.view(np.uint32), g["resized"].view(np.uint32))


def test_imresize3d_matches_oracle(api):
    g = np.load(os.path.join(GOLD, "affine.npz"))
    out = api.imresize3d(g["src"], g["resized"].shape)
    assert np.array_equal(out.view(np.uint32), g["resized"].view(np.uint32))


def test_reg3d_choice0(api):
    """regChoice 0: resample with the input matrix, or plain copy + identity (reference :2767-2774)."""
    g = np.load(os.path.join(GOLD, "affine.npz"))
    src = g["src"]
    out, tmx, rec = api.reg3d(g["tgt"], src, reg_choice=0, input_tmx=False)
    assert np.array_equal(out, src)
    assert np.array_equal(tmx, np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32))
    out, tmx, rec = api.reg3d(g["tgt"], src, i_tmx=g["mats"][4], reg_choice=0, input_tmx=True)
    assert np.array_equal(out.view(np.uint32), g["resampled"][4].view(np.uint32))
    assert np.array_equal(tmx, g["mats"][4])


def test_reg3d_error_codes(api):
    a = np.ones((8, 8, 8), dtype=f32)
    for kwargs, code in ((dict(gpu_mem_mode=0), -1), (dict(gpu_mem_mode=5), 1), (dict(reg_choice=9), 1)):
        with pytest.raises(api.LibApiError) as e:
            api.reg3d(a, a, **kwargs)
        assert e.value.code == code


def test_reg3d_self_registration_stays_at_identity(api):
    """Registering a volume with itself: ZNCC 1 at the identity and Powell leaves it there."""
    vol = do.blob_volume((24, 32, 40), n_blobs=14, seed=9)
    out, tmx, rec = api.reg3d(vol, vol, aff_method=5)
    assert abs(rec[1] - 1.0) < 1e-6 and abs(rec[3] - 1.0) < 1e-6
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32)
    assert np.max(np.abs(tmx - ident)) < 2e-2
    assert api.checkmatrix(tmx, 40, 32, 24)


def _assert_same_registration(api, dev, g, key, **call):
    """reg3d must make the oracle's evaluations one by one -- same matrices bit for bit, same costs to 1e-6, same count -- and
    hand back the same matrix, records and resampled volume (tests/golden/reg3d_traj.npz: the oracle's driver with the
    reference's compiled powell(), make_golden.py traj)."""
    want = g[key + "_trace"]
    dev.reg3d_trace_begin(len(want) + 16)
    try:
        out, tmx, rec = api.reg3d(g["tgt"], g["src"], ftol=1e-4, it_limit=3000, **call)
        got, seen = dev.reg3d_trace_read(len(want) + 16)
    finally:
        dev.reg3d_trace_begin(0)
    w_rec = g[key + "_records"]
    assert seen == len(want), (key, seen, len(want))
    same = np.all(got[:, :12].view(np.uint32) == want[:, :12].view(np.uint32), axis=1)
    assert same.all(), f"{key}: first differing evaluation {int(np.argmin(same))} of {len(want)}"
    assert np.max(np.abs(got[:, 12].astype(np.float64) - want[:, 12])) <= 1e-6
    assert rec[5] == w_rec[5] == len(want) - 1  # evaluations after the initial one (src/api_subfunc.cu:2888)
    for i in (1, 2, 3):
        assert abs(float(rec[i]) - float(w_rec[i])) <= 1e-6, (key, i, rec[i], w_rec[i])
    assert np.array_equal(tmx.view(np.uint32), g[key + "_tmx"].view(np.uint32)), key
    assert np.array_equal(out.view(np.uint32), g[key + "_reg"].view(np.uint32)), key


@pytest.mark.parametrize("method", [2, 5, 6, 7])
def test_reg3d_golden(api, dev, method):
    """End to end against the oracle cost driven by the COMPILED REFERENCE Powell, every affMethod schedule
    (src/api_subfunc.cu:2889-2957): translation -> rigid -> 12 DOF (2), direct 12 DOF (5), and the cascades 6 and 7."""
    g = np.load(os.path.join(GOLD, "reg3d_traj.npz"))
    old = np.load(os.path.join(GOLD, "reg3d.npz"))
    assert np.array_equal(g[f"m{method}_records"], old[f"m{method}_records"])  # the run reg3d.npz was made from
    _assert_same_registration(api, dev, dict(g, tgt=old["tgt"], src=old["src"]), f"m{method}", aff_method=method)


@pytest.mark.parametrize("key,method", [("in2", 2), ("in5", 5)])
def test_reg3d_with_input_matrix(api, dev, key, method):
    """flagTmx: with affMethod != 5 the source is first resampled by the input matrix, registration runs from the identity
    and the result is composed as iTmx . M (src/api_subfunc.cu:2958-2961); with affMethod 5 the matrix is the start point
    (:2818-2821).  Same evaluations as the oracle, one by one."""
    g = np.load(os.path.join(GOLD, "reg3d_traj.npz"))
    old = np.load(os.path.join(GOLD, "reg3d.npz"))
    _assert_same_registration(api, dev, dict(g, tgt=old["tgt"], src=old["src"]), key, aff_method=method, i_tmx=g["start"],
                              input_tmx=True)


def test_reg3d_different_source_size(api):
    """A differently sized source is centre-cropped / zero-padded to the target box first
    (reference src/api_reg.cpp:401-406)."""
    g = np.load(os.path.join(GOLD, "reg3d.npz"))
    src_big = np.zeros((28, 36, 46), dtype=f32)
    src_big[2:26, 2:34, 3:43] = g["src"]
    out, tmx, rec = api.reg3d(g["tgt"], src_big, reg_choice=0)
    assert np.array_equal(out, do.align_size(src_big, g["tgt"].shape))


def test_cost_matches_oracle_along_a_path(dev):
    """ZNCC cost at a sequence of matrices (as Powell would visit) vs the oracle: < 1e-6."""
    g = np.load(os.path.join(GOLD, "reg3d.npz"))
    tgt_ms, _ = ro.mean_subtract(g["tgt"])
    src_ms, _ = ro.mean_subtract(g["src"])
    sd_t = f32(np.sqrt(float(np.sum((tgt_ms * tgt_ms).astype(np.float64)))))
    dt, ds = T(tgt_ms), T(src_ms)
    rng = np.random.default_rng(0)
    for k in range(12):
        p = np.array([0, *(rng.normal(0, 1.0, 3)), *(rng.normal(0, 2.0, 3)), *(1 + rng.normal(0, 0.02, 3))], dtype=f32)
        m = ro.dof9tomatrix(p, 9)
        ss, st = dev.corr_sums(dt, ds, m)
        got = f32(f32(st / np.sqrt(ss)) / sd_t)
        want = ro.corrfunc(tgt_ms, sd_t, src_ms, m)
        assert abs(float(got) - float(want)) < 1e-6


def test_projections_and_rotation_host_api(api):
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 10, (6, 9, 13)).astype(f32)
    xy, yz, zx = api.mp2dgpu(a)
    assert np.array_equal(xy, a.max(axis=0))
    assert np.array_equal(yz, a.max(axis=2))       # [z][y]
    assert np.array_equal(zx, a.max(axis=1).T)     # [x][z]
    assert np.array_equal(api.imoperation3D(a, 1), np.ascontiguousarray(np.transpose(a, (2, 1, 0))[::-1]))
    assert np.array_equal(api.imoperation3D(a, 2), np.ascontiguousarray(np.transpose(a, (2, 1, 0))[:, :, ::-1]))
    assert np.array_equal(api.alignsize3d(a, (8, 7, 16)), do.align_size(a, (8, 7, 16)))


def test_mip3d_matches_oracle(api):
    vol = do.blob_volume((20, 36, 50), n_blobs=10, seed=4)
    for axis in (1, 2):
        got = api.mip3dgpu(vol, axis, 6)
        sz, sy, sx = vol.shape
        if axis == 1:
            rot = int(round(np.sqrt(float(sy * sy + sz * sz))))
            box = (rot, rot, sx)
        else:
            rot = int(round(np.sqrt(float(sx * sx + sz * sz))))
            box = (rot, sy, rot)
        step = f32(3.14159 * 2 / float(f32(6)))
        for i in range(6):
            m = ro.rot2matrix(f32(step * f32(i)), sx, sy, sz, axis)
            want = np.maximum(ro.affine_transform(vol, m, box).max(axis=0), f32(0))
            # rot2matrix is pinned bit for bit, the resample is bit-exact and a maximum is exact: so is the projection
            assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), f"axis {axis} angle {i}"


def test_fused_mip3d_equals_the_two_kernel_sequence(api, monkeypatch):
    """mip3dgpu runs resample + z projection as ONE kernel per angle (the rotated box never exists); MIL_MIP3D_FUSED=0 runs the
    reference's sequence affineTransform -> maxprojection (src/apifunc.cpp:610-613).  Same bits, also with negative voxels
    (the projection's accumulator starts at 0) and with more z steps than one segment of the fused kernel covers."""
    rng = np.random.default_rng(5)
    vol = (do.blob_volume((40, 70, 90), n_blobs=25, seed=9) - f32(120.0)).astype(np.float32)  # background below zero
    vol += rng.standard_normal(vol.shape).astype(np.float32)
    for axis in (1, 2):
        monkeypatch.setenv("MIL_MIP3D_FUSED", "0")
        want = api.mip3dgpu(vol, axis, 9)
        monkeypatch.delenv("MIL_MIP3D_FUSED")
        got = api.mip3dgpu(vol, axis, 9)
        assert want.min() == 0.0 and want.max() > 0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"axis {axis}"


def test_fused_resample_projection_kernels_agree(monkeypatch):
    """resample_mip_z in its three forms -- second version (csrc/resample_march.hip: taps under exact zero weights left out for
    the rotation series' matrices, z range clipped per wave), first version (MIL_MIP3D_V2=0) and the two-kernel sequence
    affine_resample + max_projection -- on rotation-series matrices about x and y, general matrices (no tap can be left out:
    only the clipping applies), shifts that put a whole face outside, and a matrix that maps everything outside."""
    import torch
    from microimagelib_amd import device as dev
    rng = np.random.default_rng(8)
    vol = (do.blob_volume((40, 70, 90), n_blobs=25, seed=9) - f32(60.0)).astype(np.float32)
    vol += rng.standard_normal(vol.shape).astype(np.float32)
    src = torch.from_numpy(vol).cuda()
    sz, sy, sx = vol.shape
    cases = []
    for axis in (1, 2):
        rot = int(round(np.sqrt(float((sy if axis == 1 else sx) ** 2 + sz * sz))))
        box = (rot, rot, sx) if axis == 1 else (rot, sy, rot)
        for deg in (0, 10, 40, 90, 130, 180, 270, 350):
            cases.append((ro.rot2matrix(f32(np.deg2rad(deg)), sx, sy, sz, axis), box))
    ident = np.eye(3, 4, dtype=f32).reshape(12)
    for k in range(6):
        m = ident + rng.normal(0, 0.1 * (k + 1), 12).astype(f32)
        m[[3, 7, 11]] = rng.uniform(-20, 20, 3).astype(f32)
        cases.append((m, (50, 64, 100)))
    far = ident.copy()
    far[[3, 7, 11]] = (500.0, 0.0, 0.0)
    cases.append((far, (40, 70, 90)))
    half = ident.copy()
    half[[3, 7, 11]] = (0.5, -30.0, 0.25)  # x row (1, 0, 0, 0.5): NOT an integer shift, no tap may be dropped
    cases.append((half, (40, 70, 90)))
    flat = ident.copy()
    flat[10] = 0.0  # every z of the box on one source plane
    flat[11] = 7.25
    cases.append((flat, (33, 70, 90)))
    for i, (m, box) in enumerate(cases):
        want = dev.max_projection(dev.affine_resample(src, m, box), 1)
        got = dev.resample_mip_z(src, m, box)
        monkeypatch.setenv("MIL_MIP3D_V2", "0")
        old = dev.resample_mip_z(src, m, box)
        monkeypatch.delenv("MIL_MIP3D_V2")
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), f"case {i}: second version"
        assert torch.equal(old.view(torch.int32), want.view(torch.int32)), f"case {i}: first version"


# ---------------------------------------------------------------- regChoice 4: 2-D MIP shift search
def _shifted_pair(shape=(24, 40, 48), shift=(3.0, -2.0, 1.6)):
    tgt = do.blob_volume(shape, n_blobs=15, seed=3)
    m = np.array([1, 0, 0, -shift[0], 0, 1, 0, -shift[1], 0, 0, 1, -shift[2]], dtype=np.float32)
    return tgt, ro.affine_transform(tgt, m, tgt.shape)


def test_mip_shift_search_matches_oracle():
    import torch
    from microimagelib_amd import device
    tgt, src = _shifted_pair()
    got = device.reg3d_mip_shift(torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda())
    want = ro.mip_shift_search(tgt, src)
    assert np.array_equal(got, want), (got, want)  # same grid point on every axis
    assert np.all(np.abs(got - np.array([3.0, -2.0, 1.6])) < 0.5)  # within one grid step of the true shift


def test_reg3d_regchoice4_recovers_shift():
    """regChoice 4 through libapi.h: MIP shifts seed the affine stage (src/api_reg.cpp:448-497)."""
    from microimagelib_amd import api
    tgt, src = _shifted_pair()
    reg, tmx, rec = api.reg3d(tgt, src, reg_choice=4, aff_method=1, it_limit=300)
    assert rec[3] > rec[1] and rec[3] > 0.9, rec  # final ZNCC (the shifted copy has zero-filled borders)
    assert abs(tmx[3] - 3.0) < 0.3 and abs(tmx[7] + 2.0) < 0.3 and abs(tmx[11] - 1.6) < 0.3, tmx
    assert np.allclose(tmx[[0, 5, 10]], 1.0) and np.allclose(tmx[[1, 2, 4, 6, 8, 9]], 0.0)


# ---------------------------------------------------------------- regChoice 1 / 3: phasor registration
@pytest.mark.parametrize("shift", [(3, -2, 1), (20, -15, 9), (-30, 4, -8), (0, 0, 0)])
@pytest.mark.parametrize("wrap", [True, False])
def test_phasor_shift_matches_oracle(shift, wrap):
    import torch
    from microimagelib_amd import device
    tgt = do.blob_volume((24, 40, 48), n_blobs=15, seed=3, background=0.0)
    src = ro.circshift(tgt, *shift) if wrap else ro.imshift(tgt, *shift)
    got = device.reg3d_phasor_shift(torch.from_numpy(tgt).cuda(), torch.from_numpy(np.ascontiguousarray(src)).cuda())
    want = ro.reg3d_phasor(tgt, src)
    assert np.array_equal(got, want), (got, want)
    if wrap and max(abs(shift[0]) * 4 // 48, abs(shift[1]) * 4 // 40, abs(shift[2]) * 4 // 24) == 0:
        assert tuple(got) == shift  # small circular shifts are recovered exactly


def test_shift_volume_matches_oracle():
    import torch
    from microimagelib_amd import device
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1, (7, 9, 20)).astype(np.float32)
    for d in ((3, -2, 1), (-19, 8, 6), (0, 0, 0), (25, 0, -9)):
        for wrap in (True, False):
            got = device.shift_volume(torch.from_numpy(a).cuda(), *d, wrap=wrap).cpu().numpy()
            want = ro.circshift(a, *d) if wrap else ro.imshift(a, *d)
            assert np.array_equal(got, want), (d, wrap)


def test_reg3d_regchoice1_and_3():
    """regChoice 1: result = source shifted back, matrix = pure translation (src/api_reg.cpp:418-427);
    regChoice 3: that translation seeds the affine stage (:434-446)."""
    from microimagelib_amd import api
    tgt = do.blob_volume((24, 40, 48), n_blobs=15, seed=3, background=0.0)
    src = np.ascontiguousarray(ro.circshift(tgt, 5, -3, 2))
    reg, tmx, _ = api.reg3d(tgt, src, reg_choice=1)
    assert np.array_equal(tmx, np.array([1, 0, 0, 5, 0, 1, 0, -3, 0, 0, 1, 2], dtype=np.float32))
    assert np.array_equal(reg, ro.imshift(src, -5, 3, -2))
    reg3, tmx3, rec3 = api.reg3d(tgt, src, reg_choice=3, aff_method=1, it_limit=200)
    # the matrix is that of Powell's LAST line-search trial (reference quirk), within Brent's tolerance of the optimum
    assert abs(tmx3[3] - 5) < 0.5 and abs(tmx3[7] + 3) < 0.5 and abs(tmx3[11] - 2) < 0.5, tmx3
    assert rec3[3] > 0.9, rec3


# ---------------------------------------------------------------- reg2d
def _pair_2d(shift=(3.3, -2.2)):
    tgt = do.blob_volume((1, 48, 64), n_blobs=12, seed=5, background=0.0)[0]
    a = np.array([1, 0, -shift[0], 0, 1, -shift[1]], dtype=np.float32)
    return tgt, ro.affine_transform_2d(tgt, a, tgt.shape)


@pytest.mark.parametrize("choice", [0, 1, 2, 3])
def test_reg2d_matches_oracle(choice):
    from microimagelib_amd import api
    tgt, src = _pair_2d()
    given = np.array([1, 0, 2.5, 0, 1, -1.5], dtype=np.float32)
    powell = po.ref_powell if po.ref_available() else po.powell
    use = choice == 0
    reg, tmx, rec = api.reg2d(tgt, src, i_tmx=given if use else None, reg_choice=choice, input_tmx=use, ftol=1e-4,
                              it_limit=500)
    w_reg, w_tmx, w_rec = ro.reg2d(tgt, src, choice, powell_fn=powell, flag_tmx=use, i_tmx=given, ftol=1e-4, it_limit=500)
    assert np.allclose(tmx, w_tmx, rtol=0, atol=1e-5), (tmx, w_tmx)
    assert np.allclose(reg, w_reg, rtol=0, atol=1e-4 * float(np.max(np.abs(w_reg))))
    if choice == 2:
        assert rec[5] == w_rec[5] and abs(rec[3] - w_rec[3]) < 1e-5  # same evaluation count and optimum
        assert rec[3] > rec[1]  # Powell improved the ZNCC (the 6-DOF optimum need not be the pure shift)
    if choice in (1, 3):
        assert abs(tmx[2] - 3.3) < 1.0 and abs(tmx[5] + 2.2) < 1.0


# ---------------------------------------------------------------- benchmark size against the oracle's fixture
def test_full_size_cost_and_resample_vs_oracle_fixture(dev, api):
    """BASELINE.json config 4 AT the benchmark size (512 x 512 x 256): correlation-cost sums for six matrices through
    the general kernel and every padded-source kernel (include/cukernel.cuh:526-556), the resampled source for two
    of them (include/cukernel.cuh:501-524) and the initial ZNCC of reg3d_affine1 (src/api_subfunc.cu:2879-2881),
    against oracle outputs committed as a fixture (tests/golden/make_golden_fullsize.py)."""
    import torch
    from tests import fullsize_inputs as fi
    g = np.load(os.path.join(GOLD, "fullsize_reg.npz"))
    shape = fi.FULL_SHAPE
    tgt_raw, src_raw = fi.hash_volume(shape, 1), fi.hash_volume(shape, 3)
    tgt_ms, tgt_sum = ro.mean_subtract(tgt_raw)
    src_ms, src_sum = ro.mean_subtract(src_raw)
    assert tgt_sum == float(g["tgt_sum"]) and src_sum == float(g["src_sum"])  # the inputs are the fixture's inputs
    t_dev, s_dev = T(tgt_ms), T(src_ms)
    pad = dev.PaddedSource(s_dev)
    for i, m in enumerate(g["mats"]):
        wss, wst = g["corr_sums"][i]
        ss, st = dev.corr_sums(t_dev, s_dev, m)
        assert abs(ss - wss) <= 1e-12 * wss and abs(st - wst) <= 1e-12 * wss, (i, ss, st, wss, wst)
        for kern in (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
            dev.corr_padded_force_kernel(kern)
            assert _same_sums(dev.corr_sums(t_dev, pad, m), (ss, st), kern), f"matrix {i}, padded kernel {kern}"
    dev.corr_padded_force_kernel(-1)
    pad.close()
    del t_dev, s_dev
    src_dev = T(src_raw)
    idx = torch.from_numpy(fi.sample_indices(int(np.prod(shape)))).cuda()
    for i in (1, 3):
        reg = dev.affine_resample(src_dev, g["mats"][i], shape)
        got = reg.reshape(-1)[idx].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), g[f"resample{i}_samples"].view(np.uint32)), f"matrix {i}"
        planes = reg.double().sum(dim=(1, 2)).cpu().numpy()
        want = g[f"resample{i}_planes"]
        assert np.all(np.abs(planes - want) <= 1e-12 * np.maximum(np.abs(want), 1.0)), f"matrix {i}"
        assert int(torch.count_nonzero(reg)) == int(g[f"resample{i}_nonzero"])
    # the same resample through the reference's entry point (regChoice 0 + input matrix, src/api_reg.cpp:264)
    out, tmx, _ = api.reg3d(tgt_raw, src_raw, i_tmx=g["mats"][1], reg_choice=0, input_tmx=True)
    assert np.array_equal(out.reshape(-1)[fi.sample_indices(out.size)].view(np.uint32),
                          g["resample1_samples"].view(np.uint32))
    # records[1] of the optimiser = ZNCC at the identity = (st / sqrt(ss)) / sd_t in float32
    _, _, rec = dev.reg3d_affine(T(tgt_raw), src_dev, aff_method=1, it_limit=5)
    ss0, st0 = g["corr_sums"][0]
    want = f32(f32(st0 / np.sqrt(ss0)) / f32(np.sqrt(float(g["tgt_sumsq"]))))
    assert abs(rec[1] - want) <= 2e-7 * abs(want), (rec[1], want)


@pytest.mark.parametrize("fixture", ["fullsize_reg_traj.npz", "fullsize_reg_traj_full.npz"])
def test_full_size_registration_trajectory_vs_oracle_fixture(dev, fixture):
    """BASELINE.json config 4 END TO END at 512 x 512 x 256: reg3d_affine1's 3 -> 6 -> 9 -> 12 DOF cascade
    (src/api_subfunc.cu:2889-2957, affMethod 7) on the closed-form sheared pair of tests/fullsize_inputs.py.  The oracle
    (its driver + the reference's compiled powell(), every cost evaluated by the NumPy restatement of corrkernel) recorded
    every matrix it evaluated and the cost it saw (tests/golden/make_golden_fullsize.py traj / trajfull); the HIP path
    must visit the SAME matrices bit for bit, see the same costs to 1e-6, stop after the same number of evaluations and
    hand back the same matrix and records.  itLimit 40 (the stages hand over after their first line minimisations) and
    the cascade run to convergence."""
    import torch
    from tests import fullsize_inputs as fi
    path = os.path.join(GOLD, fixture)
    if not os.path.exists(path):
        pytest.skip(f"{fixture} not generated")
    g = np.load(path)
    shape = fi.FULL_SHAPE
    tgt, src = T(fi.hash_volume(shape, 1)), T(fi.displaced_source(shape))
    want = g["trace"]
    dev.reg3d_trace_begin(len(want) + 16)
    reg, tmx, rec = dev.reg3d_affine(tgt, src, aff_method=int(g["aff_method"]), ftol=1e-4, it_limit=int(g["it_limit"]))
    got, seen = dev.reg3d_trace_read(len(want) + 16)
    dev.reg3d_trace_begin(0)
    assert seen == len(want), (seen, len(want))
    same = np.all(got[:, :12].view(np.uint32) == want[:, :12].view(np.uint32), axis=1)
    assert same.all(), f"first differing evaluation: {int(np.argmin(same))} of {len(want)}"
    assert np.max(np.abs(got[:, 12].astype(np.float64) - want[:, 12])) <= 1e-6
    assert np.array_equal(tmx.view(np.uint32), g["tmx"].view(np.uint32))
    w_rec = g["records"]
    assert rec[5] == w_rec[5] == len(want) - 1  # evaluations after the initial one (src/api_subfunc.cu:2888)
    for i in (1, 2, 3):
        assert abs(float(rec[i]) - float(w_rec[i])) <= 1e-6, (i, rec[i], w_rec[i])
    idx = torch.from_numpy(fi.sample_indices(int(np.prod(shape)))).cuda()
    assert np.array_equal(reg.reshape(-1)[idx].cpu().numpy().view(np.uint32), g["reg_samples"].view(np.uint32))
    planes = reg.double().sum(dim=(1, 2)).cpu().numpy()
    assert np.all(np.abs(planes - g["reg_planes"]) <= 1e-12 * np.maximum(np.abs(g["reg_planes"]), 1.0))


def test_registration_orders_itself_after_pending_stream_work(dev):
    """Inputs still in flight on torch's current stream (a non_blocking copy from pinned memory, a kernel) when the call
    is made: the entry points order their private stream after them -- no host synchronisation by the caller."""
    import torch
    rng = np.random.default_rng(3)
    shape = (64, 96, 128)
    host = torch.from_numpy(rng.uniform(10, 500, shape).astype(f32)).pin_memory()
    want = None
    for attempt in range(3):
        big = torch.empty((256, 512, 512), device="cuda")
        big.normal_()  # keeps stream 0 busy while the copy below is queued behind it
        vol = torch.empty(shape, device="cuda")
        vol.copy_(host, non_blocking=True)
        moved = vol * 1.0
        _, tmx, rec = dev.reg3d_affine(vol, moved, aff_method=1, it_limit=30)
        if want is None:
            torch.cuda.synchronize()
            _, want_tmx, want = dev.reg3d_affine(vol, moved, aff_method=1, it_limit=30)
        assert rec[1] > 0.999999 and rec[1] == want[1], (attempt, rec[1])
        del big


def test_concurrent_registrations_share_a_device(api):
    """Four host threads register on ONE device at the same time (the batch tool with a device list such as 0,0,0,0): every
    call succeeds and returns what the same call returns alone, bit for bit.  Found with four workers at 512 x 512 x 256: the
    completion counters of a cost workspace were zeroed with hipMemset on the NULL stream, which is asynchronous to the host
    and not ordered with the registration's own stream -- under load it landed inside the first evaluations, no workgroup was
    "last", and the call failed with "the cost kernel did not publish its result"."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    pairs = [_shifted_pair(shape=(40, 56, 64), shift=(1.0 + 0.5 * k, -2.0 + 0.25 * k, 0.4 * k)) for k in range(4)]
    want = [api.reg3d(t, s, aff_method=6, it_limit=400) for t, s in pairs]
    load = torch.empty((256, 512, 512), device="cuda")

    def work(k):
        out = []
        for rep in range(6):  # a fresh workspace (and its zeroing) per call
            if k == 0:
                load.normal_()  # the NULL stream stays busy too
            out.append(api.reg3d(*pairs[k], aff_method=6, it_limit=400))
        return out

    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(work, range(4)))
    for k in range(4):
        for reg, tmx, rec in got[k]:
            assert np.array_equal(tmx, want[k][1]), (k, tmx, want[k][1])
            assert np.array_equal(reg, want[k][0]) and rec[5] == want[k][2][5]


# ---------------------------------------------------------------- benchmark size, size-independent properties
def test_full_size_registration_properties(dev):
    """512x512x256 (BASELINE.json config 4), beside the fixture comparison above: size-independent properties.
    Identity resample is exact; ZNCC of a volume with itself is 1; the cost is linear in a target scale and blind to a
    target offset after mean removal; a known integer shift is what the cost prefers; Powell started at the optimum of
    a self-registration stays there."""
    import torch
    shape = (256, 512, 512)
    g = torch.Generator(device="cuda").manual_seed(5)
    vol = torch.rand(shape, device="cuda", generator=g) * 500 + 20
    ident = np.eye(3, 4, dtype=np.float32).reshape(12)
    assert torch.equal(dev.affine_resample(vol, ident, shape), vol)
    ms = vol - vol.mean()
    ss, st = dev.corr_sums(ms, ms, ident)
    assert abs(st / ss - 1.0) < 1e-12 and ss > 0
    ss2, st2 = dev.corr_sums(ms * 2.0, ms, ident)  # linear in the target
    pad = dev.PaddedSource(ms)  # the border-padded form reg3d evaluates: same bits at full size
    rot = ident.copy()
    c, s_ = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
    rot[[0, 2, 8, 10]] = np.array([c, -s_, s_, c], dtype=np.float32) * np.float32(1.02)
    rot[[3, 7, 11]] = (7.0, -3.0, -11.0)
    rot20 = ident.copy()
    c, s_ = np.cos(np.deg2rad(20.0)), np.sin(np.deg2rad(20.0))
    rot20[[0, 1, 4, 5]] = np.array([c, -s_, s_, c], dtype=np.float32)
    rot20[[3, 7]] = (90.0, -80.0)
    for m in (ident, rot, rot20):
        want = dev.corr_sums(ms, ms, m)
        for kern in (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
            dev.corr_padded_force_kernel(kern)
            assert _same_sums(dev.corr_sums(ms, pad, m), want, kern), f"kernel {kern}"
    dev.corr_padded_force_kernel(-1)
    # The last workgroup to finish sums the partials of all the others, which reach it through agent-scope stores and
    # loads instead of fences (csrc/corr_finish.h): hundreds of evaluations at full size must all publish the right bits.
    # Consecutive evaluations alternate between matrices, so a partial left over from the previous evaluation (a stale
    # read) would change the sum.
    mats = (rot, ident, rot20)
    want = [dev.corr_sums(ms, pad, m) for m in mats]
    for i in range(300):
        assert dev.corr_sums(ms, pad, mats[i % 3]) == want[i % 3], f"evaluation {i}"
    want = [dev.corr_sums(ms, ms, m) for m in mats]
    for i in range(60):
        assert dev.corr_sums(ms, ms, mats[i % 3]) == want[i % 3], f"evaluation {i} (general kernel)"
    pad.close()
    assert abs(ss2 - ss) <= 1e-12 * ss and abs(st2 - 2.0 * st) <= 1e-9 * abs(st)
    shift = ident.copy()
    shift[3], shift[7], shift[11] = 4.0, -3.0, 2.0
    moved = dev.affine_resample(vol, shift, shape)  # moved(p) = vol(p + d)
    back = ident.copy()
    back[3], back[7], back[11] = -4.0, 3.0, -2.0
    t_ms, m_ms = vol - vol.mean(), moved - moved.mean()

    def zncc(m):
        a, b = dev.corr_sums(t_ms, m_ms, m)
        return b / np.sqrt(a)
    assert zncc(back) > zncc(ident) and zncc(back) > zncc(shift)
    reg, tmx, rec = dev.reg3d_affine(vol, vol, aff_method=1, it_limit=40)
    assert rec[1] > 0.999999 and rec[3] >= rec[1] - 1e-6
    assert np.allclose(tmx, ident, atol=0.05)


def test_memo_of_repeated_evaluations_changes_nothing(dev, monkeypatch):
    """Powell re-evaluates points it has just visited; reg3d answers those from its last 32 results instead of another
    pass over the volumes (csrc/reg.cpp).  Same matrix, same ZNCC, same evaluation count as with the memo off -- only the
    number of kernel launches differs."""
    g = np.load(os.path.join(GOLD, "reg3d.npz"))
    tgt, src = T(g["tgt"]), T(g["src"])
    monkeypatch.delenv("MIL_REG_NOMEMO", raising=False)
    _, tmx, rec = dev.reg3d_affine(tgt, src, aff_method=6, it_limit=600)
    launches = dev.reg3d_last_cost_launches()
    monkeypatch.setenv("MIL_REG_NOMEMO", "1")
    _, tmx2, rec2 = dev.reg3d_affine(tgt, src, aff_method=6, it_limit=600)
    launches2 = dev.reg3d_last_cost_launches()
    assert np.array_equal(tmx.view(np.uint32), tmx2.view(np.uint32))
    assert rec[3] == rec2[3] and rec[5] == rec2[5] and rec[1] == rec2[1]
    assert launches < launches2 == int(rec2[5])  # without the memo every evaluation launches


def test_texture_weight_emulation_switch(dev, api):
    """mil_set_texture_weight_bits(8) / MIL_TEX_WEIGHT_BITS=8: interpolation weights rounded to 8 fractional bits like
    the CUDA texture unit the reference samples through (include/cukernel.cuh:510-522, 545-546).
    Resample and cost sums equal the oracle's with the same quantisation, differ from the exact-weight results by
    what 1/512 of a voxel can do, and the switch restores the default bits."""
    rng = np.random.default_rng(21)
    src = rng.uniform(0, 500, (20, 34, 70)).astype(f32)
    tgt = rng.uniform(0, 500, (18, 30, 64)).astype(f32)
    m = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=f32) + rng.normal(0, 0.03, 12).astype(f32)
    m[[3, 7, 11]] = (1.37, -0.61, 2.22)
    ds, dt = T(src), T(tgt)
    exact = dev.affine_resample(ds, m, tgt.shape).cpu().numpy()
    exact_sums = dev.corr_sums(dt, ds, m)
    assert dev.set_texture_weight_bits(8) == 8
    try:
        ro.WEIGHT_BITS = 8
        got = dev.affine_resample(ds, m, tgt.shape).cpu().numpy()
        want = ro.affine_transform(src, m, tgt.shape)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        ss, st = dev.corr_sums(dt, ds, m)
        wss, wst = ro.corr_sums(tgt, src, m)
        assert abs(ss - wss) <= 1e-12 * wss and abs(st - wst) <= 1e-11 * wss
        diff = np.abs(got - exact)
        assert 0 < diff.max() < 500 * 3 / 512 + 1e-3  # three axes, each weight off by at most 1/512
        # the registration falls back to the general cost kernel and still converges on a self-registration
        vol = T(do.blob_volume((32, 48, 64), n_blobs=10, seed=4))
        _, tmx, rec = dev.reg3d_affine(vol, vol, aff_method=1, it_limit=40)
        assert rec[1] > 0.999999 and np.allclose(tmx, np.eye(3, 4, dtype=f32).reshape(12), atol=0.05)
        out = api.atrans3dgpu(src, m, tgt.shape)  # the libapi.h entry points follow the same switch
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    finally:
        ro.WEIGHT_BITS = 0
        assert dev.set_texture_weight_bits(0) == 0
    again = dev.affine_resample(ds, m, tgt.shape).cpu().numpy()
    assert np.array_equal(again.view(np.uint32), exact.view(np.uint32)) and dev.corr_sums(dt, ds, m) == exact_sums


def test_reg3d_on_a_ragged_width_takes_the_same_path_as_the_one_voxel_kernels(api, dev):
    """A target 250 voxels wide -- not a multiple of 128 -- runs the chunked two-voxel cost kernels with the lanes beyond the row
    kept out of the sums (csrc/corr_padded.hip, RAGGED; round 6); until then such volumes fell back to the one-voxel tiled kernels
    (forced here: kernel 0).  Same voxels, same float32 values, float64 sums within 1e-15: the whole 6 -> 12-DOF cascade must
    make the same number of evaluations and return the same matrix and the same registered volume, bit for bit."""
    rng = np.random.default_rng(21)
    shape = (24, 40, 250)
    tgt = do.blob_volume(shape, n_blobs=40, seed=5) + rng.normal(0, 1.0, shape).astype(f32)
    m = ro.dof9tomatrix(np.array([0, 1.5, -0.8, 0.6, 1.2, -0.7, 2.0, 1.01, 0.99, 1.0], dtype=f32), 9)
    src = ro.affine_transform(tgt, m, shape) + rng.normal(0, 1.0, shape).astype(f32)
    dev.corr_padded_force_kernel(0)
    try:
        want = api.reg3d(tgt, src, reg_choice=2, aff_method=6)
    finally:
        dev.corr_padded_force_kernel(-1)
    got = api.reg3d(tgt, src, reg_choice=2, aff_method=6)
    assert got[2][5] == want[2][5] and got[2][5] > 100, (got[2][5], want[2][5])  # evaluations
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), (got[1], want[1])
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert got[2][3] > 0.5, got[2]  # and it did register (final ZNCC)


def test_fenced_and_fence_free_tails_agree(dev, monkeypatch):
    """The cost kernels hand their partial sums from workgroup to workgroup with agent-scope (sc1) stores / loads and
    `s_waitcnt vmcnt(0)` instead of release / acquire fences (csrc/corr_finish.h).  MIL_CORR_FENCES=1 adds the fences:
    both forms must publish the same bits, for every kernel, evaluation after evaluation."""
    import torch
    shape = (128, 256, 256)
    g = torch.Generator(device="cuda").manual_seed(11)
    vol = torch.rand(shape, device="cuda", generator=g) * 300 - 100
    tgt = torch.rand(shape, device="cuda", generator=g) * 300 - 100
    pad = dev.PaddedSource(vol)
    ident = np.eye(3, 4, dtype=np.float32).reshape(12)
    rot = ident.copy()
    c, s_ = np.cos(np.deg2rad(4.0)), np.sin(np.deg2rad(4.0))
    rot[[0, 2, 8, 10]] = np.array([c, -s_, s_, c], dtype=np.float32) * np.float32(0.98)
    rot[[3, 7, 11]] = (3.0, -2.0, 6.0)
    for kern in (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
        dev.corr_padded_force_kernel(kern)
        for m in (ident, rot):
            dev.corr_set_fences(0)
            free = [dev.corr_sums(tgt, pad, m) for _ in range(3)]
            dev.corr_set_fences(1)
            fenced = [dev.corr_sums(tgt, pad, m) for _ in range(3)]
            assert len(set(free + fenced)) == 1, (kern, free, fenced)
    dev.corr_padded_force_kernel(-1)
    dev.corr_set_fences(1)
    a = dev.corr_sums(tgt, vol, rot)  # the general kernel (csrc/affine.hip) shares the tail
    dev.corr_set_fences(-1)
    assert dev.corr_sums(tgt, vol, rot) == a
    assert _same_sums(dev.corr_sums(tgt, pad, rot), a, -1)
    pad.close()


def test_cost_tail_sums_many_workgroups(dev):
    """196 608 workgroups (> 2048 partials per group of the two-level tail, csrc/corr_finish.h) on a 4096 x 4096 x 48
    volume, which is also beyond the padded kernels' 32-bit offsets: a constant volume under the identity has sums that
    are exact in float64, N * c * c."""
    import torch
    shape = (48, 4096, 4096)
    c = 0.5
    vol = torch.full(shape, c, device="cuda")
    ident = np.eye(3, 4, dtype=np.float32).reshape(12)
    assert not dev.PaddedSource.eligible(shape, shape)
    n = float(np.prod(shape))
    for _ in range(3):
        ss, st = dev.corr_sums(vol, vol, ident)
        assert ss == n * c * c and st == n * c * c
