"""GPU: the device-resident batch pipeline (reg3d -> dual-view RL -> XY MIP) on one rank, checked
against the oracle per time point."""
import numpy as np
import pytest

from oracle import decon_oracle as do

pytestmark = pytest.mark.gpu


def test_fusion_pipeline_single_rank():
    import torch
    assert torch.cuda.is_available()
    from microimagelib_amd import batch

    shape = (32, 64, 64)
    psf_a = do.gaussian_psf((9, 9, 9), (2.0, 1.2, 1.2))
    psf_b = do.gaussian_psf((9, 9, 9), (1.2, 1.2, 2.0))
    views = {t: (do.synthetic_view(shape, psf_a, seed=t, noise_seed=10 + t, n_blobs=12),
                 do.synthetic_view(shape, psf_b, seed=t, noise_seed=20 + t, n_blobs=12)) for t in (1, 2, 3)}
    pipe = batch.FusionPipeline(shape, torch.from_numpy(psf_a).cuda(), torch.from_numpy(psf_b).cuda(), iters=4)
    decons = {}

    def load_pair(t):
        a, b = views[t]
        return torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()

    def sink(t, decon, tmx, rec):
        decons[t] = decon.cpu().numpy().copy()
        assert np.allclose(tmx, np.eye(3, 4, dtype=np.float32).reshape(12))  # regMode 0 without matrix: copy

    stack = batch.run_sharded([1, 2, 3], load_pair, pipe.fuse_pair, batch.RegPolicy(reg_mode=0), pipe.reg_fn,
                              pipe.check_fn, (shape[1], shape[2]), sink=sink, device="cuda")
    pipe.close()
    assert stack.shape == (3, shape[1], shape[2])
    for i, t in enumerate((1, 2, 3)):
        want = do.decon_dualview(views[t][0], views[t][1], psf_a, psf_b, 4)
        floor = float(np.mean(np.abs(want)))
        assert float(np.max(np.abs(decons[t] - want) / np.maximum(np.abs(want), floor))) < 1e-5
        assert np.array_equal(stack[i].cpu().numpy(), np.maximum(decons[t].max(axis=0), 0))


def test_fusion_pipeline_with_independent_registration():
    """regMode 3 on one rank: every time point registered (3-DOF, regChoice 2) before the joint RL; the pipeline's
    result equals the same calls through the host-pointer API."""
    import torch
    from microimagelib_amd import api, batch

    shape = (32, 64, 64)
    psf_a = do.gaussian_psf((9, 9, 9), (2.0, 1.2, 1.2))
    psf_b = do.gaussian_psf((9, 9, 9), (1.2, 1.2, 2.0))
    views = {t: (do.synthetic_view(shape, psf_a, seed=t, noise_seed=10 + t, n_blobs=12),
                 do.synthetic_view(shape, psf_b, seed=t, noise_seed=20 + t, n_blobs=12)) for t in (4, 5)}
    pipe = batch.FusionPipeline(shape, torch.from_numpy(psf_a).cuda(), torch.from_numpy(psf_b).cuda(), iters=3)
    got = {}

    def sink(t, decon, tmx, rec):
        got[t] = (decon.cpu().numpy().copy(), tmx.copy(), rec.copy())

    pol = batch.RegPolicy(reg_mode=3, reg_choice=2, aff_method=1, it_limit=60)
    stack = batch.run_sharded([4, 5], lambda t: tuple(torch.from_numpy(v).cuda() for v in views[t]), pipe.fuse_pair, pol,
                              pipe.reg_fn, pipe.check_fn, (shape[1], shape[2]), sink=sink, device="cuda")
    pipe.close()
    assert stack.shape == (2, shape[1], shape[2])
    for t in (4, 5):
        a, b = views[t]
        b_reg, tmx, rec = api.reg3d(a, b, reg_choice=2, aff_method=1, it_limit=60)
        want, _ = api.decon_dualview(a, b_reg, psf_a, psf_b, 3)
        assert np.array_equal(got[t][1], tmx)
        assert got[t][2][5] == rec[5]  # same number of cost evaluations
        assert np.array_equal(got[t][0], want)


def test_overlapped_pipeline_equals_serial():
    """run_sharded(overlap=True) (read-ahead thread + RL/sink worker) returns the same stack, matrices and records as
    the serial order, and the sink sees the time points in order."""
    import torch
    from microimagelib_amd import batch

    shape = (32, 64, 64)
    psf_a = do.gaussian_psf((9, 9, 9), (2.0, 1.2, 1.2))
    psf_b = do.gaussian_psf((9, 9, 9), (1.2, 1.2, 2.0))
    tps = [1, 2, 3, 4]
    views = {t: (do.synthetic_view(shape, psf_a, seed=t, noise_seed=10 + t, n_blobs=12),
                 do.synthetic_view(shape, psf_b, seed=t, noise_seed=20 + t, n_blobs=12)) for t in tps}
    pol = batch.RegPolicy(reg_mode=3, reg_choice=2, aff_method=1, it_limit=60)
    runs = {}
    for overlap in (False, True):
        pipe = batch.FusionPipeline(shape, torch.from_numpy(psf_a).cuda(), torch.from_numpy(psf_b).cuda(), iters=3)
        seen = []

        def sink(t, decon, tmx, rec, seen=seen):
            seen.append((t, decon.cpu().numpy().copy(), tmx.copy(), rec.copy()))

        stack = batch.run_sharded(tps, lambda t: tuple(torch.from_numpy(v).cuda() for v in views[t]), pipe.fuse_pair, pol,
                                  pipe.reg_fn, pipe.check_fn, (shape[1], shape[2]), sink=sink, device="cuda",
                                  overlap=overlap)
        torch.cuda.synchronize()
        pipe.close()
        runs[overlap] = (stack.cpu().numpy().copy(), seen)
    assert [s[0] for s in runs[True][1]] == tps
    assert np.array_equal(runs[False][0], runs[True][0])
    for a, b in zip(runs[False][1], runs[True][1]):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert a[3][5] == b[3][5]


def test_pipeline_projection_products_match_mp2dgpu_and_mip3dgpu():
    """FusionPipeline(mips="all") leaves the five products of a time point in HBM (what the batch gathers over RCCL): the
    same bits as the host-pointer entry points mp2dgpu / mip3dgpu (include/libapi.h) give on the same volume."""
    import torch
    from microimagelib_amd import api, batch
    shape = (20, 36, 48)
    psf = torch.from_numpy(do.gaussian_psf((7, 7, 7), (1.5, 1.2, 1.2))).cuda()
    pipe = batch.FusionPipeline(shape, psf, psf, iters=1, mips="all", project_num=36)
    try:
        vol = torch.from_numpy(do.blob_volume(shape, n_blobs=10, seed=11)).cuda()
        got = pipe.project(vol)
        assert {k: tuple(v.shape) for k, v in got.items()} == pipe.mip_shape
        xy, yz, zx = api.mp2dgpu(vol.cpu().numpy())
        assert np.array_equal(got["xy"].cpu().numpy(), xy)
        assert np.array_equal(got["yz"].cpu().numpy(), yz)
        assert np.array_equal(got["zx"].cpu().numpy(), zx)
        for name, axis in (("mip3d_x", 1), ("mip3d_y", 2)):
            want = api.mip3dgpu(vol.cpu().numpy(), axis, 36)
            assert got[name].shape == want.shape and np.array_equal(got[name].cpu().numpy(), want), name
        # and through the gather of a one-rank world: time-ordered stacks per product
        stacks = batch.gather_mip_stack({7: got, 5: got}, [5, 7], pipe.mip_shape, device="cuda")
        assert all(torch.equal(stacks[k][0], got[k]) and torch.equal(stacks[k][1], got[k]) for k in got)
        assert batch.last_gather["bytes_per_time_point"] == 4 * sum(v.numel() for v in got.values())
    finally:
        pipe.close()
