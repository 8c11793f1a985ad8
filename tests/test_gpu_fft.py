"""GPU parity of the fused transform backend (csrc/fft.hip) against NumPy's pocketfft:
forward transforms + private ordering (through the OTF), the full fused convolution, and
agreement of the two engine backends on the RL loop."""
import numpy as np
import pytest

from oracle import decon_oracle as do

pytestmark = pytest.mark.gpu
f32 = np.float32

SHAPES = [(16, 16, 16), (32, 16, 64), (16, 64, 32), (64, 32, 128), (128, 128, 128), (32, 256, 512), (256, 64, 16),
          # the 64*k boxes of snapTransformSize: radices 3, 5, 7, 11, 13 on every axis
          (16, 192, 320), (192, 16, 384), (320, 64, 192), (64, 448, 448), (16, 704, 576), (576, 16, 704),
          (16, 832, 640), (896, 16, 832), (16, 960, 896), (640, 16, 960), (16, 768, 768), (768, 32, 64),
          # every remaining instantiation of csrc/fft.hip's dispatch tables, in y AND z position: col_kernel<512>
          # (the benchmark's y pass), <384>, <1024>, y 320 / 576 / 640 / 896, z 448 / 704 / 832 / 960 ...
          (16, 512, 32), (512, 16, 32), (16, 384, 32), (384, 16, 32), (16, 1024, 32), (1024, 16, 32),
          (16, 320, 64), (16, 576, 32), (16, 640, 32), (16, 896, 32),
          (448, 16, 32), (704, 16, 32), (832, 16, 32), (960, 16, 32),
          # ... and x_kernel<128>, <512>, <1024> (nx = 256, 1024, 2048)
          (16, 16, 256), (16, 16, 1024), (16, 16, 2048),
          # BASELINE.json configs 2-4: the benchmark box itself against a float64 DFT
          (256, 512, 512)]


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from microimagelib_amd import device
    return device


def T(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_transform_matches_rfftn(dev, shape):
    """OTF of a box-sized random PSF == rfftn of the normalised, origin-shifted PSF (oracle gen_otf)."""
    rng = np.random.default_rng(sum(shape))
    psf = rng.uniform(0.0, 1.0, shape).astype(f32)
    eng = dev.DeconEngine(shape, backend=dev.DeconEngine.FUSED)
    assert eng.backend == dev.DeconEngine.FUSED
    for flip in (False, True):
        got = eng.otf_to_natural(eng.gen_otf(T(psf), flip=flip))
        eng.sync()
        want = do.gen_otf(do.flip3(psf) if flip else psf, shape)
        scale = float(np.max(np.abs(want)))
        assert float(np.max(np.abs(got.cpu().numpy() - want))) / scale < 2e-6
    eng.close()


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_convolution_matches_numpy(dev, shape):
    rng = np.random.default_rng(7 + sum(shape))
    vol = rng.uniform(0.0, 100.0, shape).astype(f32)
    psf = do.gaussian_psf((9, 9, 9), (2.0, 1.0, 1.5))
    eng = dev.DeconEngine(shape, backend=dev.DeconEngine.FUSED)
    otf = eng.gen_otf(T(psf))
    got = eng.convolve(T(vol), otf)
    eng.sync()
    want = np.fft.irfftn(np.fft.rfftn(vol.astype(np.float64)) * do.gen_otf(psf, shape).astype(np.complex128),
                         s=shape, axes=(0, 1, 2)) * vol.size
    err = float(np.max(np.abs(got.cpu().numpy() - want))) / float(np.max(np.abs(want)))
    eng.close()
    assert err < 2e-6


def test_backends_agree_on_rl(dev):
    import torch
    shape = (64, 64, 128)
    psf = do.gaussian_psf((17, 17, 17), (3.0, 1.5, 1.5))
    img = do.synthetic_view(shape, psf, seed=3, noise_seed=4, n_blobs=20)
    outs = []
    for backend in (dev.DeconEngine.ROCFFT, dev.DeconEngine.FUSED):
        eng = dev.DeconEngine(shape, backend=backend)
        A = T(img)
        E = torch.empty_like(A)
        otf, bp = eng.gen_otf(T(psf)), eng.gen_otf(T(psf), flip=True)
        eng.init_single(A, E)
        eng.run_single(E, A, otf, bp, 10)
        eng.sync()
        outs.append(E.cpu().numpy())
        eng.close()
    want = do.decon_singleview(img, psf, 10)
    floor = float(np.mean(np.abs(want)))
    for o in outs:
        assert float(np.max(np.abs(o - want) / np.maximum(np.abs(want), floor))) < 1e-5


def test_backends_agree_at_benchmark_size(dev):
    """20 single-view and 20 dual-view RL iterations at 512 x 512 x 256 (BASELINE.json configs 2, 3): the hand-written
    transforms (csrc/fft.hip) against the rocFFT backend on the same closed-form inputs, 1e-5 relative."""
    import torch
    from tests import fullsize_inputs as fi
    shape = fi.FULL_SHAPE
    A, B = T(fi.hash_volume(shape, 1)), T(fi.hash_volume(shape, 2))
    pa, pb = T(fi.psf_a()), T(fi.psf_b())
    outs = {}
    for backend in (dev.DeconEngine.ROCFFT, dev.DeconEngine.FUSED):
        eng = dev.DeconEngine(shape, backend=backend)
        assert eng.backend == backend
        oa, ba = eng.gen_otf(pa), eng.gen_otf(pa, flip=True)
        ob, bb = eng.gen_otf(pb), eng.gen_otf(pb, flip=True)
        a, b = A.clone(), B.clone()
        E = torch.empty_like(a)
        eng.init_single(a, E)
        eng.run_single(E, a, oa, ba, 20)
        eng.sync()
        sv = E.clone()
        a = A.clone()
        eng.init_dual(a, b, E)
        eng.run_dual(E, a, b, oa, ob, ba, bb, 20)
        eng.sync()
        outs[backend] = (sv, E.clone())
        eng.close()
        del oa, ba, ob, bb
    for k in (0, 1):
        want, got = outs[dev.DeconEngine.ROCFFT][k].double(), outs[dev.DeconEngine.FUSED][k].double()
        floor = float(want.abs().mean())
        err = float(((got - want).abs() / torch.clamp(want.abs(), min=floor)).max())
        assert err < 1e-5, ("single" if k == 0 else "dual", err)


_YZ = [(128, 1024), (64, 1024),                           # (nz, ny): y tail T = ny / 64 = 16 (8-, 16-column yz tiles)
       (256, 512), (128, 512), (64, 512),                 # T = 8
       (256, 256), (128, 256), (64, 256),                 # T = 4
       (512, 128), (256, 128), (128, 128), (64, 128)]     # T = 2
TWO_PASS_BOXES = [(nz, ny, nx) for nx in (512, 256, 1024) for (nz, ny) in _YZ]  # (nz, ny, nx)
# Boxes of csrc/fft2p_any_*.hip (the 64 k extents snapTransformSize produces, reference src/api_subfunc.cu:57-87): every
# nx = 192 .. 1024 (x transforms of 16 x 6 .. 16 x 32, half waves idle for nx / 2 = 96, 160, 224 ...), every y tail
# T = ny / 64 = 2 .. 16 (one- and two-stage tails), every nz = 64 .. 1024 in multiples of 64 (radices 3 .. 13 in z), 16- and
# 8-, 4- and 2-column yz tiles up to T x nz = 4096, powers of two mixed with the others.
ANY_BOXES = [(256, 384, 384), (128, 320, 448),            # the two boxes of the round-4 review
             (64, 192, 192), (192, 192, 320), (64, 448, 576), (128, 576, 640), (192, 640, 704), (128, 704, 768),
             (64, 768, 832), (128, 832, 896), (64, 896, 960), (64, 960, 1024),
             (320, 384, 256), (384, 320, 512), (512, 256, 512), (1024, 128, 256), (256, 512, 384), (128, 1024, 320),
             (448, 256, 192), (640, 192, 384), (576, 128, 448),
             (704, 128, 192), (768, 128, 192), (832, 128, 192), (896, 128, 192), (960, 128, 192),
             (384, 384, 384), (256, 576, 320), (256, 1024, 256), (512, 512, 192), (320, 640, 256)]  # 2-column yz tiles


@pytest.mark.parametrize("shape", TWO_PASS_BOXES + ANY_BOXES, ids=lambda s: "x".join(str(v) for v in s[::-1]))
def test_two_pass_kernels_agree_with_the_classic_sequence(dev, monkeypatch, shape):
    """csrc/fft2p.hip on every box it covers (nx = 256 / 512 / 1024 = x transforms of 16 x 8 / 16 / 32; ny = 1024 / 512 / 256 /
    128 = y tails of 16 / 8 / 4 / 2 rows; every nz whose yz tile fits the LDS, plus ... x 512 x 256 on register tiles) and a
    sample of the boxes of csrc/fft2p_any_*.hip that touches every instantiated nx, y tail and nz: the yz kernel
    alone (MIL_FX_TWOPASS=1, partial classic y
    passes around it), the xy kernel alone (2) and both (3, the default on these boxes) against the classic four-kernel
    half update (0) -- 3 single-view and 2 dual-view iterations on the same inputs, the last one through the
    no-forward-transform form of the x kernel.
    Same positions, twiddles and OTFs; only the order of the float operations differs."""
    import torch
    from tests import fullsize_inputs as fi
    A, B = T(fi.hash_volume(shape, 1)), T(fi.hash_volume(shape, 2))
    pa, pb = T(fi.psf_a()), T(fi.psf_b())
    outs = {}
    for mode in (0, 1, 2, 3, 13):  # 13: mode 3 once more -- the kernels have no atomics, so the same bits are expected
        monkeypatch.setenv("MIL_FX_TWOPASS", str(mode % 10))
        eng = dev.DeconEngine(shape, backend=dev.DeconEngine.FUSED)
        assert eng.two_pass == mode % 10
        oa, ba = eng.gen_otf(pa), eng.gen_otf(pa, flip=True)
        ob, bb = eng.gen_otf(pb), eng.gen_otf(pb, flip=True)
        a, b = A.clone(), B.clone()
        E = torch.empty_like(a)
        eng.init_single(a, E)
        eng.run_single(E, a, oa, ba, 3)
        eng.sync()
        sv = E.clone()
        a = A.clone()
        eng.init_dual(a, b, E)
        eng.run_dual(E, a, b, oa, ob, ba, bb, 2)
        eng.sync()
        outs[mode] = (sv, E.clone())
        eng.close()
        del oa, ba, ob, bb
    monkeypatch.delenv("MIL_FX_TWOPASS")
    eng = dev.DeconEngine(shape)
    assert eng.two_pass == 3  # the default where the box allows it
    eng.close()
    eng = dev.DeconEngine((64, 128, 128))
    assert eng.backend == dev.DeconEngine.FUSED and eng.two_pass == 0  # other boxes keep the classic sequence
    eng.close()
    assert torch.equal(outs[3][0], outs[13][0]) and torch.equal(outs[3][1], outs[13][1])
    for mode in (1, 2, 3):
        for k in (0, 1):
            want, got = outs[0][k].double(), outs[mode][k].double()
            assert bool(torch.isfinite(got).all())
            floor = float(want.abs().mean())
            err = float(((got - want).abs() / torch.clamp(want.abs(), min=floor)).max())
            assert err < 3e-6, (mode, "single" if k == 0 else "dual", err)


@pytest.mark.parametrize("nx", [256, 512, 1024])
def test_register_yz_kernel_agrees_with_the_lds_kernel(dev, monkeypatch, nx):
    """ny = 512, nz = 256 with nx a power of two has a second yz kernel: register tiles of 8 rows x 8 columns x all z
    (csrc/fft2p.hip yz_conv_kernel, the default of rounds 2 - 5, MIL_FX_YZ_REGISTER=1 since the LDS kernel overtook it).  Same
    positions and twiddles, OTFs tiled 8 instead of 4 columns wide: 3 single-view + 2 dual-view iterations must agree to 3e-6."""
    import torch
    from tests import fullsize_inputs as fi
    shape = (256, 512, nx)
    A, B = T(fi.hash_volume(shape, 1)), T(fi.hash_volume(shape, 2))
    pa, pb = T(fi.psf_a()), T(fi.psf_b())
    outs = {}
    for reg in ("0", "1"):
        monkeypatch.setenv("MIL_FX_YZ_REGISTER", reg)
        eng = dev.DeconEngine(shape, backend=dev.DeconEngine.FUSED)
        assert eng.two_pass == 3
        oa, ba = eng.gen_otf(pa), eng.gen_otf(pa, flip=True)
        ob, bb = eng.gen_otf(pb), eng.gen_otf(pb, flip=True)
        a, b = A.clone(), B.clone()
        E = torch.empty_like(a)
        eng.init_single(a, E)
        eng.run_single(E, a, oa, ba, 3)
        eng.sync()
        sv = E.clone()
        a = A.clone()
        eng.init_dual(a, b, E)
        eng.run_dual(E, a, b, oa, ob, ba, bb, 2)
        eng.sync()
        outs[reg] = (sv, E.clone())
        eng.close()
        del oa, ba, ob, bb
    monkeypatch.delenv("MIL_FX_YZ_REGISTER")
    for k in (0, 1):
        want, got = outs["0"][k].double(), outs["1"][k].double()
        assert bool(torch.isfinite(got).all())
        floor = float(want.abs().mean())
        err = float(((got - want).abs() / torch.clamp(want.abs(), min=floor)).max())
        assert err < 3e-6, (nx, "single" if k == 0 else "dual", err)


def test_snap_sized_box_runs_rl_on_the_fused_backend(dev):
    """A 50x180x300 stack snaps to 64x192x320 (radices 3 and 5): fused backend, parity with the oracle."""
    import torch
    shape = (64, 192, 320)
    psf = do.gaussian_psf((15, 15, 15), (3.0, 1.5, 1.5))
    img = do.synthetic_view(shape, psf, seed=5, noise_seed=6, n_blobs=20)
    eng = dev.DeconEngine(shape)
    assert eng.backend == dev.DeconEngine.FUSED
    A = T(img)
    E = torch.empty_like(A)
    otf, bp = eng.gen_otf(T(psf)), eng.gen_otf(T(psf), flip=True)
    eng.init_single(A, E)
    eng.run_single(E, A, otf, bp, 8)
    eng.sync()
    got = E.cpu().numpy()
    eng.close()
    want = do.decon_singleview(img, psf, 8)
    floor = float(np.mean(np.abs(want)))
    assert float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor))) < 1e-5


def test_unsupported_box_uses_rocfft(dev):
    eng = dev.DeconEngine((32, 48, 192))
    assert eng.backend == dev.DeconEngine.ROCFFT
    eng.close()
    with pytest.raises(RuntimeError):
        dev.DeconEngine((32, 48, 192), backend=dev.DeconEngine.FUSED)
