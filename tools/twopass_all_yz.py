#!/usr/bin/env python3
"""Every (ny, nz) instantiation of the LDS yz kernel (csrc/fft2p_any_yz.hip: 126 pairs with (ny / 64) * nz <= 4096) at nx = 192,
and every nx = 192 .. 1024 at (ny, nz) = (192, 64): 3 single-view iterations with both two-pass kernels against the classic
sequence, max relative difference per box.  A one-off sweep (the test suite samples this space); prints the worst box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from microimagelib_amd import device as mdev
from tests import fullsize_inputs as fi

pa = torch.from_numpy(fi.psf_a()).cuda()
boxes = [(nz, ny, 192) for ny in range(128, 1025, 64) for nz in range(64, 1025, 64) if (ny // 64) * nz <= 4096]
boxes += [(64, 192, nx) for nx in range(256, 1025, 64)]
worst = (0.0, None)
for i, shape in enumerate(boxes):
    g = torch.Generator(device="cuda").manual_seed(i)
    A = torch.rand(shape, device="cuda", generator=g) * 1000 + 10
    outs = {}
    for mode in (0, 3):
        os.environ["MIL_FX_TWOPASS"] = str(mode)
        eng = mdev.DeconEngine(shape, backend=mdev.DeconEngine.FUSED)
        assert eng.two_pass == mode, (shape, eng.two_pass)
        oa, ba = eng.gen_otf(pa), eng.gen_otf(pa, flip=True)
        a = A.clone(); E = torch.empty_like(a)
        eng.init_single(a, E); eng.run_single(E, a, oa, ba, 3); eng.sync()
        outs[mode] = E.double()
        eng.close()
    want, got = outs[0], outs[3]
    assert bool(torch.isfinite(got).all()), shape
    err = float(((got - want).abs() / torch.clamp(want.abs(), min=float(want.abs().mean()))).max())
    if err > worst[0]:
        worst = (err, shape)
    if err > 3e-6:
        print("ABOVE 3e-6:", shape, err, flush=True)
    del A, outs
print(f"{len(boxes)} boxes, worst max relative difference {worst[0]:.2e} at {worst[1]}")
