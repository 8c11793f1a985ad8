#!/usr/bin/env python3
"""Two-pass kernels against the classic sequence on the boxes of tests/test_gpu_fft.py (ANY_BOXES by default): max relative
difference of 3 single-view iterations per MIL_FX_TWOPASS mode (1 yz kernel only, 2 xy kernel only, 3 both) and the
iterations/s of modes 0 and 3.  A diagnostic (prints, never asserts).

    python tools/twopass_any_check.py [--boxes=nz,ny,nx;nz,ny,nx...] [--iters=10]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from microimagelib_amd import device as mdev
from tests import fullsize_inputs as fi
from tests.test_gpu_fft import ANY_BOXES

opts = dict(a.lstrip("-").split("=") for a in sys.argv[1:] if "=" in a)
boxes = [tuple(int(v) for v in b.split(",")) for b in opts["boxes"].split(";")] if "boxes" in opts else ANY_BOXES
iters = int(opts.get("iters", "10"))
pa = torch.from_numpy(fi.psf_a()).cuda()
for shape in boxes:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand(shape, device="cuda", generator=g) * 1000 + 10
    outs, rate = {}, {}
    for mode in (0, 1, 2, 3):
        os.environ["MIL_FX_TWOPASS"] = str(mode)
        eng = mdev.DeconEngine(shape, backend=mdev.DeconEngine.FUSED)
        if eng.two_pass != mode:
            print(shape, "mode", mode, "not available (two_pass =", eng.two_pass, ")", flush=True)
            eng.close()
            continue
        oa, ba = eng.gen_otf(pa), eng.gen_otf(pa, flip=True)
        a = A.clone()
        E = torch.empty_like(a)
        eng.init_single(a, E)
        eng.run_single(E, a, oa, ba, 3)
        eng.sync()
        outs[mode] = E.clone()
        if mode in (0, 3):
            eng.init_single(a, E)
            eng.run_single(E, a, oa, ba, 2)
            eng.sync()
            t0 = time.perf_counter()
            eng.run_single(E, a, oa, ba, iters)
            eng.sync()
            rate[mode] = iters / (time.perf_counter() - t0)
        eng.close()
        del oa, ba
    line = [f"{shape}"]
    if 0 in outs:
        want = outs[0].double()
        floor = float(want.abs().mean())
        for mode in (1, 2, 3):
            if mode in outs:
                got = outs[mode].double()
                err = float(((got - want).abs() / torch.clamp(want.abs(), min=floor)).max()) if bool(torch.isfinite(got).all()) else float("nan")
                line.append(f"mode {mode}: {err:.2e}")
    if 0 in rate and 3 in rate:
        line.append(f"classic {rate[0]:.1f} it/s, two passes {rate[3]:.1f} it/s, x{rate[3] / rate[0]:.2f}")
    print("  ".join(line), flush=True)
    del A, outs
    torch.cuda.empty_cache()
