#!/usr/bin/env python3
"""RL iterations/s of the fused backend on one box:  python tools/bench_box.py NZ NY NX [iters]  (environment switches
such as MIL_FX_TWOPASS / MIL_FX_YZ_NARROW apply).  Prints one line; checks the result against MIL_FX_TWOPASS=0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from microimagelib_amd import device as mdev

box = tuple(int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
g = torch.Generator(device="cuda").manual_seed(1)
img = torch.rand(box, device="cuda", generator=g) * 1000 + 10
ax = [np.exp(-0.5 * ((np.arange(31) - 15) / s) ** 2) for s in (3.0, 1.5, 1.5)]
psf = torch.from_numpy((ax[0][:, None, None] * ax[1][None, :, None] * ax[2][None, None, :]).astype(np.float32)).cuda()


def run(n):
    eng = mdev.DeconEngine(box, backend=2)
    otf, bp = eng.gen_otf(psf), eng.gen_otf(psf, flip=True)
    A = img.clone(); E = torch.empty_like(A)
    eng.init_single(A, E); eng.run_single(E, A, otf, bp, 60); eng.sync()
    eng.init_single(A, E); eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_single(E, A, otf, bp, n); eng.sync()
    dt = time.perf_counter() - t0
    tp = eng.two_pass
    eng.close()
    return n / dt, E, tp

rate, E, tp = run(iters)
saved = os.environ.get("MIL_FX_TWOPASS")
os.environ["MIL_FX_TWOPASS"] = "0"
_, ref, _ = run(iters)
err = float(((E - ref).abs() / torch.clamp(ref.abs(), min=float(ref.abs().mean()))).max())
print(f"box {box} two_pass {tp} narrow {os.environ.get('MIL_FX_YZ_NARROW', '0')}: {rate:8.1f} it/s  ({1e3 / rate:.4f} ms/it)  max rel diff vs classic {err:.2e}")
