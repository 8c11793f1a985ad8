#!/usr/bin/env python3
"""A/B of one run-time switch of the fused RL backend inside ONE process (switches read per launch, e.g. MIL_FX_YZ_PERSIST):
  python tools/ab_env.py VAR VALUE_A VALUE_B [NZ NY NX]...      (default boxes: the benchmark box and four others)
Both settings timed three times, interleaved, `iters` iterations each between HIP events; results compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from microimagelib_amd import device as mdev

var, va, vb = sys.argv[1:4]
rest = [int(v) for v in sys.argv[4:]]
boxes = [tuple(rest[i:i + 3]) for i in range(0, len(rest), 3)] or [(256, 512, 512), (256, 512, 256), (256, 512, 1024), (512, 256, 512),
                                                                   (256, 448, 384), (128, 512, 512)]
ax = [np.exp(-0.5 * ((np.arange(31) - 15) / s) ** 2) for s in (3.0, 1.5, 1.5)]
psf = torch.from_numpy((ax[0][:, None, None] * ax[1][None, :, None] * ax[2][None, None, :]).astype(np.float32)).cuda()
for box in boxes:
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.rand(box, device="cuda", generator=g) * 1000 + 10
    eng = mdev.DeconEngine(box, backend=2)
    otf, bp = eng.gen_otf(psf), eng.gen_otf(psf, flip=True)
    A = img.clone(); E = torch.empty_like(A)
    vol_mb = 4 * box[0] * box[1] * box[2] / 2**20
    iters = max(20, int(100 * 256 / vol_mb))
    res, outs = {va: [], vb: []}, {}
    for rep in range(3):
        for v in (va, vb):
            os.environ[var] = v
            eng.init_single(A, E); eng.run_single(E, A, otf, bp, 10)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(eng.stream); eng.run_single(E, A, otf, bp, iters); e1.record(eng.stream); eng.sync()
            res[v].append(iters / (e0.elapsed_time(e1) * 1e-3))
            outs[v] = E.clone()
    same = bool(torch.equal(outs[va], outs[vb]))
    a, b = max(res[va]), max(res[vb])
    print(f"box {box} two_pass {eng.two_pass}: {var}={va} {a:8.1f} it/s {[round(x) for x in res[va]]} | {var}={vb} {b:8.1f} it/s "
          f"{[round(x) for x in res[vb]]} | B/A {b / a:.4f} | same bits {same}", flush=True)
    eng.close()
    os.environ.pop(var, None)
