#!/usr/bin/env python3
"""RL iterations/s with ONE and with TWO (THREE) independent 512x512x256 volumes in flight on one GPU -- each on its own engine and
stream, as the batch driver runs time points.  A single volume's four launches per iteration depend on each other; two volumes
let the memory-pattern-bound yz kernel of one overlap the latency-bound xy kernel of the other."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from microimagelib_amd import device as mdev

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
psf = torch.from_numpy(bench.gaussian_psf(bench.PSF_SHAPE, (6.0, 2.5, 2.5))).cuda()
lanes = []
for k in range(3):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng = mdev.DeconEngine(bench.SHAPE, stream=st, backend=2)
        otf, bp = eng.gen_otf(psf), eng.gen_otf(psf, flip=True)
        A = bench.synth_volume(torch, bench.SHAPE, (6.0, 2.5, 2.5), seed=100 + k).clone()
        E = torch.empty_like(A)
        eng.init_single(A, E)
        eng.run_single(E, A, otf, bp, 60)
        eng.sync()
    lanes.append((st, eng, otf, bp, A, E))
torch.cuda.synchronize()
for n in (1, 2, 3, 1, 2, 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st, eng, otf, bp, A, E in lanes[:n]:
        with torch.cuda.stream(st):
            eng.run_single(E, A, otf, bp, iters)
    for st, eng, *_ in lanes[:n]:
        eng.sync()
    dt = time.perf_counter() - t0
    print(f"{n} volume(s) in flight: {n * iters / dt:8.1f} iterations/s in total ({iters / dt:7.1f} per volume)", flush=True)
for st, eng, *_ in lanes:
    eng.close()
