#!/usr/bin/env python3
"""Soak of the registration path at 512 x 512 x 256 (bench.py's reg3D workload) and at a ragged 250 x 500 x 500: N registrations of
the same pair must return the same matrix, the same registered volume and the same number of evaluations, bit for bit -- the cost
kernels' partial sums are a function of the matrix and the (static) grid alone; then 300 marching resamples with random matrices on
two streams against the one-voxel kernel.   Usage (GPU box): python tools/soak_reg.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from microimagelib_amd import device as mdev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for shape in (bench.SHAPE, (250, 500, 500)):
    target = bench.synth_volume(torch, shape, (6.0, 2.5, 2.5), seed=100)
    c, s = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
    lin = 1.02 * np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    centre = np.array([shape[2] / 2, shape[1] / 2, shape[0] / 2])
    known = np.concatenate([lin, (centre - lin @ centre + np.array([5.0, -3.0, 2.0]))[:, None]], axis=1).astype(np.float32).reshape(12)
    source = mdev.affine_resample(target, known, shape)
    torch.cuda.synchronize()
    first, t0, evals = None, time.time(), 0
    same = True
    for i in range(n):
        reg, tmx, rec = mdev.reg3d_affine(target, source, aff_method=7, ftol=1e-4, it_limit=3000)
        torch.cuda.synchronize()
        evals += int(rec[5])
        if first is None:
            first = (reg.clone(), np.array(tmx).copy(), float(rec[5]), float(rec[3]))
        else:
            same = same and torch.equal(reg.view(torch.int32), first[0].view(torch.int32)) and \
                np.array_equal(np.asarray(tmx).view(np.uint32), first[1].view(np.uint32)) and float(rec[5]) == first[2]
    dt = time.time() - t0
    print(f"shape {shape}: {n} registrations in {dt:.2f} s ({evals} evaluations, {evals / dt:.0f} /s incl. set-up), final ZNCC {first[3]:.5f}, "
          f"{int(first[2])} evaluations each, all identical: {same}", flush=True)
    del reg, source, target
# resamples: marching kernel on two streams against the one-voxel kernel
rng = np.random.default_rng(3)
shape = (128, 256, 384)
src = torch.rand(shape, device="cuda")
ident = np.eye(3, 4, dtype=np.float32).reshape(12)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
for i in range(300):
    m = ident + rng.normal(0, 0.02 * (i % 5), 12).astype(np.float32)
    m[[3, 7, 11]] = rng.uniform(-6, 6, 3).astype(np.float32)
    os.environ["MIL_RESAMPLE_MARCH"] = "0"
    want = mdev.affine_resample(src, m, shape)
    os.environ.pop("MIL_RESAMPLE_MARCH")
    torch.cuda.synchronize()
    with torch.cuda.stream(s1 if i % 2 else s2):
        got = mdev.affine_resample(src, m, shape)
    torch.cuda.synchronize()
    bad += int(not torch.equal(got.view(torch.int32), want.view(torch.int32)))
print(f"300 marching resamples on two streams against the one-voxel kernel: {300 - bad} identical, {bad} different")
