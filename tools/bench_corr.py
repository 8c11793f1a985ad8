#!/usr/bin/env python3
"""Micro-benchmark of the registration cost kernels at 512x512x256: the border-padded form reg3d evaluates
(mil_corr_sums_padded, default) or, with --general, the general kernel (mil_corr_sums)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from microimagelib_amd import device as mdev

shape = (256, 512, 512)
g = torch.Generator(device="cuda").manual_seed(0)
tgt = torch.rand(shape, device="cuda", generator=g)
src = torch.rand(shape, device="cuda", generator=g)
general = "--general" in sys.argv
source = src if general else mdev.PaddedSource(src)
for a in sys.argv:
    if a.startswith("--kernel="):  # 0 plain + reuse | 1 row pairs | 2 row pairs + reuse, +3: two voxels per lane (default: per matrix)
        mdev.corr_padded_force_kernel(int(a.split("=")[1]))
print("kernel:", "corr_kernel (general)" if general else "padded-source kernels " + " ".join(sys.argv[1:]))

def matrix(deg, scale, shift):
    c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
    lin = scale * np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    centre = np.array([256.0, 256.0, 128.0])
    t = centre - lin @ centre + np.array(shift)
    return np.concatenate([lin, t[:, None]], axis=1).astype(np.float32).reshape(12)

cases = [("identity", matrix(0, 1, (0, 0, 0))), ("rot3_s1.02", matrix(3, 1.02, (5, -3, 2))),
         ("rot10", matrix(10, 1.0, (0, 0, 0))), ("rot45", matrix(45, 1.0, (0, 0, 0)))]
if "--sweep" in sys.argv:
    cases = [(f"rot{d}", matrix(d, 1.0, (0, 0, 0))) for d in (0.1, 0.25, 0.5, 1, 2, 5, 20)] + \
            [(f"scale{s}", matrix(0, s, (0, 0, 0))) for s in (1.002, 1.01, 1.05)]
for name, m in cases:
    ss, st = mdev.corr_sums(tgt, source, m)
    torch.cuda.synchronize()
    n = int(os.environ.get('BENCH_CORR_N', '50'))
    t0 = time.perf_counter()
    for _ in range(n):
        r = mdev.corr_sums(tgt, source, m)
    dt = (time.perf_counter() - t0) / n
    vox = tgt.numel()
    print(f"{name:12s} {dt*1e3:8.3f} ms/eval  {vox/dt/1e9:8.1f} Gvoxel/s  {2*4*vox/dt/1e9:8.1f} GB/s algorithmic  ss={float(ss).hex()} st={float(st).hex()}")
