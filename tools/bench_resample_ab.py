#!/usr/bin/env python3
"""A/B of the resample and 3-D MIP kernels at 512x512x256 in one process: marching resample (csrc/resample_march.hip) against
the one-voxel kernel (MIL_RESAMPLE_MARCH=0), second against first fused MIP kernel (MIL_MIP3D_V2=0); ms per call (HIP events)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from microimagelib_amd import device as mdev

img = bench.synth_volume(torch, bench.SHAPE, (6.0, 2.5, 2.5), seed=100)
nz, ny, nx = img.shape


def matrix(deg, scale, shift, axis="y"):
    c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
    rot = {"y": [[c, 0, -s], [0, 1, 0], [s, 0, c]], "x": [[1, 0, 0], [0, c, -s], [0, s, c]], "z": [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[axis]
    lin = scale * np.array(rot)
    centre = np.array([nx / 2, ny / 2, nz / 2])
    t = centre - lin @ centre + np.array(shift)
    return np.concatenate([lin, t[:, None]], axis=1).astype(np.float32).reshape(12)


cases = [("identity", matrix(0, 1, (0, 0, 0))), ("rot3y_s1.02", matrix(3, 1.02, (5, -3, 2))), ("rot2y", matrix(2, 1.0, (3, -2, 1))),
         ("rot2x_s.99", matrix(2, 0.99, (3, -2, 1), "x")), ("rot10z", matrix(10, 1.0, (0, 0, 0), "z")), ("rot45y", matrix(45, 1.0, (0, 0, 0)))]
for rep in range(2):
    for env in (None, "0"):
        if env is None:
            os.environ.pop("MIL_RESAMPLE_MARCH", None)
        else:
            os.environ["MIL_RESAMPLE_MARCH"] = env
        for wgs in ((None,) if env == "0" else (None, "4", "5", "8")):
            if wgs is None:
                os.environ.pop("MIL_RESAMPLE_WGS_PER_CU", None)
            else:
                os.environ["MIL_RESAMPLE_WGS_PER_CU"] = wgs
            row = [bench.timed(torch, lambda m=m: mdev.affine_resample(img, m, img.shape), reps=10) * 1e3 for _, m in cases]
            print(f"resample {'march' if env is None else 'one-voxel':9s} wgs/cu={wgs}: " + " ".join(f"{n}={v:.4f}" for (n, _), v in zip(cases, row)), flush=True)
    os.environ.pop("MIL_RESAMPLE_WGS_PER_CU", None)
    for env in (None, "0"):
        if env is None:
            os.environ.pop("MIL_MIP3D_V2", None)
        else:
            os.environ["MIL_MIP3D_V2"] = env
        for axis in (1, 2):
            sec = bench.timed(torch, lambda: mdev.mip3d(img, axis, 36), reps=2)
            print(f"mip3d axis {axis} x36 {'v2' if env is None else 'v1'}: {sec * 1e3:.3f} ms", flush=True)
os.environ.pop("MIL_MIP3D_V2", None)
os.environ.pop("MIL_RESAMPLE_MARCH", None)
