#!/usr/bin/env python3
"""ms per 36-angle 3-D MIP series (about x, about y) at 512x512x256, three repetitions -- for A/B runs over build/variants
(MIL_LIBAPI_PATH) of the fused resample + projection kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from microimagelib_amd import device as mdev
img = bench.synth_volume(torch, bench.SHAPE, (6.0, 2.5, 2.5), seed=100)
ref = {}
for rep in range(3):
    row = []
    for axis in (1, 2):
        sec = bench.timed(torch, lambda: mdev.mip3d(img, axis, 36), reps=3)
        row.append(sec * 1e3)
    print("mip3d x36: about x %.3f ms, about y %.3f ms" % tuple(row), flush=True)
print("checksum", float(mdev.mip3d(img, 1, 36).double().sum()), float(mdev.mip3d(img, 2, 36).double().sum()))
