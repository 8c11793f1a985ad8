#!/usr/bin/env python3
"""Timing of mil_affine_resample at 512x512x256 for the library MIL_LIBAPI_PATH points at (build/variants/...)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from microimagelib_amd import device as mdev
img = torch.rand(bench.SHAPE, device="cuda")
ident = np.eye(3, 4, dtype=np.float32).reshape(12)
sh = ident.copy(); sh[[3, 7, 11]] = (5.3, -3.2, 2.1)
for name, m in (("identity", ident), ("shift", sh)):
    for w in (None, "4", "5", "6"):
        if w: os.environ["MIL_RESAMPLE_WGS_PER_CU"] = w
        else: os.environ.pop("MIL_RESAMPLE_WGS_PER_CU", None)
        print(name, "wgs/cu", w, f"{bench.timed(torch, lambda: mdev.affine_resample(img, m, img.shape), reps=20) * 1e3:.4f} ms", flush=True)
