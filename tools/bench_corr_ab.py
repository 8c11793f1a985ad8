#!/usr/bin/env python3
"""A/B of the registration cost kernels in ONE process at 512x512x256: every forced kernel (tiled 3-6, chunked 7-10) on
the four standard matrices, the chunked ones at several workgroups-per-CU settings, interleaved twice.  Prints ms per
evaluation (launch + poll included) and the relative difference of the two sums to kernel 0's.

    python tools/bench_corr_ab.py [--kernels=3,7,6,10] [--wgs=4,5,6] [--n=40]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from microimagelib_amd import device as mdev

opts = dict(a.lstrip("-").split("=") for a in sys.argv[1:] if "=" in a)
kernels = [int(k) for k in opts.get("kernels", "3,7,6,10,5,9").split(",")]
wgs = [int(k) for k in opts.get("wgs", "5").split(",")]
n = int(opts.get("n", "40"))
shape = tuple(int(v) for v in opts.get("shape", "256,512,512").split(","))

g = torch.Generator(device="cuda").manual_seed(0)
tgt = torch.rand(shape, device="cuda", generator=g)
src = torch.rand(shape, device="cuda", generator=g)
pad = mdev.PaddedSource(src)


def matrix(deg, scale, shift, axis="y"):
    c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
    rot = {"y": [[c, 0, -s], [0, 1, 0], [s, 0, c]], "x": [[1, 0, 0], [0, c, -s], [0, s, c]], "z": [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[axis]
    lin = scale * np.array(rot)
    centre = np.array([shape[2] / 2, shape[1] / 2, shape[0] / 2])
    t = centre - lin @ centre + np.array(shift)
    return np.concatenate([lin, t[:, None]], axis=1).astype(np.float32).reshape(12)


cases = [("identity", matrix(0, 1, (0, 0, 0))), ("rot3_s1.02", matrix(3, 1.02, (5, -3, 2))), ("rot2y", matrix(2, 1.0, (3, -2, 1))),
         ("rot2x_s.99", matrix(2, 0.99, (3, -2, 1), "x")), ("rot10", matrix(10, 1.0, (0, 0, 0)))]

mdev.corr_padded_force_kernel(0)
ref = {name: mdev.corr_sums(tgt, pad, m) for name, m in cases}


def run(kern, per_cu):
    mdev.corr_set_chunk_wgs_per_cu(per_cu or 0)
    mdev.corr_padded_force_kernel(kern)
    out = []
    for name, m in cases:
        ss, st = mdev.corr_sums(tgt, pad, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            mdev.corr_sums(tgt, pad, m)
        dt = (time.perf_counter() - t0) / n
        rss, rst = ref[name]
        err = max(abs(ss - rss), abs(st - rst)) / max(abs(rss), 1e-300)
        out.append((dt * 1e3, err))
    return out


print("ms per evaluation (max relative difference of the two sums to kernel 0)")
print(f"{'kernel':>14s} " + " ".join(f"{name:>20s}" for name, _ in cases))
for rep in range(2):
    for k in kernels:
        for w in (wgs if k >= 7 else [None]):
            res = run(k, w)
            tag = f"{k}" + (f" wgs/cu={w}" if w is not None else "")
            print(f"{tag:>14s} " + " ".join(f"{ms:10.4f} ({err:7.1e})" for ms, err in res), flush=True)
mdev.corr_padded_force_kernel(-1)
