#!/usr/bin/env python3
"""Many back-to-back marching resamples (six matrices, 512x512x256) -- run under `rocprofv3 --kernel-trace` to see whether a
single LAUNCH is ever slow (the sweep's rare 2 - 4 ms averages), or the host between launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from microimagelib_amd import device as mdev
img = bench.synth_volume(torch, bench.SHAPE, (6.0, 2.5, 2.5), seed=100)
nz, ny, nx = img.shape
def matrix(deg, scale, shift, axis="y"):
    c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
    rot = {"y": [[c, 0, -s], [0, 1, 0], [s, 0, c]], "x": [[1, 0, 0], [0, c, -s], [0, s, c]], "z": [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[axis]
    lin = scale * np.array(rot); centre = np.array([nx / 2, ny / 2, nz / 2])
    return np.concatenate([lin, (centre - lin @ centre + np.array(shift))[:, None]], axis=1).astype(np.float32).reshape(12)
mats = [matrix(0, 1, (0, 0, 0)), matrix(3, 1.02, (5, -3, 2)), matrix(10, 1.0, (0, 0, 0), "z"), matrix(45, 1.0, (0, 0, 0))]
worst = []
for rep in range(40):
    for m in mats:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            mdev.affine_resample(img, m, img.shape)
        torch.cuda.synchronize()
        worst.append(((time.perf_counter() - t0) / 20 * 1e3, mats.index(m) if False else len(worst) % len(mats), rep))
for k in range(len(mats)):
    g = sorted(w[0] for w in worst if w[1] == k)
    slow = [(round(w[0], 3), w[2]) for w in worst if w[1] == k and w[0] > 1.5 * g[len(g) // 2]]
    print("matrix %d: ms per call over groups of 20: median %.4f, max %.4f; slow groups (ms, repetition): %s" % (k, g[len(g) // 2], g[-1], slow))
