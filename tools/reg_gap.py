#!/usr/bin/env python3
"""How much of a registration is NOT cost-kernel time?  bench.py's reg3D workload (affMethod 7) once; run under
`rocprofv3 --kernel-trace` and compare the sum of the cost kernels' durations with records[6] (optimisation seconds):
the difference per launched evaluation is the host round trip (publish -> poll -> Powell -> launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from microimagelib_amd import device as mdev
SHAPE = bench.SHAPE
img = bench.synth_volume(torch, SHAPE, (6.0, 2.5, 2.5), seed=100)
c, s = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
lin = 1.02 * np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
centre = np.array([SHAPE[2] / 2, SHAPE[1] / 2, SHAPE[0] / 2])
known = np.concatenate([lin, (centre - lin @ centre + np.array([5.0, -3.0, 2.0]))[:, None]], axis=1).astype(np.float32).reshape(12)
src = mdev.affine_resample(img, known, SHAPE)
torch.cuda.synchronize()
reg, tmx, rec = mdev.reg3d_affine(img, src, aff_method=7, ftol=1e-4, it_limit=3000)
l = mdev.reg3d_last_cost_launches()
print(f"REG evaluations {rec[5]:.0f} launches {l} optimisation_s {float(rec[6]):.6f} us_per_launch {1e6 * float(rec[6]) / l:.2f}")
