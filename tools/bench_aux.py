#!/usr/bin/env python3
"""The kernels either side of the hot path at 512x512x256, a few launches each -- run under
`rocprofv3 --kernel-trace --stats` (tools/gpu_profile_run_r04.sh) so that profiles/<round>/aux_kernel_stats.csv carries their
names and durations next to the figures bench.py reports in extra.auxiliary_kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json

import torch

import bench
from microimagelib_amd import device as mdev

img = bench.synth_volume(torch, bench.SHAPE, (6.0, 2.5, 2.5), seed=100)
psf = torch.from_numpy(bench.gaussian_psf(bench.PSF_SHAPE, (6.0, 2.5, 2.5))).cuda()
eng = mdev.DeconEngine(bench.SHAPE)
print(json.dumps(bench.auxiliary_kernels(torch, mdev, eng, img, psf)))
eng.close()
