#!/usr/bin/env python3
"""Where does the full-size pipeline test's deconvolved volume differ from the oracle's, and do the two transform backends
differ from each other by as much?  (tests/golden/fullsize_pipeline.npz, time point 0)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from microimagelib_amd import device as mdev
from tests import fullsize_inputs as fi
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize_pipeline.npz"))
shape = fi.FULL_SHAPE
idx_np = fi.sample_indices(int(np.prod(shape)))
idx = torch.from_numpy(idx_np).cuda()
A0 = torch.from_numpy(fi.hash_volume(shape, 1)).cuda()
src = torch.from_numpy(fi.displaced_source(shape, 3)).cuda()
Breg = mdev.affine_resample(src, g["tmx"], shape)
print("resampled view equals the fixture:", np.array_equal(Breg.reshape(-1)[idx].cpu().numpy(), g["tp0_reg_samples"]))
outs = {}
for backend in (2, 1):
    eng = mdev.DeconEngine(shape, backend=backend)
    pa, pb = torch.from_numpy(fi.psf_a()).cuda(), torch.from_numpy(fi.psf_b()).cuda()
    oa, ba, ob, bb = eng.gen_otf(pa), eng.gen_otf(pa, flip=True), eng.gen_otf(pb), eng.gen_otf(pb, flip=True)
    A, B = A0.clone(), Breg.clone()
    E = torch.empty_like(A)
    eng.init_dual(A, B, E)
    eng.run_dual(E, A, B, oa, ob, ba, bb, int(g["iters"]))
    eng.sync()
    outs[backend] = E.reshape(-1)[idx].cpu().numpy().astype(np.float64)
    eng.close()
want = g["tp0_dv_samples"].astype(np.float64)
floor = float(g["tp0_dv_mean_abs"])
breg = g["tp0_reg_samples"]
for name, a, b in (("fused vs oracle", outs[2], want), ("rocfft vs oracle", outs[1], want), ("fused vs rocfft", outs[2], outs[1])):
    d = np.abs(a - b)
    rel = d / np.maximum(np.abs(b), floor)
    k = int(np.argmax(rel))
    z, rem = divmod(int(idx_np[k]), shape[1] * shape[2])
    y, x = divmod(rem, shape[2])
    print(f"{name}: max {rel.max():.2e} at sample {k} (x {x}, y {y}, z {z}): values {a[k]:.4f} / {b[k]:.4f}, view B there {breg[k]:.2f}; "
          f"99.9th percentile {np.percentile(rel, 99.9):.2e}, median {np.median(rel):.2e}; samples above 1e-5: {(rel > 1e-5).sum()}")
    zero = breg == 0
    print(f"    where view B is zero ({zero.sum()} samples): max {rel[zero].max():.2e}; elsewhere: max {rel[~zero].max():.2e}")
