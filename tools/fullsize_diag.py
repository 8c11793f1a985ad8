"""Diagnostic (GPU box): distance of both engine backends from the full-size oracle fixture, and from each other."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microimagelib_amd import device as dev
from tests import fullsize_inputs as fi

g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_decon.npz"))
shape = fi.FULL_SHAPE
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
idx = torch.from_numpy(fi.sample_indices(int(np.prod(shape)))).cuda()
res = {}
for backend in (2, 1):
    eng = dev.DeconEngine(shape, backend=backend)
    A, B = T(fi.hash_volume(shape, 1)), T(fi.hash_volume(shape, 2))
    pa, pb = T(fi.psf_a()), T(fi.psf_b())
    oa, ba, ob, bb = eng.gen_otf(pa), eng.gen_otf(pa, flip=True), eng.gen_otf(pb), eng.gen_otf(pb, flip=True)
    E = torch.empty_like(A)
    for tag in ("sv", "dv"):
        if tag == "sv":
            eng.init_single(A, E); eng.run_single(E, A, oa, ba, 20)
        else:
            eng.init_dual(A, B, E); eng.run_dual(E, A, B, oa, ob, ba, bb, 20)
        eng.sync()
        s = E.reshape(-1)[idx].cpu().numpy().astype(np.float64)
        planes = E.double().sum(dim=(1, 2)).cpu().numpy()
        want = g[f"{tag}_samples"].astype(np.float64)
        floor = float(g[f"{tag}_mean_abs"])
        err = np.max(np.abs(s - want) / np.maximum(np.abs(want), floor))
        perr = np.abs(planes - g[f"{tag}_planes"]) / np.abs(g[f"{tag}_planes"])
        tot = abs(planes.sum() - g[f"{tag}_planes"].sum()) / g[f"{tag}_planes"].sum()
        print(f"backend {backend} {tag}: sample err {err:.3e}  plane err max {perr.max():.3e} mean {perr.mean():.3e}  total flux err {tot:.3e}  min {float(E.min()):.4f} max {float(E.max()):.1f}")
        res[(backend, tag)] = (s, planes)
    eng.close()
for tag in ("sv", "dv"):
    s2, p2 = res[(2, tag)]; s1, p1 = res[(1, tag)]
    floor = float(g[f"{tag}_mean_abs"])
    print(tag, "fused vs rocfft: sample", np.max(np.abs(s2 - s1) / np.maximum(np.abs(s1), floor)), "plane", np.max(np.abs(p2 - p1) / p1))
