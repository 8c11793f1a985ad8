"""Soak of the RL engine (512 x 512 x 256, or --shape=nz,ny,nx): 2 x 5000 single-view iterations on the same inputs, finite
and bit-identical.   Usage (GPU box): python tools/soak_rl.py [--shape=256,384,384]"""
import os
import sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fullsize_inputs as fi
from microimagelib_amd import device as dev
shape = fi.FULL_SHAPE
for arg in sys.argv[1:]:
    if arg.startswith("--shape="):
        shape = tuple(int(v) for v in arg[len("--shape="):].split(","))
print("shape", shape)
A = torch.from_numpy(fi.hash_volume(shape, 1)).cuda(); pa = torch.from_numpy(fi.psf_a()).cuda()
eng = dev.DeconEngine(shape)
print("two_pass", eng.two_pass)
oa, ba = eng.gen_otf(pa), eng.gen_otf(pa, flip=True)
a = A.clone(); E = torch.empty_like(a); eng.init_single(a, E)
t = time.time(); eng.run_single(E, a, oa, ba, 5000); eng.sync(); dt = time.time() - t
print("5000 iterations in %.2f s = %.0f it/s, finite %s, mean %.4f" % (dt, 5000 / dt, bool(torch.isfinite(E).all()), float(E.mean())))
E2 = torch.empty_like(a); a2 = A.clone(); eng.init_single(a2, E2); eng.run_single(E2, a2, oa, ba, 5000); eng.sync()
print("repeat identical:", bool(torch.equal(E, E2)))
