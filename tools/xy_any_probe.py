#!/usr/bin/env python3
"""Per-kernel GB/s of the two-pass kernels on a list of boxes (run under rocprofv3 --kernel-trace --stats by tools/prof_any.sh
style wrappers, or alone for wall-clock it/s): mode 3 only, MIL_FX_ANY from the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from microimagelib_amd import device as mdev
from tests import fullsize_inputs as fi
opts = dict(a.lstrip("-").split("=") for a in sys.argv[1:] if "=" in a)
boxes = [tuple(int(v) for v in b.split(",")) for b in opts["boxes"].split(";")]
iters = int(opts.get("iters", "20"))
pa = torch.from_numpy(fi.psf_a()).cuda()
for shape in boxes:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand(shape, device="cuda", generator=g) * 1000 + 10
    eng = mdev.DeconEngine(shape, backend=mdev.DeconEngine.FUSED)
    oa, ba = eng.gen_otf(pa), eng.gen_otf(pa, flip=True)
    E = torch.empty_like(A)
    eng.init_single(A, E); eng.run_single(E, A, oa, ba, 3); eng.sync()
    t0 = time.perf_counter(); eng.run_single(E, A, oa, ba, iters); eng.sync()
    print(shape, "two_pass", eng.two_pass, f"{iters / (time.perf_counter() - t0):.1f} it/s", flush=True)
    eng.close()
