"""In-kernel phase stamps of the two-pass kernels (csrc/fft2p.hip, mil_fx_set_stamps): where a workgroup's time goes.
Usage: MIL_FX_TWOPASS=3 python tools/twopass_stamps.py [--shape=nz,ny,nx]   (a -DMIL_FX_STAMPS build: tools/build_variant.sh;
stamps exist in the compile-time kernels of fft2p.hip only: nx, ny powers of two)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import fullsize_inputs as fi  # noqa: E402
from microimagelib_amd import _lib, api  # noqa: E402

lib = _lib.load()
buf = torch.zeros(2 * 2048 * 16, dtype=torch.int64, device="cuda")
shape = fi.FULL_SHAPE
for arg in sys.argv[1:]:
    if arg.startswith("--shape="):
        shape = tuple(int(v) for v in arg[len("--shape="):].split(","))
img = fi.hash_volume(shape, 1)
psf = fi.psf_a()
api.decon_singleview(img, psf, 3, gpu_mem_mode=1)  # warm
lib.mil_fx_set_stamps.argtypes = [ctypes.c_void_p]
lib.mil_fx_set_stamps(ctypes.c_void_p(buf.data_ptr()))
api.decon_singleview(img, psf, 3, gpu_mem_mode=1)
torch.cuda.synchronize()
lib.mil_fx_set_stamps(ctypes.c_void_p(0))
st = buf.cpu().numpy().reshape(2, 2048, 16)
for name, s, n in (("yz_conv (regular tiles)", st[0], 2048), ("xy, ratio launch", st[1], 2048)):
    s = s[:n].astype(np.int64)
    s = s[s[:, 6] > 0]  # workgroups that stamped (the packed-column blocks of yz_conv do not)
    if s.shape[0] == 0:
        print(name, "no stamps (this box runs a kernel without them)")
        continue
    t0 = s[:, 0].min()
    d = np.diff(s[:, :7], axis=1) / 100.0  # us
    print(name, "kernel span %.1f us" % ((s[:, 6].max() - t0) / 100.0))
    print("  mean phase us:", np.round(d.mean(axis=0), 2), " total per WG %.1f" % ((s[:, 6] - s[:, 0]).mean() / 100.0))
    first = (s[:, 0] - t0) < 500
    print("  first round:  ", np.round(d[first].mean(axis=0), 2), " later rounds:", np.round(d[~first].mean(axis=0), 2))
    print("  first-round WGs (start < 5 us):", int(((s[:, 0] - t0) < 500).sum()))
    if name.startswith("yz"):
        seq = [1, 8, 9, 10, 11, 12, 13, 2]
        dd = np.diff(s[:, seq], axis=1) / 100.0
        print("  quarter 0 (park + write-out, barrier wait, read + y/z butterflies, OTF product, butterflies back + write, barrier, read back):",
              np.round(dd.mean(axis=0), 2))
    else:
        seq = [1, 8, 9, 10, 11, 12, 13, 2]
        dd = np.diff(s[:, seq], axis=1) / 100.0
        print("  trip 0 detail (fold, barrier, x^-1 L4+L32, middle + prefetch, barrier + x L32+L4, split, barrier):", np.round(dd.mean(axis=0), 2))
    order = np.argsort(s[:, 0])
    print("  starts (us) quantiles:", np.round(np.quantile((s[:, 0] - t0) / 100.0, [0, 0.25, 0.5, 0.75, 1.0]), 1))
