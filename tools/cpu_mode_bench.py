#!/usr/bin/env python3
"""gpuMemMode 0 (host RL path) timing at 512x512x256: one thread and N threads.  No GPU needed."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from microimagelib_amd import api  # noqa: E402
from tests import fullsize_inputs as fi  # noqa: E402

shape = fi.FULL_SHAPE
img, psf = fi.hash_volume(shape, 1), fi.psf_a()
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"), flush=True)
for th, it in ((1, 2), (int(sys.argv[1]) if len(sys.argv) > 1 else 8, 6)):
    os.environ["MIL_CPU_THREADS"] = str(th)
    t = time.time()
    out, rec = api.decon_singleview(img, psf, it, gpu_mem_mode=0)
    print(f"{th} thread(s): loop {rec[8]:.2f} s -> {it / rec[8]:.3f} it/s; prepare {rec[7]:.2f} s; whole call {time.time() - t:.2f} s", flush=True)
