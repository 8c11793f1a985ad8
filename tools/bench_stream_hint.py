#!/usr/bin/env python3
"""Non-temporal hints on the once-per-iteration streams (csrc/fft2p.hip Aux<STREAM>): RL iterations/s with MIL_FX_STREAM=0 and =1
on two-pass boxes of different volume sizes, interleaved, one process per measurement (tools/bench_box.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
boxes = [(128, 256, 256), (256, 256, 256), (64, 512, 512), (128, 512, 512), (256, 512, 256), (256, 512, 512), (128, 1024, 1024), (256, 512, 1024)]
for box in boxes:
    mb = 4 * box[0] * box[1] * box[2] / 2**20
    res = {"0": [], "1": []}
    for rep in range(3):
        for flag in ("0", "1"):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_box.py")] + [str(v) for v in box] + ["100"],
                                 capture_output=True, text=True, env=dict(os.environ, MIL_FX_STREAM=flag)).stdout.strip().splitlines()[-1]
            res[flag].append(float(out.split(":")[1].split("it/s")[0]))
    a, b = sorted(res["0"])[1], sorted(res["1"])[1]
    print(f"box {box} ({mb:.0f} MB per volume): plain {res['0']} -> median {a:.1f}; stream hints {res['1']} -> median {b:.1f}  ({100 * (b / a - 1):+.1f} %)", flush=True)
