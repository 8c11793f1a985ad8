#!/usr/bin/env python3
"""Which cost kernel is fastest for the matrices a registration really visits?  Times kernels 7 (plain, z), 10 (plain, y),
9 (row pairs + reuse) -- the chunked forms that run by default since round 5 -- and 4 (row pairs, no reuse, tiled) and the
library's own per-matrix choice on matrices sampled from the
full-size trajectory fixture (tests/golden/fullsize_reg_traj_full.npz) and from bench.py's optimisation, 512 x 512 x 256.
Evidence for the thresholds of pick_kernel (csrc/corr_padded.hip); not a test."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from microimagelib_amd import device as mdev

shape = (256, 512, 512)
g = torch.Generator(device="cuda").manual_seed(0)
tgt = torch.rand(shape, device="cuda", generator=g)
src = torch.rand(shape, device="cuda", generator=g)
pad = mdev.PaddedSource(src)


def rates(m):
    fd = lambda v: abs(v - round(v))
    ph = lambda v: min(64.0, 1.0 + 64.0 * fd(v))
    px, py, pz = ph(m[0]), ph(m[4]), ph(m[8])
    rz = abs(m[2]) * px + abs(m[6]) * py + abs(m[10] - 1.0) * pz
    ry = abs(m[1]) * px + abs(m[5] - 1.0) * py + abs(m[9]) * pz
    return rz, ry


mats = []
tr = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_reg_traj_full.npz"))["trace"][:, :12]
idx = np.linspace(0, len(tr) - 1, int(os.environ.get("TUNE_N", "160"))).astype(int)
mats += [tr[i] for i in idx]
# bench.py's optimum and single-parameter perturbations of it (what Powell's coordinate directions evaluate)
c, s = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
lin = 1.02 * np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
centre = np.array([256.0, 256.0, 128.0])
base = np.concatenate([lin, (centre - lin @ centre + np.array([5.0, -3.0, 2.0]))[:, None]], axis=1).astype(np.float32).reshape(12)
mats.append(base)
rng = np.random.default_rng(0)
for k in range(12):
    for d in (0.003, -0.01, 0.03):
        m = base.copy()
        m[k] += d * (50 if k in (3, 7, 11) else 1)
        mats.append(m)
# the batch's matrices (bench.py drift_matrix: 2 deg about y + shift, no scale) and perturbations; pure translations / scales
import bench
for t in (0, 5, 15):
    b2 = bench.drift_matrix(t, shape)
    mats.append(b2)
    for k in range(12):
        for d in (0.003, -0.01):
            m = b2.copy()
            m[k] += d * (50 if k in (3, 7, 11) else 1)
            mats.append(m)
ident = np.eye(3, 4, dtype=np.float32).reshape(12)
for sh in ((0, 0, 0), (2.5, -1.25, 0.75), (-7, 3, 1)):
    m = ident.copy(); m[[3, 7, 11]] = sh; mats.append(m)
    for sc in (0.98, 1.01):
        m2 = m.copy(); m2[[0, 5, 10]] = sc; mats.append(m2)
n = 4
rows = []
for m in mats:
    t = {}
    for k in (-1, 7, 10, 9, 4):
        mdev.corr_padded_force_kernel(k)
        mdev.corr_sums(tgt, pad, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            mdev.corr_sums(tgt, pad, m)
        t[k] = (time.perf_counter() - t0) / n * 1e3
    rz, ry = rates([float(v) for v in m])
    rows.append((rz, ry, t, [float(v) for v in m]))
mdev.corr_padded_force_kernel(-1)
auto = np.mean([r[2][-1] for r in rows])
best = np.mean([min(r[2][k] for k in (7, 10, 9, 4)) for r in rows])
print(f"{len(rows)} matrices: per-matrix choice {auto:.4f} ms, best of (7, 10, 9, 4) per matrix {best:.4f} ms, "
      f"always 7: {np.mean([r[2][7] for r in rows]):.4f}, always 10: {np.mean([r[2][10] for r in rows]):.4f}, "
      f"always 9: {np.mean([r[2][9] for r in rows]):.4f}, always 4: {np.mean([r[2][4] for r in rows]):.4f}")
if os.environ.get("TUNE_JSON"):
    json.dump([{"m": r[3], "rz": r[0], "ry": r[1], "t": {str(k): v for k, v in r[2].items()}} for r in rows], open(os.environ["TUNE_JSON"], "w"))
print("  rz      ry     auto    k7     k10    k9     k4   best")
for rz, ry, t, _ in sorted(rows, key=lambda r: min(r[0], r[1])):
    b = min((7, 10, 9, 4), key=lambda k: t[k])
    print(f"{rz:6.3f} {ry:6.3f}  {t[-1]:.3f}  {t[7]:.3f}  {t[10]:.3f}  {t[9]:.3f}  {t[4]:.3f}   k{b}")
