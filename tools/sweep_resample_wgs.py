import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from microimagelib_amd import device as mdev
img = bench.synth_volume(torch, bench.SHAPE, (6.0, 2.5, 2.5), seed=100)
nz, ny, nx = img.shape
def matrix(deg, scale, shift, axis="y"):
    c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
    rot = {"y": [[c, 0, -s], [0, 1, 0], [s, 0, c]], "x": [[1, 0, 0], [0, c, -s], [0, s, c]], "z": [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[axis]
    lin = scale * np.array(rot); centre = np.array([nx / 2, ny / 2, nz / 2])
    t = centre - lin @ centre + np.array(shift)
    return np.concatenate([lin, t[:, None]], axis=1).astype(np.float32).reshape(12)
cases = [("identity", matrix(0, 1, (0, 0, 0))), ("rot3y_s1.02", matrix(3, 1.02, (5, -3, 2))), ("rot2y", matrix(2, 1.0, (3, -2, 1))),
         ("rot2x_s.99", matrix(2, 0.99, (3, -2, 1), "x")), ("rot10z", matrix(10, 1.0, (0, 0, 0), "z")), ("rot45y", matrix(45, 1.0, (0, 0, 0)))]
for rep in range(int(os.environ.get("SWEEP_REPS", "3"))):
    for wgs in os.environ.get("SWEEP_WGS", "None 6 8 10 12 16 24").split():
        wgs = None if wgs == "None" else wgs
        if wgs is None: os.environ.pop("MIL_RESAMPLE_WGS_PER_CU", None)
        else: os.environ["MIL_RESAMPLE_WGS_PER_CU"] = wgs
        row = [bench.timed(torch, lambda m=m: mdev.affine_resample(img, m, img.shape), reps=20) * 1e3 for _, m in cases]
        print(f"wgs/cu={wgs}: " + " ".join(f"{n}={v:.4f}" for (n, _), v in zip(cases, row)), flush=True)
