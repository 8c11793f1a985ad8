#!/usr/bin/env python3
"""RL iterations/s of both transform backends over a range of FFT boxes (incl. the 64*k sizes of
snapTransformSize), and their agreement on the result.  Evidence for DESIGN.md, not a test."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from microimagelib_amd import device as mdev

def gaussian_psf(shape, sig):
    g = [np.exp(-0.5 * ((np.arange(s) - s // 2) / sg) ** 2) for s, sg in zip(shape, sig)]
    return (g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]).astype(np.float32)

BOXES = [(64, 64, 64), (128, 128, 128), (128, 256, 256), (256, 256, 256), (64, 512, 512), (128, 512, 512), (256, 512, 512), (64, 256, 512), (128, 256, 512), (256, 256, 512), (64, 128, 512), (128, 128, 512), (256, 128, 512), (512, 128, 512), (256, 512, 256), (128, 512, 256), (256, 512, 1024), (128, 512, 1024), (64, 512, 1024), (256, 256, 1024), (64, 1024, 1024), (128, 1024, 1024), (256, 1024, 1024), (256, 1024, 512), (512, 512, 512), (192, 192, 192), (256, 384, 384), (128, 320, 448), (256, 576, 640), (512, 1024, 1024),
         (64, 192, 320), (64, 448, 576), (128, 576, 640), (256, 384, 512), (192, 384, 640), (320, 384, 448), (128, 768, 768), (512, 256, 512)]
for arg in sys.argv[1:]:  # --boxes=nz,ny,nx;nz,ny,nx...  (default: the list above)
    if arg.startswith("--boxes="):
        BOXES = [tuple(int(v) for v in b.split(",")) for b in arg[len("--boxes="):].split(";")]
rows = []
SPAN = 0.15  # seconds per timed run: long enough for the clocks to settle, and the three engines are timed in two interleaved rounds


def timed(eng, E, A, otf, bp, n):
    eng.sync()
    t0 = time.perf_counter()
    eng.run_single(E, A, otf, bp, n)
    eng.sync()
    return n / (time.perf_counter() - t0)


for box in BOXES:
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.rand(box, device="cuda", generator=g) * 1000 + 10
    psf = torch.from_numpy(gaussian_psf((31, 31, 31), (3.0, 1.5, 1.5))).cuda()
    engs = {}
    for name, backend in (("fused", 2), ("fused_classic", 2), ("rocfft", 1)):
        # "fused" = the library's default for this box (two passes per convolution where csrc/fft2p*.hip cover the box);
        # "fused_classic" = the four-kernel sequence forced (MIL_FX_TWOPASS=0)
        if name == "fused_classic":
            os.environ["MIL_FX_TWOPASS"] = "0"
        else:
            os.environ.pop("MIL_FX_TWOPASS", None)
        eng = mdev.DeconEngine(box, backend=backend)
        if name == "fused":
            two_pass = eng.two_pass
        if name == "fused_classic" and two_pass == 0:
            eng.close()
            continue
        otf, bp = eng.gen_otf(psf), eng.gen_otf(psf, flip=True)
        A = img.clone(); E = torch.empty_like(A)
        eng.init_single(A, E); eng.run_single(E, A, otf, bp, 2); eng.sync()
        out = E.clone()
        engs[name] = [eng, otf, bp, A, E, out, 0.0]
    os.environ.pop("MIL_FX_TWOPASS", None)
    for rnd in range(3):  # round 0: estimate + clock warm-up; rounds 1, 2: timed, best kept
        for name, st in engs.items():
            eng, otf, bp, A, E = st[:5]
            n = 10 if rnd == 0 else max(10, int(SPAN * st[6]))
            r = timed(eng, E, A, otf, bp, n)
            st[6] = r if rnd < 2 else max(st[6], r)
    res = {name: (st[6], st[5]) for name, st in engs.items()}
    if "fused_classic" not in res:
        res["fused_classic"] = res["fused"]
    for st in engs.values():
        st[0].close()
    del engs
    a, b = res["fused"][1], res["rocfft"][1]
    rel = float(((a - b).abs() / torch.clamp(b.abs(), min=float(b.abs().mean()))).max())
    vox = box[0] * box[1] * box[2]
    rows.append({"box_zyx": box, "two_pass": int(two_pass), "fused_it_s": round(res["fused"][0], 1),
                 "fused_classic_it_s": round(res["fused_classic"][0], 1), "rocfft_it_s": round(res["rocfft"][0], 1),
                 "speedup": round(res["fused"][0] / res["rocfft"][0], 2),
                 "fused_GB_s_algorithmic": round(res["fused"][0] * (21 * 4 * vox) / 1e9, 0), "max_rel_diff": rel})
    print(rows[-1], flush=True)
    del res, a, b, img
    torch.cuda.empty_cache()
json.dump(rows, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "bench_sizes.json"), "w"), indent=1)
