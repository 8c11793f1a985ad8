"""Two-pass RL step (csrc/fft2p.hip) against the classic kernel sequence at 512 x 512 x 256: same inputs, MIL_FX_TWOPASS =
0 / 1 / 2 / 3 in separate processes, compared on the full output.  Usage: python tools/twopass_check.py [iterations]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(mode, iters, out):
    import fullsize_inputs as fi
    from microimagelib_amd import api
    img = fi.hash_volume(fi.FULL_SHAPE, 1)
    psf = fi.psf_a()
    res, rec = api.decon_singleview(img, psf, iters, gpu_mem_mode=1)
    np.save(out, res)
    print("mode", mode, "done", float(res.mean()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    outs = {}
    for mode in (0, 1, 2, 3):
        path = "/tmp/twopass_%d.npy" % mode
        env = dict(os.environ, MIL_FX_TWOPASS=str(mode), MIL_DECON_CACHE="0")
        r = subprocess.run([sys.executable, __file__, "--child", str(mode), str(iters), path], env=env, capture_output=True, text=True)
        print(r.stdout.strip()[-300:], r.stderr.strip()[-600:])
        if r.returncode != 0:
            print("mode", mode, "FAILED rc", r.returncode)
            continue
        outs[mode] = np.load(path)
    ref = outs.get(0)
    for mode, o in outs.items():
        if mode == 0 or ref is None:
            continue
        d = np.abs(o - ref)
        rel = d / np.maximum(np.abs(ref), float(np.mean(np.abs(ref))))
        i = int(np.argmax(rel))
        print("mode %d vs classic: max rel %.3e at %s, mean rel %.3e, nan %d" % (mode, float(rel.max()), np.unravel_index(i, ref.shape),
                                                                         float(rel.mean()), int(np.isnan(o).sum())))
