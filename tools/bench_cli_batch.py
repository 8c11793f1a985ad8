#!/usr/bin/env python3
"""End-to-end throughput of the spimFusionBatch command-line tool on TIFF files, 512x512x256, 16-bit in / 16-bit out:
our main (microimagelib_amd/cli, read-ahead + write-behind) against the reference's own main (oracle/_ref/cli,
strictly sequential I/O), both on the same libapi.so.

    python tools/bench_cli_batch.py [--time-points 4] [--reg-mode 2] [--workdir /tmp/mil_cli_batch]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time-points", type=int, default=4)
    ap.add_argument("--reg-mode", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--workdir", default="/tmp/mil_cli_batch")
    ap.add_argument("--distinct", type=int, default=0,
                    help="write only this many distinct input pairs and symlink the other time points to them (0 = all "
                         "distinct): lets a long batch (64 time points = 17 GB of stacks) fit a scratch disk")
    ap.add_argument("--workers", default="",
                    help="also run our main with this device list (argument 34, e.g. '0,0' = two worker threads sharing "
                         "device 0; '0,1,2,3' on a multi-GPU node) and compare its stacks with the one-worker run")
    ap.add_argument("--skip-reference", action="store_true", help="do not run the reference's main")
    ap.add_argument("--repeat-workers", type=int, default=1, help="run the --workers configuration this many times")
    args = ap.parse_args()

    import torch
    import bench
    from microimagelib_amd import api, device as mdev

    w = args.workdir
    shutil.rmtree(w, ignore_errors=True)
    os.makedirs(w + "/SPIMA")
    os.makedirs(w + "/SPIMB")
    sig_a, sig_b = (6.0, 2.5, 2.5), (2.5, 2.5, 6.0)
    api.writetifstack(w + "/pa.tif", bench.gaussian_psf(bench.PSF_SHAPE, sig_a), 32)
    api.writetifstack(w + "/pb.tif", bench.gaussian_psf(bench.PSF_SHAPE, sig_b), 32)
    base_a = bench.synth_volume(torch, bench.SHAPE, sig_a, seed=7)
    c, s = np.cos(np.deg2rad(2.0)), np.sin(np.deg2rad(2.0))
    lin = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    centre = np.array([256.0, 256.0, 128.0])
    known = np.concatenate([lin, (centre - lin @ centre + np.array([3.0, -2.0, 1.0]))[:, None]], axis=1).astype(np.float32)
    base_b = mdev.affine_resample(bench.synth_volume(torch, bench.SHAPE, sig_b, seed=7), known.reshape(12), bench.SHAPE)
    for t in range(args.time_points):
        if args.distinct and t >= args.distinct:
            for name in ("SPIMA", "SPIMB"):
                os.symlink(f"{w}/{name}/{name}_{t % args.distinct}.tif", f"{w}/{name}/{name}_{t}.tif")
            continue
        g = torch.Generator(device="cuda").manual_seed(1000 + t)
        for name, base in (("SPIMA", base_a), ("SPIMB", base_b)):
            vol = (base + torch.randn(base.shape, device="cuda", generator=g) * 3.0).clamp_(min=0)
            api.writetifstack(f"{w}/{name}/{name}_{t}.tif", vol.cpu().numpy(), 16)
    del base_a, base_b
    torch.cuda.empty_cache()

    def cli_args(out, devices="0"):
        return [out + "/", w + "/SPIMA/", w + "/SPIMB/", "SPIMA_", "SPIMB_", "0", str(args.time_points - 1), "1", "0",
                "1", "1", "1", "1", "1", "1", str(args.reg_mode), "0", "0", "none", "0.0001", "3000", "0", "0",
                w + "/pa.tif", w + "/pb.tif", str(args.iters), "0", "0", "1", "0", "0", "16", "0", devices]

    res = {}
    ours_exe = os.path.join(ROOT, "microimagelib_amd", "bin", "spimFusionBatch")
    for name, exe, env in (("reference main, sequential I/O", os.path.join(ROOT, "oracle", "_ref", "cli", "spimFusionBatch"), {}),
                           ("our main, host-pointer calls (MIL_BATCH_PATH=host), read-ahead + write-behind, one worker", ours_exe, {"MIL_BATCH_PATH": "host", "MIL_BATCH_WORKERS_PER_DEVICE": "1"}),
                           ("our main, time point resident in HBM, read-ahead + write-behind, one worker", ours_exe, {"MIL_BATCH_WORKERS_PER_DEVICE": "1"}),
                           ("our main, time point resident in HBM, read-ahead + write-behind, two workers (the default)", ours_exe, {})):
        if not os.path.exists(exe) or (args.skip_reference and "reference" in name):
            res[name] = None
            continue
        out = w + "/out_" + ("ref" if "reference" in name else ("host" if "MIL_BATCH_PATH" in env else ("one" if env else "ours")))
        t0 = time.perf_counter()
        r = subprocess.run([exe] + cli_args(out), capture_output=True, text=True, timeout=1800, env=dict(os.environ, **env))
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            raise SystemExit(r.stdout[-1500:] + r.stderr[-1500:])
        if os.environ.get("MIL_BENCH_CLI_LOG"):  # per-phase seconds the tools print for every time point
            print(name, "\n".join(l for l in r.stdout.splitlines() if "ime cost" in l)[-1500:], file=sys.stderr)
        res[name] = {"seconds": round(dt, 3), "time_points_per_s": round(args.time_points / dt, 3)}
    same_workers = None
    for rep in range(args.repeat_workers if args.workers else 0):
        exe = os.path.join(ROOT, "microimagelib_amd", "bin", "spimFusionBatch")
        out = w + "/out_workers"
        shutil.rmtree(out, ignore_errors=True)
        t0 = time.perf_counter()
        r = subprocess.run([exe] + cli_args(out, args.workers), capture_output=True, text=True, timeout=1800)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            raise SystemExit(r.stdout[-1500:] + r.stderr[-1500:])
        res[f"our main, worker threads on devices {args.workers}" + (f" (run {rep + 1})" if args.repeat_workers > 1 else "")] = {
            "seconds": round(dt, 3), "time_points_per_s": round(args.time_points / dt, 3)}
        same_workers = same_workers is not False
        for t in range(args.time_points):
            same_tmx = open(f"{w}/out_workers/TMX/Matrix_{t}.tmx").read() == open(f"{w}/out_ours/TMX/Matrix_{t}.tmx").read()
            same_tif = open(f"{w}/out_workers/Decon/Decon_{t}.tif", "rb").read() == open(f"{w}/out_ours/Decon/Decon_{t}.tif", "rb").read()
            if not (same_tmx and same_tif):
                same_workers = False
                a, b = api.readtifstack(f"{w}/out_workers/Decon/Decon_{t}.tif"), api.readtifstack(f"{w}/out_ours/Decon/Decon_{t}.tif")
                print(f"run {rep + 1}, time point {t}: matrix file {'same' if same_tmx else 'DIFFERS'}, decon stack {'same' if same_tif else 'DIFFERS'} "
                      f"(max abs diff {float(np.abs(a - b).max())}, {int((a != b).sum())} voxels)", file=sys.stderr)
                if not same_tmx:
                    print(open(f"{w}/out_workers/TMX/Matrix_{t}.tmx").read(), open(f"{w}/out_ours/TMX/Matrix_{t}.tmx").read(), file=sys.stderr)
                    print("stderr of the tool:", r.stderr[-3000:], file=sys.stderr)
                    print("\n".join(l for l in r.stdout.splitlines() if "rror" in l or "libapi" in l or "fail" in l)[-3000:], file=sys.stderr)
    same = None
    if all(v for k, v in res.items() if "worker" not in k) and not args.skip_reference:
        same = all(open(f"{w}/out_ref/Decon/Decon_{t}.tif", "rb").read() == open(f"{w}/out_ours/Decon/Decon_{t}.tif", "rb").read() ==
                   open(f"{w}/out_host/Decon/Decon_{t}.tif", "rb").read() for t in range(args.time_points))
    # the device-resident pipeline (batch.run_sharded + FusionPipeline, what bench.py's config-5 figure runs) on THE SAME
    # quantised stacks, read once and kept in HBM: what the tool could reach if file I/O cost nothing
    if args.reg_mode in (0, 3):
        from microimagelib_amd import batch
        pairs = {t: (torch.from_numpy(api.readtifstack(f"{w}/SPIMA/SPIMA_{t}.tif")).cuda(),
                     torch.from_numpy(api.readtifstack(f"{w}/SPIMB/SPIMB_{t}.tif")).cuda()) for t in range(args.time_points)}
        psf_a = torch.from_numpy(bench.gaussian_psf(bench.PSF_SHAPE, sig_a)).cuda()
        psf_b = torch.from_numpy(bench.gaussian_psf(bench.PSF_SHAPE, sig_b)).cuda()
        pipes = [batch.FusionPipeline(bench.SHAPE, psf_a, psf_b, iters=args.iters, mips="xy") for _ in range(2)]
        pipe = pipes[0]
        pol = lambda: batch.RegPolicy(reg_mode=args.reg_mode, reg_choice=2, aff_method=6, ftol=1e-4, it_limit=3000)
        common = dict(load_pair=lambda t: pairs[t], policy=None, reg_fn=pipe.reg_fn, check_fn=pipe.check_fn, mip_shape=pipe.mip_shape,
                      device="cuda")
        runs = {"one time point in flight (deconvolution of i overlapped with the registration of i + 1)":
                    lambda tps: batch.run_sharded(tps, fuse_pair=pipe.fuse_pair, overlap=True, **dict(common, policy=pol())),
                "two time points in flight (two host threads, as the tool's two workers)":
                    lambda tps: batch.run_sharded(tps, fuse_pair=[q.fuse_pair for q in pipes], lanes=2, **dict(common, policy=pol()))}
        for label, run in runs.items():
            run([0, 1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(list(range(args.time_points)))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["device-resident pipeline on the same quantised stacks, no file I/O, " + label] = {
                "seconds": round(dt, 3), "time_points_per_s": round(args.time_points / dt, 3)}
        for q in pipes:
            q.close()
        del pairs
    print(json.dumps({"workload": f"spimFusionBatch CLI, {args.time_points} time points 512x512x256, 16-bit TIFF in/out, "
                                  f"regMode {args.reg_mode}, {args.iters} RL iterations, process start to exit"
                                  + (f" ({args.distinct} distinct input pairs, the rest symlinked)" if args.distinct else ""),
                      "results": res, "identical_decon_stacks": same,
                      "worker_threads_identical_to_one_worker": same_workers}))
    shutil.rmtree(w, ignore_errors=True)


if __name__ == "__main__":
    main()
