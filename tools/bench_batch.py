#!/usr/bin/env python3
"""BASELINE.json configs[4]: the spimFusionBatch loop at 512x512x256, one volume pair per GPU.

    python tools/bench_batch.py [--time-points 64] [--iters 20] [--reg-mode 3] [--host]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_batch.py ...

Every time point = register view B to view A (reg3d regChoice 2, affMethod 6 as the batch hard-codes,
src/spim_fusion_batch.cpp:557), joint dual-view RL, XY maximum-intensity projection; afterwards the MIP stack is
gathered to rank 0 (the only exchange).  Time points are dealt round-robin over the ranks (microimagelib_amd/batch.py).
Prints ONE JSON line: time points/s over all ranks, the per-stage split of rank 0 and the MIP-gather time.  With --host
the views start in pinned host memory and the result goes back to it (TIFF decode excluded).  With --tiff DIR every rank
READS its time points from 16-bit TIFF stacks on disk and WRITES the deconvolved 16-bit stack back inside the timed region,
through the library's own codec (readtifstack / writetifstack, the reference's I/O path: src/spim_fusion_batch.cpp:640-700,
:900-930) -- the honest form of config 5 on a shared host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time-points", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--reg-mode", type=int, default=3, choices=[0, 1, 2, 3])
    # functional test of the multi-rank path on a one-GPU box: all ranks on cuda:0, collectives over gloo
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--shape", type=int, nargs=3, default=[256, 512, 512], metavar=("NZ", "NY", "NX"))
    ap.add_argument("--host", action="store_true", help="views in pinned host memory, result copied back")
    ap.add_argument("--overlap", action="store_true",
                    help="deconvolve time point i on a helper thread while time point i + 1 is registered (regModes 0, 1, 3)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="time points in flight per GPU, one host thread and engine each (batch.run_sharded lanes=; "
                         "regModes 0, 1, 3; not with --tiff)")
    ap.add_argument("--tiff", metavar="DIR", default=None,
                    help="read both views of every time point from 16-bit TIFF stacks under DIR/rank<r>/ and write the "
                         "deconvolved stack there, inside the timed region (files are created before it)")
    ap.add_argument("--tiff-files", type=int, default=2, help="distinct input pairs per rank (cycled over the time points)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import bench  # synthetic data helpers of the headline bench
    from microimagelib_amd import batch, device as mdev

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench_batch.py needs a GPU")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    shape = tuple(args.shape)
    sig_a, sig_b = (6.0, 2.5, 2.5), (2.5, 2.5, 6.0)
    psf_a = torch.from_numpy(bench.gaussian_psf(bench.PSF_SHAPE, sig_a)).cuda()
    psf_b = torch.from_numpy(bench.gaussian_psf(bench.PSF_SHAPE, sig_b)).cuda()

    # one noiseless pair per rank; a time point = that pair with its own Poisson noise, view B displaced by a
    # known small affine (rotation 2 deg about y, shift) so the registration has work to do
    base_a = bench.synth_volume(torch, shape, sig_a, seed=7)
    base_b0 = bench.synth_volume(torch, shape, sig_b, seed=7)
    c, s = np.cos(np.deg2rad(2.0)), np.sin(np.deg2rad(2.0))
    lin = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    centre = np.array([shape[2] / 2, shape[1] / 2, shape[0] / 2])
    shift = centre - lin @ centre + np.array([3.0, -2.0, 1.0])
    known = np.concatenate([lin, shift[:, None]], axis=1).astype(np.float32).reshape(12)
    base_b = mdev.affine_resample(base_b0, known, shape)
    del base_b0

    if args.lanes > 1 and (args.tiff or args.overlap or args.reg_mode == 2):
        raise SystemExit("--lanes goes with views in HBM / pinned memory, regModes 0, 1, 3, without --overlap")
    pipes = [batch.FusionPipeline(shape, psf_a, psf_b, iters=args.iters) for _ in range(max(1, args.lanes))]
    pipe = pipes[0]
    split = {"load": 0.0, "register": 0.0, "fuse": 0.0, "store": 0.0}
    evals = []
    host_out = torch.empty(shape, dtype=torch.float32).pin_memory() if args.host else None
    host_pair = None
    if args.host:
        host_pair = (base_a.cpu().pin_memory(), base_b.cpu().pin_memory())

    tiff_dir = None
    if args.tiff:
        from microimagelib_amd import api
        tiff_dir = os.path.join(args.tiff, f"rank{rank}")
        os.makedirs(tiff_dir, exist_ok=True)
        for k in range(args.tiff_files):  # setup, not timed: noisy 16-bit stacks like the microscope writes
            g = torch.Generator(device="cuda").manual_seed(5000 + 97 * rank + k)
            for name, base in (("SPIMA", base_a), ("SPIMB", base_b)):
                vol = (base + torch.randn(base.shape, device="cuda", generator=g) * 3.0).clamp_(0, 65535)
                api.writetifstack(os.path.join(tiff_dir, f"{name}_{k}.tif"), vol.cpu().numpy(), bits=16)
        tiff_in = torch.empty((2, 2) + shape, dtype=torch.float32).pin_memory()  # two sets: --overlap reads one time point ahead
        tiff_out = torch.empty(shape, dtype=torch.float32).pin_memory()
        tiff_calls = [0]

    load_stream = torch.cuda.Stream()

    def dev_sync():  # per-phase seconds need a device-wide wait; with --overlap / --lanes that would serialise the threads
        if not args.overlap and args.lanes <= 1:
            torch.cuda.synchronize()

    def load_pair(t):
        t0 = time.perf_counter()
        g = torch.Generator(device="cuda").manual_seed(1000 + t)
        if tiff_dir:
            k = t % args.tiff_files
            buf = tiff_in[tiff_calls[0] % 2]
            tiff_calls[0] += 1
            for i, name in enumerate(("SPIMA", "SPIMB")):  # decode straight into pinned memory, then one DMA each
                api.readtifstack(os.path.join(tiff_dir, f"{name}_{k}.tif"), out=buf[i].numpy())
            with torch.cuda.stream(load_stream):
                a, b = buf[0].cuda(non_blocking=True), buf[1].cuda(non_blocking=True)
            load_stream.synchronize()
            split["load"] += time.perf_counter() - t0
            return a, b
        if args.host:
            a, b = host_pair[0].cuda(non_blocking=True), host_pair[1].cuda(non_blocking=True)
        else:
            a, b = base_a, base_b
        a = a + torch.randn(a.shape, device="cuda", generator=g) * 3.0
        b = b + torch.randn(b.shape, device="cuda", generator=g) * 3.0
        dev_sync()
        split["load"] += time.perf_counter() - t0
        return a, b

    def reg_fn(*a, **k):
        t0 = time.perf_counter()
        out = pipe.reg_fn(*a, **k)
        dev_sync()
        split["register"] += time.perf_counter() - t0
        evals.append(float(out[2][5]))
        return out

    def make_fuse(p):
        def fuse(a, b, t):
            t0 = time.perf_counter()
            out = p.fuse_pair(a, b, t)
            dev_sync()
            split["fuse"] += time.perf_counter() - t0
            return out
        return fuse
    fuse_pair = make_fuse(pipe) if args.lanes <= 1 else [make_fuse(p) for p in pipes]

    def sink(t, decon, tmx, rec):
        t0 = time.perf_counter()
        if tiff_dir:
            tiff_out.copy_(decon, non_blocking=True)
            torch.cuda.synchronize()
            api.writetifstack(os.path.join(tiff_dir, f"Decon_{t}.tif"), tiff_out.numpy(), bits=16)
        elif args.host:
            host_out.copy_(decon, non_blocking=True)
            torch.cuda.synchronize()
        split["store"] += time.perf_counter() - t0

    tps = list(range(args.time_points))
    policy = lambda: batch.RegPolicy(reg_mode=args.reg_mode, reg_choice=2, aff_method=6, ftol=1e-4, it_limit=3000,
                                     i_tmx=known if args.reg_mode == 0 else None)
    # warm-up: one time point per rank (plans, OTFs, clocks), not timed
    batch.run_sharded(tps[:world * max(1, args.lanes)], load_pair, fuse_pair, policy(), reg_fn, pipe.check_fn,
                      (shape[1], shape[2]), sink=sink, device=coll_dev, lanes=max(1, args.lanes))
    for k in split:
        split[k] = 0.0
    evals.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    gather_s = [0.0]
    real_gather = batch.gather_mip_stack

    def timed_gather(*a, **k):
        tg = time.perf_counter()
        out = real_gather(*a, **k)
        gather_s[0] = time.perf_counter() - tg  # includes waiting for the slowest rank
        return out
    batch.gather_mip_stack = timed_gather
    t0 = time.perf_counter()
    stack = batch.run_sharded(tps, load_pair, fuse_pair, policy(), reg_fn, pipe.check_fn, (shape[1], shape[2]),
                              sink=sink, device=coll_dev, overlap=args.overlap, lanes=max(1, args.lanes))
    torch.cuda.synchronize()
    batch.gather_mip_stack = real_gather
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n_local = len(batch.shard_time_points(tps, 0, world))
        print(json.dumps({
            "metric": "spimFusionBatch time points/s, 512x512x256 dual view (BASELINE.json configs[4])",
            "value": round(args.time_points / elapsed, 3), "unit": "time points/s", "n_gpus": world,
            "time_points": args.time_points, "rl_iterations": args.iters, "reg_mode": args.reg_mode,
            "elapsed_s": round(elapsed, 3), "overlap_deconvolution_with_next_registration": bool(args.overlap),
            "time_points_in_flight_per_gpu": max(1, args.lanes),
            "views": ("16-bit TIFF stacks on disk, read and written inside the timed region" if tiff_dir else
                      "pinned host memory" if args.host else "resident in HBM"),
            "rank0_mip_gather_s": round(gather_s[0], 4),
            "rank0_seconds_per_time_point": {k: round(v / max(n_local, 1), 4) for k, v in split.items()},
            "rank0_cost_evaluations_per_time_point": round(float(np.mean(evals)), 1) if evals else 0,
            "mip_stack": list(stack.shape), "mip_checksum": float(stack.double().sum().item()),
        }))
    for p in pipes:
        p.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
