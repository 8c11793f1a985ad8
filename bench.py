#!/usr/bin/env python3
"""Headline benchmark: Richardson-Lucy iterations/s on a 512x512x256 float32 volume.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one single-view RL iteration (forward projection, ratio, back projection,
estimate update: reference src/api_subfunc.cu:3404-3416) on one 512x512x256 volume per
GPU, inputs resident in HBM (BASELINE.json configs[1]).  N > 1: one process per GPU
(torch.distributed, RCCL), every rank deconvolves its own time point -- weak scaling, no
collective on the data path; the only exchange is the gather of the XY maximum-intensity
projections to rank 0 after the timed region.

`python bench.py --gpus N` from a plain shell starts the N ranks ITSELF (a child `python -m torch.distributed.run
--nproc-per-node N ... bench.py ...` spawned before anything touches the GPU; its output is passed through and its exit
status returned); under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      HBM roofline of the dominant kernel (z_conv of the fused backend: the top symbol of the rocprofv3
                kernel stats), timed with HIP events on the engine's stream inside the timed region; the other
                three kernel families are in extra.kernel_families;
  cpu_baseline  the library's gpuMemMode 0 path (C++ restatement of the reference's CPU twin) timed on this
                box's host cores (N = 1 only), a bounded sample: one thread like the reference + all cores;
  extra         dual-view it/s, reg3D voxels/s and the per-kernel-family time split.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPE = (256, 512, 512)  # (nz, ny, nx): imSize = {512, 512, 256}
PSF_SHAPE = (64, 64, 64)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def synth_clean(torch, shape, psf, seed):
    """Noise-free blob phantom blurred by the view PSF (sigma z, y, x) over background 100, generated on the device (data
    synthesis only; torch.fft is not part of the measured path).  Returns (volume, generator) -- the generator has produced
    the blobs and goes on to produce the noise of `synth_volume`."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    pts = torch.zeros(shape, device="cuda")
    n_blobs = 200
    idx = [torch.randint(0, s, (n_blobs,), device="cuda", generator=g) for s in shape]
    amp = torch.rand(n_blobs, device="cuda", generator=g) * 1800 + 200
    pts[idx[0], idx[1], idx[2]] = amp * 425.0  # unit-sum sigma-3 kernel peaks at 1/425: blob peak = amp (200..2000) over background 100
    # blob shape: isotropic Gaussian sigma 3; then the view PSF
    def gauss_kernel(sig):
        ax = [torch.arange(s, device="cuda", dtype=torch.float32) for s in shape]
        ax = [torch.minimum(a, s - a) for a, s in zip(ax, shape)]
        k = (torch.exp(-0.5 * (ax[0] / sig[0]) ** 2)[:, None, None] * torch.exp(-0.5 * (ax[1] / sig[1]) ** 2)[None, :, None]
             * torch.exp(-0.5 * (ax[2] / sig[2]) ** 2)[None, None, :])
        return k / k.sum()
    spec = torch.fft.rfftn(pts) * torch.fft.rfftn(gauss_kernel((3.0, 3.0, 3.0))) * torch.fft.rfftn(gauss_kernel(psf))
    vol = torch.fft.irfftn(spec, s=shape).clamp_(min=0) + 100.0
    del spec, pts
    return vol, g


def add_noise(torch, clean, noise_seed):
    """Poisson noise with its own seed: the per-time-point noise of the batch (rng(100 + t))."""
    g = torch.Generator(device="cuda").manual_seed(noise_seed)
    return torch.poisson(clean, generator=g).float().contiguous()


def synth_volume(torch, shape, psf, seed):
    """Blob phantom blurred by the PSF + Poisson noise (one generator for object and noise)."""
    vol, g = synth_clean(torch, shape, psf, seed)
    noisy = torch.poisson(vol, generator=g)
    return noisy.float().contiguous()


def gaussian_psf(shape, sigma_zyx):
    """Synthetic input only: unnormalised anisotropic Gaussian PSF centred at shape // 2."""
    g = []
    for s, sg in zip(shape, sigma_zyx):
        c = np.arange(s, dtype=np.float64) - s // 2
        g.append(np.exp(-0.5 * (c / sg) ** 2))
    return (g[0][:, None, None] * g[1][None, :, None] * g[2][None, None, :]).astype(np.float32)


def kernel_source_hashes():
    """sha256 of the kernel sources the PMC traffic figures depend on (profiles/pmc_traffic.json carries the same map)."""
    import hashlib
    out = {}
    for f in ("fft.hip", "fft2p.hip", "fft2p_any_yz.hip", "fft2p_dev.h", "fft_dev.h", "kernels.hip"):
        path = os.path.join(ROOT, "microimagelib_amd", "csrc", f)
        if os.path.exists(path):
            out[f] = hashlib.sha256(open(path, "rb").read()).hexdigest()
    return out


def cpu_baseline(img_host, psf_host, iters, threads):
    """The library's own gpuMemMode 0 path -- the C++ restatement of the reference's CPU twin (decon_singleview_OTF0,
    src/api_subfunc.cu:3310-3359: same order of operations, separate element-wise passes, bundled mixed-radix FFT in
    place of FFTW) -- through the same libapi.h entry point the GPU path is called by.  `threads` = MIL_CPU_THREADS
    (1 = the reference's behaviour).  Returns (iterations/s over the RL loop, loop seconds, whole-call seconds)."""
    from microimagelib_amd import api
    saved = os.environ.get("MIL_CPU_THREADS")
    os.environ["MIL_CPU_THREADS"] = str(threads)
    try:
        t0 = time.perf_counter()
        _, rec = api.decon_singleview(img_host, psf_host, iters, gpu_mem_mode=0)
        whole = time.perf_counter() - t0
    finally:
        if saved is None:
            os.environ.pop("MIL_CPU_THREADS", None)
        else:
            os.environ["MIL_CPU_THREADS"] = saved
    assert rec[0] == 0, "gpuMemMode 0 did not run on the host"
    loop_s = float(rec[8])
    return iters / loop_s, loop_s, whole


class Line:
    """Rank 0's ONE JSON line, and the guarantee that it is printed: every supplementary figure (`extra`) runs under
    `run`, which records a failure in extra.errors instead of losing the headline, and under a deadline -- a watchdog
    thread on EVERY rank that, when the extras have not finished in time (a rank stuck in a collective whose peer died),
    prints the line as it stands (rank 0) and ends the process with status 0."""

    def __init__(self, rank, real_stdout, deadline_s):
        import threading
        self.rank, self.real_stdout = rank, real_stdout
        self.result, self.errors, self.stage = None, [], "headline"
        self._lock, self._done = threading.Lock(), False
        self._timer = None
        self.deadline_s = deadline_s

    def start_deadline(self):
        import threading
        if self.deadline_s and self.deadline_s > 0:
            self._timer = threading.Timer(self.deadline_s, self._expired)
            self._timer.daemon = True
            self._timer.start()

    def _expired(self):
        self.errors.append({"extra": self.stage, "error": f"deadline of {self.deadline_s:.0f} s for the supplementary figures "
                                                          f"exceeded in stage '{self.stage}'; line printed by the watchdog"})
        print(f"bench.py: rank {self.rank}: extras deadline exceeded in '{self.stage}', ending the run", file=sys.stderr, flush=True)
        if self.result is not None:
            self.result["extras_truncated"] = True  # top level: a reader of the line sees that the run was cut short
        self.emit()
        # Status 0 on purpose: the headline was measured and printed in full, the truncation is in the line itself
        # (`extras_truncated`, extra.errors).  A non-zero status from ANY rank makes torch.distributed.run tear the others
        # down -- possibly rank 0 before its line is out -- and the watchdogs of all ranks fire together, so no rank is left
        # waiting in a collective for a peer that is gone.  MIL_BENCH_DEADLINE_RC overrides the status (CI that wants a
        # truncated run to fail).
        os._exit(int(os.environ.get("MIL_BENCH_DEADLINE_RC", "0")))

    def run(self, name, fn):
        """fn() with its failure recorded, never raised.  MIL_BENCH_FAIL_EXTRA=<name> raises inside it (test hook)."""
        import traceback
        self.stage = name
        try:
            if os.environ.get("MIL_BENCH_FAIL_EXTRA") == name:
                raise RuntimeError(f"MIL_BENCH_FAIL_EXTRA: deliberate failure inside extra '{name}'")
            return fn()
        except BaseException as e:  # incl. SystemExit from library code: the headline must survive
            if isinstance(e, KeyboardInterrupt):
                raise
            traceback.print_exc(file=sys.stderr)
            self.errors.append({"extra": name, "error": f"{type(e).__name__}: {e}"[:600]})
            return None

    def emit(self):
        with self._lock:
            if self._done:
                return
            self._done = True
            if self._timer is not None:
                self._timer.cancel()
            if self.rank != 0 or self.result is None:
                return
            for _ in range(5):  # the watchdog may serialise while the main thread is still adding a figure
                try:
                    self.result.setdefault("extra", {})["errors"] = list(self.errors)
                    text = json.dumps(self.result)
                    break
                except RuntimeError:
                    time.sleep(0.05)
            else:
                text = json.dumps({k: v for k, v in self.result.items() if k != "extra"})
            sys.stdout.flush()
            os.write(self.real_stdout, (text + "\n").encode())


def init_groups(dist, torch, backend, rank, world, local_rank, timeout_s):
    """Process group(s) with FINITE timeouts: a rank that never arrives becomes an exception, not a hang.  Returns the gloo
    control group (None when the main group is gloo itself)."""
    import datetime
    to = datetime.timedelta(seconds=timeout_s)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=to)
        return dist.new_group(backend="gloo", timeout=to)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=to)
    return None


def launch_ranks(n, argv):
    """`bench.py --gpus N` called directly: run the N ranks as a CHILD process tree (torch.distributed.run, one rank per
    GPU, rendezvous on 127.0.0.1) and hand back its exit status.  Nothing here has touched the GPU or imported torch, and
    the launcher is a child, never an exec -- a process that has initialised the GPU must not be replaced."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    print("bench.py: starting", n, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """Argument plumbing, rendezvous, the collectives of the timed region AND the config-5 driver (batch.run_sharded in
    regMode 3 and 2, the agreement handshake, the gather of all five products, the guarded extras and the line's watchdog) on
    the CPU over gloo, with stand-in arithmetic (FakeBatchKit): no GPU, no measurement.  Rank 0 prints one JSON line; used
    by tests/test_bench_launch.py so that everything above N = 1 cannot rot between the rare runs on a multi-GPU node."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("MIL_BENCH_DRY_RUN_FAIL") == str(rank):  # test hook: a rank that dies must fail the whole command
        raise SystemExit(3)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # gloo prints banners to fd 1
    distributed = world > 1
    control = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        control = init_groups(dist, torch, "gloo", rank, world, 0, args.dist_timeout)
        dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        slowest = float(t.item())
        parts = [torch.empty(4) for _ in range(world)] if rank == 0 else None
        dist.gather(torch.full((4,), float(rank)), gather_list=parts, dst=0)
        gathered = [float(p[0]) for p in parts] if rank == 0 else None
    else:
        slowest, gathered = 1.0, [0.0]
    line = Line(rank, real_stdout, args.extras_deadline)
    result = {"dry_run": True, "n_gpus": world, "requested_gpus": args.gpus, "steps": args.steps,
              "warmup": args.warmup, "max_over_ranks": slowest, "gathered_ranks": gathered,
              "local_rank_env": os.environ.get("LOCAL_RANK"), "extra": {}} if rank == 0 else None
    line.result = result
    line.start_deadline()

    def fence():
        if distributed:
            dist.barrier()

    if args.batch_time_points_per_gpu > 0 and not args.no_extra:
        def batch_extra():
            from microimagelib_amd import batch
            if os.environ.get("MIL_BENCH_DRY_RUN_BATCH_FAIL") == str(rank):  # test hook: ONE rank fails inside the batch
                def broken(t):
                    raise RuntimeError(f"MIL_BENCH_DRY_RUN_BATCH_FAIL: rank {rank} cannot load time point {t}")
                kit = FakeBatchKit(torch, max(1, args.batch_lanes))
                kit.load_pair = broken
            else:
                kit = FakeBatchKit(torch, max(1, args.batch_lanes))
            figs = batch_figures(kit, torch, dist, batch, args, world, rank, "cpu", fence, distributed, control)
            if rank == 0:
                result["extra"]["batch_config5_by_mode"] = figs
        line.run("batch_config5", batch_extra)
    line.run("hang", lambda: time.sleep(3600) if os.environ.get("MIL_BENCH_DRY_RUN_HANG") == str(rank) else None)
    line.run("other", lambda: result["extra"].__setitem__("other", "ran") if rank == 0 else None)
    line.stage = "shutdown"
    line.emit()
    try:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:
        print("bench.py: shutdown:", repr(e), file=sys.stderr, flush=True)


def timed(torch, fn, reps=5):
    """Average seconds of fn() on the current stream (HIP events), after one untimed call."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def auxiliary_kernels(torch, mdev, eng, img, psf):
    """The kernels either side of the RL loop and of the cost evaluation ("Resample: read R + write R"; f-row
    kernels), each with its algorithmic bytes, GB/s and fraction of the HBM peak; kernel names as rocprofv3 prints them."""
    nz, ny, nx = img.shape
    R = 4 * img.numel()

    def row(kernel, seconds, nbytes, note=None):
        d = {"kernel": kernel, "ms": round(seconds * 1e3, 4), "algorithmic_bytes": int(nbytes),
             "hbm_gbs": round(nbytes / seconds / 1e9, 1), "frac_of_peak": round(nbytes / seconds / 1e9 / HBM_PEAK_GBS, 4)}
        if note:
            d["note"] = note
        return d
    out = {}
    ident = np.eye(3, 4, dtype=np.float32).reshape(12)
    c, s = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
    lin = 1.02 * np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    ctr = np.array([nx / 2, ny / 2, nz / 2])
    m3 = np.concatenate([lin, (ctr - lin @ ctr + np.array([5.0, -3.0, 2.0]))[:, None]], axis=1).astype(np.float32).reshape(12)
    dst = torch.empty_like(img)
    lib = mdev._lib.load()

    def resample(m):
        mm = np.ascontiguousarray(m, dtype=np.float32)
        return lambda: mdev._check(lib.mil_affine_resample(dst.data_ptr(), img.data_ptr(), nx, ny, nz, nx, ny, nz,
                                                           mm.ctypes.data_as(mdev._lib.c_float_p),
                                                           int(torch.cuda.current_stream().cuda_stream)), "mil_affine_resample")
    out["resample_identity"] = row("resample_march_kernel<false, false>", timed(torch, resample(ident), reps=20), 2 * R, "affineTransform, read R + write R")
    out["resample_3deg_2pct"] = row("resample_march_kernel<true, false>", timed(torch, resample(m3), reps=20), 2 * R, "rotation 3 deg about y, scale 1.02, shift")
    # mip3dgpu: 36 x (resample into the rotated box + z projection), one fused kernel per angle: the box is never written, so
    # the bytes per angle are read R (source) + write the plane.  (The reference's two kernels move R + 2 boxes + the plane =
    # 1.6 GB per angle; on those bytes the same time would read 6.9 / 4.3 TB/s.)
    for axis in (1, 2):
        box = (nx, round((ny * ny + nz * nz) ** 0.5), round((ny * ny + nz * nz) ** 0.5)) if axis == 1 else \
              (round((nx * nx + nz * nz) ** 0.5), ny, round((nx * nx + nz * nz) ** 0.5))
        per_angle = R + 4 * box[0] * box[1]
        sec = timed(torch, lambda: mdev.mip3d(img, axis, 36), reps=2)
        kern = "mip_rot_x_kernel (x36)" if axis == 1 else "transpose_xy + mip_rot_x_kernel on the transposed volume + transpose_xy of the plane (x36)"
        out[f"mip3d_axis{axis}_36_angles"] = row(kern, sec, 36 * per_angle,
                                                 f"rotated box {box[0]}x{box[1]}x{box[2]}, resample + z projection fused; "
                                                 f"{36 * 8 * box[0] * box[1] * box[2] / 1e9:.1f} GB of box traffic avoided")
    for d, name in ((1, "xy"), (3, "yz"), (2, "zx")):
        sec = timed(torch, lambda: mdev.max_projection(img, d))
        out[f"max_projection_{name}"] = row("max_projection_kernel", sec, R, "mp2dgpu plane: read R")
    small = img[:nz - 6, :ny - 12, :nx - 12].contiguous()  # 500 x 500 x 250 in a 512 x 512 x 256 box
    padded = mdev.pad_stack(small, img.shape)
    out["pad_stack_500x500x250"] = row("pad_stack_kernel", timed(torch, lambda: mdev.pad_stack(small, img.shape)), 4 * small.numel() + R,
                                       "padstackgpu: read image + write box (torch allocation of the output included)")
    out["crop_500x500x250"] = row("crop_kernel", timed(torch, lambda: mdev.crop_stack(padded, small.shape)), 2 * 4 * small.numel(),
                                  "cropgpu: read + write image")
    out["rot_y90"] = row("rot_y90_kernel", timed(torch, lambda: mdev.rot_y90(img, 1)), 2 * R, "imoperation3D +90 deg about y")
    t0 = time.perf_counter()
    for _ in range(3):
        eng.gen_otf(psf)
    eng.sync()
    out["gen_otf"] = {"kernels": "reduce_kernel + pad_psf_kernel + forward transform (3 kernels)",
                      "ms": round((time.perf_counter() - t0) / 3 * 1e3, 3),
                      "note": "genOTFgpu of a 64x64x64 PSF into the 512x512x256 box, incl. allocation of the OTF"}
    return out


def drift_matrix(t, shape):
    """Target -> source matrix of time point t: rotation about y by 2 deg + 0.02 deg * t around the volume centre, shift
    (3, -2, 1) + t * (0.15, -0.10, 0.05) voxels -- a specimen that drifts slowly through the time-lapse, so that regMode 2's
    chained start matrix has something to follow and every registration does real work (round 4 fed every time point the
    SAME pair: each one after the first started at its own optimum)."""
    ang = np.deg2rad(2.0 + 0.02 * t)
    c, s_ = np.cos(ang), np.sin(ang)
    lin = np.array([[c, 0, -s_], [0, 1, 0], [s_, 0, c]])
    centre = np.array([shape[2] / 2, shape[1] / 2, shape[0] / 2])
    shift = np.array([3.0, -2.0, 1.0]) + t * np.array([0.15, -0.10, 0.05])
    return np.concatenate([lin, (centre - lin @ centre + shift)[:, None]], axis=1).astype(np.float32).reshape(12)


class GpuBatchKit:
    """Inputs and pipelines of the config-5 figures on the GPU: ONE object ("same pair"), every time point with
    its own Poisson noise (seeds 100 + t and 10100 + t) and its own drifted matrix; all pairs of this rank resident in HBM
    before the clock starts."""

    def __init__(self, torch, mdev, batch, psf_a, psf_b, lanes):
        self.torch, self.mdev, self.batch = torch, mdev, batch
        self.shape = SHAPE
        self.psf_a, self.psf_b, self.lanes = psf_a, psf_b, lanes
        self.pairs, self.pipes = {}, []

    def prepare(self, time_points):
        torch = self.torch
        clean_a, _ = synth_clean(torch, SHAPE, (6.0, 2.5, 2.5), seed=7)
        clean_b, _ = synth_clean(torch, SHAPE, (2.5, 2.5, 6.0), seed=7)  # the same object through the other PSF
        for t in time_points:
            a = add_noise(torch, clean_a, 100 + t)
            b = self.mdev.affine_resample(add_noise(torch, clean_b, 10100 + t), drift_matrix(t, SHAPE), SHAPE)
            self.pairs[t] = (a, b)
        del clean_a, clean_b
        self.pipes = [self.batch.FusionPipeline(SHAPE, self.psf_a, self.psf_b, iters=20, mips="all") for _ in range(self.lanes)]
        torch.cuda.synchronize()

    def load_pair(self, t):
        return self.pairs[t]

    def close(self):
        for p in self.pipes:
            p.close()
        self.pairs.clear()


class FakeBatchKit:
    """--dry-run: the same driver code (batch.run_sharded, token ring, gather of five products, agreement handshake) over
    gloo with stand-in arithmetic on tiny CPU tensors.  Nothing here is a measurement."""

    class Pipe:
        def __init__(self, torch, shape):
            self.torch, self.shape = torch, shape

        def reg_fn(self, target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit):
            tmx = np.eye(3, 4, dtype=np.float32).reshape(12)
            if input_tmx and i_tmx is not None:
                tmx = np.asarray(i_tmx, dtype=np.float32).reshape(12).copy()
            tmx[3] += 0.25
            rec = np.zeros(11, dtype=np.float32)
            rec[3], rec[5] = 0.9, 17.0
            return source, tmx, rec

        def check_fn(self, tmx):
            return True

        @property
        def mip_shape(self):
            nz, ny, nx = self.shape
            return {"xy": (ny, nx), "yz": (nz, ny), "zx": (nx, nz), "mip3d_x": (3, ny + 1, nx), "mip3d_y": (3, ny, nx + 1)}

        def fuse_pair(self, a, b_reg, t):
            torch = self.torch
            est = (a + b_reg) * 0.5
            return est, {k: torch.full(shp, float(t), dtype=torch.float32) for k, shp in self.mip_shape.items()}

        def close(self):
            pass

    def __init__(self, torch, lanes):
        self.torch, self.lanes, self.shape = torch, lanes, (4, 6, 8)
        self.pairs, self.pipes = {}, []

    def prepare(self, time_points):
        for t in time_points:
            self.pairs[t] = (self.torch.full(self.shape, float(t)), self.torch.full(self.shape, float(t) + 0.5))
        self.pipes = [FakeBatchKit.Pipe(self.torch, self.shape) for _ in range(self.lanes)]

    def load_pair(self, t):
        return self.pairs[t]

    def close(self):
        self.pairs.clear()


def batch_figures(kit, torch, dist, batch, args, world, rank, coll_dev, fence, distributed, control, tiff_leg=None):
    """BASELINE.json configs[4] per regMode: what ">= 7x at 8 GPUs" is judged on.  regMode 3 (independent time points,
    `lanes` in flight per GPU), regMode 2 (the chained mode sh_spimFusionBatch.sh:21 runs: token ring over the ranks) and
    regMode 3 with 16-bit TIFF stacks read and written inside the timed region.  Every time point has its own noise and its
    own drifted matrix (GpuBatchKit); every one ends in the FIVE projection products of the reference, gathered to rank 0
    in one exchange whose seconds and bytes are reported.  `kit` supplies volumes and pipelines (FakeBatchKit: --dry-run)."""
    lanes = max(1, args.batch_lanes)
    n_tp = args.batch_time_points_per_gpu * world
    all_tps = list(range(n_tp))
    kit.prepare(batch.shard_time_points(all_tps, rank, world))
    pipes = kit.pipes
    pipe = pipes[0]
    out = {}
    evals = []  # cost evaluations of every registration this rank made inside a timed region (records[5])

    def policy(mode):
        return batch.RegPolicy(reg_mode=mode, reg_choice=2, aff_method=6, ftol=1e-4, it_limit=3000)

    def count(t, decon, tmx, rec):
        evals.append(float(rec[5]))

    def measure(key, workload, run, warm_tps, extra=None, tps=None):
        tps = all_tps if tps is None else tps
        n_tp = len(tps)
        run(warm_tps)  # warm-up: plans, OTFs, clocks
        del evals[:]
        fence()
        tb = time.perf_counter()
        stacks = run(tps)
        fence()
        sec = time.perf_counter() - tb
        if distributed:
            tt = torch.tensor([sec], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sec = float(tt.item())
        if rank == 0:
            g = batch.last_gather
            ev = sorted(evals)
            out[key] = dict({"workload": workload, "time_points": n_tp, "time_points_per_s": round(n_tp / sec, 3),
                             "elapsed_s": round(sec, 3),
                             "registration_evaluations_per_time_point_rank0": (
                                 {"mean": round(sum(ev) / len(ev), 1), "min": ev[0], "max": ev[-1], "registrations": len(ev)}
                                 if ev else None),
                             "mip_gather": {"seconds_rank0": round(g["seconds"], 4), "products": g["products"],
                                            "bytes_per_time_point": g["bytes_per_time_point"],
                                            "bytes_received_rank0": g["bytes_received_rank0"]},
                             "mip_stacks": {k: list(v.shape) for k, v in stacks.items()}}, **(extra or {}))

    base = ("reg3d (affMethod 6) + joint dual-view RL (20 it) + XY / YZ / ZX MIPs + 36-angle 3-D MIPs about x and y per time "
            "point, 512x512x256, all projections gathered to rank 0; every time point has its own noise seed and a drifting "
            "matrix (2 deg + 0.02 deg * t about y, shift + 0.15 voxel * t)")
    # RCCL carries the gather of the projections and nothing else: the 12 / 24 floats of regMode 1 / 2 go over `control`
    common = dict(load_pair=kit.load_pair, reg_fn=pipe.reg_fn, check_fn=pipe.check_fn, mip_shape=pipe.mip_shape,
                  device=coll_dev, control_group=control, control_device="cpu" if control is not None else None, sink=count)
    if lanes > 1:
        # `lanes` time points in flight per GPU, one host thread each with its own engine: the kernels of one fill the host
        # round trips between the dependent cost evaluations of the other
        run3 = lambda tps: batch.run_sharded(tps, fuse_pair=[p.fuse_pair for p in pipes], policy=policy(3), lanes=lanes, **common)
    else:
        run3 = lambda tps: batch.run_sharded(tps, fuse_pair=pipe.fuse_pair, policy=policy(3), overlap=True, **common)
    measure("regmode3", "spimFusionBatch loop, regMode 3 (independent time points): " + base + "; views resident in HBM",
            run3, all_tps[:world * lanes], {"time_points_in_flight_per_gpu": lanes})
    # one rank: the joint RL + projections of time point i run on a helper thread while i + 1 is registered (the chain only
    # orders the registrations); several ranks: the token ring overlaps them across ranks
    run2 = lambda tps: batch.run_sharded(tps, fuse_pair=pipe.fuse_pair, policy=policy(2), overlap=(world == 1), **common)
    measure("regmode2", "spimFusionBatch loop, regMode 2 (chained start matrices, the mode of sh_spimFusionBatch.sh:21; token "
            "ring over the ranks, registrations serial by definition): " + base + "; views resident in HBM",
            run2, all_tps[:world], {"time_points_in_flight_per_gpu": 1})
    if tiff_leg is not None:
        tiff_leg(kit, pipe, common, policy, measure, base, all_tps)
    kit.close()
    return out


def gpu_tiff_leg(torch, batch, args, world, rank):
    """regMode 3 with both views READ from 16-bit TIFF stacks and the deconvolved 16-bit stack WRITTEN inside the timed region.
    Every time point has its own files (quantised copies of its own pair); at most --batch-tiff-time-points per GPU, the
    figure is then quoted on that many."""
    def leg(kit, pipe, common, policy, measure, base, all_tps):
        from microimagelib_amd import api
        tdir = os.path.join(args.batch_tiff_dir, f"rank{rank}")
        os.makedirs(tdir, exist_ok=True)
        per_gpu = max(1, min(args.batch_tiff_time_points, args.batch_time_points_per_gpu))
        tps = all_tps[:per_gpu * world]
        for t in batch.shard_time_points(tps, rank, world):  # setup, not timed
            a, b = kit.load_pair(t)
            for name, vol in (("SPIMA", a), ("SPIMB", b)):
                api.writetifstack(os.path.join(tdir, f"{name}_{t}.tif"), vol.clamp(0, 65535).cpu().numpy(), bits=16)
        bufs = torch.empty((3, 2) + SHAPE, dtype=torch.float32).pin_memory()  # overlap: three pairs are live at once
        outb = torch.empty(SHAPE, dtype=torch.float32).pin_memory()
        calls = [0]

        def load_tiff(t):
            buf = bufs[calls[0] % 3]
            calls[0] += 1
            for i, name in enumerate(("SPIMA", "SPIMB")):
                api.readtifstack(os.path.join(tdir, f"{name}_{t}.tif"), out=buf[i].numpy())
            a, b = buf[0].cuda(non_blocking=True), buf[1].cuda(non_blocking=True)
            torch.cuda.current_stream().synchronize()  # the library's streams do not wait for this thread's stream
            return a, b

        def sink(t, decon, tmx, rec):
            common["sink"](t, decon, tmx, rec)
            outb.copy_(decon, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            api.writetifstack(os.path.join(tdir, f"Decon_{t}.tif"), outb.numpy(), bits=16)
        tiff_common = dict(common, load_pair=load_tiff, sink=sink)
        runt = lambda ts: batch.run_sharded(ts, fuse_pair=pipe.fuse_pair, policy=policy(3), overlap=True, **tiff_common)
        try:
            measure("regmode3_tiff", "spimFusionBatch loop, regMode 3, both views READ from 16-bit TIFF stacks and the deconvolved "
                    "16-bit stack WRITTEN inside the timed region (library codec; decode / upload / encode on helper threads): "
                    + base, runt, tps[:world], {"time_points_in_flight_per_gpu": 1, "tiff_dir": args.batch_tiff_dir}, tps=tps)
        finally:
            for f in os.listdir(tdir):
                os.remove(os.path.join(tdir, f))
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 100 steps ~ 0.1 s of GPU time: the fixed cost of a run_single call (first pass over cold caches, clock ramp after the
    # fence, ~1.3 ms) is then ~1 % of the timed region; at 20 steps it is 6 % (measured: 915 vs 980 it/s).
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    # The GPU raises its clock over the first ~50 ms of sustained load (measured: the marginal step costs 1.13 ms in the
    # first 5 steps after idle, 1.02 ms after 50, 1.01 ms after 100).  A fixed number of untimed steps before the W
    # warm-up steps brings it to the sustained state a long deconvolution runs in, whatever W the caller picks.
    ap.add_argument("--clock-warm-steps", type=int, default=50)
    ap.add_argument("--batch-lanes", type=int, default=2,
                    help="time points in flight per GPU in the config-5 figure (host threads, batch.run_sharded(lanes=))")
    ap.add_argument("--batch-time-points-per-gpu", type=int, default=16,
                    help="time points per GPU of the config-5 figures in `extra` (register + joint RL + projections), each "
                         "with its own noise and drifted matrix; 0 = skip")
    ap.add_argument("--batch-tiff-time-points", type=int, default=4,
                    help="time points per GPU of the TIFF-inclusive config-5 figure (each has its own 2 x 134 MB files)")
    ap.add_argument("--lead-steps", type=int, default=10,
                    help="untimed steps enqueued between the opening fence and the start event of the K timed steps")
    ap.add_argument("--extras-deadline", type=float, default=900.0,
                    help="seconds the supplementary figures may take before the watchdog prints the line as it stands")
    ap.add_argument("--dist-timeout", type=float, default=300.0, help="timeout of the process groups' collectives, seconds")
    ap.add_argument("--batch-tiff-dir", default="/tmp/mil_bench_tiff",
                    help="scratch directory of the TIFF-inclusive config-5 figure ('' = skip it)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / collective plumbing only, on the CPU over gloo: no GPU, no measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=10, help="RL iterations of the single-thread CPU baseline (~1 s each)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="second CPU figure on this many threads (a 1-GPU box's share of host cores); 1 = skip")
    # functional test of the N > 1 path on a one-GPU box: all ranks on cuda:0, collectives over gloo (never a measurement)
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl", help=argparse.SUPPRESS)
    # run the collective code (barrier, all-reduce of the time, MIP gather) even in a world of ONE rank: the only way to
    # execute the RCCL branch on a one-GPU box (RCCL refuses two ranks on one device, profiles/r03/nccl_two_ranks_one_gpu.txt)
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", choices=["auto", "rocfft", "fused"], default="auto",
                    help="transform backend of the RL engine (auto = fused LDS-tiled kernels on power-of-two boxes)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # called directly: start the ranks ourselves, as a child
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run:
        return dry_run(args)

    # stdout carries the JSON line and nothing else: RCCL and gloo print banners ("RCCL version : ...", "[Gloo] Rank 0 is
    # connected to ...") to file descriptor 1 from native code, so fd 1 is pointed at stderr for the run and the line is
    # written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (got WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or args.force_dist
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"  # where the (tiny) collective payloads live
    control = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # host-side control messages of the batch (regMode-1 matrix, regMode-2 chain token: 12 / 24 floats, the agreement
        # handshake before the gather) travel over gloo even when the collectives are RCCL's: RCCL is used for the gather of
        # the projections (and this file's timing all-reduce) only.  Finite timeouts: a missing rank is an exception.
        control = init_groups(dist, torch, args.dist_backend, rank, world, local_rank, args.dist_timeout)
    line = Line(rank, real_stdout, args.extras_deadline)

    from microimagelib_amd import _lib, device as mdev
    lib = _lib.load()
    assert lib.mil_set_device(local_rank) == 0

    psf_a_host = gaussian_psf(PSF_SHAPE, (6.0, 2.5, 2.5))  # sigma (z, y, x)
    psf_b_host = gaussian_psf(PSF_SHAPE, (2.5, 2.5, 6.0))
    psf_a = torch.from_numpy(psf_a_host).cuda()
    psf_b = torch.from_numpy(psf_b_host).cuda()

    # each rank = one time point of the batch (seeded per rank)
    img_a = synth_volume(torch, SHAPE, (6.0, 2.5, 2.5), seed=100 + rank)
    n_real = img_a.numel()
    R = 4 * n_real
    C = 8 * SHAPE[0] * SHAPE[1] * (SHAPE[2] // 2 + 1)

    eng = mdev.DeconEngine(SHAPE, backend={"auto": 0, "rocfft": 1, "fused": 2}[args.backend])
    fused = eng.backend == mdev.DeconEngine.FUSED
    otf_a = eng.gen_otf(psf_a)
    bp_a = eng.gen_otf(psf_a, flip=True)
    A = img_a.clone()
    E = torch.empty_like(A)
    eng.init_single(A, E)
    eng.run_single(E, A, otf_a, bp_a, args.clock_warm_steps)  # untimed: clock ramp
    eng.run_single(E, A, otf_a, bp_a, args.warmup)            # untimed: the W warm-up steps
    eng.sync()

    ALL_TIMERS = (mdev.DeconEngine.TIMER_UPDATE, mdev.DeconEngine.TIMER_RATIO, mdev.DeconEngine.TIMER_CMUL,
                  mdev.DeconEngine.TIMER_FFT)
    # Inside the timed region only the roofline kernel is bracketed with HIP events on the engine's stream
    # (fused backend: z_conv = z transform, OTF product, inverse z transform in one kernel -- the largest single
    # symbol in the rocprofv3 kernel stats; rocFFT backend: the OTF product), and only every 8th launch of it
    # (one per four steps: 25 samples at K = 100, 5 at K = 20; one per step cost 0.6 % of the timed region).  An event pair is a pair of stream markers: the
    # rocprofv3 trace shows 0 us between back-to-back kernels and 6-10 us across a marker, so bracketing all 8
    # launches of a step would cost several percent of it.  The other families are timed in a second,
    # un-clocked pass of the same K steps.
    dom_timer = mdev.DeconEngine.TIMER_CMUL  # fused: z_conv, the largest single kernel symbol of the step

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # The K timed steps sit between two HIP events on the ENGINE's stream ("hipEvent around the loop"), behind
    # `lead` untimed steps enqueued after the opening fence: while the host was fenced the queue ran dry and the clocks
    # dropped, and the first pass after that costs ~1.3 ms -- 6 % of a 20-step region, 1 % of a 100-step one (round 4's
    # figure moved with K).  The fences of the contract (barrier + synchronize on both sides) stay where they were; the
    # host clock over the whole fenced region is reported as extra.wall_clock_it_s.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    if args.lead_steps > 0:
        eng.run_single(E, A, otf_a, bp_a, args.lead_steps)
    if not os.environ.get("MIL_BENCH_NO_EVENTS"):  # diagnostic: how much the in-region markers cost
        eng.timer_enable(dom_timer, int(os.environ.get("MIL_BENCH_EVENT_STRIDE", "8")))
    ev0.record(eng.stream)
    eng.run_single(E, A, otf_a, bp_a, args.steps)  # EXACTLY K steps
    ev1.record(eng.stream)
    eng.sync()
    fence()
    wall = time.perf_counter() - t0
    elapsed = ev0.elapsed_time(ev1) * 1e-3
    if distributed:
        t = torch.tensor([elapsed, wall], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, wall = float(t[0].item()), float(t[1].item())
    dom_ms, dom_n = eng.timer_read(dom_timer)
    eng.timer_enable(dom_timer, False)

    # second pass, all families bracketed (not part of `value`)
    for w in ALL_TIMERS:
        eng.timer_enable(w, True)
    t1 = time.perf_counter()
    eng.run_single(E, A, otf_a, bp_a, args.steps)
    eng.sync()
    instrumented = time.perf_counter() - t1
    upd_ms, upd_n = eng.timer_read(mdev.DeconEngine.TIMER_UPDATE)
    rat_ms, rat_n = eng.timer_read(mdev.DeconEngine.TIMER_RATIO)
    cm_ms, cm_n = eng.timer_read(mdev.DeconEngine.TIMER_CMUL)
    fft_ms, fft_n = eng.timer_read(mdev.DeconEngine.TIMER_FFT)
    for w in ALL_TIMERS:
        eng.timer_enable(w, False)

    mip = mdev.max_projection(E, 1).to(coll_dev)  # XY MIP of this rank's estimate, 512x512 float32
    mip_checksum = float(mip.double().sum().item())

    result = None
    if rank == 0:
        value = world * args.steps / elapsed
        # Kernel families timed with HIP events on the engine's stream inside the timed region, with the
        # ALGORITHMIC bytes one launch must move (R = real volume, C = half spectrum [nz][ny][nx/2+1] complex64;
        # DESIGN.md "Kernels").  Fused backend: every family is one hand-written kernel.
        two_pass = fused and getattr(eng, "two_pass", 0) == 3
        if two_pass:  # csrc/fft2p.hip: the y transform rides in the z kernel (its last stage) and in the x kernel
            # the spectra and OTFs of this path are PACKED: exactly nx / 2 complex columns per row (column 0 carries X[0] and
            # X[nx/2]), C'' = 8 nz ny nx / 2 = R bytes -- what the kernels really have to move
            C2 = 8 * SHAPE[0] * SHAPE[1] * (SHAPE[2] // 2)
            fam = {
                "xy_update (y head^-1, x^-1, E=max(E*T,0.01), x, y head fused)": (upd_ms, upd_n, 2 * C2 + 2 * R),
                "xy_ratio (y head^-1, x^-1, T=A/T, x, y head fused)": (rat_ms, rat_n, 2 * C2 + R),
                "yz_conv (y tail, z, *OTF, z^-1, y tail^-1 fused)": (cm_ms, cm_n, 3 * C2),
            }
            iter_bytes = 10 * C2 + 3 * R
        elif fused:
            fam = {
                "x_update (x^-1, E=max(E*T,0.01), x fused)": (upd_ms, upd_n, 2 * C + 2 * R),
                "x_ratio (x^-1, T=A/T, x fused)": (rat_ms, rat_n, 2 * C + R),
                "z_conv (z, *OTF, z^-1 fused)": (cm_ms, cm_n, 3 * C),
                "y_pass (in-place column transform)": (fft_ms, fft_n, 2 * C),
            }
            iter_bytes = 18 * C + 3 * R
        else:
            fam = {
                "rl_update (E = max(E*T, 0.01), fused multiply+clamp)": (upd_ms, upd_n, 3 * R),
                "rl_ratio (T = A/T)": (rat_ms, rat_n, 3 * R),
                "cmul_otf (S *= OTF)": (cm_ms, cm_n, 3 * C),
                "rocfft_execute (library 3-D real transform, 3-4 kernels)": (fft_ms, fft_n, R + C),
            }
            iter_bytes = 4 * (R + C) + 2 * 3 * C + 3 * R + 3 * R
        upd_key = [k for k in fam if k.startswith(("x_update", "xy_update", "rl_update"))][0]
        upd_frac = (round(fam[upd_key][2] / (upd_ms / upd_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    if upd_n and upd_ms > 0 else None)
        dom = [k for k in fam if k.startswith(("yz_conv", "z_conv") if fused else "cmul_otf")][0]
        dom_bytes = fam[dom][2]
        dom_avg_ms = dom_ms / max(dom_n, 1)  # measured inside the timed region
        achieved = dom_bytes / (dom_avg_ms * 1e-3) / 1e9 if dom_n else None
        # HBM bytes per launch from the PMC counters: a separate rocprofv3 run (tools/gpu_profile_run.sh) recorded in
        # profiles/pmc_traffic.json together with the hashes of the kernel sources it was measured on.  A kernel edited
        # since then invalidates the number: traffic is then null, not a stale constant.
        traffic, traffic_meta = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                want = tj.get("sources_sha256", {})
                now = kernel_source_hashes()
                stale = [f for f, h in want.items() if now.get(f) != h]
                traffic_meta = {"measured_on_commit": tj.get("commit"), "kernel_symbol": tj.get("symbols", {}).get(dom.split(" ")[0]),
                                "stale_sources": stale}
                if want and not stale:
                    traffic = tj.get("fused" if fused else "rocfft", {}).get(dom.split(" ")[0])
            except Exception:
                traffic = None
        result = {
            "metric": "RL iterations/s, 512x512x256 float32, single view",
            "value": round(value, 3),
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_warm_steps": args.clock_warm_steps,
            "untimed_steps_total": args.warmup + args.clock_warm_steps,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "deconSingleView 512x512x256 float32, RL iterations, 1 MI355X (BASELINE.json configs[1])",
                "psf": "64x64x64 anisotropic Gaussian, matched back projector",
                "backend": ("fused, two passes per convolution" if two_pass else "fused") if fused else "rocfft",
                "volumes_per_gpu": 1,
                "sharding": "one time-point volume per GPU, no data-path collective; XY-MIP gather over RCCL after the timed region",
            },
            "roofline": {
                "kernel": dom,
                "bound": "hbm",
                "achieved": round(achieved, 1) if achieved else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                "traffic": traffic,
                "traffic_provenance": traffic_meta,
                "algorithmic_bytes_per_launch": dom_bytes,
                "avg_launch_ms": round(dom_avg_ms, 5),
                "launches": int(dom_n),
            },
            "extra": {
                "kernel_families": {
                    k: {"ms_per_launch": round(ms / max(n, 1), 5), "launches": int(n), "algorithmic_bytes": b,
                        "hbm_gbs": round(b / (ms / max(n, 1) * 1e-3) / 1e9, 1) if n and ms > 0 else None,
                        "share_of_step": round(ms / (1e3 * instrumented), 4)}
                    for k, (ms, n, b) in fam.items()
                },
                "kernel_families_note": "second pass of the same K steps with every launch bracketed by HIP events "
                                        f"({round(1e3 * instrumented / args.steps, 4)} ms per step with the extra markers)",
                # 4 transforms in+out, 2 OTF products, ratio, update = 5 379 194 880 B
                "iteration_algorithmic_bytes": 4 * (R + C) + 2 * 3 * C + 3 * R + 3 * R,
                "iteration_hbm_roofline_frac": round((4 * (R + C) + 6 * C + 6 * R) / (elapsed / args.steps) / 1e9
                                                     / HBM_PEAK_GBS, 4),
                "iteration_hbm_roofline_frac_note": "work-equivalent: SURVEY 8(d)'s bytes of the ideal per-stage-fused "
                                                    "iteration over this backend's time -- NOT the bytes this backend moves",
                "iteration_bytes_this_backend_minimum": iter_bytes,
                # what the kernels of this backend really have to move, over the same time: the achieved HBM fraction
                "iteration_hbm_achieved_frac": round(iter_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                "mip_checksum": mip_checksum,
                # north_star: ">= 60 % of HBM-read roofline on the RL element-wise update" -- the kernel that carries it
                # (fused backend: x^-1 . E = max(E*T, eps) . x on 2C + 2R bytes; rocFFT backend: the 3R update kernel)
                "rl_update_kernel_hbm_frac": upd_frac,
            },
        }

    # From here on nothing can lose the headline: every supplementary figure runs under line.run (a failure lands in
    # extra.errors), and a watchdog prints the line as it stands should the extras not finish within --extras-deadline.
    line.result = result
    line.start_deadline()
    if rank == 0:
        result["extra"]["wall_clock_it_s"] = round(world * (args.steps + args.lead_steps) / wall, 3)
        # how `value` was timed, at the top level (advisor, round 5): rounds 1-4 and the CPU baseline are host-clock figures
        # over the fenced region; since round 5 `value` is the K steps between two HIP events behind the lead steps.  The
        # host-clock figure of THIS run is kept beside it so the two methodologies are never compared blind.
        result["timing"] = "hip_events_after_lead"
        result["wall_clock_it_s"] = result["extra"]["wall_clock_it_s"]
        result["extra"]["wall_clock_note"] = (f"host clock over the fenced region: {args.lead_steps} lead + {args.steps} timed steps, "
                                              "launch of the first pass and both fences included; `value` is the K timed steps "
                                              "between two HIP events on the engine's stream")

    # ---- gather of the XY MIPs (the kind of exchange the batch makes; here one plane per rank) ------------
    if distributed:
        def xy_gather():
            stack = [torch.empty_like(mip) for _ in range(world)] if rank == 0 else None
            dist.gather(mip, gather_list=stack, dst=0)
            if rank == 0:
                result["extra"]["mip_checksum"] = float(sum(float(m.double().sum().item()) for m in stack))
                result["extra"]["mip_checksum_ranks"] = world
        line.run("xy_mip_gather", xy_gather)

    # ---- supplementary, every N: BASELINE.json configs[4], the time-lapse batch ------------
    # Each rank registers view B to view A (reg3d affMethod 6 as the batch hard-codes), deconvolves the pair jointly (20
    # iterations) and projects; the projections are gathered to rank 0 (the only exchange).
    if args.batch_time_points_per_gpu > 0 and not args.no_extra:
        def batch_extra():
            from microimagelib_amd import batch
            kit = GpuBatchKit(torch, mdev, batch, psf_a, psf_b, max(1, args.batch_lanes))
            try:
                figs = batch_figures(kit, torch, dist, batch, args, world, rank, coll_dev, fence, distributed, control,
                                     tiff_leg=gpu_tiff_leg(torch, batch, args, world, rank) if args.batch_tiff_dir else None)
            finally:
                kit.close()
            if rank == 0:
                result["extra"]["batch_config5"] = figs.get("regmode3")
                result["extra"]["batch_config5_by_mode"] = figs
        line.run("batch_config5", batch_extra)

    # ---- supplementary: dual view + registration (N = 1 only) ---------------------------
    if world == 1 and not args.no_extra:
        def dual_extra():
            img_b = synth_volume(torch, SHAPE, (2.5, 2.5, 6.0), seed=200)
            otf_b = eng.gen_otf(psf_b)
            bp_b = eng.gen_otf(psf_b, flip=True)
            B = img_b.clone()
            eng.init_dual(A, B, E)
            eng.run_dual(E, A, B, otf_a, otf_b, bp_a, bp_b, 2)
            eng.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run_dual(E, A, B, otf_a, otf_b, bp_a, bp_b, args.steps)
            eng.sync()
            dual_s = time.perf_counter() - t0
            result["extra"]["dualview_iterations_per_s"] = round(args.steps / dual_s, 3)
            dual_bytes = 2 * result["extra"]["iteration_bytes_this_backend_minimum"]  # two view updates per iteration
            result["extra"]["dualview"] = {
                "workload": "deconDualView joint RL (2 PSFs) 512x512x256 (BASELINE.json configs[2])",
                "iterations_per_s": round(args.steps / dual_s, 3), "ms_per_iteration": round(1e3 * dual_s / args.steps, 4),
                "roofline": {"bound": "hbm", "bytes_per_iteration_this_backend": dual_bytes,
                             "achieved": round(dual_bytes / (dual_s / args.steps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(dual_bytes / (dual_s / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                             "work_equivalent_frac": round(2 * result["extra"]["iteration_algorithmic_bytes"]
                                                           / (dual_s / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}}
            del B, img_b, otf_b, bp_b

        line.run("dualview", dual_extra)

        def reg_extra():
            # registration: source = target warped by a known matrix (rotation 3 deg about y, scale 1.02, shift)
            c, s = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
            lin = 1.02 * np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
            centre = np.array([SHAPE[2] / 2, SHAPE[1] / 2, SHAPE[0] / 2])  # (x, y, z)
            shift = centre - lin @ centre + np.array([5.0, -3.0, 2.0])      # rotate / scale about the volume centre
            known = np.concatenate([lin, shift[:, None]], axis=1).astype(np.float32).reshape(12)
            target = img_a
            source = mdev.affine_resample(target, known, SHAPE)
            torch.cuda.synchronize()
            reg, tmx, rec = mdev.reg3d_affine(target, source, aff_method=7, ftol=1e-4, it_limit=3000)
            evals, opt_s = float(rec[5]), float(rec[6])
            launches = float(mdev.reg3d_last_cost_launches())  # evaluations that really passed over both volumes
            result["extra"]["reg3d"] = {
                "workload": "reg3D 12-DOF cascade (affMethod 7), Powell + correlation cost, 512x512x256",
                "voxels_per_s": round(n_real * launches / opt_s, 1) if opt_s > 0 else None,
                "cost_evaluations": int(evals),
                "cost_kernel_launches": int(launches),
                "evaluations_per_s": round(evals / opt_s, 2) if opt_s > 0 else None,
                "optimisation_s": round(opt_s, 3),
                "zncc_initial": round(float(rec[1]), 5),
                "zncc_final": round(float(rec[3]), 5),
                "hbm_roofline_frac_per_eval": round(2 * R * launches / opt_s / 1e9 / HBM_PEAK_GBS, 4) if opt_s > 0 else None,
                # per launched evaluation: read target once + source once (2R = 536 870 912 B), whole optimisation
                # time (kernel + launch + poll + Powell's host logic) in the denominator
                "roofline": {"bound": "hbm", "algorithmic_bytes_per_evaluation": 2 * R,
                             "achieved": round(2 * R * launches / opt_s / 1e9, 1) if opt_s > 0 else None, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s",
                             "frac": round(2 * R * launches / opt_s / 1e9 / HBM_PEAK_GBS, 4) if opt_s > 0 else None,
                             "us_per_launched_evaluation": round(1e6 * opt_s / launches, 2) if launches else None},
            }
            # the same pair through the direct 12-DOF search (affMethod 5), SURVEY.md section 8(d)
            reg5, tmx5, rec5 = mdev.reg3d_affine(target, source, aff_method=5, ftol=1e-4, it_limit=3000)
            ev5, s5, l5 = float(rec5[5]), float(rec5[6]), float(mdev.reg3d_last_cost_launches())
            result["extra"]["reg3d_direct_12dof"] = {
                "workload": "reg3D direct 12-DOF (affMethod 5), same volumes",
                "voxels_per_s": round(n_real * l5 / s5, 1) if s5 > 0 else None,
                "cost_evaluations": int(ev5), "cost_kernel_launches": int(l5),
                "evaluations_per_s": round(ev5 / s5, 2) if s5 > 0 else None, "optimisation_s": round(s5, 3),
                "zncc_initial": round(float(rec5[1]), 5), "zncc_final": round(float(rec5[3]), 5),
            }
            del reg, reg5, source

        line.run("reg3d", reg_extra)

        def host_pointer_extra():
            # the reference's own boundary: host pointers in / out (H2D, OTF preparation, RL loop, D2H inside the call);
            # never the headline value -- DESIGN.md "Measurement"
            from microimagelib_amd import api
            img_h = img_a.cpu().numpy()
            out_h = np.empty_like(img_h)  # caller-owned output buffer, as in the C API; touched by the warm-up call
            api.decon_singleview(img_h, psf_a_host, 1, out=out_h)  # first call: engine, OTFs and buffers are built and kept
            calls = {}
            for iters in sorted({20, args.steps}):  # 20 = BASELINE.json configs[1]; K = this run's step count
                t0 = time.perf_counter()
                _, rec = api.decon_singleview(img_h, psf_a_host, iters, out=out_h)
                call_s = time.perf_counter() - t0
                import ctypes
                split = (ctypes.c_double * 5)()
                lib.mil_decon_last_call_split(split)
                calls[str(iters)] = {"iterations": iters, "whole_call_s": round(call_s, 4),
                                     "iterations_per_s_pcie_inclusive": round(iters / call_s, 2),
                                     "records_prepare_s": round(float(rec[7]), 4), "records_loop_s": round(float(rec[8]), 4),
                                     # mil_decon_last_call_split: where the call's seconds went (pageable host buffers)
                                     "split_s": {"engine_and_buffers": round(split[0], 5), "otfs": round(split[1], 5),
                                                 "h2d_pad_init": round(split[2], 5), "rl_loop": round(split[3], 5),
                                                 "crop_d2h": round(split[4], 5)},
                                     "h2d_gbs": round(img_h.nbytes / max(split[2], 1e-9) / 1e9, 1),
                                     "d2h_gbs": round(out_h.nbytes / max(split[4], 1e-9) / 1e9, 1)}
            result["extra"]["libapi_decon_singleview_host_pointers"] = dict(calls[str(args.steps)], by_iterations=calls)
            del img_h, out_h
        line.run("libapi_host_pointers", host_pointer_extra)
        line.run("auxiliary_kernels",
                 lambda: result["extra"].__setitem__("auxiliary_kernels", auxiliary_kernels(torch, mdev, eng, img_a, psf_a)))

    if world == 1 and not args.no_cpu_baseline and rank == 0:
        result["cpu_baseline"] = None

        def cpu_extra():
            img_host = img_a.cpu().numpy()
            torch.cuda.empty_cache()
            its, secs, whole = cpu_baseline(img_host, psf_a_host, args.cpu_iters, 1)
            result["cpu_baseline"] = {
                "value": round(its, 4),
                "unit": "iterations/s",
                "cores": 1,
                "kind": "port",
                "implementation": "libapi.so decon_singleview(gpuMemMode 0): C++ restatement of the reference's CPU twin "
                                  "(decon_singleview_OTF0 order of operations, separate single-op element-wise passes), "
                                  "bundled mixed-radix FFT (FFTW is not on this box)",
                "sample": f"{args.cpu_iters} RL iterations of the same 512x512x256 volume and PSF on 1 thread like the "
                          f"reference: {secs:.1f} s in the loop, {whole:.1f} s for the whole call (OTFs included)",
                "host_cores_available": os.cpu_count(),
            }
            n_thr = min(args.cpu_threads, os.cpu_count() or 1)
            if n_thr > 1:
                it_mt = 3 * args.cpu_iters
                its_mt, secs_mt, whole_mt = cpu_baseline(img_host, psf_a_host, it_mt, n_thr)
                result["cpu_baseline"]["all_cores"] = {
                    "value": round(its_mt, 4), "unit": "iterations/s", "cores": n_thr,
                    "sample": f"{it_mt} iterations of the same call with MIL_CPU_THREADS={n_thr} (every pass threaded): "
                              f"{secs_mt:.1f} s in the loop, {whole_mt:.1f} s whole"}
        line.run("cpu_baseline", cpu_extra)
    elif rank == 0:
        result["cpu_baseline"] = None

    line.stage = "shutdown"
    line.emit()
    try:
        eng.close()
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:  # the line is out; a peer that died must not turn into a non-zero status here
        print("bench.py: shutdown:", repr(e), file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
