"""Time-lapse batch: the spimFusionBatch loop sharded one volume pair per GPU.

The reference processes time points strictly one after another on one device
(src/spim_fusion_batch.cpp:613-966): resize / rotate, register view B to view A
according to ``regMode``, joint dual-view RL, write the result, 2-D and 3-D MIPs.
Time points are independent except in ``regMode 2`` (the starting matrix of time
point t is a running 0.8 / 0.2 average of the earlier results, :786-808).

Here: one process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests), time points dealt round-robin, OTFs computed
once per rank and reused for all its time points.  There is NO collective on the data
path; the only exchange is

  * regMode 1: the matrix found on the test time point is broadcast from rank 0
    (12 floats), mirroring "register one, apply to all" (:719-751);
  * the final gather of the per-time-point maximum-intensity projections to rank 0: the XY MIP alone, or all five
    products the reference writes per time point (XY, YZ, ZX and the 36-angle 3-D MIPs about x and y), device-resident,
    one record per time point, ONE gather (``gather_mip_stack``).

``regMode 2`` carries a loop dependence between time points: on several ranks the chain state
(two matrices, 24 floats) is passed from the owner of one time point to the owner of the next, point
to point; each owner registers, hands the state on, then deconvolves -- registrations serial, everything
else overlapped (``_run_chained``).  All volume arithmetic happens in ``libapi.so``; this module is host logic.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import threading
import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.distributed as dist


# --------------------------------------------------------------------------- sharding / exchange
def shard_time_points(time_points: Sequence[int], rank: int, world: int) -> List[int]:
    """Round-robin deal: rank r gets time_points[r], [r + world], ...  (1 volume pair per GPU at a time)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    return list(time_points[rank::world])


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def broadcast_matrix(tmx: Optional[np.ndarray], src: int = 0, device="cpu", group=None) -> np.ndarray:
    """regMode 1: every rank applies the matrix registered on the test time point (12 floats from `src`)."""
    rank, world = _world(group)
    t = torch.zeros(12, dtype=torch.float32, device=device)
    if rank == src:
        t.copy_(torch.as_tensor(np.asarray(tmx, dtype=np.float32).reshape(12)))
    if world > 1:
        dist.broadcast(t, src=src, group=group)
    return t.cpu().numpy()


MipShape = Union[Tuple[int, ...], Dict[str, Tuple[int, ...]]]
# what the last gather_mip_stack of this process moved (bench.py reports it): seconds include waiting for the slowest rank
last_gather = {"seconds": 0.0, "bytes_per_time_point": 0, "bytes_received_rank0": 0, "products": []}


def _products(mip_shape: MipShape):
    """[(name, shape, floats)] of the projection products one time point contributes, in gather order."""
    if isinstance(mip_shape, dict):
        items = [(k, tuple(int(v) for v in shp)) for k, shp in mip_shape.items()]
    else:
        items = [(None, tuple(int(v) for v in mip_shape))]
    return [(k, shp, int(np.prod(shp))) for k, shp in items]


def _clone_mips(m):
    return {k: v.clone() for k, v in m.items()} if isinstance(m, dict) else m.clone()


def gather_mip_stack(local: Dict[int, object], time_points: Sequence[int], mip_shape: MipShape,
                     device="cpu", group=None):
    """Gather the per-time-point projections into time-ordered stacks on rank 0 -- the ONE exchange of the batch.

    `mip_shape` is either one shape (rank-local dict t -> tensor: the XY MIP) or a dict product -> shape (rank-local dict
    t -> {product: tensor}): the reference writes five products per time point (src/spim_fusion_batch.cpp:915-940: XY, YZ,
    ZX maximum projections and the 36-angle 3-D MIPs about x and y, ~43 MB at 512x512x256).  All products of a time point
    travel as ONE float32 record; every rank contributes a fixed-size slab (ceil(n / world) records, unused ones zero) so a
    single ``gather`` moves everything and rank 0 re-orders the records into time order.  Returns [n, *shape] (or
    {product: [n, *shape]}) on rank 0, None elsewhere."""
    rank, world = _world(group)
    time_points = list(time_points)
    n = len(time_points)
    prods = _products(mip_shape)
    rec = sum(c for _, _, c in prods)
    slots = (n + world - 1) // world
    mine = shard_time_points(time_points, rank, world)
    t0 = time.perf_counter()
    slab = torch.zeros((max(slots, 1), rec), dtype=torch.float32, device=device)
    for i, t in enumerate(mine):
        off = 0
        for name, shp, c in prods:
            src = local[t] if name is None else local[t][name]
            if tuple(src.shape) != shp:
                raise ValueError(f"time point {t}: product {name or 'mip'} has shape {tuple(src.shape)}, expected {shp}")
            slab[i, off:off + c].copy_(src.reshape(-1))
            off += c
    if world == 1:
        parts = [slab]
    else:
        parts = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
        dist.gather(slab, gather_list=parts, dst=0, group=group)
    last_gather.update(seconds=time.perf_counter() - t0, bytes_per_time_point=4 * rec,
                       bytes_received_rank0=4 * rec * (n - len(mine)) if rank == 0 else 0,
                       products=[name or "xy" for name, _, _ in prods])
    if rank != 0:
        return None
    out = {name: torch.empty((n,) + shp, dtype=torch.float32, device=device) for name, shp, _ in prods}
    index = {t: k for k, t in enumerate(time_points)}
    del slab  # world > 1: rank 0's own contribution is parts[0] by now
    for r in range(world):
        part, parts[r] = parts[r], None  # each received slab is released as soon as its records are in place: rank 0 holds
        for i, t in enumerate(shard_time_points(time_points, r, world)):  # the stacks + ONE slab, not the stacks twice
            off = 0
            for name, shp, c in prods:
                out[name][index[t]].copy_(part[i, off:off + c].reshape(shp))
                off += c
        del part
    last_gather["seconds"] = time.perf_counter() - t0
    return out if isinstance(mip_shape, dict) else out[None]


# --------------------------------------------------------------------------- registration policy
@dataclass
class RegPolicy:
    """The reference's per-time-point registration logic (src/spim_fusion_batch.cpp:714-843)."""
    reg_mode: int = 0           # 0 apply input matrix | 1 register test point, apply to all | 2 chained | 3 independent
    reg_choice: int = 2         # reg3d regChoice for the first attempt
    aff_method: int = 6         # batch hard-codes 6 (:557)
    ftol: float = 1e-4
    it_limit: int = 3000
    cost_bar: float = 0.1       # ZNCC threshold of the retry heuristics (costValueBar)
    i_tmx: Optional[np.ndarray] = None   # input matrix (flagiTmx)
    test_time_point: Optional[int] = None
    # state of regMode 2
    weighted: Optional[np.ndarray] = field(default=None, repr=False)
    previous: Optional[np.ndarray] = field(default=None, repr=False)


RegFn = Callable[..., Tuple[object, np.ndarray, np.ndarray]]
# reg_fn(target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit) -> (registered, tmx[12], records[11])
CheckFn = Callable[[np.ndarray], bool]


def register_pair(policy: RegPolicy, reg_fn: RegFn, check_fn: CheckFn, target, source, first: bool):
    """One time point's registration with the reference's fall-back ladder.  Returns (registered, tmx, records)."""
    p = policy
    have_input = p.i_tmx is not None

    def attempt(choice, tmx, use_input):
        return reg_fn(target, source, tmx, choice, p.aff_method, use_input, p.ftol, p.it_limit)

    def bad(tmx, rec):
        return (not check_fn(tmx)) or (rec[3] < p.cost_bar)

    if p.reg_mode == 0:  # regChoice forced to 0: apply the input matrix, or copy (:614-616, :717-721)
        return attempt(0, p.i_tmx, have_input)
    if p.reg_mode in (1, 3) or (p.reg_mode == 2 and first):
        reg, tmx, rec = attempt(p.reg_choice, p.i_tmx, have_input)
        if bad(tmx, rec):  # change scheme and redo (:730-735, :762-767, :815-820)
            alt = 2 if p.reg_choice == 4 else 4
            reg, tmx, rec = attempt(alt, None, False)
            if bad(tmx, rec) and have_input:  # fall back to the input matrix (:736-746)
                reg, tmx, rec = attempt(0, p.i_tmx, True)
        if p.reg_mode == 2:
            p.weighted = np.array(tmx, dtype=np.float32)
            p.previous = np.array(tmx, dtype=np.float32)
        return reg, tmx, rec
    if p.reg_mode == 2:  # start from the running average, fall back to the previous matrix (:786-808)
        reg, tmx, rec = attempt(2, p.weighted, True)
        if bad(tmx, rec):
            reg, tmx, rec = attempt(0, p.previous, True)
        # double arithmetic, stored as float (src/spim_fusion_batch.cpp:805-807)
        p.weighted = (0.8 * p.weighted.astype(np.float64) + 0.2 * np.asarray(tmx, dtype=np.float64)).astype(np.float32)
        p.previous = np.array(tmx, dtype=np.float32)
        return reg, tmx, rec
    raise ValueError(f"regMode {p.reg_mode} is not one of 0, 1, 2, 3")


# --------------------------------------------------------------------------- the sharded loop
def run_sharded(time_points: Sequence[int], load_pair: Callable[[int], Tuple[object, object]],
                fuse_pair: Callable[[object, object, int], Tuple[object, torch.Tensor]],
                policy: RegPolicy, reg_fn: RegFn, check_fn: CheckFn, mip_shape: Tuple[int, int],
                sink: Optional[Callable[[int, object, np.ndarray, np.ndarray], None]] = None,
                device="cpu", group=None, overlap: bool = False, lanes: int = 1, control_group=None,
                control_device=None) -> Optional[torch.Tensor]:
    """Run the batch over this rank's share of `time_points`.

    load_pair(t) -> (view A, view B) already resized / rotated; fuse_pair(A, B_registered, t) -> (decon, XY MIP);
    sink(t, decon, tmx, reg_records) persists per-time-point outputs (rank-local, no exchange).
    overlap: deconvolution + sink of time point i run on a helper thread while this thread registers time point i + 1
    (independent streams on the GPU: the deconvolution fills the gaps between Powell's cost evaluations), and
    load_pair(i + 2) runs on a second helper thread (file decode + upload hidden behind the registration).  BUFFER
    CONTRACT with overlap: THREE pairs are live at once -- the pair of time point i is still being read by the
    deconvolution worker while the pair of i + 1 is registered and load_pair(i + 2) is already filling its buffers -- so
    load_pair must hand out buffers that stay valid for the next TWO calls (rotate at least three sets, or allocate), and
    reg_fn must return a registered volume that does not alias the one it returned for the previous time point (the worker
    reads b_reg of i while i + 1 is registered).  fuse_pair may reuse one output buffer: one deconvolution is in flight at
    a time and its MIP is cloned before the next starts.  Results are the same, only the order of host-side side effects
    within a rank changes.  Modes 0, 1, 3, and mode 2 on ONE rank (the chain orders the registrations, which stay on the calling
    thread; across ranks the token ring does the overlapping).
    lanes > 1 (modes 0, 1, 3): this rank works on `lanes` time points AT ONCE, one host thread each -- the same idea as the
    command-line tool's worker threads (cli/spim_fusion_batch_main.cpp: a device listed twice).  A registration is a chain
    of ~10^3 dependent cost evaluations with a host round trip between any two, so one time point alone leaves the GPU idle
    in the gaps; a second time point's kernels fill them.  `fuse_pair` must then be a SEQUENCE of `lanes` callables (each
    with its own engine / estimate buffer: FusionPipeline per lane); `load_pair`, `reg_fn`, `check_fn` and `sink` are
    called from several threads.  Every time point is processed exactly as with one lane: same results.
    control_group / control_device: where the few floats of host-side state travel (the regMode-1 matrix, the regMode-2 chain
    token): by default the same group / device as the gather.  On the GPU node the driver passes a gloo group and "cpu", so
    that RCCL carries the gather of the projections and nothing else (north_star: "RCCL ... used only to gather the final MIP
    stack") and a 24-float token never waits behind a device-side point-to-point kernel.
    Before the gather the ranks AGREE over the control group that all of them got there (`_agree`): a rank whose time point
    raised reports it and every rank raises, instead of the healthy ones waiting in the gather for a peer that will never
    come.  Returns the time-ordered MIP stack on rank 0."""
    rank, world = _world(group)
    time_points = list(time_points)
    if control_group is None:
        control_group = group
    if control_device is None:
        control_device = device if control_group is group else "cpu"
    failure = None
    local = None
    try:
        if policy.reg_mode == 2 and world > 1:
            local = _run_chained(time_points, load_pair, fuse_pair, policy, reg_fn, check_fn, mip_shape, sink, device, group,
                                 control_group, control_device)
        else:
            local = _run_local(time_points, load_pair, fuse_pair, policy, reg_fn, check_fn, sink, device, group, overlap,
                               lanes, control_group, control_device)
    except Exception as e:  # reported to the peers first, raised below
        failure = e
    _agree(failure, control_group, control_device)
    return gather_mip_stack(local, time_points, mip_shape, device=device, group=group)


def _agree(failure, group, device):
    """All ranks or none: MIN all-reduce of an "I am fine" flag over the control group (gloo on the GPU node: its timeout
    turns a dead peer into an exception).  Raises this rank's own failure, or a RuntimeError naming the situation."""
    rank, world = _world(group)
    if world > 1:
        flag = torch.tensor([0.0 if failure is not None else 1.0], dtype=torch.float32, device=device)
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        except Exception:
            if failure is not None:
                raise failure
            raise
        if failure is None and float(flag.item()) < 1.0:
            raise RuntimeError("batch: another rank failed in its share of the time points; nothing was gathered")
    if failure is not None:
        raise failure


def _run_local(time_points, load_pair, fuse_pair, policy, reg_fn, check_fn, sink, device, group, overlap, lanes,
               control_group, control_device):
    """run_sharded without the token ring: this rank's share, `local` = time point -> projections."""
    rank, world = _world(group)
    if policy.reg_mode == 1:
        # register the test time point once (rank 0), then every rank applies that matrix (:719-751)
        tmx = None
        if rank == 0:
            t_test = policy.test_time_point if policy.test_time_point is not None else time_points[0]
            a, b = load_pair(t_test)
            _, tmx, _ = register_pair(policy, reg_fn, check_fn, a, b, first=True)
        tmx = broadcast_matrix(tmx, 0, device=control_device, group=control_group)
        policy = RegPolicy(reg_mode=0, reg_choice=0, aff_method=policy.aff_method, ftol=policy.ftol,
                           it_limit=policy.it_limit, cost_bar=policy.cost_bar, i_tmx=tmx)
    local: Dict[int, torch.Tensor] = {}
    if lanes > 1:
        if policy.reg_mode == 2:
            raise ValueError("lanes > 1 needs independent time points (regMode 0, 1 or 3)")
        fuse_list = list(fuse_pair)
        if len(fuse_list) != lanes:
            raise ValueError("lanes > 1: fuse_pair must be a sequence of one callable per lane")
        import concurrent.futures
        import dataclasses
        mine = shard_time_points(time_points, rank, world)
        dev_index = torch.cuda.current_device() if torch.cuda.is_available() else None

        stop = threading.Event()  # set by the first lane that fails: the others finish their time point and return

        def lane(l):
            if dev_index is not None:
                torch.cuda.set_device(dev_index)  # the current device is per thread
            pol = dataclasses.replace(policy)
            try:
                for j, t in enumerate(mine[l::lanes]):
                    if stop.is_set():
                        return
                    a, b = load_pair(t)
                    b_reg, tmx, rec = register_pair(pol, reg_fn, check_fn, a, b, first=(j == 0))
                    decon, mip = fuse_list[l](a, b_reg, t)
                    local[t] = _clone_mips(mip)
                    if sink is not None:
                        sink(t, decon, np.asarray(tmx, dtype=np.float32), np.asarray(rec, dtype=np.float32))
            except BaseException:
                stop.set()
                raise

        # A failing lane stops the others at their next time point; its exception is raised once all lanes have drained.
        # Under torch.distributed the healthy RANKS would then wait in the gather for ever: a rank that raises here must
        # take the process group down (let the exception end the process -- torch.distributed.run then stops the others).
        with concurrent.futures.ThreadPoolExecutor(max_workers=lanes) as pool:
            futures = [pool.submit(lane, l) for l in range(lanes)]
            concurrent.futures.wait(futures, return_when=concurrent.futures.FIRST_EXCEPTION)
            stop_now = any(f.done() and f.exception() is not None for f in futures)
            if stop_now:
                stop.set()
            concurrent.futures.wait(futures)
            for f in futures:
                f.result()
        return local

    def finish(a, b_reg, t, tmx, rec):
        decon, mip = fuse_pair(a, b_reg, t)
        local[t] = _clone_mips(mip) if overlap else mip
        if sink is not None:
            sink(t, decon, np.asarray(tmx, dtype=np.float32), np.asarray(rec, dtype=np.float32))

    worker, loader, pending, ahead = None, None, None, None
    mine = shard_time_points(time_points, rank, world)
    if overlap:
        import concurrent.futures
        dev_index = torch.cuda.current_device() if torch.cuda.is_available() else None

        def init_worker():  # the current device is per thread
            if dev_index is not None:
                torch.cuda.set_device(dev_index)
        worker = concurrent.futures.ThreadPoolExecutor(max_workers=1, initializer=init_worker)
        loader = concurrent.futures.ThreadPoolExecutor(max_workers=1, initializer=init_worker)
        if mine:
            ahead = loader.submit(load_pair, mine[0])
    try:
        for i, t in enumerate(mine):
            if loader is None:
                a, b = load_pair(t)
            else:
                a, b = ahead.result()
                ahead = loader.submit(load_pair, mine[i + 1]) if i + 1 < len(mine) else None
            b_reg, tmx, rec = register_pair(policy, reg_fn, check_fn, a, b, first=(i == 0))
            if worker is None:
                finish(a, b_reg, t, tmx, rec)
            else:
                if pending is not None:
                    pending.result()  # one deconvolution in flight: the pipeline owns one estimate buffer
                pending = worker.submit(finish, a, b_reg, t, tmx, rec)
        if pending is not None:
            pending.result()
    finally:
        for ex in (worker, loader):
            if ex is not None:
                ex.shutdown(wait=True)
    return local


def _run_chained(time_points, load_pair, fuse_pair, policy, reg_fn, check_fn, mip_shape, sink, device, group,
                 control_group=None, control_device=None):
    """regMode 2 on several ranks: a token ring.  The matrix chain (running 0.8 / 0.2 average, fall-back to the previous
    matrix, src/spim_fusion_batch.cpp:786-808) is inherently sequential: time point i cannot register before i - 1 has.
    The chain STATE -- the weighted and the previous matrix, 24 floats -- travels from the owner of time point i - 1 to the
    owner of time point i (ranks own time points round-robin); the owner, which has already loaded its pair, registers
    it, passes the state on and only then deconvolves.  So every pair is loaded once, no volume ever crosses ranks, every
    rank does the same kind of work, and the deconvolution of time point i overlaps with the registration of i + 1 on
    the next rank.  Registrations stay serial (the reference's semantics), results equal the single-rank run's.
    Throughput bound with registration share r of a time point: 1 / (r + (1 - r) / N) of one GPU's."""
    rank, world = _world(group)
    if control_group is None:
        control_group, control_device = group, device
    n = len(time_points)
    local: Dict[int, torch.Tensor] = {}
    sends = []
    for i, t in enumerate(time_points):
        if i % world != rank:
            continue
        a, b = load_pair(t)  # before the token arrives: overlaps with the registrations upstream
        if i > 0:
            state = torch.zeros(24, dtype=torch.float32, device=control_device)
            dist.recv(state, src=(i - 1) % world, group=control_group)
            got = state.cpu().numpy()
            policy.weighted, policy.previous = got[:12].copy(), got[12:].copy()
        b_reg, tmx, rec = register_pair(policy, reg_fn, check_fn, a, b, first=(i == 0))
        if i + 1 < n:
            state = torch.zeros(24, dtype=torch.float32, device=control_device)
            state[:12].copy_(torch.from_numpy(np.asarray(policy.weighted, dtype=np.float32).reshape(12)))
            state[12:].copy_(torch.from_numpy(np.asarray(policy.previous, dtype=np.float32).reshape(12)))
            sends.append((dist.isend(state, dst=(i + 1) % world, group=control_group), state))  # keep the tensor alive
        decon, mip = fuse_pair(a, b_reg, t)
        local[t] = mip
        if sink is not None:
            sink(t, decon, np.asarray(tmx, dtype=np.float32), np.asarray(rec, dtype=np.float32))
    for req, _ in sends:
        req.wait()
    return local


# --------------------------------------------------------------------------- GPU pipeline (device-resident)
class FusionPipeline:
    """reg3d -> decon_dualview -> XY MIP on device-resident volumes of one GPU (libapi.so kernels only)."""

    def __init__(self, shape, psf_a: torch.Tensor, psf_b: torch.Tensor, iters: int, psf_a_bp: Optional[torch.Tensor] = None,
                 psf_b_bp: Optional[torch.Tensor] = None, mips: str = "xy", project_num: int = 36):
        """mips = "xy": fuse_pair returns the XY MIP (a tensor); "all": the five projection products the reference writes
        per time point (src/spim_fusion_batch.cpp:915-940, projectNum 36 as it hard-codes) as a dict, see `mip_shape`."""
        from . import device as mdev
        self._mdev = mdev
        self.shape = tuple(int(s) for s in shape)
        self.iters = int(iters)
        if mips not in ("xy", "all"):
            raise ValueError("mips must be 'xy' or 'all'")
        self.mips, self.project_num = mips, int(project_num)
        self.engine = mdev.DeconEngine(self.shape)
        # four OTFs once per rank (the reference recomputes them for every time point, src/api_decon.cpp:524-547)
        self.otf_a = self.engine.gen_otf(psf_a)
        self.otf_b = self.engine.gen_otf(psf_b)
        self.bp_a = self.engine.gen_otf(psf_a_bp) if psf_a_bp is not None else self.engine.gen_otf(psf_a, flip=True)
        self.bp_b = self.engine.gen_otf(psf_b_bp) if psf_b_bp is not None else self.engine.gen_otf(psf_b, flip=True)
        self.estimate = torch.empty(self.shape, dtype=torch.float32, device=psf_a.device)

    def reg_fn(self, target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit):
        """reg3d on device volumes: regChoice 0 (apply / copy), 2 (affine) and 4 (2-D MIP shift search, then
        affine from that translation: src/api_reg.cpp:448-497), 1 / 3 (phasor shift; 3 continues with affine)."""
        method = 0 if reg_choice == 0 else aff_method
        if reg_choice in (1, 3):
            sx, sy, sz = (float(v) for v in self._mdev.reg3d_phasor_shift(target, source))
            i_tmx, input_tmx = np.array([1, 0, 0, sx, 0, 1, 0, sy, 0, 0, 1, sz], dtype=np.float32), True
            if reg_choice == 1:  # the shifted source is the result (src/api_reg.cpp:418-427)
                reg = self._mdev.shift_volume(source, -int(sx), -int(sy), -int(sz), wrap=False)
                return reg, i_tmx, np.zeros(11, dtype=np.float32)
        if reg_choice == 4:
            sx, sy, sz = self._mdev.reg3d_mip_shift(target, source)
            i_tmx, input_tmx = np.array([1, 0, 0, sx, 0, 1, 0, sy, 0, 0, 1, sz], dtype=np.float32), True
        return self._mdev.reg3d_affine(target, source, aff_method=method, i_tmx=i_tmx, flag_tmx=bool(input_tmx),
                                       ftol=ftol, it_limit=it_limit)

    def check_fn(self, tmx) -> bool:
        from . import api
        nz, ny, nx = self.shape
        return api.checkmatrix(tmx, nx, ny, nz)

    def fuse_pair(self, a: torch.Tensor, b_reg: torch.Tensor, t: int):
        A, B = a.clone(), b_reg.clone()  # the engine clamps its inputs in place
        self.engine.init_dual(A, B, self.estimate)
        self.engine.run_dual(self.estimate, A, B, self.otf_a, self.otf_b, self.bp_a, self.bp_b, self.iters)
        self.engine.sync()
        return self.estimate, self.project(self.estimate)

    @property
    def mip_shape(self):
        """What `gather_mip_stack` needs: the XY shape, or product -> shape for mips="all" (mp2dgpu / mip3dgpu layouts,
        src/apifunc.cpp:485-505, 576-644)."""
        nz, ny, nx = self.shape
        if self.mips == "xy":
            return (ny, nx)
        rot_x = int(round(np.sqrt(float(ny * ny + nz * nz))))  # rotation about x: the y-z diagonal
        rot_y = int(round(np.sqrt(float(nx * nx + nz * nz))))
        return {"xy": (ny, nx), "yz": (nz, ny), "zx": (nx, nz), "mip3d_x": (self.project_num, rot_x, nx),
                "mip3d_y": (self.project_num, ny, rot_y)}

    def project(self, vol: torch.Tensor):
        """The projections of one deconvolved volume, device-resident."""
        if self.mips == "xy":
            return self._mdev.max_projection(vol, 1)
        return {"xy": self._mdev.max_projection(vol, 1), "yz": self._mdev.max_projection(vol, 3),
                "zx": self._mdev.max_projection(vol, 2), "mip3d_x": self._mdev.mip3d(vol, 1, self.project_num),
                "mip3d_y": self._mdev.mip3d(vol, 2, self.project_num)}

    def close(self):
        self.engine.close()
