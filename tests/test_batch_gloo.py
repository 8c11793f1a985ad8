"""Host logic of the sharded time-lapse batch (microimagelib_amd/batch.py) on CPU:
world_size-2 ``gloo`` runs of the sharding, the regMode-1 matrix broadcast and the MIP gather,
plus the reference's registration fall-back ladder (src/spim_fusion_batch.cpp:714-843).
No volume arithmetic happens here: the compute callbacks are stand-ins, the GPU pipeline
behind them is covered by the -m gpu tests."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from microimagelib_amd import batch

MIP_SHAPE = (4, 6)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_reg(log):
    def reg_fn(target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit):
        tmx = np.zeros(12, dtype=np.float32)
        if reg_choice == 0:
            if input_tmx:
                tmx[:] = np.asarray(i_tmx, dtype=np.float32)
            else:
                tmx[[0, 5, 10]] = 1
        else:  # "registration": a matrix that encodes the time point it was computed on
            tmx[[0, 5, 10]] = 1
            tmx[3] = float(target)
        rec = np.zeros(11, dtype=np.float32)
        rec[3] = 0.9
        log.append((int(target), int(reg_choice), bool(input_tmx)))
        return source, tmx, rec
    return reg_fn


def _worker(rank, world, port, outdir, reg_mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tps = [3, 4, 5, 6, 7]
        log, sunk = [], {}
        policy = batch.RegPolicy(reg_mode=reg_mode, reg_choice=2, test_time_point=5)

        def load_pair(t):
            return t, t  # stand-in volumes: the time point itself

        def fuse_pair(a, b_reg, t):
            return None, torch.full(MIP_SHAPE, float(10 * t + 1))

        def sink(t, decon, tmx, rec):
            sunk[t] = tmx.tolist()

        err = None
        stack = None
        try:
            stack = batch.run_sharded(tps, load_pair, fuse_pair, policy, _fake_reg(log), lambda m: True, MIP_SHAPE,
                                      sink=sink)
        except ValueError as e:
            err = str(e)
        json.dump({"log": log, "sunk": sunk, "err": err, "stack": None if stack is None else stack[:, 0, 0].tolist(),
                   "stack_shape": None if stack is None else list(stack.shape)},
                  open(os.path.join(outdir, f"rank{rank}.json"), "w"))
    finally:
        dist.destroy_process_group()


def _run(tmp_path, reg_mode, world=2):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), reg_mode), nprocs=world, join=True)
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


def test_shard_is_a_partition():
    tps = list(range(1, 65))
    for world in (1, 2, 3, 8):
        parts = [batch.shard_time_points(tps, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == tps
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        batch.shard_time_points(tps, 2, 2)


@pytest.mark.timeout(120)
def test_independent_registration_two_ranks(tmp_path):
    """regMode 3: every time point registered on the rank that owns it; MIPs gathered in time order."""
    r0, r1 = _run(tmp_path, reg_mode=3)
    assert sorted(int(t) for t in r0["sunk"]) == [3, 5, 7] and sorted(int(t) for t in r1["sunk"]) == [4, 6]
    assert r1["stack"] is None
    assert r0["stack_shape"] == [5, *MIP_SHAPE]
    assert r0["stack"] == [31.0, 41.0, 51.0, 61.0, 71.0]
    for r in (r0, r1):  # each owned time point registered once with regChoice 2, no input matrix
        assert all(c == 2 and not inp for (_, c, inp) in r["log"])
        for t, m in r["sunk"].items():
            assert m[3] == float(t)


@pytest.mark.timeout(120)
def test_register_once_apply_to_all_two_ranks(tmp_path):
    """regMode 1: rank 0 registers the test time point, the 12 floats are broadcast, every rank applies them."""
    r0, r1 = _run(tmp_path, reg_mode=1)
    assert r0["log"][0] == [5, 2, False]  # the registration itself, on the test time point
    for r in (r0, r1):
        applied = [e for e in r["log"] if e[1] == 0]
        assert len(applied) == len(r["sunk"]) and all(inp for (_, _, inp) in applied)
        for m in r["sunk"].values():
            assert m[3] == 5.0 and m[0] == 1.0  # the matrix found on time point 5
    assert r0["stack"] == [31.0, 41.0, 51.0, 61.0, 71.0]


@pytest.mark.timeout(120)
def test_chained_mode_two_ranks_matches_single_rank(tmp_path):
    """regMode 2: the chain state travels from owner to owner (token ring); every rank registers ITS time points, in
    global order, each pair is loaded and registered exactly once; matrices and MIP stack equal the single-rank run."""
    r0, r1 = _run(tmp_path, reg_mode=2)
    assert r0["err"] is None and r1["err"] is None
    # time points 3, 5, 7 on rank 0 and 4, 6 on rank 1: the first with the configured scheme and no input matrix, the
    # others chained (regChoice 2 starting from the running average they were handed)
    assert r0["log"] == [[3, 2, False], [5, 2, True], [7, 2, True]]
    assert r1["log"] == [[4, 2, True], [6, 2, True]]
    assert sorted(int(t) for t in r0["sunk"]) == [3, 5, 7] and sorted(int(t) for t in r1["sunk"]) == [4, 6]
    (tmp_path / "single").mkdir()
    (s0,) = _run(tmp_path / "single", reg_mode=2, world=1)
    merged = {**r0["sunk"], **r1["sunk"]}
    assert merged == s0["sunk"]
    assert r0["stack"] == s0["stack"] == [31.0, 41.0, 51.0, 61.0, 71.0] and r1["stack"] is None


@pytest.mark.timeout(120)
def test_chained_mode_three_ranks_carries_the_running_average(tmp_path):
    """The state handed on is the reference's: weighted = 0.8 * weighted + 0.2 * result (double arithmetic stored as
    float), previous = result -- checked through the start matrices a three-rank ring registers from."""
    rs = _run(tmp_path, reg_mode=2, world=3)
    (tmp_path / "single").mkdir()
    (s0,) = _run(tmp_path / "single", reg_mode=2, world=1)
    merged = {}
    for r in rs:
        assert r["err"] is None
        merged.update(r["sunk"])
    assert merged == s0["sunk"] and rs[0]["stack"] == s0["stack"]
    assert [len(r["log"]) for r in rs] == [2, 2, 1]  # 5 time points dealt 3, 4, 5, 6, 7 -> ranks 0, 1, 2, 0, 1


@pytest.mark.parametrize("reg_mode", [3, 2])
def test_overlapped_pipeline_gives_the_same_results(reg_mode):
    """(regMode 2 on one rank: the chain orders the registrations, which stay on the calling thread.)  overlap=True: fuse + sink of time point i on a helper thread while i + 1 registers -- same matrices, same MIPs,
    every time point sunk exactly once, and the helper really ran concurrently with a registration."""
    import threading
    import time as _time

    def run(overlap):
        log, sunk, seen = [], {}, {"concurrent": False, "registering": False}

        def reg_fn(target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit):
            seen["registering"] = True
            _time.sleep(0.02)
            seen["registering"] = False
            tmx = np.eye(3, 4, dtype=np.float32).reshape(12)
            tmx[3] = float(target)
            rec = np.zeros(11, dtype=np.float32)
            rec[3] = 0.9
            log.append(int(target))
            return source, tmx, rec

        def fuse_pair(a, b_reg, t):
            for _ in range(20):
                if seen["registering"]:
                    seen["concurrent"] = True
                _time.sleep(0.002)
            return None, torch.full(MIP_SHAPE, float(10 * t + 1))

        def sink(t, decon, tmx, rec):
            assert t not in sunk
            sunk[t] = (tmx.tolist(), threading.current_thread().name)
        stack = batch.run_sharded([3, 4, 5, 6], lambda t: (t, t), fuse_pair, batch.RegPolicy(reg_mode=reg_mode, reg_choice=2),
                                  reg_fn, lambda m: True, MIP_SHAPE, sink=sink, overlap=overlap)
        return log, sunk, stack, seen["concurrent"]

    log0, sunk0, stack0, conc0 = run(False)
    log1, sunk1, stack1, conc1 = run(True)
    assert log0 == log1 == [3, 4, 5, 6]
    assert {t: v[0] for t, v in sunk0.items()} == {t: v[0] for t, v in sunk1.items()}
    assert torch.equal(stack0, stack1)
    assert not conc0 and conc1
    assert all(v[1] == threading.current_thread().name for v in sunk0.values())
    assert all(v[1] != threading.current_thread().name for v in sunk1.values())


def test_fallback_ladder_matches_reference():
    """Low ZNCC on the first attempt -> other scheme without input matrix -> input matrix (regChoice 0)."""
    calls = []

    def reg_fn(target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit):
        calls.append((reg_choice, input_tmx))
        rec = np.zeros(11, dtype=np.float32)
        rec[3] = 0.05 if reg_choice != 0 else 0.5
        tmx = np.eye(3, 4, dtype=np.float32).reshape(12) if i_tmx is None else np.asarray(i_tmx, dtype=np.float32)
        return source, tmx, rec

    given = (np.eye(3, 4, dtype=np.float32) * 1.01).reshape(12)
    pol = batch.RegPolicy(reg_mode=3, reg_choice=2, i_tmx=given)
    _, tmx, rec = batch.register_pair(pol, reg_fn, lambda m: True, 0, 0, first=True)
    assert calls == [(2, True), (4, False), (0, True)]
    assert np.allclose(tmx, given) and rec[3] == 0.5
    # regChoice 4 first -> the alternative scheme is 2
    calls.clear()
    batch.register_pair(batch.RegPolicy(reg_mode=3, reg_choice=4), reg_fn, lambda m: True, 0, 0, first=True)
    assert calls == [(4, False), (2, False)]


def test_chained_mode_running_average_single_rank():
    """regMode 2 on one rank: matrix of time point t starts from 0.8*weighted + 0.2*previous result."""
    starts = []

    def reg_fn(target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit):
        starts.append(None if i_tmx is None else float(np.asarray(i_tmx)[3]))
        tmx = np.eye(3, 4, dtype=np.float32).reshape(12)
        tmx[3] = float(target)  # "found" shift = time point
        rec = np.zeros(11, dtype=np.float32)
        rec[3] = 0.9
        return source, tmx, rec

    pol = batch.RegPolicy(reg_mode=2, reg_choice=2)
    stack = batch.run_sharded([10, 20, 30], lambda t: (t, t), lambda a, b, t: (None, torch.zeros(MIP_SHAPE)), pol,
                              reg_fn, lambda m: True, MIP_SHAPE)
    assert stack.shape == (3, *MIP_SHAPE)
    assert starts[0] is None and starts[1] == 10.0 and abs(starts[2] - (0.8 * 10 + 0.2 * 20)) < 1e-5


def test_lanes_process_time_points_concurrently_with_the_same_results():
    """run_sharded(lanes=2): two time points in flight per rank, one host thread each, each with its own fuse callable;
    matrices, sinks and the MIP stack are those of the one-lane run."""
    import threading
    tps = list(range(20, 31))
    results = {}
    for lanes in (1, 2, 3):
        log, sunk, seen_threads = [], {}, set()
        lock = threading.Lock()

        def reg_fn(target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit):
            tmx = np.zeros(12, dtype=np.float32)
            tmx[[0, 5, 10]] = 1
            tmx[3] = float(target) * 0.5
            rec = np.zeros(11, dtype=np.float32)
            rec[3] = 0.9
            with lock:
                log.append(int(target))
                seen_threads.add(threading.get_ident())
            import time
            time.sleep(0.002)  # long enough for the lanes to overlap
            return source, tmx, rec

        def make_fuse(l):
            def fuse(a, b_reg, t):
                return None, torch.full(MIP_SHAPE, float(10 * t + 1))
            return fuse

        def sink(t, decon, tmx, rec):
            with lock:
                sunk[t] = tmx.tolist()

        fuse = make_fuse(0) if lanes == 1 else [make_fuse(l) for l in range(lanes)]
        stack = batch.run_sharded(tps, lambda t: (t, t), fuse, batch.RegPolicy(reg_mode=3, reg_choice=2), reg_fn,
                                  lambda m: True, MIP_SHAPE, sink=sink, lanes=lanes)
        assert sorted(log) == tps and 1 <= len(seen_threads) <= lanes
        results[lanes] = (stack[:, 0, 0].tolist(), sunk)
    assert results[1] == results[2] == results[3]
    with pytest.raises(ValueError):
        batch.run_sharded(tps, lambda t: (t, t), [None, None], batch.RegPolicy(reg_mode=2), None, None, MIP_SHAPE, lanes=2)


# ---- all five projection products of a time point in one gather (src/spim_fusion_batch.cpp:915-940)
PRODUCTS = {"xy": (4, 6), "yz": (3, 4), "zx": (6, 3), "mip3d_x": (5, 5, 6), "mip3d_y": (5, 4, 7)}


def _product_values(t):
    return {k: torch.arange(int(np.prod(s)), dtype=torch.float32).reshape(s) * (i + 1) + 1000.0 * t
            for i, (k, s) in enumerate(PRODUCTS.items())}


def _products_worker(rank, world, port, outdir, reg_mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tps = [3, 4, 5, 6, 7]
        # the chain token / the regMode-1 matrix travel over their own (gloo) group, as bench.py sets it up next to RCCL
        control = dist.new_group(backend="gloo")
        stacks = batch.run_sharded(tps, lambda t: (t, t), lambda a, b, t: (None, _product_values(t)),
                                   batch.RegPolicy(reg_mode=reg_mode, reg_choice=2), _fake_reg([]), lambda m: True, PRODUCTS,
                                   control_group=control, control_device="cpu")
        if rank == 0:
            torch.save({"stacks": stacks, "gather": dict(batch.last_gather)}, os.path.join(outdir, "stacks.pt"))
        else:
            assert stacks is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("reg_mode", [3, 2, 1])
def test_all_projection_products_two_ranks_match_single_rank(tmp_path, reg_mode):
    """XY, YZ, ZX and both 36-angle 3-D MIP stacks travel as ONE record per time point in ONE gather; rank 0 ends up with
    the five time-ordered stacks of the single-rank run (also through the regMode-2 token ring)."""
    port = _free_port()
    mp.spawn(_products_worker, args=(2, port, str(tmp_path), reg_mode), nprocs=2, join=True)
    got = torch.load(tmp_path / "stacks.pt")
    tps = [3, 4, 5, 6, 7]
    single = batch.run_sharded(tps, lambda t: (t, t), lambda a, b, t: (None, _product_values(t)),
                               batch.RegPolicy(reg_mode=reg_mode, reg_choice=2), _fake_reg([]), lambda m: True, PRODUCTS)
    assert set(got["stacks"]) == set(PRODUCTS) == set(single)
    for k, shp in PRODUCTS.items():
        assert tuple(got["stacks"][k].shape) == (len(tps),) + shp
        assert torch.equal(got["stacks"][k], single[k])
        for i, t in enumerate(tps):
            assert torch.equal(single[k][i], _product_values(t)[k])
    rec = sum(int(np.prod(s)) for s in PRODUCTS.values())
    assert got["gather"]["bytes_per_time_point"] == 4 * rec
    assert got["gather"]["bytes_received_rank0"] == 4 * rec * 2  # time points 4 and 6 came from rank 1
    assert got["gather"]["products"] == list(PRODUCTS)


def test_gather_rejects_a_product_of_the_wrong_shape():
    with pytest.raises(ValueError, match="mip3d_x"):
        bad = _product_values(3)
        bad["mip3d_x"] = bad["mip3d_x"][:, :, :5]
        batch.gather_mip_stack({3: bad}, [3], PRODUCTS)


def test_a_failing_lane_stops_the_others_at_their_next_time_point():
    """lanes=2: the first exception sets a stop flag; the healthy lane finishes the time point it is on and returns, the
    exception is raised after the lanes have drained (it used to surface only after ALL remaining time points)."""
    import threading
    import time as _time
    done, lock = [], threading.Lock()

    def reg_fn(target, source, i_tmx, reg_choice, aff_method, input_tmx, ftol, it_limit):
        if int(target) == 21:
            raise RuntimeError("registration of time point 21 failed")
        _time.sleep(0.05)
        with lock:
            done.append(int(target))
        tmx = np.eye(3, 4, dtype=np.float32).reshape(12)
        rec = np.zeros(11, dtype=np.float32)
        rec[3] = 0.9
        return source, tmx, rec

    fuse = [lambda a, b, t: (None, torch.zeros(MIP_SHAPE))] * 2
    with pytest.raises(RuntimeError, match="time point 21"):
        batch.run_sharded(list(range(20, 60)), lambda t: (t, t), fuse, batch.RegPolicy(reg_mode=3, reg_choice=2), reg_fn,
                          lambda m: True, MIP_SHAPE, lanes=2)
    assert len(done) <= 2, done  # lane 0 (20, 22, ...) stopped right away instead of working through 20 time points
