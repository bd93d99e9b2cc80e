"""N > 1 path on CPU: two and EIGHT gloo ranks shard window ids exactly as bench.py does (weak and strong partition), solve their
own windows (with the CPU oracle standing in for the device), exchange the per-window result records with the SAME
all-gather bench.py issues over RCCL (sharding.gather_records), and agree on the max step time.  Checks that the shards
are disjoint and cover the batch, that every rank ends up with every window's record, and that a window's result does
not depend on the rank count (windows are independent: no collective inside the solve)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_LM = 30


def _record(w, s):
    """the layout isv_batch_pack_results writes on the device (include/isvins_backend.h)"""
    lam = np.zeros(N_LM); lam[: w.L] = w.para_Feature[: w.L]
    return np.concatenate([w.para_Pose.ravel(), w.para_SpeedBias.ravel(), lam,
                           [s.final_cost, s.initial_cost, s.iterations, s.termination, s.num_successful, 0.0, w.header0, w.L]])


def _solve(lib, cfg, ids):
    from isvins_amd import abi, synth
    recs = []
    for w in synth.make_windows(ids, n_landmarks=N_LM):
        s = abi.isv_summary_t(); mg = abi.isv_marg_result_t()
        lib.isvo_optimize(C.byref(cfg), C.byref(w.c()), C.byref(s), C.byref(mg))
        recs.append(_record(w, s))
    return np.stack(recs)


def _worker(rank, world, per_rank, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import isvins_loader; isvins_loader.load()
    import torch
    import torch.distributed as dist
    from isvins_amd import abi, sharding
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = oracle_lib.load()
    cfg = abi.make_config(11, 5)
    out = {}
    for mode, total in (("weak", None), ("strong", world * per_rank)):
        ids = list(sharding.shard_window_ids(rank, world, per_rank, total))
        recs = torch.from_numpy(_solve(lib, cfg, ids))
        gathered = torch.empty((world * len(ids), recs.shape[1]), dtype=torch.float64)
        sharding.gather_records(recs, gathered, dist)
        out[mode] = (ids, gathered.numpy().copy())
    dt = sharding.max_over_ranks(0.5 + rank, dist)      # pretend rank r took 0.5 + r seconds
    q.put((rank, dt, out))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, per_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, per_rank, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        rank, dt, out = q.get(timeout=300)
        got[rank] = (dt, out)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from isvins_amd import abi
    ref = _solve(oracle_lib.load(), abi.make_config(11, 5), range(world * per_rank))      # the same windows in ONE process
    for mode in ("weak", "strong"):
        ids = [i for r in range(world) for i in got[r][1][mode][0]]
        assert sorted(ids) == list(range(world * per_rank)) and len(set(ids)) == len(ids)       # disjoint cover
        for r in range(world):
            dt, out = got[r]
            assert dt == 0.5 + world - 1                                                         # max over ranks
            # every rank holds every window's record, in rank order = window-id order, bitwise the single-process result
            assert np.array_equal(out[mode][1], ref)
            assert np.array_equal(out[mode][1][:, -2], np.arange(world * per_rank, dtype=float))  # header0 = window id


def test_two_rank_sharding_gather_and_timing():
    _run_world(2, 3)


def test_eight_rank_sharding_gather_and_timing():
    """the driver's real rank count: eight gloo ranks, one window each per partition mode"""
    _run_world(8, 1)


def test_config4_partition_over_eight_ranks_and_the_uneven_refusal():
    """BASELINE config 4 as written: 1024 windows block-partitioned over 8 ranks = 128 each, disjoint, covering; bench.py refuses a
    strong partition the ranks cannot share equally (the all-gather needs equal record counts) before it touches torch or a GPU"""
    import subprocess
    sys.path.insert(0, ROOT)
    import isvins_loader; isvins_loader.load()
    from isvins_amd import sharding
    shards = [list(sharding.shard_window_ids(r, 8, 0, 1024)) for r in range(8)]
    assert all(len(s_) == 128 for s_ in shards) and sorted(i for s_ in shards for i in s_) == list(range(1024))
    assert [list(sharding.shard_window_ids(r, 8, 128, None))[0] for r in range(8)] == [128 * r for r in range(8)]     # weak: 128 per rank
    env = dict(os.environ, RANK="0", WORLD_SIZE="3", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    o = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--scaling", "strong", "--windows", "1024"], env=env, capture_output=True, text=True, timeout=120)
    assert o.returncode != 0 and "divisible" in (o.stderr + o.stdout)
    o = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--windows", "1024"], env=env, capture_output=True, text=True, timeout=120)
    assert o.returncode != 0 and "WORLD_SIZE" in (o.stderr + o.stdout)          # --gpus must match the launcher's rank count
