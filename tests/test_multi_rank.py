"""N > 1 path on CPU: two gloo ranks shard window ids exactly as bench.py does, solve their own
windows (with the CPU oracle standing in for the device), and agree on the max step time.  Checks
that the shards are disjoint, cover the batch, and that a window's result does not depend on the
rank count (windows are independent: no data-path collective)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, per_rank, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import isvins_loader; isvins_loader.load()
    import torch.distributed as dist
    from isvins_amd import abi, sharding, synth
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = list(sharding.shard_window_ids(rank, world, per_rank))
    lib = oracle_lib.load()
    cfg = abi.make_config(11, 5)
    costs = []
    for w in synth.make_windows(ids, n_landmarks=30):
        s = abi.isv_summary_t(); mg = abi.isv_marg_result_t()
        lib.isvo_optimize(C.byref(cfg), C.byref(w.c()), C.byref(s), C.byref(mg))
        costs.append(s.final_cost)
    dt = sharding.max_over_ranks(0.5 + rank, dist)      # pretend rank r took 0.5 + r seconds
    gathered = [None] * world
    dist.all_gather_object(gathered, (ids, costs))
    if rank == 0:
        q.put((dt, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world, per_rank = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, per_rank, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    dt, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert dt == 1.5                                    # max over ranks
    ids = [i for g in gathered for i in g[0]]
    assert sorted(ids) == list(range(world * per_rank)) and len(set(ids)) == len(ids)
    # same windows solved in a single process give the same costs (independence)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from isvins_amd import abi, synth
    lib = oracle_lib.load(); cfg = abi.make_config(11, 5)
    ref = []
    for w in synth.make_windows(range(world * per_rank), n_landmarks=30):
        s = abi.isv_summary_t(); mg = abi.isv_marg_result_t()
        lib.isvo_optimize(C.byref(cfg), C.byref(w.c()), C.byref(s), C.byref(mg))
        ref.append(s.final_cost)
    got = dict(zip(ids, [c for g in gathered for c in g[1]]))
    assert [got[i] for i in range(world * per_rank)] == ref
