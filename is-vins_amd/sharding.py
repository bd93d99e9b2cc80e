"""Multi-GPU sharding of independent sliding windows: static block partition over the ranks, no
collective inside the solve; ONE exchange step at the end -- an all-gather of the per-window result records (RCCL over
xGMI on GPUs, gloo on CPU) -- and a max-reduction of the step time."""


def shard_window_ids(rank, world, per_rank, total=None):
    """weak scaling (total=None): rank r owns window ids [r * per_rank, (r + 1) * per_rank).
    strong scaling (total given): the `total` windows are block-partitioned, rank r owns [r * total // world, (r + 1) * total // world)"""
    if total is None:
        return range(rank * per_rank, (rank + 1) * per_rank)
    return range(rank * total // world, (rank + 1) * total // world)


def max_over_ranks(value, dist=None, device="cpu"):
    """max-reduce a python float over the process group (RCCL on GPUs, gloo on CPU)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(records, out, dist=None):
    """the exchange step: all-gather every rank's [n_local, rec] result records into out [world * n_local, rec]
    (ncclAllGather on device tensors).  Every rank must hold EQUALLY many records: unequal shards are refused here
    rather than left to hang in the collective (bench.py rejects --scaling strong with windows % ranks != 0).
    Enqueued on the current stream; returns `out`."""
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size()
    if out.shape[0] != world * records.shape[0] or out.shape[1:] != records.shape[1:]:
        raise ValueError(f"gather_records: out {tuple(out.shape)} is not world ({world}) x records {tuple(records.shape)}")
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out.copy_(records)
        return out
    try:
        dist.all_gather_into_tensor(out, records)
    except (RuntimeError, NotImplementedError):           # a backend without the flat form: gather into views of `out`
        dist.all_gather(list(out.chunk(dist.get_world_size(), dim=0)), records)
    return out
