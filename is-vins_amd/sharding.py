"""Multi-GPU sharding of independent sliding windows: static block partition, no
data-path collective; ranks only agree on the slowest rank's step time."""


def shard_window_ids(rank, world, per_rank):
    """weak scaling: rank r owns window ids [r * per_rank, (r + 1) * per_rank)"""
    return range(rank * per_rank, (rank + 1) * per_rank)


def max_over_ranks(value, dist=None, device="cpu"):
    """max-reduce a python float over the process group (RCCL on GPUs, gloo on CPU)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
