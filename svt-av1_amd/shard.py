"""Frame/stream sharding across GPUs: the hot path has no cross-GPU exchange —
streams (or frames of the open-loop stages) are independent, so rank r simply owns streams
r, r+world, ...; torch.distributed is used only for the barrier and the max-over-ranks time."""


def streams_of_rank(n_streams, world, rank):
    return list(range(rank, n_streams, world))


def max_over_ranks(seconds, dist=None, device=None):
    """Whole-job time = slowest rank. Works with gloo (CPU tensors) and nccl/RCCL (device tensors)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(units_per_rank, seconds, dist=None, device=None):
    """value = units processed by ALL ranks / max-over-ranks time."""
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    if world == 1:
        return units_per_rank / seconds
    import torch
    u = torch.tensor([float(units_per_rank)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / max_over_ranks(seconds, dist, device)
