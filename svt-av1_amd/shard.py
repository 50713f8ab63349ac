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


def gather_floats(value, dist=None, device=None):
    """[value of rank 0, value of rank 1, ...] on every rank (the per-rank figures of a multi-GPU bench line); None values travel as NaN."""
    v = float("nan") if value is None else float(value)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [v]
    import torch
    world = dist.get_world_size()
    t = torch.tensor([v], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def rank_env(gpus, environ):
    """(rank, local_rank, world) of this process from a launcher's environment, checked against --gpus: a 1-GPU measurement labelled N must not happen."""
    rank, local_rank, world = int(environ.get("RANK", "0")), int(environ.get("LOCAL_RANK", "0")), int(environ.get("WORLD_SIZE", "1"))
    if world != max(1, gpus):
        raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}: launch {gpus} ranks (torch.distributed.run --nproc-per-node {gpus}) or drop the launcher")
    if not 0 <= rank < world or not 0 <= local_rank < world:
        raise SystemExit(f"bench.py: RANK={rank} LOCAL_RANK={local_rank} outside WORLD_SIZE={world}")
    return rank, local_rank, world


def _cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_of_pci(pci_bus_id, sysfs="/sys"):
    """NUMA node of a PCI device ("0000:c1:00.0") and that node's CPUs, from sysfs; (None, []) when the platform does not say (one node, a VM, no sysfs)."""
    import os
    try:
        with open(os.path.join(sysfs, "bus/pci/devices", pci_bus_id.lower(), "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None, []
        with open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")) as f:
            return node, _cpulist(f.read())
    except (OSError, ValueError):
        return None, []


def pin_rank_to_gpu_numa(pci_bus_id, sysfs="/sys", setaffinity=None):
    """Keeps a rank's threads (and with them the first touch of its page-locked staging buffers) on the CPUs next to its GPU: eight ranks on one host otherwise
    copy through each other's memory controllers.  Returns {"node": n, "cpus": count} or None when nothing was pinned."""
    import os
    node, cpus = numa_of_pci(pci_bus_id, sysfs)
    if node is None or not cpus:
        return None
    try:
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return None
        (setaffinity or os.sched_setaffinity)(0, allowed)
    except (AttributeError, OSError):
        return None
    return {"node": node, "cpus": len(allowed)}
