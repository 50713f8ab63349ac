"""CPU, world_size 2, gloo: the N>1 path of bench.py — independent streams per rank, no data-path
collective, whole-job throughput = sum of units / max-over-ranks time."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_package


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    load_package()
    import importlib
    shard = importlib.import_module("svt_av1_amd.shard")
    mine = shard.streams_of_rank(8, world, rank)
    t = shard.max_over_ranks(1.0 + rank, dist)
    v = shard.aggregate_throughput(2040 * len(mine), 1.0 + rank, dist)
    q.put((rank, mine, t, v))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps: p.join(60)
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    assert res[0][2] == res[1][2] == 2.0
    assert abs(res[0][3] - 2040 * 8 / 2.0) < 1e-6 and res[0][3] == res[1][3]
