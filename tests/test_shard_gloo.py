"""CPU, world_size 2, gloo: the N>1 path of bench.py — independent streams per rank, no data-path collective, whole-job throughput = sum of units /
max-over-ranks time.  Each rank DRIVES KERNELS: its streams' frames go through a chained step of the C ABI (ME -> deblock -> CDEF search -> strength
decision -> CDEF apply, tests/shard_common.py) on the CPU test double of the library; the union of what the two ranks produce must be exactly what one
process produces for all streams, and the throughput arithmetic is checked on the real unit counts."""
import os
import socket
import time

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_package
import shard_common as sc

N_STREAMS = 5   # an odd count: rank 0 owns 3 streams, rank 1 owns 2


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    load_package()
    import importlib
    shard = importlib.import_module("svt_av1_amd.shard")
    mine = shard.streams_of_rank(N_STREAMS, world, rank)
    dev = sc.Dev(sc.load(sc.MOCK_LIB))
    dist.barrier()
    t0 = time.perf_counter()
    out, units = {}, 0
    for s in mine:
        n_sb, res = sc.stream_step(dev, s)
        out[s] = res; units += n_sb
    dist.barrier()
    elapsed = time.perf_counter() - t0
    dev.close()
    t = shard.max_over_ranks(elapsed, dist)
    v = shard.aggregate_throughput(units, elapsed, dist)
    q.put((rank, mine, out, units, elapsed, t, v))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(not os.path.exists(sc.MOCK_LIB), reason="oracle/_ref/mock/libsvtav1_hip.so not built (make -f oracle/Makefile.enc; needs /root/reference)")
def test_two_ranks_drive_their_streams():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in ps: p.join(60)
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    # one process, all streams: the ranks' outputs are exactly these
    dev = sc.Dev(sc.load(sc.MOCK_LIB))
    total_units = 0
    for s_id in range(N_STREAMS):
        n_sb, want = sc.stream_step(dev, s_id)
        total_units += n_sb
        got = res[s_id % 2][2][s_id]
        assert got == want, (s_id, got, want)
    dev.close()
    assert len({tuple(sorted(r.items())) for rk in res for r in rk[2].values()}) == N_STREAMS   # streams really differ (seeded per stream)
    # whole-job figures: max over ranks, units summed over ranks
    t_max = max(res[0][4], res[1][4])
    assert res[0][5] == res[1][5] and abs(res[0][5] - t_max) < 1e-9
    assert res[0][3] + res[1][3] == total_units
    assert abs(res[0][6] - total_units / t_max) < 1e-6 * total_units / t_max and res[0][6] == res[1][6]
