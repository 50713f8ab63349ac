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


def _fake_sysfs(tmp_path):
    """two GPUs on two NUMA nodes: 0000:c1:00.0 -> node 0 (CPUs 0-3, 8), 0000:c2:00.0 -> node 1 (CPUs 4-7)"""
    root = tmp_path / "sys"
    for bdf, node in (("0000:c1:00.0", 0), ("0000:c2:00.0", 1), ("0000:c3:00.0", -1)):
        d = root / "bus/pci/devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpus in ((0, "0-3,8"), (1, "4-7")):
        d = root / "devices/system/node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    return str(root)


def test_numa_helpers(tmp_path):
    load_package()
    import importlib
    shard = importlib.import_module("svt_av1_amd.shard")
    fs = _fake_sysfs(tmp_path)
    assert shard.numa_of_pci("0000:C1:00.0", fs) == (0, [0, 1, 2, 3, 8]) and shard.numa_of_pci("0000:c2:00.0", fs) == (1, [4, 5, 6, 7])
    assert shard.numa_of_pci("0000:c3:00.0", fs) == (None, []) and shard.numa_of_pci("0000:ff:00.0", fs) == (None, [])   # "no node" / unknown device: nothing to pin to
    calls = []
    got = shard.pin_rank_to_gpu_numa("0000:c2:00.0", fs, setaffinity=lambda pid, cpus: calls.append((pid, sorted(cpus))))
    allowed = sorted(set([4, 5, 6, 7]) & set(os.sched_getaffinity(0)))
    assert (got == {"node": 1, "cpus": len(allowed)} and calls == [(0, allowed)]) if allowed else (got is None and not calls)
    assert shard.pin_rank_to_gpu_numa("0000:c3:00.0", fs, setaffinity=lambda *a: calls.append(a)) is None
    assert shard.gather_floats(3.5) == [3.5] and shard.gather_floats(None)[0] != shard.gather_floats(None)[0]
    with pytest.raises(SystemExit):
        shard.rank_env(2, {"WORLD_SIZE": "1"})
    with pytest.raises(SystemExit):
        shard.rank_env(2, {"WORLD_SIZE": "2", "RANK": "2"})
    assert shard.rank_env(1, {}) == (0, 0, 1) and shard.rank_env(4, {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}) == (3, 3, 4)


@pytest.mark.skipif(not os.path.exists(sc.MOCK_LIB), reason="oracle/_ref/mock/libsvtav1_hip.so not built (make -f oracle/Makefile.enc; needs /root/reference)")
def test_launcher_to_json_line_two_ranks(tmp_path):
    """`--gpus 2` end to end without GPUs: the launcher bench.py uses (python -m torch.distributed.run, 127.0.0.1) starts two ranks of tests/shard_bench_driver.py, which
    goes through the same svt_av1_amd.shard calls as bench.py (rank_env, barrier, per-rank gather, max over ranks, whole-job aggregate, NUMA pinning against a fake sysfs)
    with gloo and the CPU test double; rank 0's JSON line must carry n_gpus 2, one entry per rank, and value = all units / the slowest rank's time."""
    import json
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    drv = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_bench_driver.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port), drv, "--gpus", "2", "--steps", "2",
           "--fake-sysfs", _fake_sysfs(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    pr = d["per_rank"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(pr["ms_per_step"]) == 2 and pr["numa_node"] == [0, 1]
    assert pr["units"][0] > pr["units"][1] > 0                                   # five streams: three on rank 0, two on rank 1
    assert abs(d["ms_per_step"] - max(pr["ms_per_step"])) < 1e-6 * d["ms_per_step"]
    assert abs(d["value"] - sum(pr["units"]) / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    # a launcher whose world size disagrees with --gpus is refused by every rank
    r = subprocess.run([sys.executable, drv, "--gpus", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
