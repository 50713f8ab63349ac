"""One rank of tests/test_shard_gloo.py::test_launcher_to_json_line_two_ranks: the rank / world plumbing of bench.py (svt_av1_amd.shard: rank_env, the barrier, per-rank
gather, max over ranks, whole-job aggregate, NUMA pinning) driven exactly as bench.py drives it, under torch.distributed.run with the gloo backend, each rank running the
chained step of the C ABI on the CPU test double (tests/shard_common.py).  Rank 0 prints ONE JSON line of the shape bench.py prints."""
import argparse
import importlib
import json
import os
import sys
import time

import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package   # noqa: E402
import shard_common as sc   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--fake-sysfs", default="")
a = ap.parse_args()
load_package()
shard = importlib.import_module("svt_av1_amd.shard")
rank, local_rank, world = shard.rank_env(a.gpus, os.environ)
pinned = []
numa = shard.pin_rank_to_gpu_numa("0000:%02x:00.0" % (0xc1 + local_rank), a.fake_sysfs or "/sys", setaffinity=lambda pid, cpus: pinned.append(sorted(cpus))) if a.fake_sysfs else None
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
D = dist if world > 1 else None
dev = sc.Dev(sc.load(sc.MOCK_LIB))
mine = shard.streams_of_rank(2 * world + 1, world, rank)
if D: D.barrier()
t0 = time.perf_counter()
units = 0
for _ in range(a.steps):
    for s in mine:
        n_sb, _res = sc.stream_step(dev, s)
        units += n_sb
if D: D.barrier()
own = time.perf_counter() - t0
dev.close()
elapsed = shard.max_over_ranks(own, D)
per_rank_ms = [t / a.steps * 1e3 for t in shard.gather_floats(own, D)]
per_rank_units = shard.gather_floats(units, D)
per_rank_numa = [None if n != n else int(n) for n in shard.gather_floats(numa["node"] if numa else None, D)]
value = shard.aggregate_throughput(units, own, D)
if rank == 0:
    print(json.dumps({"metric": "test double", "value": value, "n_gpus": world, "steps": a.steps, "ms_per_step": elapsed / a.steps * 1e3, "scaling": "weak",
                      "per_rank": {"ms_per_step": per_rank_ms, "units": per_rank_units, "numa_node": per_rank_numa, "pinned": pinned}}))
if D:
    D.barrier()
    D.destroy_process_group()
