#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X SVT-AV1 hot path (BASELINE.json metric).

One "step" = one pass of every implemented kernel class of the hot path over ONE synthetic
4K (3840x2160) 8-bit 4:2:0 frame = 2040 superblocks, inputs resident in HBM before the timed
region.  `value` = superblocks per second over the whole job (all ranks).

Multi-GPU: frames/streams are independent, so rank i simply processes its own
frame on GPU i — no data-path collective; torch.distributed (RCCL) is used only for the barrier
and the max-over-ranks time.  Scaling is "weak" (per-GPU work fixed).

PyTorch is plumbing only here (device memory, streams, distributed); every kernel is launched
through the C ABI of libsvtav1_hip.so (include/svt_hip.h) on torch's current stream.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "encoded 4K 8-bit SB/s (ME+txfm+quant+loopfilter) per GPU; bit-exact vs C ref"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists in the product path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from conftest import load_package, ptr
    import me_common as mc
    pkg = load_package()
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    ctx = pkg.Context(local_rank)
    L = ctx.L
    stream = torch.cuda.current_stream()
    ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(stream.cuda_stream)))

    W, H = args.width, args.height
    PAD = mc.synth.PAD
    # ---- synthetic inputs (seeded per rank = per stream), uploaded before the timed region
    cur, ref = mc.synth.make_luma_pair(W, H, seed=11 + 100 * rank)
    cur_p, ref_p = mc.synth.pad_plane(cur), mc.synth.pad_plane(ref)
    stride = cur_p.shape[1]
    sbs = mc.windows(orc, W, H, 64, 64)
    n_sb = len(sbs)
    dev = torch.device("cuda", local_rank)
    d_cur = torch.from_numpy(cur_p).to(dev)
    d_ref = torch.from_numpy(ref_p).to(dev)
    d_sbs = torch.from_numpy(np.frombuffer(bytes(sbs), dtype=np.uint8).copy()).to(dev)
    d_sad = torch.zeros((n_sb, 85), dtype=torch.int32, device=dev)
    d_mv = torch.zeros((n_sb, 85), dtype=torch.int32, device=dev)

    # ---- kernel classes of the step.  bytes_per_sb = SURVEY.md 8(d) algorithmic HBM bytes per SB.
    def run_me():
        ctx.check(L.svt_hip_me_fullpel_frame_dev(ctx.h, d_cur.data_ptr(), d_ref.data_ptr(), stride, PAD, PAD,
                                                 d_sbs.data_ptr(), n_sb, 0, d_sad.data_ptr(), d_mv.data_ptr()), "me")

    stages = [
        dict(name="me_fullpel_85pu", run=run_me, bytes_per_sb=8872, kernel="me_fullpel_85pu_kernel",
             work_per_sb=4096 * 4096, work_unit="px-SAD"),
    ]

    def step():
        for st in stages:
            st["run"]()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    import importlib
    shard = importlib.import_module("svt_av1_amd.shard")
    elapsed = shard.max_over_ranks(elapsed, dist if world > 1 else None, dev)

    # ---- per-kernel device time with HIP events on the launch stream (outside the headline timing)
    per_kernel = {}
    for st in stages:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(5, args.steps)
        e0.record(stream)
        for _ in range(reps):
            st["run"]()
        e1.record(stream)
        e1.synchronize()
        per_kernel[st["name"]] = e0.elapsed_time(e1) / reps  # ms per launch
    dominant = max(stages, key=lambda s: per_kernel[s["name"]])
    dom_ms = per_kernel[dominant["name"]]
    achieved_gbs = dominant["bytes_per_sb"] * n_sb / (dom_ms * 1e-3) / 1e9

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- parity spot check of what was just timed (first SB row) — the checker, not the product
    g_sad = d_sad.cpu().numpy().view(np.uint32)
    g_mv = d_mv.cpu().numpy().view(np.uint32)
    nrow = (W + 63) // 64
    o_sad, o_mv = mc.oracle_frame(orc, cur_p, ref_p, stride, PAD, sbs, 0, 0, min(nrow, 8))
    parity_ok = bool(np.array_equal(o_sad[:min(nrow, 8)], g_sad[:min(nrow, 8)]) and
                     np.array_equal(o_mv[:min(nrow, 8)], g_mv[:min(nrow, 8)]))

    # ---- CPU baseline: the oracle port of the same stage chain on a bounded SB sample, all host cores
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cores = max(1, min(os.cpu_count() or 1, 64))
        per_thread = 12
        sample = min(n_sb, cores * per_thread)
        chunks = [(i * sample // cores, (i + 1) * sample // cores) for i in range(cores)]

        def work(b_e):
            mc.oracle_frame(orc, cur_p, ref_p, stride, PAD, sbs, 0, b_e[0], b_e[1])
        t1 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, chunks))
        cpu_s = time.perf_counter() - t1
        cpu = dict(value=sample / cpu_s, unit="SB/s", cores=cores, kind="port",
                   sample=f"{sample} of {n_sb} SBs of the same 4K frame, oracle C port (scalar, -O2), "
                          f"{cores} threads x {per_thread} SBs, {cpu_s:.1f} s wall")

    total_sb = n_sb * args.steps * world
    out = {
        "metric": METRIC, "value": total_sb / elapsed, "unit": "SB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{W}x{H} 8-bit 4:2:0 synthetic frame, {n_sb} SBs/frame/GPU, 1 reference, "
                               f"64x64 integer search area; stages: " + ",".join(s["name"] for s in stages),
                   "stages_ms": per_kernel, "parity_spot_check": parity_ok},
        "roofline": {"bound": "hbm", "kernel": dominant["kernel"], "achieved": achieved_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
                     "note": "algorithmic bytes/SB (SURVEY 8d) x SBs / HIP-event launch time; this kernel is "
                             "integer-VALU bound, see valu",
                     "valu": {"achieved": dominant["work_per_sb"] * n_sb / (dom_ms * 1e-3) / 1e12,
                              "peak": 1024 * 64 * 16 / 16.0 * 2.4e9 / 1e12, "unit": "T px-SAD/s",
                              "note": "peak = 1024 SIMDs x 64 lanes x 16 abs-diff per v_qsad_pk_u16_u8 / 16 cyc x 2.4 GHz"}},
        "cpu_baseline": cpu,
    }
    out["roofline"]["valu"]["frac"] = out["roofline"]["valu"]["achieved"] / out["roofline"]["valu"]["peak"]
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
