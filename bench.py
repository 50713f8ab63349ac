#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X SVT-AV1 hot path (BASELINE.json metric).

One "step" = one pass of every implemented kernel class of the hot path over ONE synthetic
4K (3840x2160) 8-bit 4:2:0 frame = 2040 superblocks, inputs resident in
HBM before the timed region:
    HME pyramids + variance pyramid -> HME L0/L1/L2 -> integer ME (85 PUs, 64x64 search area, 1 ref) ->
    sub-pel convolve (every 16x16 luma block) -> residual + fwd txfm + quantize (all planes, per-SB
    square tiling 4..64) -> inverse txfm + recon -> deblock (3 planes, V then H) -> CDEF strength
    search (64 strengths, 3 planes) -> CDEF apply -> SGR search (16 parameter sets, 3 planes) -> SGR apply.
`value` = superblocks per second over the whole job (all ranks).

Multi-GPU: frames/streams are independent, so rank i processes its own frame on
GPU i — no data-path collective; torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks time.  Scaling is "weak" (per-GPU work fixed).

PyTorch is plumbing only (device memory, streams, distributed); every kernel is launched through the
C ABI of libsvtav1_hip.so (include/svt_hip.h) on torch's current stream.
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
P3, I3 = C.c_void_p * 3, C.c_int * 3

# algorithmic HBM bytes per SB of each kernel class (8-bit 4:2:0, luma + chroma where the stage covers chroma)
BYTES_PER_SB = {
    "me_fullpel_85pu": 8872,                 # 4096 src + 4096 ref (amortised) + 85*8 out
    "fwd_txfm_quant": 61440 + 128,           # (src+pred 2*6144) + qcoeff+dqcoeff 2*4*6144 + eob   (luma+chroma)
    "inv_txfm_recon": 36864,                 # dqcoeff 4*6144 + pred 6144 + recon 6144
    "deblock": 2 * (6144 + 6144) + 2560,     # two passes (V, H): planes R+W each + edge descriptors
    "cdef_search": 13312,                    # recon 6144 + source 6144 R + 2*64*8 W
    "cdef_apply": 12288,                     # 6144 R + 6144 W
    "pyramids": 5376 + 4351,                 # decimation 4096 R + 1024 + 256 W ; variance pyramid 4096 R + 85*3 W
    "hme_l0_l1_l2": 256 + 1024 + 4096 + 3 * 12,   # source blocks of the three levels + results (windows are cache-resident)
    "subpel_convolve": 12560,                # 16 blocks x (16+7)^2 R + 4096 W (luma)
    "sgr_search": 12288 + 640,               # dgd 6144 + source 6144 R + sums
    "sgr_apply": 12288,                      # 6144 R + 6144 W
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "reference", "port"],
                    help="reference = the reference's own SIMD kernels (oracle/_ref SIMD flavour); port = the oracle's scalar C; auto = reference when built")
    ap.add_argument("--cpu-port-too", action="store_true", help="with the reference baseline, also time the scalar port")
    ap.add_argument("--serial", action="store_true", help="issue every launch on one stream (no intra-step concurrency)")
    ap.add_argument("--lanes", type=int, default=3, help="streams for the independent launches of a stage (debug)")
    ap.add_argument("--tx-multi", default="inv", help="which transform stages use the mixed-size launch (debug): fwd,inv / fwd / inv / none")
    ap.add_argument("--side-keys", default="pyr,hme,me,subpel", help="stages issued on the side stream (debug; must be source-side stages)")
    ap.add_argument("--me-waves", type=int, default=4, help="svt_hip_me_set_waves_per_sb value (debug)")
    ap.add_argument("--no-side", action="store_true", help="keep the source-side chain on the main stream (debug)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one captured HIP graph per step")
    ap.add_argument("--stages", default="all", help="comma list (debug): pyr,hme,me,subpel,txfm,inv,dlf,cdef_search,cdef_apply,sgr_search,sgr_apply")
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
    import importlib
    import me_common as mc
    import workload
    import txfm_common as tc
    pkg = load_package()
    shard = importlib.import_module("svt_av1_amd.shard")
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    ctx = pkg.Context(local_rank)
    L = ctx.L
    stream = torch.cuda.current_stream()
    ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(stream.cuda_stream)))
    dev = torch.device("cuda", local_rank)
    ctx.check(L.svt_hip_me_set_waves_per_sb(ctx.h, args.me_waves))

    W, H = args.width, args.height
    F = workload.Frame(W, H, seed=11 + 100 * rank)   # one stream per rank
    n_sb, PAD = F.n_sb, F.pad

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    # ---------------------------------------------------------------- device-resident inputs / outputs
    d_cur_p, d_ref_p = T(F.cur_y_p), T(F.ref_y_p)
    sbs = mc.windows_product(L, W, H, 64, 64)   # the product's own restatement of integer_search_sb's window clamp
    d_sbs = T(np.frombuffer(bytes(sbs), dtype=np.uint8).copy())
    d_sad = torch.zeros((n_sb, 85), dtype=torch.int32, device=dev)
    d_mv = torch.zeros((n_sb, 85), dtype=torch.int32, device=dev)
    d_cur = [T(p) for p in F.cur]
    d_pred = [T(p) for p in F.ref]                       # prediction = co-located reference (zero MV)
    d_recon = [torch.zeros_like(p) for p in d_pred]
    d_cdef_out = [torch.zeros_like(p) for p in d_pred]
    strides = [p.shape[1] for p in F.cur]
    tx_jobs = []   # one launch per (plane, tx size)
    keep = []
    for (kind, ts), descs in sorted(F.descs.items()):
        nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
        d_desc = T(descs)
        isc = [T(s) if s is not None else None for s in F.scan_tables(ts)]
        keep.append(isc)
        st = pkg.ScanTables()
        for c in range(3):
            st.iscan[c] = isc[c].data_ptr() if isc[c] is not None else None
        for plane in ([0] if kind == 0 else [1, 2]):
            qs = pkg.QuantParams()
            qp = F.qp[plane]
            for name, row in (("zbin", qp[0]), ("round", qp[1]), ("quant", qp[2]), ("quant_shift", qp[3]), ("dequant", qp[4])):
                getattr(qs, name)[0] = int(row[0]); getattr(qs, name)[1] = int(row[1])
            qs.log_scale = tc.TX_SCALE[ts]; qs.variant = 0
            n = len(descs)
            tx_jobs.append(dict(ts=ts, plane=plane, n=n, desc=d_desc, qs=qs, st=st,
                                q=torch.zeros(n * nk, dtype=torch.int32, device=dev), dq=torch.zeros(n * nk, dtype=torch.int32, device=dev),
                                eob=torch.zeros(n, dtype=torch.int16, device=dev), cul=torch.zeros(n, dtype=torch.int32, device=dev)))
    d_edges = [(T(ev), T(eh), ev.shape[1], ev.shape[0]) for ev, eh in F.edges]
    d_skip8 = T(F.skip8)
    d_mse = torch.zeros((2, n_sb, 64), dtype=torch.int64, device=dev)
    d_dir = torch.zeros(n_sb * 64, dtype=torch.uint8, device=dev)
    d_var = torch.zeros(n_sb * 64, dtype=torch.int32, device=dev)
    d_cy, d_cuv = T(F.cdef_y), T(F.cdef_uv)
    # pyramids / HME: 1/4 and 1/16 resolution source + reference, variance pyramid
    PADQ, PADS = 32, 16
    qw, qh, sw_, sh_ = W // 2, H // 2, W // 4, H // 4
    d_cur_q = torch.zeros((qh + 2 * PADQ, qw + 2 * PADQ), dtype=torch.uint8, device=dev); d_ref_q = torch.zeros_like(d_cur_q)
    d_cur_s = torch.zeros((sh_ + 2 * PADS, sw_ + 2 * PADS), dtype=torch.uint8, device=dev); d_ref_s = torch.zeros_like(d_cur_s)
    d_ymean = torch.zeros((n_sb, 85), dtype=torch.uint8, device=dev); d_yvar = torch.zeros((n_sb, 85), dtype=torch.int16, device=dev)
    # aligned copy of the current luma for the variance pyramid (needs an 8-byte aligned origin/stride and 64 px of slack)
    vp = np.zeros((F.sb_rows * 64 + 64, F.sb_cols * 64 + 64), np.uint8); vp[:H, :W] = F.cur[0]
    d_vp = T(vp)
    hme_host = workload.hme_jobs(F)
    hme = [dict(S=T(np.frombuffer(bytes(S), np.uint8).copy()), sad=torch.zeros(n_sb, dtype=torch.int32, device=dev),
                xy=torch.zeros((n_sb, 2), dtype=torch.int16, device=dev)) for S in hme_host]
    # sub-pel: every 16x16 luma block at an eighth-pel MV (EIGHTTAP_REGULAR), written into a prediction plane
    CB, nblk16 = workload.conv_jobs(F, 14 + rank)
    k = nblk16
    rng = np.random.default_rng(15 + rank)
    d_cb = T(np.frombuffer(bytes(CB), np.uint8)[:k * C.sizeof(pkg.ConvBlk)].copy())
    d_subpel = torch.zeros((H, W), dtype=torch.uint8, device=dev)
    # SGR: 3-px extended copies of the CDEF output, projection sums for all 16 sets, apply with fixed per-unit sets
    EXT = 3
    d_ext = [torch.zeros((p.shape[0] + 2 * EXT, p.shape[1] + 2 * EXT + ((-(p.shape[1] + 2 * EXT)) % 4), ), dtype=torch.uint8, device=dev) for p in d_pred]
    US = [256, 256, 256]   # restoration unit size per plane: what the reference picks above CIF (set_restoration_unit_size, EbPictureControlSet.c:31-47)
    n_units = [max((F.cur[p].shape[1] + US[p] // 2) // US[p], 1) * max((F.cur[p].shape[0] + US[p] // 2) // US[p], 1) for p in range(3)]
    d_sgr_sums = [torch.zeros((n_units[p], 16, 5), dtype=torch.int64, device=dev) for p in range(3)]
    d_unit_ep = [T(rng.integers(0, 16, n_units[p]).astype(np.uint8)) for p in range(3)]
    d_unit_xqd = [T(np.stack([rng.integers(-96, 32, n_units[p]), rng.integers(-32, 96, n_units[p])], 1).astype(np.int32)) for p in range(3)]
    d_sgr_out = [torch.zeros_like(p) for p in d_pred]

    # ---------------------------------------------------------------- streams
    # Independent launches of a stage (planes, transform sizes) and the two data-independent halves of a step (the source-side
    # chain pyramids -> HME -> ME -> sub-pel, and the reconstruction-side chain transform -> deblock -> CDEF -> restoration) are
    # issued on separate HIP streams, forked from and joined back into the stream that carries the step; inside a captured graph
    # these become parallel branches, so short kernels fill each other's tails and the VALU-bound search kernels overlap the
    # HBM-bound transform / filter passes.  --serial keeps everything on one stream.
    S = {"cur": stream}
    n_lanes = 1 if args.serial else args.lanes
    lanes = [torch.cuda.Stream() for _ in range(n_lanes)] if n_lanes > 1 else []
    side = torch.cuda.Stream() if not (args.serial or args.no_side) else None

    class on:
        def __init__(self, st):
            self.st = st

        def __enter__(self):
            self.prev = S["cur"]
            S["cur"] = self.st
            ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(self.st.cuda_stream)))
            self.t = torch.cuda.stream(self.st)
            self.t.__enter__()

        def __exit__(self, *a):
            self.t.__exit__(*a)
            S["cur"] = self.prev
            ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(self.prev.cuda_stream)))

    def parallel(jobs):
        """Run the callables round-robin on the lane streams, forked from / joined into the current stream."""
        if not lanes or len(jobs) < 2 or S["cur"] is side:   # the lanes belong to the main chain
            for j in jobs:
                j()
            return
        base = S["cur"]
        used = lanes[:min(len(lanes), len(jobs))]
        for ln in used:
            ln.wait_stream(base)
        for i, j in enumerate(jobs):
            with on(used[i % len(used)]):
                j()
        for ln in used:
            base.wait_stream(ln)

    # ---------------------------------------------------------------- the kernel classes of a step
    def run_me():
        ctx.check(L.svt_hip_me_fullpel_frame_dev(ctx.h, d_cur_p.data_ptr(), d_ref_p.data_ptr(), F.cur_y_p.shape[1], PAD, PAD,
                                                 d_sbs.data_ptr(), n_sb, 0, d_sad.data_ptr(), d_mv.data_ptr()), "me")

    # one mixed-size launch per 16 (size, plane) job lists (svt_hip_*_multi_dev): the 19 lists of a frame are 400-4000 blocks each
    FJ = (pkg.FwdTxJob * len(tx_jobs))(); IJ = (pkg.InvTxJob * len(tx_jobs))()
    for k, j in enumerate(tx_jobs):
        p = j["plane"]
        FJ[k] = pkg.FwdTxJob(j["ts"], j["n"], d_cur[p].data_ptr(), strides[p], d_pred[p].data_ptr(), strides[p], j["desc"].data_ptr(), j["qs"], j["st"],
                             None, j["q"].data_ptr(), j["dq"].data_ptr(), j["eob"].data_ptr(), j["cul"].data_ptr(), None)
        IJ[k] = pkg.InvTxJob(j["ts"], j["n"], j["dq"].data_ptr(), d_pred[p].data_ptr(), strides[p], d_recon[p].data_ptr(), strides[p], j["desc"].data_ptr())

    def txfm_job(j):
        p = j["plane"]
        ctx.check(L.svt_hip_fwd_txfm_quant_batch_dev(ctx.h, j["ts"], 1, d_cur[p].data_ptr(), strides[p], d_pred[p].data_ptr(), strides[p],
                                                     j["desc"].data_ptr(), j["n"], C.byref(j["qs"]), C.byref(j["st"]), None,
                                                     j["q"].data_ptr(), j["dq"].data_ptr(), j["eob"].data_ptr(), j["cul"].data_ptr(), None), "fwd")

    def inv_job(j):
        p = j["plane"]
        ctx.check(L.svt_hip_inv_txfm_add_batch_dev(ctx.h, j["ts"], 1, 8, j["dq"].data_ptr(), d_pred[p].data_ptr(), strides[p],
                                                   d_recon[p].data_ptr(), strides[p], j["desc"].data_ptr(), j["n"]), "inv")

    def run_txfm():
        if "fwd" in args.tx_multi:
            ctx.check(L.svt_hip_fwd_txfm_quant_multi_dev(ctx.h, 1, FJ, len(tx_jobs)), "fwd")
        else:
            parallel([lambda j=j: txfm_job(j) for j in tx_jobs])

    def run_inv():
        if "inv" in args.tx_multi:
            ctx.check(L.svt_hip_inv_txfm_add_multi_dev(ctx.h, 1, 8, IJ, len(tx_jobs)), "inv")
        else:
            parallel([lambda j=j: inv_job(j) for j in tx_jobs])

    def run_dlf():   # all three planes: one launch per direction
        ctx.check(L.svt_hip_deblock_frame_dev(ctx.h, P3(*[p.data_ptr() for p in d_recon]), 1, I3(*strides), 8, P3(*[d_edges[p][0].data_ptr() for p in range(3)]),
                                              P3(*[d_edges[p][1].data_ptr() for p in range(3)]), I3(*[d_edges[p][2] for p in range(3)]),
                                              I3(*[d_edges[p][3] for p in range(3)]), 0), "dlf")

    def dlf_plane(p):
        if True:
            ev, eh, uw, uh = d_edges[p]
            ctx.check(L.svt_hip_deblock_plane_dev(ctx.h, d_recon[p].data_ptr(), 1, strides[p], 8, ev.data_ptr(), eh.data_ptr(), uw, uh, 0), "dlf")

    def run_cdef_search():
        ctx.check(L.svt_hip_cdef_search_frame_dev(ctx.h, 1, P3(*[p.data_ptr() for p in d_recon]), I3(*strides), P3(*[p.data_ptr() for p in d_cur]),
                                                  I3(*strides), W, H, d_skip8.data_ptr(), F.cdef_damping, 8, d_mse.data_ptr(), d_dir.data_ptr(),
                                                  d_var.data_ptr()), "cdef search")

    def run_cdef_apply():
        for p in range(3):
            d_cdef_out[p].copy_(d_recon[p])   # destination starts as a copy of the pre-CDEF picture (device-to-device)
        ctx.check(L.svt_hip_cdef_apply_frame_dev(ctx.h, 1, P3(*[p.data_ptr() for p in d_recon]), P3(*[p.data_ptr() for p in d_cdef_out]),
                                                 I3(*strides), W, H, d_skip8.data_ptr(), d_cy.data_ptr(), d_cuv.data_ptr(), F.cdef_damping, 8,
                                                 d_dir.data_ptr(), d_var.data_ptr()), "cdef apply")

    def pyr_job(src_p, dst_t, pad_, step_):
        org = src_p.data_ptr() + PAD * F.cur_y_p.shape[1] + PAD
        ctx.check(L.svt_hip_downsample_2d_dev(ctx.h, org, F.cur_y_p.shape[1], W, H, dst_t.data_ptr() + pad_ * dst_t.shape[1] + pad_, dst_t.shape[1], step_, 1), "ds")

    def run_pyramids():
        parallel([lambda: pyr_job(d_cur_p, d_cur_q, PADQ, 2), lambda: pyr_job(d_cur_p, d_cur_s, PADS, 4),
                  lambda: pyr_job(d_ref_p, d_ref_q, PADQ, 2), lambda: pyr_job(d_ref_p, d_ref_s, PADS, 4),
                  lambda: ctx.check(L.svt_hip_variance_pyramid_dev(ctx.h, d_vp.data_ptr(), d_vp.shape[1], F.sb_cols, n_sb, 0, d_ymean.data_ptr(), d_yvar.data_ptr()), "varpyr")])

    def run_hme():
        for lvl, (cur_t, ref_t) in enumerate(((d_cur_s, d_ref_s), (d_cur_q, d_ref_q), (d_cur_p, d_ref_p))):
            j = hme[lvl]
            ctx.check(L.svt_hip_sad_loop_batch_dev(ctx.h, cur_t.data_ptr(), cur_t.shape[1], ref_t.data_ptr(), ref_t.shape[1], j["S"].data_ptr(), n_sb,
                                                   j["sad"].data_ptr(), j["xy"].data_ptr()), "hme")

    def run_subpel():
        ctx.check(L.svt_hip_subpel_predict_batch_dev(ctx.h, 1, 8, d_ref_p.data_ptr() + PAD * F.ref_y_p.shape[1] + PAD, F.ref_y_p.shape[1],
                                                     d_subpel.data_ptr(), W, d_cb.data_ptr(), nblk16), "subpel")

    def sgr_extend(p):
        # svt_extend_frame equivalent (device-to-device, torch slicing = plumbing): 3-px edge replication of the CDEF output
        h_, w_ = d_cdef_out[p].shape
        e = d_ext[p]
        e[EXT:EXT + h_, EXT:EXT + w_] = d_cdef_out[p]
        e[EXT:EXT + h_, :EXT] = d_cdef_out[p][:, :1]; e[EXT:EXT + h_, EXT + w_:EXT + w_ + EXT] = d_cdef_out[p][:, -1:]
        e[:EXT, :] = e[EXT:EXT + 1, :]; e[EXT + h_:EXT + h_ + EXT, :] = e[EXT + h_ - 1:EXT + h_, :]

    def sgr_search_plane(p):
        sgr_extend(p)
        d_sgr_sums[p].zero_()
        h_, w_ = d_cdef_out[p].shape
        ctx.check(L.svt_hip_sgr_search_plane_dev(ctx.h, 1, 8, d_ext[p].data_ptr() + EXT * d_ext[p].shape[1] + EXT, d_ext[p].shape[1], d_cur[p].data_ptr(),
                                                 strides[p], w_, h_, US[p], int(p > 0), 0xFFFF, d_sgr_sums[p].data_ptr()), "sgr search")

    def run_sgr_search():
        parallel([lambda p=p: sgr_search_plane(p) for p in range(3)])

    def sgr_apply_plane(p):
        h_, w_ = d_cdef_out[p].shape
        ctx.check(L.svt_hip_sgr_apply_plane_dev(ctx.h, 1, 8, d_ext[p].data_ptr() + EXT * d_ext[p].shape[1] + EXT, d_ext[p].shape[1], d_sgr_out[p].data_ptr(),
                                                strides[p], w_, h_, US[p], int(p > 0), d_recon[p].data_ptr(), strides[p],   # stripe context rows from the deblocked picture
                                                d_unit_ep[p].data_ptr(), d_unit_xqd[p].data_ptr()), "sgr apply")

    def run_sgr_apply():
        parallel([lambda p=p: sgr_apply_plane(p) for p in range(3)])

    all_stages = [
        dict(key="pyr", name="pyramids", run=run_pyramids, kernel="downsample_kernel+variance_pyramid_kernel"),
        dict(key="hme", name="hme_l0_l1_l2", run=run_hme, kernel="sad_loop_kernel"),
        dict(key="me", name="me_fullpel_85pu", run=run_me, kernel="me_fullpel_85pu_kernel"),
        dict(key="subpel", name="subpel_convolve", run=run_subpel, kernel="subpel_predict_kernel"),
        dict(key="txfm", name="fwd_txfm_quant", run=run_txfm, kernel="fwd_txfm_quant_multi_kernel"),
        dict(key="inv", name="inv_txfm_recon", run=run_inv, kernel="inv_txfm_add_multi_kernel"),
        dict(key="dlf", name="deblock", run=run_dlf, kernel="deblock_pass_kernel"),
        dict(key="cdef_search", name="cdef_search", run=run_cdef_search, kernel="cdef_search_luma_kernel"),
        dict(key="cdef_apply", name="cdef_apply", run=run_cdef_apply, kernel="cdef_apply_kernel"),
        dict(key="sgr_search", name="sgr_search", run=run_sgr_search, kernel="sgr_search8_kernel"),
        dict(key="sgr_apply", name="sgr_apply", run=run_sgr_apply, kernel="lr_apply8_kernel"),
    ]
    want = None if args.stages == "all" else set(args.stages.split(","))
    stages = [s for s in all_stages if want is None or s["key"] in want]

    # read only the source / reference pictures: independent of the reconstruction chain
    SOURCE_SIDE = tuple(k for k in args.side_keys.split(",") if k in ("pyr", "hme", "me", "subpel"))

    def step():
        if side is None:
            for st in stages:
                st["run"]()
            return
        base = S["cur"]
        side.wait_stream(base)
        with on(side):
            for st in stages:
                if st["key"] in SOURCE_SIDE:
                    st["run"]()
        for st in stages:
            if st["key"] not in SOURCE_SIDE:
                st["run"]()
        base.wait_stream(side)

    def capture(fn, reps=1):
        """One HIP graph of `reps` back-to-back calls of fn(): a frame step is ~80 short launches, replaying a captured
        graph takes the host (Python, ctypes) out of the timed region.  The library's _dev entry points only enqueue
        work on the context's stream, so they are capture-safe; the context is pointed at the capture stream meanwhile."""
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(stream)
        ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(cap.cuda_stream)))
        S["cur"] = cap
        try:
            with torch.cuda.graph(g, stream=cap):
                for _ in range(reps):
                    fn()
        finally:
            S["cur"] = stream
            ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(stream.cuda_stream)))
        return g

    step()                      # eager once: first-touch, lazy module loads
    torch.cuda.synchronize()
    use_graph = not args.no_graph
    if use_graph:
        g_step = capture(step)
        do_step = g_step.replay
    else:
        do_step = step
    for _ in range(args.warmup):
        do_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        do_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None, dev)

    # ---- per-stage device time with HIP events on the launch stream (outside the headline timing): `reps` back-to-back
    #      passes of one stage (one captured graph unless --no-graph), so the figure is kernel time, not launch gaps
    per_stage = {}
    for st in stages:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(5, args.steps)
        if use_graph:
            g = capture(st["run"], reps)
            g.replay()
            torch.cuda.synchronize()
            e0.record(stream)
            g.replay()
            e1.record(stream)
        else:
            e0.record(stream)
            for _ in range(reps):
                st["run"]()
            e1.record(stream)
        e1.synchronize()
        per_stage[st["name"]] = e0.elapsed_time(e1) / reps  # ms per frame
    dominant = max(stages, key=lambda s: per_stage[s["name"]])
    dom_ms = per_stage[dominant["name"]]
    achieved_gbs = BYTES_PER_SB[dominant["name"]] * n_sb / (dom_ms * 1e-3) / 1e9
    # HBM traffic of the dominant stage per frame from the PMC passes of the latest profiled round (tools/collect_profiles.sh:
    # FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs; summary committed as profiles/<round>/pmc_traffic.json)
    traffic, traffic_src, valu_busy = None, None, None
    try:
        rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "pmc_traffic.json")))
        if rounds:
            traffic_src = f"profiles/{rounds[-1]}/pmc_traffic.json"
            pt = json.load(open(os.path.join(ROOT, traffic_src)))
            frames = max((e["launches"] for k, e in pt.items() if k.startswith("me_fullpel_85pu_kernel")), default=0)
            names = dominant["kernel"].split("+")
            tot = sum((e.get("fetch_bytes_per_launch", 0.0) + e.get("write_bytes_per_launch", 0.0)) * e["launches"]
                      for k, e in pt.items() if any(k.startswith(nm) for nm in names))
            traffic = tot / frames if frames and tot else None
            # VALU issue utilisation of the same launches: SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs
            act = sum(e.get("sq", {}).get("SQ_ACTIVE_INST_VALU", 0.0) * e["launches"] for k, e in pt.items() if any(k.startswith(nm) for nm in names))
            dur = sum(e["avg_us"] * e["launches"] for k, e in pt.items() if any(k.startswith(nm) for nm in names))
            valu_busy = (act * 4.0) / (1024 * dur * 1e-6 * 2.4e9) if dur else None
    except (OSError, ValueError, KeyError):
        traffic = None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- parity spot check of what was just timed (first SB row of ME) — the checker, not the product
    parity_ok = None
    if any(s["key"] == "me" for s in stages):
        g_sad = d_sad.cpu().numpy().view(np.uint32); g_mv = d_mv.cpu().numpy().view(np.uint32)
        k = min(F.sb_cols, 8)
        o_sad, o_mv = mc.oracle_frame(orc, F.cur_y_p, F.ref_y_p, F.cur_y_p.shape[1], PAD, sbs, 0, 0, k)
        parity_ok = bool(np.array_equal(o_sad[:k], g_sad[:k]) and np.array_equal(o_mv[:k], g_mv[:k]))
        simd_lib = os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")
        if os.path.exists(simd_lib) and world == 1:   # the whole frame against the reference's own (SIMD) kernels: 2040 SBs x 85 PUs
            refb = C.CDLL(simd_lib)
            refb.refb_setup.restype = C.c_uint64; refb.refb_setup.argtypes = [C.c_uint64]; refb.refb_setup(0xFFFFFFFFFFFFFFFF)
            refb.refb_parallel.restype = C.c_double; refb.refb_parallel.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
            r_sad = np.zeros((n_sb, 85), np.uint32); r_mv = np.zeros((n_sb, 85), np.uint32)
            slots = (C.c_int64 * 10)(F.cur_y_p.ctypes.data, F.ref_y_p.ctypes.data, F.cur_y_p.shape[1], PAD, PAD, C.addressof(sbs), n_sb, 0, r_sad.ctypes.data, r_mv.ctypes.data)
            refb.refb_parallel(0, C.addressof(slots), n_sb, 1, min(len(os.sched_getaffinity(0)), 128), 1)
            parity_ok = bool(parity_ok and np.array_equal(r_sad, g_sad) and np.array_equal(r_mv, g_mv))

    # ---- CPU baseline: the oracle C port of the same stage chain, every host core, on a bounded sample of the
    #      same frame; composite SB/s = 1 / sum_k (seconds per SB of stage k)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        jobs = dict(hme=hme_host, conv=(CB, nblk16), unit=US[0])
        simd = os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")
        if args.cpu_baseline in ("auto", "reference") and os.path.exists(simd):
            cpu = cpu_baseline_reference(C.CDLL(simd), orc, F, sbs, mc, tc, stages, jobs)
            if args.cpu_baseline == "auto" and args.cpu_port_too:
                cpu["port"] = cpu_baseline(orc, F, sbs, mc, tc, stages, jobs)
        elif args.cpu_baseline == "reference":
            raise SystemExit("oracle/_ref/libsvtav1_ref_simd.so is not built (make -f oracle/Makefile.ref simd)")
        else:
            cpu = cpu_baseline(orc, F, sbs, mc, tc, stages, jobs)

    total_sb = n_sb * args.steps * world
    out = {
        "metric": METRIC, "value": total_sb / elapsed, "unit": "SB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "launch": ("eager" if not use_graph else "hip_graph_replay") + (", single stream" if args.serial else f", 1+1+{n_lanes} forked streams per step"),
        "config": {"workload": f"{W}x{H} 8-bit 4:2:0 synthetic frame, {n_sb} SBs/frame/GPU; stages: " + ",".join(s["name"] for s in stages)
                               + "; HME L0 64x32 / L1,L2 16x16 windows; ME 1 ref 64x64 search area; sub-pel 2d_sr on every 16x16; square tx tiling "
                                 "4..64 per SB, quantize_b qindex 60; deblock levels (20,20,12,12); CDEF full 64-strength search; SGR 16 sets, restoration units 256",
                   "stages_ms": per_stage, "parity_spot_check": parity_ok},   # ME: first SB row vs the oracle + (when oracle/_ref is present) the whole frame vs the reference's SIMD kernels
        "roofline": {"bound": "hbm", "kernel": dominant["kernel"], "achieved": achieved_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "valu_busy": valu_busy,   # fraction of VALU issue cycles used by the dominant stage's kernels (profiled round, 2.4 GHz)
                     "traffic_note": None if traffic is None else f"bytes per frame of the stage's launches, FETCH_SIZE + WRITE_SIZE from {traffic_src} "
                                     "(raw counters x 1024; narrow loads are uncalibrated on gfx950, Infinity-Cache hits included)",
                     "note": "algorithmic bytes/SB (SURVEY 8d) x SBs / HIP-event stage time of the slowest stage; the ME and CDEF-search "
                             "kernels are integer-VALU bound (DESIGN.md), see per_stage_gbs for the HBM-bound ones",
                     "per_stage_gbs": {s["name"]: BYTES_PER_SB[s["name"]] * n_sb / (per_stage[s["name"]] * 1e-3) / 1e9 for s in stages}},
        "cpu_baseline": cpu,
    }
    if any(s["key"] == "me" for s in stages):
        ms = per_stage["me_fullpel_85pu"]
        out["roofline"]["valu_me"] = {"achieved": 4096 * 4096 * n_sb / (ms * 1e-3) / 1e12, "peak": 1024 * 64 * 16 / 16.0 * 2.4e9 / 1e12,
                                      "unit": "T px-SAD/s", "note": "peak = 1024 SIMDs x 64 lanes x 16 abs-diff per v_qsad_pk_u16_u8 / 16 cyc x 2.4 GHz"}
        out["roofline"]["valu_me"]["frac"] = out["roofline"]["valu_me"]["achieved"] / out["roofline"]["valu_me"]["peak"]
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(orc, F, sbs, mc, tc, stages, jobs):
    """Oracle port, all host cores, bounded sample (about 10-30 s of CPU time in total)."""
    from conftest import ptr
    cores = max(1, min(os.cpu_count() or 1, 64))
    keys = {s["key"] for s in stages}
    sec_per_sb = {}

    def par(fn, chunks):
        t = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(fn, chunks))
        return time.perf_counter() - t

    def split(n):
        return [(i * n // cores, (i + 1) * n // cores) for i in range(cores) if (i + 1) * n // cores > i * n // cores]

    if "me" in keys:
        n = min(F.n_sb, cores * 8)
        t = par(lambda be: mc.oracle_frame(orc, F.cur_y_p, F.ref_y_p, F.cur_y_p.shape[1], F.pad, sbs, 0, be[0], be[1]), split(n))
        sec_per_sb["me_fullpel_85pu"] = t / n
    if "txfm" in keys or "inv" in keys:
        # forward + quant + inverse chain on every block list, first `frac` of each list
        recon = [np.zeros_like(p) for p in F.ref]
        total_blocks_px, t_sum = 0, 0.0
        for (kind, ts), descs in sorted(F.descs.items()):
            n = max(cores, len(descs) // 8)
            n = min(n, len(descs))
            scans = F.scans(ts)
            SC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in scans])
            for plane in ([0] if kind == 0 else [1, 2]):
                qp = F.qp[plane]
                def work(be, plane=plane, descs=descs, ts=ts, qp=qp, SC=SC):
                    orc.orc_txfm_chain_8bit(ptr(F.cur[plane]), F.cur[plane].shape[1], ptr(F.ref[plane]), F.ref[plane].shape[1],
                                            ptr(recon[plane]), recon[plane].shape[1], ptr(descs), be[0], be[1], ts, 0, ptr(qp), SC,
                                            tc.TX_SCALE[ts], None, None)
                t_sum += par(work, split(n))
                total_blocks_px += n * tc.TXW[ts] * tc.TXH[ts]
        sec_per_sb["fwd_txfm_quant+inv_txfm_recon"] = t_sum / (total_blocks_px / 6144.0)   # 6144 px per SB (4:2:0)
    if "dlf" in keys:
        band = min(F.h, 64 * 6)
        t = 0.0
        for p in range(3):
            ev, eh = F.edges[p]
            ph = band >> (p > 0)
            img = np.ascontiguousarray(F.ref[p][:ph + 8].copy())
            uh = ph // 4
            t0 = time.perf_counter()
            orc.orc_deblock_plane(ptr(img), 1, img.shape[1], 8, ptr(np.ascontiguousarray(ev[:uh])), ptr(np.ascontiguousarray(eh[:uh])), ev.shape[1], uh, 0)
            t += time.perf_counter() - t0
        sec_per_sb["deblock"] = t / (F.sb_cols * (band // 64)) / cores   # single-threaded measurement scaled to `cores` (rows are independent)
    if "cdef_search" in keys:
        P3, I3 = C.c_void_p * 3, C.c_int * 3
        n = min(F.n_sb, cores * 2)
        mse = np.zeros((2, F.n_sb, 64), np.uint64)
        def work(be):
            orc.orc_cdef_search_frame(P3(*[p.ctypes.data for p in F.ref]), I3(*[p.shape[1] for p in F.ref]), P3(*[p.ctypes.data for p in F.cur]),
                                      I3(*[p.shape[1] for p in F.cur]), 1, F.w, F.h, ptr(F.skip8), F.cdef_damping, 8, 0, ptr(mse), be[0], be[1])
        t = par(work, split(n))
        sec_per_sb["cdef_search"] = t / n
    if "cdef_apply" in keys:
        # whole-frame call on a 4-SB-row crop of the picture (the function has no fb range): single-threaded, scaled to `cores`
        P3, I3 = C.c_void_p * 3, C.c_int * 3
        rows = min(F.h, 256)
        ins = [np.ascontiguousarray(F.ref[p][:rows >> (p > 0)]) for p in range(3)]
        outs = [a.copy() for a in ins]
        nfb = F.sb_cols * (rows // 64)
        t0 = time.perf_counter()
        orc.orc_cdef_apply_frame(P3(*[a.ctypes.data for a in ins]), P3(*[a.ctypes.data for a in outs]), I3(*[a.shape[1] for a in ins]), 1, F.w, rows,
                                 ptr(np.ascontiguousarray(F.skip8[:(rows // 8) * (F.w // 8)])), ptr(F.cdef_y[:nfb].copy()), ptr(F.cdef_uv[:nfb].copy()), F.cdef_damping, 8)
        sec_per_sb["cdef_apply"] = (time.perf_counter() - t0) / nfb / cores
    if "pyr" in keys or "hme" in keys:
        W_, H_, st = F.w, F.h, F.cur_y_p.shape[1]
        org = F.pad * st + F.pad
        PQ, PS = 32, 16
        planes = {}
        t0 = time.perf_counter()
        for name, src_p in (("cur", F.cur_y_p), ("ref", F.ref_y_p)):
            q = np.zeros((H_ // 2 + 2 * PQ, W_ // 2 + 2 * PQ), np.uint8); s_ = np.zeros((H_ // 4 + 2 * PS, W_ // 4 + 2 * PS), np.uint8)
            orc.orc_downsample_2d(C.c_void_p(src_p.ctypes.data + org), st, W_, H_, C.c_void_p(q.ctypes.data + PQ * q.shape[1] + PQ), q.shape[1], 2, 1)
            orc.orc_downsample_2d(C.c_void_p(src_p.ctypes.data + org), st, W_, H_, C.c_void_p(s_.ctypes.data + PS * s_.shape[1] + PS), s_.shape[1], 4, 1)
            planes[name] = (s_, q, src_p)
        t_ds = time.perf_counter() - t0
        nv = min(F.n_sb, 256)
        mean, var = np.zeros(85, np.uint8), np.zeros(85, np.uint16)
        t0 = time.perf_counter()
        for i in range(nv):
            sx, sy = (i % F.sb_cols) * 64, (i // F.sb_cols) * 64
            orc.orc_variance_pyramid_sb(C.c_void_p(F.cur_y_p.ctypes.data + org + sy * st + sx), st, 0, ptr(mean), ptr(var))
        t_vp = (time.perf_counter() - t0) / nv
        if "pyr" in keys:
            sec_per_sb["pyramids"] = (t_ds / F.n_sb + t_vp) / cores   # single-threaded, rows / SBs independent
        if "hme" in keys:
            # the same three SvtHipSadLoop job lists the GPU stage gets, one C call per thread and level
            n = min(F.n_sb, cores * 8)
            t = 0.0
            for lvl in range(3):
                cur_t, ref_t = planes["cur"][lvl], planes["ref"][lvl]
                S_ = jobs["hme"][lvl]
                sad_o = np.zeros(F.n_sb, np.uint32); xy_o = np.zeros((F.n_sb, 2), np.int16)
                t += par(lambda be: orc.orc_sad_loop_batch(ptr(cur_t), cur_t.shape[1], ptr(ref_t), ref_t.shape[1], S_, be[0], be[1], ptr(sad_o), ptr(xy_o)), split(n))
            sec_per_sb["hme_l0_l1_l2"] = t / n
    if "subpel" in keys:
        CB_, nb = jobs["conv"]   # the SvtHipConvBlk list of the GPU stage (16 blocks per SB, raster order)
        n = min(nb, cores * 8 * 16)
        st = F.ref_y_p.shape[1]
        dst = np.zeros((F.h, F.w), np.uint8)
        t = par(lambda be: orc.orc_subpel_predict_batch(1, 8, C.c_void_p(F.ref_y_p.ctypes.data + F.pad * st + F.pad), st, ptr(dst), F.w, CB_, be[0], be[1]), split(n))
        sec_per_sb["subpel_convolve"] = t / (n / 16.0)
    if "sgr_search" in keys or "sgr_apply" in keys:
        # luma band of 4 unit rows, one unit column per work item; chroma adds half as many samples (x 1.5)
        EXT_ = 3
        rows = min(F.h, 256)
        ext = np.ascontiguousarray(np.pad(F.ref[0][:rows], EXT_, mode="edge")); est = ext.shape[1]
        eoff = EXT_ * est + EXT_
        ncol = F.w // 64
        nunit = ncol * (rows // 64)
        if "sgr_search" in keys:
            def work(c):
                sums = np.zeros((rows // 64, 16, 5), np.int64)
                orc.orc_sgr_search_plane(C.c_void_p(ext.ctypes.data + eoff + 64 * c), 1, est, C.c_void_p(F.cur[0].ctypes.data + 64 * c), F.cur[0].shape[1], 64, rows, 0, 0, 64, 8, 0xFFFF, ptr(sums))
            sec_per_sb["sgr_search"] = par(work, list(range(ncol))) / nunit * 1.5
        if "sgr_apply" in keys:
            uep = np.full(rows // 64, 3, np.uint8); uxqd = np.tile(np.array([-30, 40], np.int32), (rows // 64, 1)).copy()
            dst = np.zeros((rows, F.w), np.uint8)
            work_ext = [ext.copy() for _ in range(cores)]
            def work(c):
                e = work_ext[c % cores]
                orc.orc_sgr_apply_plane(C.c_void_p(F.ref[0].ctypes.data + 64 * c), F.ref[0].shape[1], C.c_void_p(e.ctypes.data + eoff + 64 * c), est, 1, 64, rows, 0, 0, 64, 8,
                                        ptr(uep), ptr(uxqd), C.c_void_p(dst.ctypes.data + 64 * c), F.w)
            sec_per_sb["sgr_apply"] = par(work, list(range(ncol))) / nunit * 1.5
    total = sum(sec_per_sb.values())
    return dict(value=1.0 / total if total > 0 else None, unit="SB/s", cores=cores, kind="port",
                sample="oracle C port (scalar, gcc -O2) of the same stage chain on a bounded sample of the same frame, "
                       f"{cores} threads; seconds per SB per stage: " + ", ".join(f"{k}={v:.2e}" for k, v in sec_per_sb.items())
                       + " (deblock, cdef_apply, pyramids timed single-threaded and divided by cores; SGR timed on luma and scaled x1.5 for 4:2:0)")


def cpu_baseline_reference(refb, orc, F, sbs, mc, tc, stages, jobs):
    """The reference's own kernels as its x86 build dispatches them (SSE2 .. AVX2 / AVX-512 through setup_common_rtcd_internal /
    setup_rtcd_internal with this host's CPU flags), driven by oracle/ref_bench.c over the SAME job lists as the HIP stages, on every
    hardware thread (a pthread pool inside ref_bench.c: no Python in the timed region), the WHOLE frame per stage, best of 3.
    tests/test_ref_bench.py shows these loops produce the oracle's outputs bit for bit."""
    from conftest import ptr
    refb.refb_setup.restype = C.c_uint64; refb.refb_setup.argtypes = [C.c_uint64]
    refb.refb_parallel.restype = C.c_double
    refb.refb_parallel.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    flags = refb.refb_setup(0xFFFFFFFFFFFFFFFF)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 512))
    keys = {s["key"] for s in stages}
    sec = {}
    keep = []

    def adr(x):
        if x is None: return 0
        if isinstance(x, np.ndarray):
            keep.append(x)
            return x.ctypes.data
        if isinstance(x, int): return x
        keep.append(x)
        return C.addressof(x)

    def run(stage, slots, n, chunk, reps=3):
        a = (C.c_int64 * len(slots))(*[adr(v) for v in slots])
        return refb.refb_parallel(stage, C.addressof(a), n, chunk, cores, reps)

    W_, H_, n_sb = F.w, F.h, F.n_sb
    st = F.cur_y_p.shape[1]
    org = F.pad * st + F.pad
    if "me" in keys:
        sad = np.zeros((n_sb, 85), np.uint32); mv = np.zeros((n_sb, 85), np.uint32)
        sec["me_fullpel_85pu"] = run(0, [F.cur_y_p, F.ref_y_p, st, F.pad, F.pad, sbs, n_sb, 0, sad, mv], n_sb, 1) / n_sb
    if "pyr" in keys or "hme" in keys:
        PQ, PS = 32, 16
        planes = {}
        t0 = time.perf_counter()
        for name, src_p in (("cur", F.cur_y_p), ("ref", F.ref_y_p)):
            q = np.zeros((H_ // 2 + 2 * PQ, W_ // 2 + 2 * PQ), np.uint8); s_ = np.zeros((H_ // 4 + 2 * PS, W_ // 4 + 2 * PS), np.uint8)
            orc.orc_downsample_2d(C.c_void_p(src_p.ctypes.data + org), st, W_, H_, C.c_void_p(q.ctypes.data + PQ * q.shape[1] + PQ), q.shape[1], 2, 1)
            orc.orc_downsample_2d(C.c_void_p(src_p.ctypes.data + org), st, W_, H_, C.c_void_p(s_.ctypes.data + PS * s_.shape[1] + PS), s_.shape[1], 4, 1)
            planes[name] = (s_, q, src_p)
        t_ds = time.perf_counter() - t0
        if "pyr" in keys:   # decimation + variance pyramid: the oracle's scalar C (tens of ns per SB either way), single-threaded / threads
            nv = min(n_sb, 256)
            mean, var = np.zeros(85, np.uint8), np.zeros(85, np.uint16)
            t0 = time.perf_counter()
            for i in range(nv):
                sx, sy = (i % F.sb_cols) * 64, (i // F.sb_cols) * 64
                orc.orc_variance_pyramid_sb(C.c_void_p(F.cur_y_p.ctypes.data + org + sy * st + sx), st, 0, ptr(mean), ptr(var))
            sec["pyramids"] = (t_ds / n_sb + (time.perf_counter() - t0) / nv) / cores
        if "hme" in keys:
            t = 0.0
            for lvl in range(3):
                cur_t, ref_t = planes["cur"][lvl], planes["ref"][lvl]
                sad_o = np.zeros(n_sb, np.uint32); xy_o = np.zeros((n_sb, 2), np.int16)
                t += run(1, [cur_t, cur_t.shape[1], ref_t, ref_t.shape[1], jobs["hme"][lvl], sad_o, xy_o], n_sb, 2)
            sec["hme_l0_l1_l2"] = t / n_sb
    if "subpel" in keys:
        CB_, nb = jobs["conv"]
        dst = np.zeros((H_, W_), np.uint8)
        sec["subpel_convolve"] = run(2, [F.ref_y_p.ctypes.data + org, st, dst, W_, CB_], nb, 32) / n_sb
    if "txfm" in keys or "inv" in keys:
        recon = [np.zeros_like(p) for p in F.ref]
        t_sum = 0.0
        for (kind, ts), descs in sorted(F.descs.items()):
            sc, isc = F.scans(ts), F.scan_tables(ts)
            px = tc.TXW[ts] * tc.TXH[ts]
            for plane in ([0] if kind == 0 else [1, 2]):
                t_sum += run(3, [F.cur[plane], F.cur[plane].shape[1], F.ref[plane], F.ref[plane].shape[1], recon[plane], recon[plane].shape[1], descs, ts, F.qp[plane],
                                 tc.TX_SCALE[ts], sc[0], sc[1], sc[2], isc[0], isc[1], isc[2]], len(descs), max(1, 4096 // px))
        sec["fwd_txfm_quant+inv_txfm_recon"] = t_sum / n_sb
    if "dlf" in keys:
        t = 0.0
        for p in range(3):
            ev, eh = F.edges[p]
            uh, uw = ev.shape
            slots = []
            nb_ = min(uh // 4, cores * 2) or 1   # private copies of row bands (+ 8 rows of margin): bands are filtered independently
            for i in range(nb_):
                u0, u1 = i * uh // nb_, (i + 1) * uh // nb_
                r0, r1 = max(4 * u0 - 8, 0), min(4 * u1 + 8, F.ref[p].shape[0])
                img = np.ascontiguousarray(F.ref[p][r0:r1]); keep.append(img)
                slots += [img.ctypes.data + (4 * u0 - r0) * img.shape[1], img.shape[1], np.ascontiguousarray(ev[u0:u1]), np.ascontiguousarray(eh[u0:u1]), uw, u1 - u0] + [0] * 10
            t += run(4, slots, nb_, 1, reps=1)
        sec["deblock"] = t / n_sb
    if "cdef_search" in keys:
        mse = np.zeros((2, n_sb, 64), np.uint64)
        sec["cdef_search"] = run(5, [F.ref[0], F.ref[1], F.ref[2]] + [p.shape[1] for p in F.ref] + [F.cur[0], F.cur[1], F.cur[2]] + [p.shape[1] for p in F.cur]
                                 + [W_, H_, F.skip8, F.cdef_damping, mse], n_sb, 1) / n_sb
    if "cdef_apply" in keys:
        outs = [p.copy() for p in F.ref]
        sec["cdef_apply"] = run(6, [F.ref[0], F.ref[1], F.ref[2]] + [p.shape[1] for p in F.ref] + outs + [W_, H_, F.skip8, F.cdef_y, F.cdef_uv, F.cdef_damping], n_sb, 2) / n_sb
    if "sgr_search" in keys or "sgr_apply" in keys:
        US = jobs.get("unit", 256)
        EXT_ = 3
        t_search = t_apply = 0.0
        for p in range(3):
            ssub = int(p > 0)
            ph, pw = F.ref[p].shape
            if "sgr_search" in keys:
                ext = np.ascontiguousarray(np.pad(F.ref[p], EXT_, mode="edge")); est = ext.shape[1]; eoff = EXT_ * est + EXT_
                nu = max((pw + US // 2) // US, 1) * max((ph + US // 2) // US, 1)
                lim = np.zeros((nu, 4), np.int32)
                orc.orc_rest_unit_limits(pw, ph, ssub, US, ptr(lim))
                xq = np.zeros((nu, 16, 2), np.int32)
                t_search += run(7, [ext.ctypes.data + eoff, est, F.cur[p], F.cur[p].shape[1], lim, 64 >> ssub, 64 >> ssub, 0xFFFF, xq], nu, 1, reps=2)
                keep.append(ext)
            if "sgr_apply" in keys:
                # horizontal bands of the plane, each filtered as a picture of its own by the reference's frame-level restoration
                # (boundary-line save, svt_av1_loop_restoration_filter_unit per unit / stripe): the same work per sample as one big picture
                band_h = US
                slots, nbands = [], 0
                y0 = 0
                while y0 < ph:
                    y1 = ph if ph - (y0 + band_h) < band_h else y0 + band_h
                    cdef_b = np.ascontiguousarray(np.pad(F.ref[p][y0:y1], EXT_, mode="edge")); dbl_b = np.ascontiguousarray(F.cur[p][y0:y1])
                    hb = y1 - y0
                    nu_b = max((pw + US // 2) // US, 1) * max((hb + US // 2) // US, 1)
                    uep = np.full(nu_b, 3, np.uint8); uxqd = np.tile(np.array([-30, 40], np.int32), (nu_b, 1)).copy(); dst = np.zeros((hb, pw), np.uint8)
                    slots += [p, pw << ssub, hb << ssub, dbl_b, pw, cdef_b.ctypes.data + EXT_ * cdef_b.shape[1] + EXT_, cdef_b.shape[1], dst, pw, US, uep, uxqd] + [0] * 4
                    keep.append(cdef_b)
                    nbands += 1
                    y0 = y1
                t_apply += run(8, slots, nbands, 1, reps=1)
        if "sgr_search" in keys: sec["sgr_search"] = t_search / n_sb
        if "sgr_apply" in keys: sec["sgr_apply"] = t_apply / n_sb
    total = sum(sec.values())
    return dict(value=1.0 / total if total > 0 else None, unit="SB/s", cores=cores, kind="reference",
                sample="the reference's own kernels as its x86 build dispatches them on this host (cpu flags 0x%x: SSE2..AVX2%s; oracle/_ref SIMD flavour built by "
                       "oracle/Makefile.ref from the reference sources, NASM-only helpers stubbed in C and not on this path), driven by oracle/ref_bench.c over the "
                       "same job lists as the HIP stages, whole frame per stage, %d pthreads (all hardware threads), best of 3; seconds per SB per stage: " % (
                           flags, " + AVX-512" if flags & (1 << 9) else "", cores)
                       + ", ".join(f"{k}={v:.2e}" for k, v in sec.items())
                       + " (pyramids: the oracle's scalar C / threads; deblock and restoration apply run on independent row bands; restoration search is "
                         "one work item per restoration unit, as in the reference's rest segments)")


if __name__ == "__main__":
    main()
