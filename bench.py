#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X SVT-AV1 hot path (BASELINE.json metric).

One "step" = one pass of every kernel class of the hot path over ONE BATCH of `--frames` (default 4) distinct synthetic 4K (3840x2160)
8-bit 4:2:0 frames = 2040 superblocks each, inputs resident in HBM before the timed region.  north_star batches
"all 64x64 superblocks of one (or many concurrent) frames": the frames of a batch are independent pictures (different streams / different
pictures of the open-loop stages), every kernel class still runs once per frame, the frames' launches are parallel branches of one HIP
graph.  `--frames 1` is the single-frame step; the JSON line carries the F = 1 / 4 / 8 sweep.  Consecutive steps rotate over
`--groups` batches of different frames, so the 256 MB Infinity Cache cannot hold a step's working set.

Per frame, in the encoder's data flow:
    source side (open loop, runs ahead of the coding loop):  HME pyramids + variance pyramid -> HME L0/L1/L2 -> integer ME (85 PUs, 64x64
        search area, 1 reference)
    reconstruction side: sub-pel prediction of every 16x16 luma block at THIS frame's integer ME vector of the block + a random eighth-pel phase
        (SURVEY 8(d) config 3 ii; the job list is built on the device from the ME table) -> residual + fwd txfm + quantize against THAT prediction
        (luma; chroma predicts from the co-located reference) -> inverse txfm + recon -> deblock (3 planes, V then H) -> CDEF 64-strength search
        -> finish_cdef_search's decision on the device (four strength-pair searches, count of pairs by RDCOST, every filter block's pair) -> CDEF
        apply with exactly those strengths -> the COMPLETE self-guided search of every restoration unit, 16 sets
        (sums, 2x2 solve, encode_xq, finer search, best set — all on the device) -> restoration apply with the sets / xqd that search chose.
`value` = superblocks per second over the whole job (all ranks, all frames of a step).

Multi-GPU: frames/streams are independent, so rank i processes its own frames on GPU i — no data-path collective;
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.  Scaling is "weak" (per-GPU work fixed).

PyTorch is plumbing only (device memory, streams, graph capture, distributed); every kernel is launched through the C ABI of
libsvtav1_hip.so (include/svt_hip.h).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

# The step's four frame chains are four branches of one HIP graph; ROCm spreads a graph's branches over its hardware queues (GPU_MAX_HW_QUEUES, default 4).  Measured on
# the MI355X in round 6 (profiles/r06/NOTES.md): 4 queues 7.8 ms per step, 5-8 queues 10.3 ms, 16 queues 12.8 ms -- more queues put more of the searches, each of which
# holds a compute unit's whole register file, side by side.  The default is what the figures are quoted on; it is pinned here so that an inherited setting cannot move them.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "encoded 4K 8-bit SB/s (ME+txfm+quant+loopfilter) per GPU; bit-exact vs C ref"
P3, I3 = C.c_void_p * 3, C.c_int * 3
# algorithmic bytes per SB, kernel names per stage and the roofline arithmetic: tools/roofline_defs.py (shared with tools/summarize_profiles.py)
CDEF_LAMBDA = 55473                  # av1_lambda_mode_decision8_bit_sse[120] (EbLambdaRateTables.h:227): full lambda of a key frame at the workload's base_q_idx
EXT = 3                              # RESTORATION_BORDER: recon / CDEF / restoration planes carry a 3-sample border


class Env:
    pass


class Pipeline:
    """The device-resident buffers of ONE frame and the launches of its stages (all through the C ABI)."""

    def __init__(self, E, F, rank):
        import torch
        self.E, self.F = E, F
        ctx, L, pkg, dev, tc, mc, workload = E.ctx, E.L, E.pkg, E.dev, E.tc, E.mc, E.workload
        W, H = F.w, F.h
        self.n_sb, PAD = F.n_sb, F.pad

        def T(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.T = T
        self.d_cur_p, self.d_ref_p = T(F.cur_y_p), T(F.ref_y_p)
        self.sbs = mc.windows_product(L, W, H, 64, 64)   # the product's own restatement of integer_search_sb's window clamp
        self.d_sbs = T(np.frombuffer(bytes(self.sbs), dtype=np.uint8).copy())
        self.d_sad = torch.zeros((self.n_sb, 85), dtype=torch.int32, device=dev)
        self.d_mv = torch.zeros((self.n_sb, 85), dtype=torch.int32, device=dev)
        self.d_cur = [T(p) for p in F.cur]
        self.strides = [p.shape[1] for p in F.cur]
        self.d_subpel = torch.zeros((H, W), dtype=torch.uint8, device=dev)
        # prediction of the transform chain: luma = this frame's sub-pel prediction plane, chroma = co-located reference
        self.d_pred = [self.d_subpel, T(F.ref[1]), T(F.ref[2])]
        # reconstruction / CDEF output / restoration output: one geometry (stride, 3-sample border) so that whole-buffer copies and the
        # in-place border extension need no repacking
        self.xs = [((p.shape[1] + 2 * EXT + 63) // 64) * 64 for p in F.cur]
        mk = lambda: [torch.zeros((p.shape[0] + 2 * EXT, self.xs[i]), dtype=torch.uint8, device=dev) for i, p in enumerate(F.cur)]
        self.b_recon, self.b_cdef, self.b_rest = mk(), mk(), mk()
        # deblocking: one out-of-place launch (both directions, three planes) into b_dbl; SVT_BENCH_DLF=twopass keeps the two in-place launches (A/B)
        self.dlf_fused = os.environ.get("SVT_BENCH_DLF", "fused") != "twopass"
        self.b_dbl = mk() if self.dlf_fused else self.b_recon
        org = lambda b, i: b[i].data_ptr() + EXT * self.xs[i] + EXT
        self.p_recon = [org(self.b_recon, i) for i in range(3)]
        self.p_dbl = [org(self.b_dbl, i) for i in range(3)]
        self.p_cdef = [org(self.b_cdef, i) for i in range(3)]
        self.p_rest = [org(self.b_rest, i) for i in range(3)]
        self.tx_jobs, self.keep = [], []
        for (kind, ts), descs in sorted(F.descs.items()):
            nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
            d_desc = T(descs)
            isc = [T(s) if s is not None else None for s in F.scan_tables(ts)]
            self.keep.append(isc)
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
                self.tx_jobs.append(dict(ts=ts, plane=plane, n=n, desc=d_desc, qs=qs, st=st,
                                         q=torch.zeros(n * nk, dtype=torch.int32, device=dev), dq=torch.zeros(n * nk, dtype=torch.int32, device=dev),
                                         eob=torch.zeros(n, dtype=torch.int16, device=dev), cul=torch.zeros(n, dtype=torch.int32, device=dev)))
        self.FJ = (pkg.FwdTxJob * len(self.tx_jobs))(); self.IJ = (pkg.InvTxJob * len(self.tx_jobs))()
        for k, j in enumerate(self.tx_jobs):
            p = j["plane"]
            self.FJ[k] = pkg.FwdTxJob(j["ts"], j["n"], self.d_cur[p].data_ptr(), self.strides[p], self.d_pred[p].data_ptr(), self.strides[p], j["desc"].data_ptr(), j["qs"], j["st"],
                                      None, j["q"].data_ptr(), j["dq"].data_ptr(), j["eob"].data_ptr(), j["cul"].data_ptr(), None)
            self.IJ[k] = pkg.InvTxJob(j["ts"], j["n"], j["dq"].data_ptr(), self.d_pred[p].data_ptr(), self.strides[p], self.p_recon[p], self.xs[p], j["desc"].data_ptr())
        self.EJ = (pkg.EncTxJob * len(self.tx_jobs))()   # the fused form: no dequantised coefficients in memory
        for k, j in enumerate(self.tx_jobs):
            p = j["plane"]
            self.EJ[k].fwd = pkg.FwdTxJob(j["ts"], j["n"], self.d_cur[p].data_ptr(), self.strides[p], self.d_pred[p].data_ptr(), self.strides[p], j["desc"].data_ptr(), j["qs"],
                                          j["st"], None, j["q"].data_ptr(), None, j["eob"].data_ptr(), j["cul"].data_ptr(), None)
            self.EJ[k].d_recon = self.p_recon[p]; self.EJ[k].recon_stride = self.xs[p]
        self.d_edges = [(T(ev), T(eh), ev.shape[1], ev.shape[0]) for ev, eh in F.edges]
        self.d_skip8 = T(F.skip8)
        self.d_mse = torch.zeros((2, self.n_sb, 64), dtype=torch.int64, device=dev)
        self.d_dir = torch.zeros(self.n_sb * 64, dtype=torch.uint8, device=dev)
        self.d_var = torch.zeros(self.n_sb * 64, dtype=torch.int32, device=dev)
        # CDEF strengths per filter block: written by the strength decision of THIS step (finish_cdef_search on the device), read by the apply
        self.d_cy = torch.zeros(self.n_sb, dtype=torch.uint8, device=dev); self.d_cuv = torch.zeros(self.n_sb, dtype=torch.uint8, device=dev)
        self.d_sel_state = torch.zeros(pkg.CDEF_SELECT_STATE_BYTES, dtype=torch.uint8, device=dev)
        self.d_fin = torch.zeros(88, dtype=torch.uint8, device=dev); self.d_sel_gi = torch.zeros(self.n_sb, dtype=torch.int32, device=dev)
        # pyramids / HME: 1/4 and 1/16 resolution source + reference, variance pyramid
        PADQ, PADS = workload.PADQ, workload.PADS
        qw, qh, sw_, sh_ = W // 2, H // 2, W // 4, H // 4
        self.d_cur_q = torch.zeros((qh + 2 * PADQ, qw + 2 * PADQ), dtype=torch.uint8, device=dev); self.d_ref_q = torch.zeros_like(self.d_cur_q)
        self.d_cur_s = torch.zeros((sh_ + 2 * PADS, sw_ + 2 * PADS), dtype=torch.uint8, device=dev); self.d_ref_s = torch.zeros_like(self.d_cur_s)
        self.d_ymean = torch.zeros((self.n_sb, 85), dtype=torch.uint8, device=dev); self.d_yvar = torch.zeros((self.n_sb, 85), dtype=torch.int16, device=dev)
        # aligned copy of the current luma for the variance pyramid (needs an 8-byte aligned origin/stride and 64 px of slack)
        vp = np.zeros((F.sb_rows * 64 + 64, F.sb_cols * 64 + 64), np.uint8); vp[:H, :W] = F.cur[0]
        self.d_vp = T(vp)
        self.hme_host = workload.hme_jobs(F)
        self.hme = [dict(S=T(np.frombuffer(bytes(S), np.uint8).copy()), sad=torch.zeros(self.n_sb, dtype=torch.int32, device=dev),
                         xy=torch.zeros((self.n_sb, 2), dtype=torch.int16, device=dev)) for S in self.hme_host]
        # sub-pel: every whole 16x16 luma block at its integer ME vector + a random eighth-pel phase (EIGHTTAP_REGULAR), written into the prediction
        # plane; the SvtHipConvBlk list is built on the device from this step's ME table
        self.nblk16 = (W // 16) * (H // 16)
        self.d_frac = T(np.random.default_rng(14 + rank + 1000 * F.seed).integers(0, 16, (self.nblk16, 2)).astype(np.uint8))
        self.d_cb = torch.zeros(self.nblk16 * C.sizeof(pkg.ConvBlk), dtype=torch.uint8, device=dev)
        # restoration: the complete per-unit self-guided search (device scratch from the library's own size query) and the apply fed by it
        self.US = [256, 256, 256]   # restoration unit size per plane: what the reference picks above CIF (set_restoration_unit_size, EbPictureControlSet.c:31-47)
        self.n_units = [max((F.cur[p].shape[1] + self.US[p] // 2) // self.US[p], 1) * max((F.cur[p].shape[0] + self.US[p] // 2) // self.US[p], 1) for p in range(3)]
        L.svt_hip_sgr_search_units_scratch_bytes.restype = C.c_size_t
        self.scr_bytes = [L.svt_hip_sgr_search_units_scratch_bytes(F.cur[p].shape[1], F.cur[p].shape[0], self.US[p]) for p in range(3)]
        self.d_scr = [torch.zeros(n, dtype=torch.uint8, device=dev) for n in self.scr_bytes]
        self.d_uxqd = [torch.zeros((n, 16, 2), dtype=torch.int32, device=dev) for n in self.n_units]
        self.d_uerr = [torch.zeros((n, 16), dtype=torch.int64, device=dev) for n in self.n_units]
        self.d_ubest = [torch.zeros(n, dtype=torch.uint8, device=dev) for n in self.n_units]
        self.d_ubx = [torch.zeros((n, 2), dtype=torch.int32, device=dev) for n in self.n_units]
        self.SGRJ = (pkg.SgrUnitsPlaneDev * 3)()
        for p in range(3):
            ph_, pw_ = F.cur[p].shape
            self.SGRJ[p] = pkg.SgrUnitsPlaneDev(self.p_cdef[p], self.xs[p], self.d_cur[p].data_ptr(), self.strides[p], pw_, ph_, self.US[p], int(p > 0), 0xFFFF,
                                                self.d_uxqd[p].data_ptr(), self.d_uerr[p].data_ptr(), self.d_ubest[p].data_ptr(), self.d_ubx[p].data_ptr(),
                                                self.d_scr[p].data_ptr(), self.scr_bytes[p])
        self.stage_fns = dict(pyr=self.run_pyramids, hme=self.run_hme, me=self.run_me, subpel=self.run_subpel, txfm=self.run_txfm, inv=self.run_inv, enc_txfm=self.run_enc_txfm,
                              dlf=self.run_dlf,
                              cdef_search=self.run_cdef_search, cdef_pick=self.run_cdef_pick, cdef_apply=self.run_cdef_apply, sgr_units=self.run_sgr_units, sgr_apply=self.run_sgr_apply)

    # ---------------------------------------------------------------- the kernel classes of a step
    def chk(self, rc, what):
        self.E.ctx.check(rc, what)

    def run_me(self):
        E, F = self.E, self.F
        self.chk(E.L.svt_hip_me_fullpel_frame_dev(E.ctx.h, self.d_cur_p.data_ptr(), self.d_ref_p.data_ptr(), F.cur_y_p.shape[1], F.pad, F.pad,
                                                  self.d_sbs.data_ptr(), self.n_sb, 0, self.d_sad.data_ptr(), self.d_mv.data_ptr()), "me")

    def run_txfm(self):   # one mixed-size launch per 16 (size, plane) job lists: the 19 lists of a frame are 400-4000 blocks each
        self.chk(self.E.L.svt_hip_fwd_txfm_quant_multi_dev(self.E.ctx.h, 1, self.FJ, len(self.tx_jobs)), "fwd")

    def run_enc_txfm(self):   # residual -> forward -> quantize -> inverse -> reconstruction, one launch per 16 job lists
        self.chk(self.E.L.svt_hip_enc_txfm_multi_dev(self.E.ctx.h, 1, 8, self.EJ, len(self.tx_jobs)), "enc txfm")

    def run_inv(self):
        self.chk(self.E.L.svt_hip_inv_txfm_add_multi_dev(self.E.ctx.h, 1, 8, self.IJ, len(self.tx_jobs)), "inv")

    def run_dlf(self):   # all three planes, both directions: one launch (out of place), or one launch per direction (in place)
        e = self.d_edges
        if self.dlf_fused:
            F = self.F
            self.chk(self.E.L.svt_hip_deblock_frame_fused_dev(self.E.ctx.h, P3(*self.p_recon), P3(*self.p_dbl), 1, I3(*self.xs), 8, I3(*[p.shape[1] for p in F.cur]),
                                                              I3(*[p.shape[0] for p in F.cur]), P3(*[e[p][0].data_ptr() for p in range(3)]),
                                                              P3(*[e[p][1].data_ptr() for p in range(3)]), I3(*[e[p][2] for p in range(3)]), I3(*[e[p][3] for p in range(3)]), 0), "dlf fused")
            return
        self.chk(self.E.L.svt_hip_deblock_frame_dev(self.E.ctx.h, P3(*self.p_recon), 1, I3(*self.xs), 8, P3(*[e[p][0].data_ptr() for p in range(3)]),
                                                    P3(*[e[p][1].data_ptr() for p in range(3)]), I3(*[e[p][2] for p in range(3)]), I3(*[e[p][3] for p in range(3)]), 0), "dlf")

    def run_cdef_search(self):
        F = self.F
        self.chk(self.E.L.svt_hip_cdef_search_frame_dev(self.E.ctx.h, 1, P3(*self.p_dbl), I3(*self.xs), P3(*[p.data_ptr() for p in self.d_cur]), I3(*self.strides), F.w, F.h,
                                                        self.d_skip8.data_ptr(), F.cdef_damping, 8, self.d_mse.data_ptr(), self.d_dir.data_ptr(), self.d_var.data_ptr()), "cdef search")

    def run_cdef_pick(self):
        """finish_cdef_search on the distortion table of the search that just ran: 1 + 40 + 1 launches, no host step; every filter block is listed (none is all-skip here)"""
        L, h, n = self.E.L, self.E.ctx.h, self.n_sb
        m0, m1 = self.d_mse.data_ptr(), self.d_mse.data_ptr() + n * 64 * 8
        self.chk(L.svt_hip_cdef_strength_select_dev(h, m0, m1, n, 0, 64, self.d_sel_state.data_ptr(), self.E.pkg.CDEF_SELECT_STATE_BYTES), "cdef select")
        self.chk(L.svt_hip_cdef_finish_dev(h, m0, m1, n, self.d_sel_state.data_ptr(), CDEF_LAMBDA, None, self.d_fin.data_ptr(), self.d_sel_gi.data_ptr(), self.d_cy.data_ptr(),
                                           self.d_cuv.data_ptr()), "cdef finish")

    @staticmethod
    def run_cdef_pick_batch(batch):
        """the strength decision of every frame of the batch in ONE set of launches (svt_hip_cdef_strength_select_multi_dev: the stage is 80 dependent launches of ~60 us
        of arithmetic per picture; four pictures share them), then each frame's finish_cdef_search tail"""
        P0 = batch[0]
        L, h, n = P0.E.L, P0.E.ctx.h, P0.n_sb
        VP = C.c_void_p * len(batch)
        m0 = VP(*[P.d_mse.data_ptr() for P in batch]); m1 = VP(*[P.d_mse.data_ptr() + n * 64 * 8 for P in batch]); st = VP(*[P.d_sel_state.data_ptr() for P in batch])
        P0.chk(L.svt_hip_cdef_strength_select_multi_dev(h, len(batch), m0, m1, n, 0, 64, st, P0.E.pkg.CDEF_SELECT_STATE_BYTES), "cdef select (batch)")

    def run_cdef_finish(self):
        L, h, n = self.E.L, self.E.ctx.h, self.n_sb
        m0, m1 = self.d_mse.data_ptr(), self.d_mse.data_ptr() + n * 64 * 8
        self.chk(L.svt_hip_cdef_finish_dev(h, m0, m1, n, self.d_sel_state.data_ptr(), CDEF_LAMBDA, None, self.d_fin.data_ptr(), self.d_sel_gi.data_ptr(), self.d_cy.data_ptr(),
                                           self.d_cuv.data_ptr()), "cdef finish")

    def run_cdef_apply(self):
        F, L, h = self.F, self.E.L, self.E.ctx.h
        self.chk(L.svt_hip_cdef_apply_frame_dev(h, 1, P3(*self.p_dbl), P3(*self.p_cdef), I3(*self.xs), F.w, F.h, self.d_skip8.data_ptr(), self.d_cy.data_ptr(),
                                                self.d_cuv.data_ptr(), F.cdef_damping, 8, self.d_dir.data_ptr(), self.d_var.data_ptr()), "cdef apply")

    def run_pyramids(self):
        E, F = self.E, self.F
        PAD, st = F.pad, F.cur_y_p.shape[1]
        for src_p, dst_t, pad_, step_ in ((self.d_cur_p, self.d_cur_q, E.workload.PADQ, 2), (self.d_cur_p, self.d_cur_s, E.workload.PADS, 4),
                                          (self.d_ref_p, self.d_ref_q, E.workload.PADQ, 2), (self.d_ref_p, self.d_ref_s, E.workload.PADS, 4)):
            self.chk(E.L.svt_hip_downsample_2d_dev(E.ctx.h, src_p.data_ptr() + PAD * st + PAD, st, F.w, F.h, dst_t.data_ptr() + pad_ * dst_t.shape[1] + pad_, dst_t.shape[1], step_, 1), "ds")
        self.chk(E.L.svt_hip_variance_pyramid_dev(E.ctx.h, self.d_vp.data_ptr(), self.d_vp.shape[1], F.sb_cols, self.n_sb, 0, self.d_ymean.data_ptr(), self.d_yvar.data_ptr()), "varpyr")

    def run_hme(self):
        for lvl, (cur_t, ref_t) in enumerate(((self.d_cur_s, self.d_ref_s), (self.d_cur_q, self.d_ref_q), (self.d_cur_p, self.d_ref_p))):
            j = self.hme[lvl]
            self.chk(self.E.L.svt_hip_sad_loop_batch_dev(self.E.ctx.h, cur_t.data_ptr(), cur_t.shape[1], ref_t.data_ptr(), ref_t.shape[1], j["S"].data_ptr(), self.n_sb,
                                                         j["sad"].data_ptr(), j["xy"].data_ptr()), "hme")

    def run_subpel(self):
        F = self.F
        self.chk(self.E.L.svt_hip_subpel_jobs_from_me_dev(self.E.ctx.h, self.d_mv.data_ptr(), F.sb_cols, F.w, F.h, self.d_frac.data_ptr(), self.d_cb.data_ptr()), "subpel jobs")
        self.chk(self.E.L.svt_hip_subpel_predict_batch_dev(self.E.ctx.h, 1, 8, self.d_ref_p.data_ptr() + F.pad * F.ref_y_p.shape[1] + F.pad, F.ref_y_p.shape[1],
                                                           self.d_subpel.data_ptr(), F.w, self.d_cb.data_ptr(), self.nblk16), "subpel")

    def run_sgr_units(self):
        """svt_extend_frame of the CDEF output (in place: the planes carry the border) + search_selfguided_restoration of every unit"""
        L, h, F = self.E.L, self.E.ctx.h, self.F
        for p in range(3):
            ph, pw = F.cur[p].shape
            self.chk(L.svt_hip_generate_padding_dev(h, self.p_cdef[p], 1, self.xs[p], pw, ph, EXT, EXT), "extend")
        self.chk(L.svt_hip_sgr_search_units_picture_dev(h, 1, 8, 3, self.SGRJ), "sgr units")   # one sums launch per plane, one walk launch for the picture

    def run_sgr_extend(self):
        L, h, F = self.E.L, self.E.ctx.h, self.F
        for p in range(3):
            ph, pw = F.cur[p].shape
            self.chk(L.svt_hip_generate_padding_dev(h, self.p_cdef[p], 1, self.xs[p], pw, ph, EXT, EXT), "extend")

    @staticmethod
    def run_sgr_units_batch(batch):
        """the restoration unit search of every frame of the batch in ONE pair of launches: the planes of up to four pictures share the sums / difference-plane launch and
        the walk launch (include/svt_hip.h: SVT_HIP_SGR_MAX_PLANES)"""
        P0 = batch[0]
        n = 3 * len(batch)
        arr = (type(P0.SGRJ[0]) * n)()
        for i, P in enumerate(batch):
            for p in range(3): arr[3 * i + p] = P.SGRJ[p]
        P0.chk(P0.E.L.svt_hip_sgr_search_units_picture_dev(P0.E.ctx.h, 1, 8, n, arr), "sgr units (batch)")

    def run_sgr_apply(self):   # every unit filtered with the set / xqd its search chose (device arrays), stripe context rows from the deblocked picture
        L, h, F = self.E.L, self.E.ctx.h, self.F
        for p in range(3):
            ph, pw = F.cur[p].shape
            self.chk(L.svt_hip_sgr_apply_plane_dev(h, 1, 8, self.p_cdef[p], self.xs[p], self.p_rest[p], self.xs[p], pw, ph, self.US[p], int(p > 0), self.p_dbl[p], self.xs[p],
                                                   self.d_ubest[p].data_ptr(), self.d_ubx[p].data_ptr()), "sgr apply")


STAGGER_DEFAULT = "off"
ALL_STAGES = [("pyr", "pyramids"), ("hme", "hme_l0_l1_l2"), ("me", "me_fullpel_85pu"), ("subpel", "subpel_convolve"), ("enc_txfm", "fwd_quant_inv_recon"),
              ("txfm", "fwd_txfm_quant"), ("inv", "inv_txfm_recon"), ("dlf", "deblock"), ("cdef_search", "cdef_search"), ("cdef_pick", "cdef_strength_select"), ("cdef_apply", "cdef_apply"), ("sgr_units", "sgr_units_search"),
              ("sgr_apply", "sgr_apply")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--frames", type=int, default=4, help="independent frames per step (their launches are parallel branches of one HIP graph)")
    ap.add_argument("--groups", type=int, default=0, help="distinct batches the steps rotate over (default: enough for >= 4 distinct frames)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-sweep", action="store_true", help="skip the F = 1 / 4 / 8 sweep")
    ap.add_argument("--no-transfers", action="store_true", help="skip the PCIe-inclusive measurement")
    ap.add_argument("--no-1080p", action="store_true", help="skip the extra 1920x1080 measurement (a second, short run of this script)")
    ap.add_argument("--no-variants", action="store_true", help="skip the configs[1] / configs[3] sub-lines (tools/bench_variants.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "reference", "port"],
                    help="reference = the reference's own SIMD kernels (oracle/_ref SIMD flavour); port = the oracle's scalar C; auto = reference when built")
    ap.add_argument("--me-waves", type=int, default=0, help="svt_hip_me_set_waves_per_sb value; 0 = 2 waves per superblock when several frames are in flight, 4 for one frame "
                    "(measured on the MI355X in round 6: alone the 85-PU search is fastest with 4 waves per superblock -- 0.38 against 0.41 ms -- but with four frames in flight the "
                    "128-thread workgroups share the chip better with the other frames' kernels: 7.66-7.70 against 7.77-7.79 ms per step)")
    ap.add_argument("--no-side", action="store_true", help="(ignored; kept for older command lines: a frame's chain is one stream -- the sub-pel stage reads this step's ME "
                                                            "table, and a second stream per frame for the source side measured slower anyway, 8.60 vs 8.26 ms)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one captured HIP graph per step")
    ap.add_argument("--stages", default="all", help="comma list (debug): " + ",".join(k for k, _ in ALL_STAGES))
    args = ap.parse_args()

    # --gpus N without a launcher: spawn the N ranks here (one process per GPU, torch.distributed.run on 127.0.0.1) and relay rank 0's JSON line.
    # Under a launcher (WORLD_SIZE set) the two must agree: a silent 1-GPU measurement labelled N would be worse than no number.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from conftest import load_package
    load_package()
    import importlib
    shard = importlib.import_module("svt_av1_amd.shard")
    rank, local_rank, world = shard.rank_env(args.gpus, os.environ)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists in the product path)")
    if torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    # every rank next to its GPU: threads and the first touch of the page-locked staging buffers on the GPU's NUMA node (the PCIe-inclusive rate of eight ranks on one
    # host depends on it); a platform that does not publish the topology leaves the rank where the launcher put it
    numa = None
    if world > 1 and not os.environ.get("SVT_BENCH_NO_NUMA"):
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            numa = shard.pin_rank_to_gpu_numa("%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id))
        except Exception:   # noqa: BLE001
            numa = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import me_common as mc
    import workload
    import txfm_common as tc
    E = Env()
    E.pkg = load_package()
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))   # the checker: parity spot check and (port) CPU baseline only, after the timed region
    E.ctx = E.pkg.Context(local_rank)
    E.L, E.mc, E.tc, E.workload = E.ctx.L, mc, tc, workload
    L, ctx = E.L, E.ctx
    # the step runs on a stream of its own, not on the (legacy, implicitly synchronising) default stream: copies queued on other streams overlap it
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)
    ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(stream.cuda_stream)))
    E.dev = dev = torch.device("cuda", local_rank)
    auto_me_waves = not args.me_waves
    if auto_me_waves: args.me_waves = 2 if args.frames > 1 else 4
    ctx.check(L.svt_hip_me_set_waves_per_sb(ctx.h, args.me_waves))
    ctx.check(L.svt_hip_me_set_big_windows(ctx.h, 0))   # every window of this workload is 64 x 64 candidates: no strip-walking launch needed

    W, H = args.width, args.height
    nF = max(1, args.frames)
    sweep_fs = [f for f in (1, 4, 8) if not args.no_sweep or f == nF]
    n_groups = args.groups if args.groups > 0 else max(2, (4 + nF - 1) // nF)
    n_pipes = max(nF * n_groups, 8 if (8 in sweep_fs and not args.no_sweep) else 0, 4)
    frames = [workload.Frame(W, H, seed=11 + 100 * rank + 7 * i) for i in range(n_pipes)]   # distinct pictures
    pipes = [Pipeline(E, F, rank) for F in frames]
    n_sb = pipes[0].n_sb
    want = None if args.stages == "all" else set(args.stages.split(","))
    # default: the fused transform stage; the separate forward / inverse launches remain selectable (--stages ...,txfm,inv,...)
    stages = [(k, n) for k, n in ALL_STAGES if (k not in ("txfm", "inv") if want is None else k in want)]

    # ---------------------------------------------------------------- streams / graphs
    # Each frame of the batch gets its own HIP stream, forked from and joined back into the stream that carries the step; inside a captured graph
    # these are parallel branches, so the short memory-side kernels of one frame fill the gaps next to the VALU-bound searches of another.
    cur = {"s": stream}

    class on:
        def __init__(self, st):
            self.st = st

        def __enter__(self):
            self.prev = cur["s"]
            cur["s"] = self.st
            ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(self.st.cuda_stream)))
            self.t = torch.cuda.stream(self.st)
            self.t.__enter__()

        def __exit__(self, *a):
            self.t.__exit__(*a)
            cur["s"] = self.prev
            ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(self.prev.cuda_stream)))

    max_f = max([nF] + (sweep_fs if not args.no_sweep else []))
    main_streams = [torch.cuda.Stream() for _ in range(max_f)]

    # SVT_BENCH_STAGGER="<stage>:<lag>": frame i of a step (i >= lag) starts its chain when frame i - lag has finished <stage>.  Four chains that start together run the
    # same stage at the same time, so all four sit in the latency-bound strength decision together (80 dependent launches, ~0.65 ms with the chip nearly idle); a
    # staggered start would put one frame's decision beside another frame's searches.  Measured on the MI355X (profiles/r03/bench_stagger_ab.txt): every staggered start is
    # SLOWER (8.51 ms unstaggered; me:1 9.01, enc_txfm:1 9.41, me:2 9.87, subpel:2 10.1, cdef_search:2 10.5, cdef_pick:1 11.9 ms) - the step is a fork / join, and what a
    # late chain gains beside the others' decisions it loses alone at the tail.  "off" (default): every chain starts at once.
    stagger = os.environ.get("SVT_BENCH_STAGGER", STAGGER_DEFAULT)
    stagger_stage, stagger_lag = (stagger.split(":")[0], int(stagger.split(":")[1])) if stagger not in ("", "off") else (None, 0)
    # SVT_BENCH_SGR_JOINT=1 (experiment): the restoration unit search of the step's frames as ONE pair of launches (twelve planes; the frame chains join before it).
    # Measured on the MI355X (gpurun_out/sgr_joint_ab.txt): 7.68-7.69 against 7.32-7.35 ms per step -- like the joint strength decision, the join idles the chains.
    joint_sgr = bool(os.environ.get("SVT_BENCH_SGR_JOINT"))
    joint_pick = bool(os.environ.get("SVT_BENCH_PICK_JOINT")) and not joint_sgr   # measured slower (9.9 vs 9.5 ms): the join idles the other streams for the length of the chain   # A/B: the strength decision per frame, inside each frame's own chain (the round-3 first form)

    # CDEF strength selection: the one-launch (resident) form when ONE frame is in flight, the launch-per-step form when several are (include/svt_hip.h: the
    # resident form's workgroups wait for each other, so selections serialise and other frames' kernels delay it; measured 12.1 against 9.5 ms at four frames,
    # 2.96 against 3.25 ms at one).  SVT_BENCH_SELECT_FORM=steps|resident forces one form everywhere.
    forced_form = {"steps": 0, "resident": 1}.get(os.environ.get("SVT_BENCH_SELECT_FORM", ""), None)

    def select_form(frames_in_flight):
        form = forced_form if forced_form is not None else (1 if frames_in_flight == 1 else 0)
        ctx.check(L.svt_hip_set_cdef_select_form(ctx.h, form))
        return form

    def batch_step(batch, stages=stages, pre=None):
        """every frame's chain on its own stream (SVT_BENCH_PICK_JOINT: the strength decision of the whole batch as one joint stage between the two halves of the chains)"""
        base = cur["s"]
        select_form(len(batch))
        keys = [k for k, _ in stages]
        split = keys.index("cdef_pick") if (joint_pick and "cdef_pick" in keys and len(batch) > 1) else None
        if joint_sgr and "sgr_units" in keys and len(batch) > 1 and split is None: split = keys.index("sgr_units")
        halves = [keys] if split is None else [keys[:split], keys[split + 1:]]
        for hi, half in enumerate(halves):
            used = []
            released = {}
            for i, P in enumerate(batch):
                ms = main_streams[i]
                ms.wait_stream(base)
                if stagger_stage in half and i - stagger_lag in released:
                    ms.wait_event(released[i - stagger_lag])
                used.append(ms)
                with on(ms):
                    if hi == 0 and pre is not None: pre(P)
                    if hi == 1 and not joint_sgr: P.run_cdef_finish()
                    for k in half:
                        P.stage_fns[k]()
                        if k == stagger_stage and len(batch) > 1:
                            released[i] = torch.cuda.Event()
                            released[i].record(ms)
            for st in used:
                base.wait_stream(st)
            if hi == 0 and split is not None:
                if joint_sgr:   # on the base stream: the border extension of every frame's CDEF output, then ONE unit search for all of them
                    for P in batch: P.run_sgr_extend()
                    Pipeline.run_sgr_units_batch(batch)
                else:
                    Pipeline.run_cdef_pick_batch(batch)   # on the base stream: after every frame's CDEF search, before every frame's CDEF apply

    def capture(fn, reps=1):
        """One HIP graph of `reps` back-to-back calls of fn(): a step is a few hundred short launches, replaying a captured graph takes the host
        (Python, ctypes) out of the timed region.  The library's _dev entry points only enqueue work on the context's stream, so they are
        capture-safe; the context is pointed at the capture stream meanwhile."""
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(stream)
        ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(cap.cuda_stream)))
        cur["s"] = cap
        try:
            with torch.cuda.graph(g, stream=cap):
                for _ in range(reps):
                    fn()
        finally:
            cur["s"] = stream
            ctx.check(L.svt_hip_set_stream(ctx.h, C.c_void_p(stream.cuda_stream)))
        return g

    use_graph = not args.no_graph

    def make_steps(f, stage_list=stages, pre=None):
        """the rotating step functions for batches of f frames over all pipelines (pre(P): extra launches at the head of a frame's chain)"""
        batches = [pipes[i:i + f] for i in range(0, len(pipes) - f + 1, f)]
        for b in batches:
            batch_step(b, stage_list, pre)      # eager once: first-touch, lazy module loads
        torch.cuda.synchronize()
        if use_graph:
            return [capture(lambda b=b: batch_step(b, stage_list, pre)).replay for b in batches]
        return [lambda b=b: batch_step(b, stage_list, pre) for b in batches]

    # SVT_BENCH_OVERLAP=1 (experiment, graph replays only): consecutive steps -- they work on distinct batches -- are replayed on two alternating streams, so that one
    # step's latency-bound phases (the strength decision) can sit beside the next step's searches; SVT_BENCH_OVERLAP_LAG_US=<n> delays the second stream once by n us.
    overlap = bool(os.environ.get("SVT_BENCH_OVERLAP")) and use_graph
    alt = [torch.cuda.Stream(), torch.cuda.Stream()] if overlap else None
    lag_cycles = int(float(os.environ.get("SVT_BENCH_OVERLAP_LAG_US", "0")) * 100)   # torch.cuda._sleep counts in units of ~10 ns here (100 MHz timer)

    def run_steps(fns, n):
        if not overlap or len(fns) < 2:
            for i in range(n):
                fns[i % len(fns)]()
            return
        for st in alt:
            st.wait_stream(torch.cuda.current_stream())
        if lag_cycles:
            with torch.cuda.stream(alt[1]):
                torch.cuda._sleep(lag_cycles)
        for i in range(n):
            with torch.cuda.stream(alt[i % 2]):
                fns[i % len(fns)]()
        for st in alt:
            torch.cuda.current_stream().wait_stream(st)

    def timed(fns, steps, warmup, barrier):
        run_steps(fns, warmup)
        torch.cuda.synchronize()
        if barrier and world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(fns, steps)
        torch.cuda.synchronize()
        if barrier and world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # SVT_BENCH_FREERUN=1 (experiment): no fork / join per step -- every frame slot is a stream of its own that replays ITS frame chain (one captured graph per frame) step
    # after step; the slots start a quarter of a frame apart and drift freely, so at any moment the frames in flight are in different stages (one frame's strength
    # decision -- 80 short dependent launches -- sits beside the others' searches instead of beside their decisions).  Same work, same outputs; a "step" is still
    # nF frames, the timed region ends when every slot has finished its last frame.  Measured on the MI355X (gpurun_out/freerun_ab.txt, profiles/r06/NOTES.md): SLOWER, 8.46-8.59
    # against 7.39-7.42 ms per step at any start lag -- a frame's chain of short dependent launches queues behind the other frames' long search workgroups (a walk
    # workgroup lasts ~70 us) at every launch, while chains that run in lockstep find the chip idle together.  The fork / join per step is what keeps them in lockstep.
    free_run = bool(os.environ.get("SVT_BENCH_FREERUN")) and use_graph and nF > 1
    free_lag_us = float(os.environ.get("SVT_BENCH_FREERUN_LAG_US", "-1"))

    def make_free(f):
        batches = [pipes[i:i + f] for i in range(0, len(pipes) - f + 1, f)]
        select_form(f)
        for b in batches:
            batch_step(b)      # eager once: first-touch, lazy module loads
        torch.cuda.synchronize()
        graphs = [[capture(lambda P=P: [P.stage_fns[k]() for k, _ in stages]) for P in b] for b in batches]
        return graphs

    def timed_free(graphs, steps, warmup, barrier, f):
        nbat = len(graphs)

        def run(n, lag):
            for i in range(f):
                main_streams[i].wait_stream(stream)
                if lag > 0 and i:
                    with torch.cuda.stream(main_streams[i]):
                        torch.cuda._sleep(int(lag * i * 100))   # ~10 ns units (100 MHz timer)
            for s in range(n):
                for i in range(f):
                    with torch.cuda.stream(main_streams[i]):
                        graphs[s % nbat][i].replay()
            for i in range(f):
                stream.wait_stream(main_streams[i])
        run(warmup, 0)
        torch.cuda.synchronize()
        if barrier and world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps, free_lag_us)
        torch.cuda.synchronize()
        if barrier and world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    step_fns = make_steps(nF)
    free_graphs = make_free(nF) if free_run else None
    if free_run and free_lag_us < 0: free_lag_us = timed_free(free_graphs, 8, 2, False, nF) / 8 / nF * 1e6   # us per frame, measured: the slots start this far apart
    elapsed = timed_free(free_graphs, args.steps, args.warmup, True, nF) if free_run else timed(step_fns, args.steps, args.warmup, True)   # this rank's own time ...
    per_rank_ms = [t / args.steps * 1e3 for t in shard.gather_floats(elapsed, dist if world > 1 else None, dev)]
    elapsed = shard.max_over_ranks(elapsed, dist if world > 1 else None, dev)   # ... the job's time is the slowest rank's

    # what the TIMED replays left in frame 0's buffers (device -> host copies only; compared with the gated pass further down)
    timed_final = final_outputs(pipes[0]) if rank == 0 else None

    # ---- the same step at other batch sizes (rank 0 reports; a few dozen steps each)
    sweep = {str(nF): {"ms_per_step": elapsed / args.steps * 1e3, "sb_per_s": nF * n_sb * args.steps / elapsed}}
    if not args.no_sweep:
        for f in sweep_fs:
            if f == nF or f > len(pipes):
                continue
            if auto_me_waves: ctx.check(L.svt_hip_me_set_waves_per_sb(ctx.h, 2 if f > 1 else 4))   # read at capture time: each batch size runs the setting it would be used with
            fns = make_steps(f)
            n = max(10, min(args.steps, 40))
            t = timed(fns, n, 3, False)
            sweep[str(f)] = {"ms_per_step": t / n * 1e3, "sb_per_s": f * n_sb * n / t}
            del fns
        if auto_me_waves: ctx.check(L.svt_hip_me_set_waves_per_sb(ctx.h, args.me_waves))
    # ---- the step without the restoration stages = BASELINE.json configs[2] (ME + sub-pel + transform + deblock + CDEF on 4K 8-bit); the self-guided
    #      search and filter are configs[3]'s work, included in the headline step because the round-1 review asked for the complete search there
    subsets = {}
    if not args.no_sweep:
        sub = [(k, n_) for k, n_ in stages if not k.startswith("sgr")]
        if 0 < len(sub) < len(stages):
            fns = make_steps(nF, sub)
            n = max(10, min(args.steps, 40))
            t = timed(fns, n, 3, False)
            subsets["without_restoration (BASELINE configs[2]: " + ",".join(n_ for _, n_ in sub) + ")"] = {"ms_per_step": t / n * 1e3, "sb_per_s": nF * n_sb * n / t, "frames_per_step": nF}
            del fns

    # ---- the same step on 1920x1080 frames (north_star: "synthetic 1080p / 4K 4:2:0"): a short second run of this script, rank 0 of a 1-GPU job only
    also_1080p = None
    if world == 1 and not args.no_1080p and not args.no_sweep and (args.width, args.height) == (3840, 2160):
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--width", "1920", "--height", "1080", "--frames", str(4 * nF), "--steps", "30", "--warmup", "3", "--no-sweep",
                                "--no-transfers", "--no-cpu-baseline", "--no-1080p", "--no-variants"], capture_output=True, text=True, timeout=300)
            d2 = json.loads(r.stdout.strip().splitlines()[-1])
            also_1080p = {"value": d2["value"], "unit": d2["unit"], "ms_per_step": d2["ms_per_step"], "frames_per_step": 4 * nF, "workload": d2["config"]["workload"].split(";")[0],
                          "stages_ms": d2["config"]["stages_ms"]}
        except Exception as ex:   # the headline run must not depend on it
            also_1080p = {"error": str(ex)[:200]}

    # ---- the BASELINE.json configurations the headline step does not cover, as sub-lines (tools/bench_variants.py, one short subprocess each):
    #      configs[1] (1080p: ME with FULL / SUB_SAD search, transform chain 4..32 with quantize_b / quantize_fp at four q-indices) and configs[3] (4K 10-bit:
    #      HBD SAD / variance, the 64-point transform chain, the self-guided search and filter on 16-bit planes)
    also_10bit = config1_variants = config2_subpel = None
    if world == 1 and not args.no_variants and not args.no_sweep and (args.width, args.height) == (3840, 2160):
        import subprocess
        for which in ("10bit", "config1", "config2"):
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_variants.py"), "--which", which], capture_output=True, text=True, timeout=420)
                d2 = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as ex:   # the headline run must not depend on it
                d2 = {"error": str(ex)[:200]}
            if which == "10bit": also_10bit = d2
            elif which == "config1": config1_variants = d2
            else: config2_subpel = d2   # BASELINE configs[2]'s sub-pel part: the eight-neighbour probes + the x-only / y-only / copy convolves, each gated against the reference

    # ---- per-stage device time with HIP events on the launch stream (outside the headline timing): `reps` back-to-back passes of one stage of
    #      ONE frame (one captured graph unless --no-graph), so the figure is that stage's kernel time alone on an otherwise idle GPU
    per_stage = {}
    select_forms_ms = {}
    P0 = pipes[0]
    timing_list = list(stages) + ([("cdef_pick", "cdef_strength_select/steps_form")] if any(k == "cdef_pick" for k, _ in stages) and forced_form is None else [])
    for k, name in timing_list:
        form = 0 if name.endswith("/steps_form") else select_form(1)
        if name.endswith("/steps_form"):
            ctx.check(L.svt_hip_set_cdef_select_form(ctx.h, 0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        fn = P0.stage_fns[k]
        if use_graph:
            g = capture(fn, reps)
            g.replay()
            torch.cuda.synchronize()
            e0.record(stream)
            g.replay()
            e1.record(stream)
        else:
            fn()
            torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(reps):
                fn()
            e1.record(stream)
        e1.synchronize()
        if k == "cdef_pick":
            select_forms_ms["resident" if form == 1 else "steps"] = e0.elapsed_time(e1) / reps
        if not name.endswith("/steps_form"):
            per_stage[name] = e0.elapsed_time(e1) / reps  # ms per frame

    # ---- PCIe-inclusive rate (SURVEY 8(d) "with and without transfers"): every step additionally uploads its batch's source pictures from pinned
    #      host memory (padded luma + U + V = what a new input picture is; references are earlier inputs and already resident) and downloads its
    #      results (ME SAD / MV tables, CDEF distortion table, restoration search results, the restored picture) on a copy stream; uploads of batch
    #      i+1 and downloads of batch i-1 overlap the compute of batch i
    with_transfers = None
    if not args.no_transfers and len(step_fns) >= 2:   # every rank at the same time when there are several: the host side (page-locked copies over PCIe) is what they share
        def unpad(P):   # the un-padded views the transform / filter stages read: device-to-device, at the head of the frame's own chain (a kernel on the
            # copy stream would queue behind a whole step's launches: ROCm runs the streams on a few in-order hardware queues)
            P.d_cur[0].copy_(P.d_cur_p[P.F.pad:P.F.pad + P.F.h, P.F.pad:P.F.pad + P.F.w], non_blocking=True)
            P.d_vp[:P.F.h, :P.F.w].copy_(P.d_cur[0], non_blocking=True)
        if world > 1:
            dist.barrier()
        with_transfers = measure_with_transfers(torch, stream, pipes, nF, make_steps(nF, stages, unpad), n_sb, max(10, min(args.steps, 40)))
    per_rank_with_transfers = shard.gather_floats(with_transfers["value"] if isinstance(with_transfers, dict) and "value" in with_transfers else None, dist if world > 1 else None, dev) if world > 1 else None
    per_rank_numa = None
    if world > 1:
        per_rank_numa = [None if n != n else int(n) for n in shard.gather_floats(numa["node"] if numa else None, dist, dev)]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- parity spot check of what was just timed — the checker, not the product
    parity_ok = None
    F0 = frames[0]
    if any(k == "me" for k, _ in stages):
        g_sad = P0.d_sad.cpu().numpy().view(np.uint32); g_mv = P0.d_mv.cpu().numpy().view(np.uint32)
        k8 = min(F0.sb_cols, 8)
        o_sad, o_mv = mc.oracle_frame(orc, F0.cur_y_p, F0.ref_y_p, F0.cur_y_p.shape[1], F0.pad, P0.sbs, 0, 0, k8)
        parity_ok = bool(np.array_equal(o_sad[:k8], g_sad[:k8]) and np.array_equal(o_mv[:k8], g_mv[:k8]))
        simd_lib = os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")
        if os.path.exists(simd_lib) and world == 1:   # the whole frame against the reference's own (SIMD) kernels: 2040 SBs x 85 PUs
            refb = C.CDLL(simd_lib)
            refb.refb_setup.restype = C.c_uint64; refb.refb_setup.argtypes = [C.c_uint64]; refb.refb_setup(0xFFFFFFFFFFFFFFFF)
            refb.refb_parallel.restype = C.c_double; refb.refb_parallel.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
            r_sad = np.zeros((n_sb, 85), np.uint32); r_mv = np.zeros((n_sb, 85), np.uint32)
            slots = (C.c_int64 * 10)(F0.cur_y_p.ctypes.data, F0.ref_y_p.ctypes.data, F0.cur_y_p.shape[1], F0.pad, F0.pad, C.addressof(P0.sbs), n_sb, 0, r_sad.ctypes.data, r_mv.ctypes.data)
            refb.refb_parallel(0, C.addressof(slots), n_sb, 8, min(len(os.sched_getaffinity(0)), 128), 1)
            parity_ok = bool(parity_ok and np.array_equal(r_sad, g_sad) and np.array_equal(r_mv, g_mv))
    me_ok = parity_ok
    walk_stats = None
    if any(k == "sgr_units" for k, _ in stages):   # diagnostics the search leaves at the start of its scratch: passes / points per walk, unfinished walks
        st3 = [P0.d_scr[p][:128].cpu().numpy().view(np.uint32) for p in range(3)]
        nw = 16 * sum(P0.n_units)
        walk_stats = {"passes_per_walk": float(sum(int(s[0]) for s in st3)) / nw, "points_per_walk": float(sum(int(s[1]) for s in st3)) / nw,
                      "unfinished": int(sum(int(s[2]) for s in st3)),
                      # shader cycles per walk and plane as the walking wave sees them: solve, replay, wait for the unit to become resident, evaluation, total
                      # (zeros unless SVT_HIP_SGR_WALK_CLOCKS=1: the clocks cost 1.2 % of the stage and are off in the timed configuration)
                      "phase_cycles_per_walk": [{k: round(64.0 * int(s[24 + i]) / (16 * P0.n_units[p])) for i, k in enumerate(("solve", "replay", "load_wait", "evaluate", "total"))}
                                                | {"wave1_candidate_loops": round(64.0 * int(s[29]) / (16 * P0.n_units[p]))}
                                                | {"passes": round(int(s[0]) / (16 * P0.n_units[p]), 2), "points": round(int(s[1]) / (16 * P0.n_units[p]), 2)} for p, s in enumerate(st3)]}
        parity_ok = bool(parity_ok is not False and walk_stats["unfinished"] == 0)
        if os.environ.get("SVT_BENCH_SGR_RANGE"):   # development probe: the value ranges of the difference planes the unit search stores (per plane, per set)
            rng_out = []
            for p in range(3):
                ph_, pw_ = F0.cur[p].shape
                nu = P0.n_units[p]; dstride = (pw_ + 63) & ~63; dplane = dstride * ph_
                al = lambda v: (v + 255) & ~255
                o = al(128); o = al(o + 8 * nu * 16 * 5); o = al(o + 8 * nu); o = al(o + 4 * nu); sd_off = o; o = al(o + 2 * dplane); pairs_off = o
                sd = P0.d_scr[p][sd_off:sd_off + 2 * dplane].cpu().numpy().view(np.int16).reshape(ph_, dstride)[:, :pw_]
                row = {"plane": p, "sd_absmax": int(np.abs(sd).max())}
                US = P0.US[p]
                for ep in (0, 4, 9, 10, 14):
                    pr = P0.d_scr[p][pairs_off + 4 * dplane * ep: pairs_off + 4 * dplane * (ep + 1)].cpu().numpy().view(np.uint32).reshape(ph_, dstride)[:, :pw_]
                    f0 = (pr & 0xffff).astype(np.uint16).view(np.int16); f1 = (pr >> 16).astype(np.uint16).view(np.int16)
                    a0 = np.abs(f0.astype(np.int32)); a1 = np.abs(f1.astype(np.int32))
                    nuy, nux = max((ph_ + US // 2) // US, 1), max((pw_ + US // 2) // US, 1)
                    over = {"f0>2047": 0, "f0>1023": 0, "f1>1023": 0, "f1>511": 0}
                    for uy in range(nuy):
                        for ux in range(nux):
                            y1 = ph_ if uy == nuy - 1 else (uy + 1) * US; x1 = pw_ if ux == nux - 1 else (ux + 1) * US
                            m0 = a0[uy * US:y1, ux * US:x1].max(); m1 = a1[uy * US:y1, ux * US:x1].max()
                            over["f0>2047"] += int(m0 > 2047); over["f0>1023"] += int(m0 > 1023); over["f1>1023"] += int(m1 > 1023); over["f1>511"] += int(m1 > 511)
                    row[f"ep{ep}"] = {"f0_absmax": int(a0.max()), "f1_absmax": int(a1.max()), "f0_p999": int(np.percentile(a0, 99.9)), "f1_p999": int(np.percentile(a1, 99.9)), "units": nuy * nux, **over}
                rng_out.append(row)
            walk_stats["difference_plane_ranges"] = rng_out

    gate_detail = {}
    # ---- THE PARITY GATE: frame 0's chain once more, stage by stage, every stage's output against the reference's kernels run on the device's own
    #      input of that stage (tools/parity_gate.py); the timed replays' final buffers must equal this pass's.  No number without it.
    gate = None
    full_chain = [k for k, _ in stages] == ["pyr", "hme", "me", "subpel", "enc_txfm", "dlf", "cdef_search", "cdef_pick", "cdef_apply", "sgr_units", "sgr_apply"]
    if full_chain and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")):
        gate = run_parity_gate(E, P0, F0, stages, select_form, timed_final, orc, torch)
        gate_detail.update(gate["stages"])
        gate_detail["timed_step_outputs_equal_gated_pass"] = gate["timed_equal"]
        if gate["differences"]: gate_detail["gate_differences"] = gate["differences"]
        parity_ok = bool(parity_ok is not False and all(gate["stages"].values()) and gate["timed_equal"])
    else:
        gate_detail["gate"] = "not run: " + ("a subset of the stages was selected" if not full_chain else "oracle/_ref/libsvtav1_ref_simd.so is not built")

    # ---- CPU baseline: the reference's own kernels over the same job lists (or the oracle port), all hardware threads and one thread
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        stage_keys = [dict(key=k, name=n) for k, n in stages]
        CB0 = (E.pkg.ConvBlk * P0.nblk16).from_buffer_copy(P0.d_cb.cpu().numpy().tobytes())   # the job list the device built from frame 0's ME table
        jobs = dict(hme=P0.hme_host, conv=(CB0, P0.nblk16), unit=P0.US[0], cdef_mse=P0.d_mse.cpu().numpy().view(np.uint64), cdef_lambda=CDEF_LAMBDA,
                    cdef_strengths=(P0.d_cy.cpu().numpy(), P0.d_cuv.cpu().numpy()))
        simd = os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")
        try:   # a reported baseline, not the thing measured: whatever goes wrong in it must not cost the run its line
            if args.cpu_baseline in ("auto", "reference") and os.path.exists(simd):
                cpu = cpu_baseline_reference(C.CDLL(simd), orc, F0, P0.sbs, mc, tc, stage_keys, jobs)
            elif args.cpu_baseline == "reference":
                raise SystemExit("oracle/_ref/libsvtav1_ref_simd.so is not built (make -f oracle/Makefile.ref simd)")
            else:
                cpu = cpu_baseline(orc, F0, P0.sbs, mc, tc, stage_keys, jobs)
        except Exception as ex:   # noqa: BLE001
            cpu = {"value": None, "unit": "SB/s", "cores": 0, "kind": "reference", "sample": "not measured", "error": f"{type(ex).__name__}: {str(ex)[:300]}"}
    # the strength decision of frame 0 against the reference's (svt_search_one_dual x 75 + the RDCOST choice) on the same distortion table: the C functions decide
    # parity ("bit-exact vs C ref"); the dispatched SIMD kernels are compared as well and reported
    parity_detail = {"me_85pu_vs_reference": me_ok, "sgr_walks_unfinished": None if walk_stats is None else walk_stats["unfinished"]}
    parity_detail.update(gate_detail)
    if world == 1 and any(k == "cdef_pick" for k, _ in stages):
        m = np.ascontiguousarray(P0.d_mse.cpu().numpy().view(np.uint64)).reshape(2, n_sb, 64)
        g_fin = P0.d_fin.cpu().numpy(); g_sel = P0.d_sel_gi.cpu().numpy()
        parity_detail["cdef_distortion_table_max"] = [int(m[0].max()), int(m[1].max())]
        st_head = P0.d_sel_state[:3400].cpu().numpy()
        parity_detail["cdef_select_err_flag"] = int(st_head[3396:3400].view(np.uint32)[0])   # the one-launch form's "a gather timed out" flag
        g_bits = int(g_fin[:4].view(np.int32)[0]); g_y = g_fin[8:40].view(np.int32); g_uv = g_fin[40:72].view(np.int32)
        for flavour, libname in (("c", "libsvtav1_ref.so"), ("simd", "libsvtav1_ref_simd.so")):
            path = os.path.join(ROOT, "oracle", "_ref", libname)
            if not os.path.exists(path):
                continue
            rl = C.CDLL(path)
            rl.refb_setup.restype = C.c_uint64; rl.refb_setup.argtypes = [C.c_uint64]; rl.refb_setup(0xFFFFFFFFFFFFFFFF)
            rl.refb_cdef_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
            r_fin = np.zeros(17, np.int32); r_sel = np.zeros(n_sb, np.int32)
            rl.refb_cdef_finish(m[0].ctypes.data, m[1].ctypes.data, n_sb, CDEF_LAMBDA, r_fin.ctypes.data, r_sel.ctypes.data)
            same = bool(g_bits == int(r_fin[0]) and np.array_equal(g_y, r_fin[1:9]) and np.array_equal(g_uv, r_fin[9:17]) and np.array_equal(g_sel, r_sel))
            parity_detail["cdef_strength_decision_vs_reference_" + flavour] = same
            if not same:
                parity_detail["cdef_decision_" + flavour + "_diff"] = {"device": [g_bits] + g_y.tolist() + g_uv.tolist(), "reference": r_fin.tolist(),
                                                                    "per_block_differences": int(np.count_nonzero(g_sel != r_sel))}
            if flavour == "c":
                parity_ok = bool(parity_ok is not False and same)

    out = {
        "metric": METRIC, "value": (nF * n_sb * args.steps * world / elapsed) if parity_ok is not False else None, "unit": "SB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "launch": ("eager" if not use_graph else "hip_graph_replay") + f", {nF} frames per step, one forked stream per frame, "
                  f"steps rotate over {len(step_fns)} batches = {len(step_fns) * nF} distinct frames",
        "per_rank": {"ms_per_step": per_rank_ms, "sb_per_s_with_transfers": per_rank_with_transfers, "numa_node": per_rank_numa,
                     "note": "value = the units of all ranks / the slowest rank's time; with several ranks the PCIe-inclusive run of every rank is measured at the same time"},
        "value_with_transfers": with_transfers, "stage_subsets": subsets, "also_1080p": also_1080p, "also_10bit": also_10bit, "config1_variants": config1_variants, "config2_subpel": config2_subpel,
        "frames_per_step_sweep": sweep,
        "config": {"workload": f"{W}x{H} 8-bit 4:2:0 synthetic frames, {n_sb} SBs/frame, {nF} independent frames per step per GPU (F = 1 / 4 / 8 in frames_per_step_sweep); "
                               "stages per frame: " + ",".join(n for _, n in stages)
                               + "; HME L0 64x32 / L1,L2 16x16 windows; ME 1 ref 64x64 search area; sub-pel 2d_sr on every 16x16 at this step's ME vector of the block + a random "
                                 "eighth-pel phase; transform: luma residual against THAT sub-pel prediction, chroma against the co-located reference, "
                                 "square tx tiling 4..64 per SB, quantize_b qindex 60; deblock levels (20,20,12,12) on the reconstruction; CDEF full 64-strength search on the "
                                 "deblocked picture, finish_cdef_search's strength decision on that table (lambda of base_q_idx 120), apply with the strengths it chose; "
                                 "restoration: complete search_selfguided_restoration of every unit (16 sets, units 256, solve + finer search on the device) on the CDEF "
                                 "output, apply with the sets it chose; generator: numpy default_rng (PCG64) seeded with SURVEY 8(d)'s seed numbers (11 + 100 rank + 7 frame ...), not std::mt19937 - "
                                 "the device and the checkers read the same arrays, so parity does not depend on the generator (stated deviation from SURVEY 8(d))",
                   "stages_ms": per_stage, "stages_ms_note": "one frame, stage alone on an idle GPU (HIP events around 10 back-to-back passes); cdef_strength_select in the form a one-frame step uses",
                   "cdef_strength_select_forms_ms": select_forms_ms,
                   "cdef_strength_select_form": {"frames_per_step == 1": "resident (one launch)", "frames_per_step > 1": "steps (80 launches)"} if forced_form is None
                                                else os.environ.get("SVT_BENCH_SELECT_FORM"),
                   "me_waves_per_superblock": args.me_waves, "parity_spot_check": parity_ok, "parity_detail": parity_detail, "sgr_walk": walk_stats},
        "cpu_baseline": cpu,
    }
    out["roofline"] = roofline(per_stage, stages, n_sb)
    if isinstance(config2_subpel, dict) and "convolve_sr_16x16_blocks" in config2_subpel:   # the sub-line's kernels next to the step's stages (memory-bound: against HBM)
        for name, e in list(config2_subpel["convolve_sr_16x16_blocks"].items()) + [("upsampled_pred_variance_8_neighbours", config2_subpel["upsampled_pred_variance_8_neighbours"])]:
            out["roofline"]["stages"]["config2/" + name] = {k: e.get(k) for k in ("ms", "algorithmic_bytes", "algorithmic_GBps", "hbm_frac", "traffic_bytes")}
        if config2_subpel.get("gate") is False:
            out["value"] = None
            out["config"]["parity_spot_check"] = False
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if isinstance(config2_subpel, dict) and config2_subpel.get("gate") is False:
        raise SystemExit("bench.py: the configs[2] sub-line differs from the reference's kernels - the timing is not a result (config2_subpel.gate_vs_reference_simd says where)")
    if parity_ok is False:
        raise SystemExit("bench.py: the parity gate failed - the timing is not a result (parity_detail says which stage)")


def _interior(P, bufs, p):
    ph, pw = P.F.cur[p].shape
    return bufs[p][EXT:EXT + ph, EXT:EXT + pw].cpu().numpy().copy()


def final_outputs(P):
    """the buffers a finished step leaves behind for one frame (host copies)"""
    return {"sad": P.d_sad.cpu().numpy().copy(), "mv": P.d_mv.cpu().numpy().copy(), "mse": P.d_mse.cpu().numpy().copy(), "sel": P.d_sel_gi.cpu().numpy().copy(),
            "ubest": [t.cpu().numpy().copy() for t in P.d_ubest], "ubx": [t.cpu().numpy().copy() for t in P.d_ubx], "rest": [_interior(P, P.b_rest, p) for p in range(3)],
            "cdef_border": [t.cpu().numpy().copy() for t in P.b_cdef]}


def run_parity_gate(E, P, F, stages, select_form, timed_final, orc, torch):
    """Frame 0's chain once more, eagerly, one stage at a time; after every stage its outputs come to the host.  tools/parity_gate.py then recomputes every stage with
    the reference's kernels from the DEVICE's input of that stage and compares.  -> {"stages": {name: bool}, "timed_equal": bool, "differences": {...}}"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import parity_gate as G
    refb = G.setup_refb(C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")))
    select_form(1)
    S = {"sbs": P.sbs, "frac": P.d_frac.cpu().numpy()}
    for k, _ in stages:
        P.stage_fns[k]()
        torch.cuda.synchronize()
        if k == "pyr":
            S.update(cur_2=P.d_cur_q.cpu().numpy(), cur_4=P.d_cur_s.cpu().numpy(), ref_2=P.d_ref_q.cpu().numpy(), ref_4=P.d_ref_s.cpu().numpy(), ymean=P.d_ymean.cpu().numpy(),
                     yvar=P.d_yvar.cpu().numpy().view(np.uint16))
        elif k == "hme":
            for lvl, j in enumerate(P.hme):
                S[f"hme_sad_{lvl}"] = j["sad"].cpu().numpy().view(np.uint32); S[f"hme_xy_{lvl}"] = j["xy"].cpu().numpy()
        elif k == "me":
            S.update(sad=P.d_sad.cpu().numpy().view(np.uint32), mv=P.d_mv.cpu().numpy().view(np.uint32))
        elif k == "subpel":
            S.update(conv_jobs=P.d_cb.cpu().numpy(), subpel=P.d_subpel.cpu().numpy())
        elif k == "enc_txfm":
            for i, j in enumerate(P.tx_jobs):
                S[f"q_{i}"] = j["q"].cpu().numpy().reshape(j["n"], -1); S[f"eob_{i}"] = j["eob"].cpu().numpy().view(np.uint16)
            for p in range(3): S[f"recon_{p}"] = _interior(P, P.b_recon, p)
        elif k == "dlf":
            for p in range(3): S[f"dbl_{p}"] = _interior(P, P.b_dbl, p)
        elif k == "cdef_search":
            S["mse"] = P.d_mse.cpu().numpy().view(np.uint64)
        elif k == "cdef_pick":
            fin = P.d_fin.cpu().numpy()
            S.update(cdef_fin=np.concatenate([fin[:4].view(np.int32), fin[8:40].view(np.int32), fin[40:72].view(np.int32)]), cdef_sel=P.d_sel_gi.cpu().numpy(),
                     cdef_y=P.d_cy.cpu().numpy(), cdef_uv=P.d_cuv.cpu().numpy())
        elif k == "cdef_apply":
            for p in range(3): S[f"cdef_{p}"] = _interior(P, P.b_cdef, p)
        elif k == "sgr_units":
            for p in range(3):
                S[f"unit_ep_{p}"] = P.d_ubest[p].cpu().numpy(); S[f"unit_xqd_{p}"] = P.d_ubx[p].cpu().numpy()
        elif k == "sgr_apply":
            for p in range(3): S[f"rest_{p}"] = _interior(P, P.b_rest, p)
    threads = max(1, min(len(os.sched_getaffinity(0)), 128))
    ok, bad = G.check_chain(F, S, refb, orc, E.pkg, E.tc, E.workload, threads, CDEF_LAMBDA, unit_size=P.US[0])
    now = final_outputs(P)
    same = all(np.array_equal(now[k], timed_final[k]) for k in ("sad", "mv", "mse", "sel")) and all(
        np.array_equal(a, b) for k in ("ubest", "ubx", "rest", "cdef_border") for a, b in zip(now[k], timed_final[k]))
    return {"stages": {"gate_" + k: v for k, v in ok.items()}, "timed_equal": bool(same), "differences": bad}


def spawn_ranks(n):
    """python bench.py --gpus N: N ranks on this node through torch.distributed.run (RCCL only carries the barrier and the max-over-ranks time)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def measure_with_transfers(torch, stream, pipes, nF, step_fns, n_sb, steps):
    batches = [pipes[i:i + nF] for i in range(0, len(pipes) - nF + 1, nF)][:len(step_fns)]
    up_s = down_s = torch.cuda.Stream()   # one in-order copy stream (see run())
    if os.environ.get("SVT_BENCH_XFER_STREAMS") == "2": down_s = torch.cuda.Stream()   # experiment: uploads and downloads on streams of their own
    # the next step's upload is enqueued BEFORE this step's graph launch: the copy stream shares the four in-order hardware queues with the frame chains, and an
    # upload enqueued after the launch waits behind a whole step's kernels of the chain it shares a queue with (measured, gpurun_out/xfer_ab2.txt: uploads only
    # 8.50 -> 7.96 ms per step, uploads + downloads 9.97 -> 9.20; a second copy stream is slower: 11.8).  SVT_BENCH_XFER_ORDER=late restores the old order.
    early_up = os.environ.get("SVT_BENCH_XFER_ORDER", "early") == "early"
    pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host = []
    up_bytes = down_bytes = 0
    for P in pipes:
        ups = [(P.d_cur_p, pin(P.d_cur_p).copy_(P.d_cur_p.cpu())), (P.d_cur[1], pin(P.d_cur[1]).copy_(P.d_cur[1].cpu())), (P.d_cur[2], pin(P.d_cur[2]).copy_(P.d_cur[2].cpu()))]
        downs = [(P.d_sad, pin(P.d_sad)), (P.d_mv, pin(P.d_mv)), (P.d_mse, pin(P.d_mse))] + [(t, pin(t)) for t in P.d_ubest + P.d_ubx] + [(t, pin(t)) for t in P.b_rest]
        host.append((ups, downs))
        up_bytes = sum(d.numel() * d.element_size() for d, _ in ups); down_bytes = sum(d.numel() * d.element_size() for d, _ in downs)
    idx = {id(P): i for i, P in enumerate(pipes)}
    up_done = [torch.cuda.Event() for _ in batches]
    comp_done = [torch.cuda.Event() for _ in batches]
    down_done = [torch.cuda.Event() for _ in batches]

    def upload(b):
        with torch.cuda.stream(up_s):
            up_s.wait_event(comp_done[b])          # the previous compute on these buffers has finished (the downloads only read result buffers)
            for P in batches[b]:
                for d, h in host[idx[id(P)]][0]:
                    d.copy_(h, non_blocking=True)
            up_done[b].record(up_s)

    def download(b):
        with torch.cuda.stream(down_s):
            down_s.wait_event(comp_done[b])
            for P in batches[b]:
                for d, h in host[idx[id(P)]][1]:
                    h.copy_(d, non_blocking=True)
            down_done[b].record(down_s)

    nb = len(batches)
    for b in range(nb):
        comp_done[b].record(stream)
    torch.cuda.synchronize()

    mode = os.environ.get("SVT_BENCH_XFER", "both")   # diagnosis: up | down | both | none
    do_up, do_down = mode in ("up", "both"), mode in ("down", "both")

    def run(n):
        # ONE copy stream, commands in the order they can run: the next batch's upload (it only waits for the step before this one) goes in ahead of
        # this step's download (which waits for this step), so the upload is never queued behind a wait it does not depend on
        if do_up: upload(0)
        for i in range(n):
            b = i % nb
            if do_up: stream.wait_event(up_done[b])
            if do_down and i >= nb: stream.wait_event(down_done[b])   # the results of this batch's previous step have left the device
            if early_up and do_up and i + 1 < n: upload((i + 1) % nb)
            step_fns[b]()
            comp_done[b].record(stream)
            if not early_up and do_up and i + 1 < n: upload((i + 1) % nb)
            if do_down: download(b)
        torch.cuda.synchronize()

    run(3)
    t0 = time.perf_counter()
    run(steps)
    t = time.perf_counter() - t0
    return {"value": nF * n_sb * steps / t, "unit": "SB/s", "ms_per_step": t / steps * 1e3, "h2d_bytes_per_frame": up_bytes, "d2h_bytes_per_frame": down_bytes,
            "note": "every step uploads its frames' source pictures (padded luma, U, V) from pinned host memory and downloads ME tables, CDEF distortion table, restoration "
                    "search results and the restored picture, on one copy stream (the next batch's upload is enqueued ahead of this step's launch, this batch's download behind it) next to the neighbouring steps' compute"}


def roofline(per_stage, stages, n_sb):
    """tools/roofline_defs.py holds the arithmetic (shared with tools/summarize_profiles.py, which recomputes the same object from the committed profile): the
    dominant stage against the integer-VALU peak by what the ISA issued (SQ_INSTS_VALU x 64 from the latest profiles/<round>/pmc_traffic.json) and by the useful
    work count, plus algorithmic GB/s, counter traffic and traffic / algorithmic for every stage."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import roofline_defs as rd
    pmc, src = None, None
    try:
        rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "pmc_traffic.json")))
        if rounds:
            src = f"profiles/{rounds[-1]}/pmc_traffic.json"
            pmc = json.load(open(os.path.join(ROOT, src)))
    except (OSError, ValueError):
        pmc = None
    return rd.roofline({n: per_stage[n] for _, n in stages}, n_sb, pmc, src)


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
    if "txfm" in keys or "inv" in keys or "enc_txfm" in keys:
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
                                 ptr(np.ascontiguousarray(F.skip8[:(rows // 8) * (F.w // 8)])), ptr(jobs["cdef_strengths"][0][:nfb].copy()), ptr(jobs["cdef_strengths"][1][:nfb].copy()), F.cdef_damping, 8)
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
    if "sgr_units" in keys or "sgr_apply" in keys:
        # luma band of 4 unit rows, one unit column per work item; chroma adds half as many samples (x 1.5)
        EXT_ = 3
        rows = min(F.h, 256)
        ext = np.ascontiguousarray(np.pad(F.ref[0][:rows], EXT_, mode="edge")); est = ext.shape[1]
        eoff = EXT_ * est + EXT_
        ncol = F.w // 64
        nunit = ncol * (rows // 64)
        if "sgr_units" in keys:   # the complete unit search (64 x 64 units here: same work per sample)
            def work(c):
                nu_ = rows // 64
                xqd = np.zeros((nu_, 16, 2), np.int32); err = np.zeros((nu_, 16), np.int64); best = np.zeros(nu_, np.uint8)
                orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + eoff + 64 * c), 1, est, C.c_void_p(F.cur[0].ctypes.data + 64 * c), F.cur[0].shape[1], 64, rows, 0, 0, 64, 8,
                                               0xFFFF, ptr(xqd), ptr(err), ptr(best))
            sec_per_sb["sgr_units_search"] = par(work, list(range(ncol))) / nunit * 1.5
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
    hardware thread (a pthread pool inside ref_bench.c: no Python in the timed region), the WHOLE frame per stage, median of 5 (the stages that take seconds: once).
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

    sec1 = {}   # the same on ONE thread (BASELINE.md asks for T in {1, nproc})

    def run(stage, slots, n, chunk, reps=-5, name=None, one_thread_items=None):   # reps < 0: median of -reps; the stages of several seconds run once (reps=1)
        a = (C.c_int64 * len(slots))(*[adr(v) for v in slots])
        t = refb.refb_parallel(stage, C.addressof(a), n, chunk, cores, reps)
        if name is not None:   # one thread, a bounded prefix of the same items, scaled to the whole frame
            n1 = min(n, one_thread_items or n)
            sec1[name] = sec1.get(name, 0.0) + refb.refb_parallel(stage, C.addressof(a), n1, chunk, 1, 1) * (n / n1)
        return t

    W_, H_, n_sb = F.w, F.h, F.n_sb
    st = F.cur_y_p.shape[1]
    org = F.pad * st + F.pad
    if "me" in keys:
        sad = np.zeros((n_sb, 85), np.uint32); mv = np.zeros((n_sb, 85), np.uint32)
        sec["me_fullpel_85pu"] = run(0, [F.cur_y_p, F.ref_y_p, st, F.pad, F.pad, sbs, n_sb, 0, sad, mv], n_sb, 8, name="me_fullpel_85pu", one_thread_items=512) / n_sb   # 8 SBs per work item
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
                t += run(1, [cur_t, cur_t.shape[1], ref_t, ref_t.shape[1], jobs["hme"][lvl], sad_o, xy_o], n_sb, 8, name="hme_l0_l1_l2", one_thread_items=512)
            sec["hme_l0_l1_l2"] = t / n_sb
    if "subpel" in keys:
        CB_, nb = jobs["conv"]
        dst = np.zeros((H_, W_), np.uint8)
        sec["subpel_convolve"] = run(2, [F.ref_y_p.ctypes.data + org, st, dst, W_, CB_], nb, 64, name="subpel_convolve", one_thread_items=16 * 512) / n_sb
    if "txfm" in keys or "inv" in keys or "enc_txfm" in keys:
        recon = [np.zeros_like(p) for p in F.ref]
        t_sum = 0.0
        for (kind, ts), descs in sorted(F.descs.items()):
            sc, isc = F.scans(ts), F.scan_tables(ts)
            px = tc.TXW[ts] * tc.TXH[ts]
            for plane in ([0] if kind == 0 else [1, 2]):
                t_sum += run(3, [F.cur[plane], F.cur[plane].shape[1], F.ref[plane], F.ref[plane].shape[1], recon[plane], recon[plane].shape[1], descs, ts, F.qp[plane],
                                 tc.TX_SCALE[ts], sc[0], sc[1], sc[2], isc[0], isc[1], isc[2]], len(descs), max(1, 4096 // px), name="fwd_txfm_quant+inv_txfm_recon",
                             one_thread_items=max(64, len(descs) // 4))
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
                img = F.ref[p][r0:r1].copy(); keep.append(img)   # a private copy (ascontiguousarray of a row band is a VIEW: the filter would run in place on the frame the parity gate reads)
                slots += [img.ctypes.data + (4 * u0 - r0) * img.shape[1], img.shape[1], np.ascontiguousarray(ev[u0:u1]), np.ascontiguousarray(eh[u0:u1]), uw, u1 - u0] + [0] * 10
            t += run(4, slots, nb_, 1, reps=1, name="deblock", one_thread_items=max(1, nb_ // 8))
        sec["deblock"] = t / n_sb
    if "cdef_search" in keys:
        mse = np.zeros((2, n_sb, 64), np.uint64)
        sec["cdef_search"] = run(5, [F.ref[0], F.ref[1], F.ref[2]] + [p.shape[1] for p in F.ref] + [F.cur[0], F.cur[1], F.cur[2]] + [p.shape[1] for p in F.cur]
                                 + [W_, H_, F.skip8, F.cdef_damping, mse], n_sb, 4, name="cdef_search", one_thread_items=256) / n_sb
    if "cdef_pick" in keys:
        # finish_cdef_search's decision is ONE thread's work per picture in the reference (75 dispatched svt_search_one_dual steps + the RDCOST choice): timed as
        # `cores` pictures side by side (the device's own distortion table of frame 0 for each), i.e. as if the encoder kept that many pictures in flight
        m = np.ascontiguousarray(jobs["cdef_mse"]).reshape(2, n_sb, 64)
        npic = cores
        fin = np.zeros((npic, 17), np.int32); sel = np.zeros((npic, n_sb), np.int32)
        t = run(10, [m[0], m[1], n_sb, int(jobs["cdef_lambda"]), fin, sel], npic, 1, reps=1)
        sec["cdef_strength_select"] = t / npic / n_sb   # one picture per thread, all threads busy
        one = (C.c_int64 * 6)(*[adr(v) for v in [m[0], m[1], n_sb, int(jobs["cdef_lambda"]), fin, sel]])   # a NAMED array: the address of a temporary is a dangling pointer by the time the call reads it (an intermittent segmentation fault of this line until round 4)
        sec1["cdef_strength_select"] = refb.refb_parallel(10, C.addressof(one), 1, 1, 1, 1)
    if "cdef_apply" in keys:
        outs = [p.copy() for p in F.ref]
        cy_, cuv_ = jobs["cdef_strengths"]
        sec["cdef_apply"] = run(6, [F.ref[0], F.ref[1], F.ref[2]] + [p.shape[1] for p in F.ref] + outs + [W_, H_, F.skip8, cy_, cuv_, F.cdef_damping], n_sb, 4, name="cdef_apply", one_thread_items=256) / n_sb
    if "sgr_units" in keys or "sgr_apply" in keys:
        US = jobs.get("unit", 256)
        EXT_ = 3
        t_search = t_apply = 0.0
        for p in range(3):
            ssub = int(p > 0)
            ph, pw = F.ref[p].shape
            if "sgr_units" in keys:   # the reference's complete search_selfguided_restoration per unit (oracle/ref_shim_restpick.c), 16 sets
                ext = np.ascontiguousarray(np.pad(F.ref[p], EXT_, mode="edge")); est = ext.shape[1]; eoff = EXT_ * est + EXT_
                nu = max((pw + US // 2) // US, 1) * max((ph + US // 2) // US, 1)
                lim = np.zeros((nu, 4), np.int32)
                orc.orc_rest_unit_limits(pw, ph, ssub, US, ptr(lim))
                res = np.zeros((nu, 3), np.int32)
                t_search += run(9, [ext.ctypes.data + eoff, est, F.cur[p], F.cur[p].shape[1], lim, 64 >> ssub, 64 >> ssub, res], nu, 1, reps=1, name="sgr_units_search",
                                one_thread_items=4)
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
                t_apply += run(8, slots, nbands, 1, reps=1, name="sgr_apply", one_thread_items=2)
        if "sgr_units" in keys: sec["sgr_units_search"] = t_search / n_sb
        if "sgr_apply" in keys: sec["sgr_apply"] = t_apply / n_sb
    total = sum(sec.values())
    total1 = sum(sec1.values()) / n_sb + sec.get("pyramids", 0.0) * cores
    return dict(value=1.0 / total if total > 0 else None, unit="SB/s", cores=cores, kind="reference",
                value_one_thread=1.0 / total1 if total1 > 0 else None,
                one_thread_note="the same drivers on 1 thread over a bounded prefix of each stage's work items, scaled to the frame; seconds per SB per stage: "
                                + ", ".join(f"{k}={v / n_sb:.2e}" for k, v in sec1.items()),
                sample="the reference's own kernels as its x86 build dispatches them on this host (cpu flags 0x%x: SSE2..AVX2%s; oracle/_ref SIMD flavour built by "
                       "oracle/Makefile.ref from the reference sources, NASM-only helpers stubbed in C and not on this path), driven by oracle/ref_bench.c over the "
                       "same job lists as the HIP stages (the restoration search is the complete search_selfguided_restoration per unit), whole frame per stage, %d pthreads (all hardware "
                       "threads; ME and HME in work items of 8 SBs), median of 5 (deblocking, the strength decision and the restoration stages: one run); seconds per SB per stage: " % (
                           flags, " + AVX-512" if flags & (1 << 9) else "", cores)
                       + ", ".join(f"{k}={v:.2e}" for k, v in sec.items())
                       + " (pyramids: the oracle's scalar C / threads; deblock and restoration apply run on independent row bands; restoration search is "
                         "one work item per restoration unit, as in the reference's rest segments)")


if __name__ == "__main__":
    main()
