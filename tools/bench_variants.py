#!/usr/bin/env python3
"""Sub-lines of bench.py for the BASELINE.json configurations its headline step does not cover (run by bench.py as a subprocess on rank 0 of a 1-GPU job;
each prints ONE JSON object):

  --which config1   configs[1]: 1920x1080 8-bit 4:2:0, all 510 SBs — full-pel 85-PU ME with FULL_SAD_SEARCH and SUB_SAD_SEARCH, and the transform
                    chain residual -> fwd txfm2d 4..32 -> quantize -> inverse -> reconstruction with svt_aom_quantize_b and svt_av1_quantize_fp at
                    qindex 20 / 60 / 120 / 200 (SURVEY 8(d) config 2's variants)
  --which config2   configs[2], the sub-pel part: svt_upsampled_pref_error (svt_aom_upsampled_pred + svt_aom_variance16x16) of the eight half-pel neighbours of every
                    16x16 block's vector of a 4K frame, and the four svt_av1_convolve_{2d,x,y,2d_copy}_sr kernels on every 16x16 block, each gated against the reference's SIMD kernels
  --which 10bit     configs[3]: 3840x2160 10-bit 4:2:0 (16-bit planes) — sad_16b_kernel over a 64x64 window for every 64x64 block, HBD SAD + highbd_10
                    variance of every 64x64 / 32x32 pair, the 64-point transform chain (64x64, 64x32, 32x64, 64x16, 16x64; highbd quantize_b / quantize_fp)
                    of the whole luma plane, the complete self-guided unit search (16 sets, three planes) and the restoration apply with the sets it chose

Inputs are resident in HBM before the timed regions; every figure is HIP-event time of `--reps` back-to-back launches of the stage on one stream.  Bit-exactness
of every entry point used here is what tests/test_config4_hbd_gpu.py, tests/test_full4k_vs_reference_gpu.py and tests/test_txfm_gpu.py establish; this file only
measures."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def timed(torch, stream, fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def setup():
    import torch
    from conftest import load_package
    import me_common as mc
    import txfm_common as tc
    import workload
    import bench
    E = bench.Env()
    E.pkg = load_package()
    E.ctx = E.pkg.Context(0)
    E.L, E.mc, E.tc, E.workload = E.ctx.L, mc, tc, workload
    stream = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(stream)
    E.ctx.check(E.L.svt_hip_set_stream(E.ctx.h, C.c_void_p(stream.cuda_stream)))
    E.dev = torch.device("cuda", 0)
    E.ctx.check(E.L.svt_hip_me_set_big_windows(E.ctx.h, 0))
    return torch, bench, E, stream


def config1(reps):
    torch, bench, E, stream = setup()
    W, H = 1920, 1080
    out = {"config": "BASELINE configs[1]: 1920x1080 8-bit 4:2:0, 510 SBs: full-pel 85-PU ME (64x64 search area, 1 reference) + residual / fwd txfm2d 4..32 / quantize / "
                     "inverse / reconstruction of every block of the frame's tiling (luma + chroma)", "unit": "ms per frame (HIP events)", "me": {}, "txfm_chain": {}}
    n_sb = None
    for q in (20, 60, 120, 200):
        F = E.workload.Frame(W, H, seed=1, qindex=q)
        P = bench.Pipeline(E, F, 0)
        n_sb = P.n_sb
        if q == 20:
            for name, sub in (("FULL_SAD_SEARCH", 0), ("SUB_SAD_SEARCH", 1)):
                fn = lambda sub=sub: E.ctx.check(E.L.svt_hip_me_fullpel_frame_dev(E.ctx.h, P.d_cur_p.data_ptr(), P.d_ref_p.data_ptr(), F.cur_y_p.shape[1], F.pad, F.pad,
                                                                                   P.d_sbs.data_ptr(), P.n_sb, sub, P.d_sad.data_ptr(), P.d_mv.data_ptr()), "me")
                ms = timed(torch, stream, fn, reps)
                out["me"][name] = {"ms": ms, "sb_per_s": n_sb / (ms * 1e-3)}
            P.run_me(); P.run_subpel()   # the prediction the transform chain codes against
        else:
            P.run_me(); P.run_subpel()
        keep = [k for k, j in enumerate(P.tx_jobs) if j["ts"] <= 3]   # TX_4X4 .. TX_32X32
        for vname, variant in (("svt_aom_quantize_b", 0), ("svt_av1_quantize_fp", 2)):
            EJ = (E.pkg.EncTxJob * len(keep))()
            for i, k in enumerate(keep):
                EJ[i] = P.EJ[k]
                qp = F.qp[P.tx_jobs[k]["plane"]]
                EJ[i].fwd.qp.variant = variant
                if variant == 2:   # round_fp_qtx / quant_fp_qtx in round / quant (include/svt_hip.h)
                    for c in range(2):
                        EJ[i].fwd.qp.round[c] = int(qp[5][c]); EJ[i].fwd.qp.quant[c] = int(qp[6][c])
            fn = lambda EJ=EJ: E.ctx.check(E.L.svt_hip_enc_txfm_multi_dev(E.ctx.h, 1, 8, EJ, len(keep)), "enc txfm")
            ms = timed(torch, stream, fn, reps)
            nblk = sum(P.tx_jobs[k]["n"] for k in keep)
            out["txfm_chain"].setdefault(vname, {})[f"qindex_{q}"] = {"ms": ms, "blocks": nblk, "sb_per_s": n_sb / (ms * 1e-3)}
        del P
    out["n_sb"] = n_sb
    print(json.dumps(out))


def config2(reps):
    """BASELINE configs[2] / SURVEY 8(d) config 3 (ii), the sub-pel part the headline step only samples: (a) svt_upsampled_pref_error of the EIGHT neighbours of every
    16x16 block's vector (svt_first_level_check's probes of the first round, half-pel: svt_aom_upsampled_pred + svt_aom_variance16x16, mcomp.c:102-156,
    variance.c:212-269), (b) the four svt_av1_convolve_{2d,x,y,2d_copy}_sr kernels on every 16x16 block of the frame at fixed phases.  Every timed launch is then
    recomputed by the reference's SIMD kernels (oracle/_ref/libsvtav1_ref_simd.so, after the timed regions) and compared bit for bit."""
    torch, bench, E, stream = setup()
    pkg, L, ctx = E.pkg, E.L, E.ctx
    W, H = 3840, 2160
    F = E.workload.Frame(W, H, seed=11)
    n_sb = F.n_sb
    PAD = F.pad
    refp, cur = F.ref_y_p, F.cur[0]                     # padded reference luma (pad = 68), unpadded source luma
    st = refp.shape[1]
    rng = np.random.default_rng(5)
    nbx, nby = W // 16, H // 16
    nblk = nbx * nby
    mvx = rng.integers(-12, 13, nblk).astype(np.int32) * 8; mvy = rng.integers(-12, 13, nblk).astype(np.int32) * 8   # full-pel vectors, eighth-pel units
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(E.dev)
    d_ref, d_cur = T(refp), T(cur)
    out = {"config": "BASELINE configs[2] sub-pel part on one 3840x2160 8-bit frame (SURVEY 8(d) config 3 ii): 32 400 luma 16x16 blocks", "unit": "ms per frame (HIP events)", "n_sb": n_sb}
    HBM = 8.0e12
    # ---- (a) the eight half-pel neighbours of every block's vector: upsampled_pred (8-tap) + variance16x16
    jobs = (pkg.UpsampledBlk * (8 * nblk))() if hasattr(pkg, "UpsampledBlk") else None
    assert jobs is not None
    src_off = np.zeros(8 * nblk, np.int32)
    pairs = (pkg.BlkPair * (8 * nblk))()
    k = 0
    for b in range(nblk):
        bx, by = (b % nbx) * 16, (b // nbx) * 16
        for dy in (-4, 0, 4):
            for dx in (-4, 0, 4):
                if not dx and not dy: continue
                cx, cy = int(mvx[b]) + dx, int(mvy[b]) + dy
                ro = (by + PAD + (cy >> 3)) * st + bx + PAD + (cx >> 3)
                jobs[k] = pkg.UpsampledBlk(ro, k * 256, 16, 16, cx & 7, cy & 7, 0)
                src_off[k] = by * W + bx
                pairs[k] = pkg.BlkPair(0, k * 16, bx, by, 16, 16)   # prediction k is a 16-wide strip of a packed [8 nblk * 16][16] plane
                k += 1
    d_jobs, d_pairs = T(np.frombuffer(bytes(jobs), np.uint8).copy()), T(np.frombuffer(bytes(pairs), np.uint8).copy())
    d_pred = torch.zeros(8 * nblk * 256, dtype=torch.uint8, device=E.dev)
    d_var = torch.zeros(8 * nblk, dtype=torch.int32, device=E.dev); d_sse = torch.zeros(8 * nblk, dtype=torch.int32, device=E.dev)

    def probes_two_launches():   # round 4's form: every prediction written to HBM by one launch and read back by the next
        ctx.check(L.svt_hip_upsampled_pred_batch_dev(ctx.h, d_ref.data_ptr(), st, d_pred.data_ptr(), d_jobs.data_ptr(), 8 * nblk), "upsampled pred")
        ctx.check(L.svt_hip_block_variance_batch_dev(ctx.h, 1, 8, d_pred.data_ptr(), 16, d_cur.data_ptr(), W, d_pairs.data_ptr(), 8 * nblk, d_var.data_ptr(), d_sse.data_ptr()), "variance")
    ms_two = timed(torch, stream, probes_two_launches, reps)
    v2, s2 = d_var.cpu().numpy().view(np.uint32).copy(), d_sse.cpu().numpy().view(np.uint32).copy()
    # the fused form: one workgroup per block stages the window once, the nine half-pel positions' predictions never leave the chip (svt_hip_md_halfpel_grid_picture_dev)
    sb_cols, sb_rows = (W + 63) // 64, (H + 63) // 64
    pus16 = (pkg.MdPu * 16)(*[pkg.MdPu(16 * (i % 4), 16 * (i // 4), 16, 16) for i in range(16)])
    mvtab = np.full((sb_rows * sb_cols, 16), (0x8000 << 16) | 0x8000, np.uint32)
    slot_of = np.zeros(nblk, np.int64)
    for b in range(nblk):
        bx, by = (b % nbx) * 16, (b // nbx) * 16
        sbi, pui = (by // 64) * sb_cols + bx // 64, ((by % 64) // 16) * 4 + (bx % 64) // 16
        mvtab[sbi, pui] = ((int(mvy[b]) >> 3) & 0xffff) << 16 | ((int(mvx[b]) >> 3) & 0xffff)
        slot_of[b] = sbi * 16 + pui
    d_mvtab = T(mvtab)
    planes = (pkg.MdRefPlane * 1)(pkg.MdRefPlane(d_ref.data_ptr() + PAD * st + PAD, st, -PAD, -PAD, W + PAD, H + PAD))
    d_grid = torch.zeros(sb_rows * sb_cols * 16 * 18, dtype=torch.int32, device=E.dev)

    def probes():
        ctx.check(L.svt_hip_md_halfpel_grid_picture_dev(ctx.h, d_cur.data_ptr(), W, W, H, sb_cols, sb_rows * sb_cols, 16, pus16, 1, planes, d_mvtab.data_ptr(), 0, d_grid.data_ptr()), "half-pel grid")
    ms = timed(torch, stream, probes, reps)
    alg = nblk * (25 * 25 + 256 + 8 * 8)   # per block: the (16 + 8 + 1)^2 window its eight probes share + the source block, read once; 8 x (variance, sse) out
    out["upsampled_pred_variance_8_neighbours"] = {"ms": ms, "candidates": 8 * nblk, "algorithmic_bytes": alg, "algorithmic_GBps": alg / (ms * 1e-3) / 1e9, "hbm_frac": alg / (ms * 1e-3) / HBM,
                                                   "sb_per_s": n_sb / (ms * 1e-3), "traffic_bytes": None, "launches": 1, "two_launch_form_ms": ms_two,
                                                   "note": "one workgroup per 16x16 block: window staged once in LDS, horizontal pass once per horizontal offset, nine positions' statistics "
                                                           "on chip (also computes the centre); the round-4 form wrote 66 MB of predictions to HBM between two launches (two_launch_form_ms)"}
    grid = d_grid.cpu().numpy().view(np.uint32).reshape(-1, 9, 2)[slot_of]          # [nblk][9][2]
    order = [k for k in range(9) if k != 4]                                          # the eight neighbours in the job list's order (dy outer, dx inner)
    g_var, g_sse = grid[:, order, 0].reshape(-1), grid[:, order, 1].reshape(-1)
    two_launch_same = bool(np.array_equal(v2, g_var) and np.array_equal(s2, g_sse))
    # ---- (b) the four single-reference convolves on every 16x16 block at fixed phases (regular 8-tap)
    CB = (pkg.ConvBlk * nblk)()
    d_dst = torch.zeros((H, W), dtype=torch.uint8, device=E.dev)
    conv = {}
    conv_out = {}
    for name, sx, sy, alg_blk in (("svt_av1_convolve_2d_sr", 5, 11, 23 * 23 + 256), ("svt_av1_convolve_x_sr", 5, 0, 16 * 23 + 256), ("svt_av1_convolve_y_sr", 0, 11, 23 * 16 + 256),
                                  ("svt_av1_convolve_2d_copy_sr", 0, 0, 256 + 256)):
        for b in range(nblk):
            bx, by = (b % nbx) * 16, (b // nbx) * 16
            CB[b] = pkg.ConvBlk(bx + (int(mvx[b]) >> 3), by + (int(mvy[b]) >> 3), bx, by, 16, 16, 0, 0, sx, sy, 0, 0)
        d_cb = T(np.frombuffer(bytes(CB), np.uint8).copy())
        off = PAD * st + PAD
        fn = lambda d_cb=d_cb: ctx.check(L.svt_hip_subpel_predict_batch_dev(ctx.h, 1, 8, d_ref.data_ptr() + off, st, d_dst.data_ptr(), W, d_cb.data_ptr(), nblk), "convolve")
        ms = timed(torch, stream, fn, reps)
        alg = nblk * alg_blk
        conv[name] = {"ms": ms, "algorithmic_bytes": alg, "algorithmic_GBps": alg / (ms * 1e-3) / 1e9, "hbm_frac": alg / (ms * 1e-3) / HBM, "sb_per_s": n_sb / (ms * 1e-3), "traffic_bytes": None}
        conv_out[name] = (d_dst.cpu().numpy().copy(), np.frombuffer(bytes(CB), np.uint8).copy())
    out["convolve_sr_16x16_blocks"] = conv
    # ---- the gate (after every timed region): the reference's own SIMD kernels on the same jobs
    import ctypes as Cc
    from concurrent.futures import ThreadPoolExecutor
    simd = os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")
    gate = {}
    if os.path.exists(simd):
        R = Cc.CDLL(simd)
        R.refb_setup.restype = Cc.c_uint64; R.refb_setup.argtypes = [Cc.c_uint64]
        R.refb_setup(0xffffffffffffffff)
        nt = min(32, os.cpu_count() or 1)
        e_var, e_sse = np.zeros(8 * nblk, np.uint32), np.zeros(8 * nblk, np.uint32)
        jb = np.frombuffer(bytes(jobs), np.uint8).copy()
        vp = lambda a: a.ctypes.data_as(Cc.c_void_p)
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(lambda be: R.refb_upsampled_var_batch(vp(refp), st, vp(cur), W, vp(jb), vp(src_off), be[0], be[1], vp(e_var), vp(e_sse)),
                        [(i * 8 * nblk // nt, (i + 1) * 8 * nblk // nt) for i in range(nt)]))
        gate["upsampled_pred_variance_8_neighbours"] = bool(np.array_equal(e_var, g_var) and np.array_equal(e_sse, g_sse))
        gate["upsampled_pred_variance_two_launch_form"] = two_launch_same and gate["upsampled_pred_variance_8_neighbours"]
        for name, (got, cb) in conv_out.items():
            exp = np.zeros((H, W), np.uint8)
            base = Cc.c_void_p(refp.ctypes.data + PAD * st + PAD)
            with ThreadPoolExecutor(nt) as ex:
                list(ex.map(lambda be: R.refb_subpel_predict_batch(base, st, vp(exp), W, vp(cb), be[0], be[1]), [(i * nblk // nt, (i + 1) * nblk // nt) for i in range(nt)]))
            gate[name] = bool(np.array_equal(exp, got))
        out["gate_vs_reference_simd"] = gate
        out["gate"] = all(gate.values())
    else:
        out["gate"] = None
    print(json.dumps(out))


def hbd(reps):
    torch, bench, E, stream = setup()
    pkg, L, ctx, tc = E.pkg, E.L, E.ctx, E.tc
    W, H, BD = 3840, 2160, 10
    n_sb = ((W + 63) // 64) * ((H + 63) // 64)
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    base = 480 + 300 * np.sin(xx / 97.0) * np.cos(yy / 61.0) + 60 * (((xx // 16).astype(np.int32) + (yy // 16).astype(np.int32)) % 2)
    cur = np.clip(base + rng.normal(0, 9, (H, W)), 0, 1023).astype(np.uint16)
    ref = np.clip(np.roll(base, (3, -5), (0, 1)) + rng.normal(0, 9, (H, W)), 0, 1023).astype(np.uint16)
    planes_cur = [cur, np.ascontiguousarray(512 + (cur[::2, ::2].astype(np.int32) - 512) // 2).astype(np.uint16), np.ascontiguousarray(512 - (cur[::2, ::2].astype(np.int32) - 512) // 3).astype(np.uint16)]
    planes_rec = [ref, np.ascontiguousarray(512 + (ref[::2, ::2].astype(np.int32) - 512) // 2).astype(np.uint16), np.ascontiguousarray(512 - (ref[::2, ::2].astype(np.int32) - 512) // 3).astype(np.uint16)]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16) if a.dtype == np.uint16 else np.ascontiguousarray(a)).to(E.dev)
    out = {"config": "BASELINE configs[3]: 3840x2160 10-bit 4:2:0 (16-bit planes), 2040 SBs", "unit": "ms per frame (HIP events)", "dtype": "u16", "stages_ms": {}}
    PAD = 64
    refp = np.ascontiguousarray(np.pad(ref, PAD, mode="edge")); curp = np.ascontiguousarray(np.pad(cur, PAD, mode="edge"))
    d_curp, d_refp = T(curp), T(refp)
    st = refp.shape[1]
    # (1) sad_16b_kernel over a 64 x 64 window centred on every 64x64 block (svt_sad_loop_kernel's order): 2040 searches of 4096 candidates
    S = (pkg.SadLoop * n_sb)()
    k = 0
    for by in range(0, H, 64):
        for bx in range(0, W, 64):
            bh = min(64, H - by)
            S[k] = pkg.SadLoop(bx + PAD, by + PAD, bx + PAD - 32, by + PAD - 32, 64, bh, 64, 64, 1, 0); k += 1
    d_S = T(np.frombuffer(bytes(S), np.uint8).copy())
    d_bs = torch.zeros(n_sb, dtype=torch.int32, device=E.dev); d_bxy = torch.zeros((n_sb, 2), dtype=torch.int16, device=E.dev)
    out["stages_ms"]["hbd_sad_window_64x64"] = timed(torch, stream, lambda: ctx.check(L.svt_hip_sad_loop16_batch_dev(ctx.h, d_curp.data_ptr(), st, d_refp.data_ptr(), st, d_S.data_ptr(), n_sb,
                                                                                                                      d_bs.data_ptr(), d_bxy.data_ptr()), "sad16 loop"), max(2, reps // 2))
    # (2) HBD SAD + highbd_10 variance of every whole 64x64 and its four 32x32
    pairs = []
    for by in range(0, H - 63, 64):
        for bx in range(0, W - 63, 64):
            pairs.append((bx, by, bx + 1, by + 2, 64, 64))
            for qd in range(4):
                pairs.append((bx + 32 * (qd & 1), by + 32 * (qd >> 1), bx + 1 + 32 * (qd & 1), by + 2 + 32 * (qd >> 1), 32, 32))
    Pp = (pkg.BlkPair * len(pairs))(*[pkg.BlkPair(*p) for p in pairs])
    d_p = T(np.frombuffer(bytes(Pp), np.uint8).copy())
    d_cur, d_ref = T(cur), T(np.ascontiguousarray(np.pad(ref, ((0, 8), (0, 8)), mode="edge")))
    d_o = [torch.zeros(len(pairs), dtype=torch.int32, device=E.dev) for _ in range(3)]

    def sadvar():
        ctx.check(L.svt_hip_block_sad_batch_dev(ctx.h, 2, d_cur.data_ptr(), W, d_ref.data_ptr(), W + 8, d_p.data_ptr(), len(pairs), d_o[0].data_ptr()), "sad16")
        ctx.check(L.svt_hip_block_variance_batch_dev(ctx.h, 2, BD, d_cur.data_ptr(), W, d_ref.data_ptr(), W + 8, d_p.data_ptr(), len(pairs), d_o[1].data_ptr(), d_o[2].data_ptr()), "var10")
    out["stages_ms"]["hbd_sad_variance_64x64_32x32"] = timed(torch, stream, sadvar, reps)
    # (3) the 64-point transform chain of the whole luma plane, five sizes, both high-bit-depth quantizers
    g = np.load(os.path.join(ROOT, "tests", "golden", "txfm_tables.npz"))
    qp = g["qp/10/60/0"]
    d_prd = T(ref); d_rec = torch.zeros((H, W), dtype=torch.int16, device=E.dev)
    keep = []
    for ts, (tw, th) in ((4, (64, 64)), (12, (64, 32)), (11, (32, 64)), (18, (64, 16)), (17, (16, 64))):
        descs = np.array([pkg.tx_desc(x, y, 0) for y in range(0, H - th + 1, th) for x in range(0, W - tw + 1, tw)], np.uint32)
        isc = T(g[f"iscan/{ts}/0"].astype(np.int16)); keep.append(isc)
        stt = pkg.ScanTables(); stt.iscan[0] = isc.data_ptr()
        d_desc = T(descs); keep.append(d_desc)
        nk = min(tw, 32) * min(th, 32)
        d_q = torch.zeros(len(descs) * nk, dtype=torch.int32, device=E.dev); d_eob = torch.zeros(len(descs), dtype=torch.int16, device=E.dev); keep += [d_q, d_eob]
        for vname, variant in (("highbd_quantize_b", 1), ("highbd_quantize_fp", 3)):
            qs = pkg.QuantParams()
            for name, row in (("zbin", qp[0]), ("round", qp[5] if variant == 3 else qp[1]), ("quant", qp[6] if variant == 3 else qp[2]), ("quant_shift", qp[3]), ("dequant", qp[4])):
                getattr(qs, name)[0] = int(row[0]); getattr(qs, name)[1] = int(row[1])
            qs.log_scale = tc.TX_SCALE[ts]; qs.variant = variant
            EJ = (pkg.EncTxJob * 1)()
            EJ[0].fwd = pkg.FwdTxJob(ts, len(descs), d_cur.data_ptr(), W, d_prd.data_ptr(), W, d_desc.data_ptr(), qs, stt, None, d_q.data_ptr(), None, d_eob.data_ptr(), None, None)
            EJ[0].d_recon = d_rec.data_ptr(); EJ[0].recon_stride = W
            ms = timed(torch, stream, lambda EJ=EJ: ctx.check(L.svt_hip_enc_txfm_multi_dev(ctx.h, 2, BD, EJ, 1), "enc txfm 10"), reps)
            out["stages_ms"][f"txfm_chain_{tw}x{th}_{vname}"] = ms
    # (4) the complete self-guided unit search of the three planes (16 sets) + apply with the sets it chose
    EXT, US = 3, [256, 256, 256]
    xs = [((p.shape[1] + 2 * EXT + 63) // 64) * 64 for p in planes_rec]
    b_cdef = [torch.zeros((p.shape[0] + 2 * EXT, xs[i]), dtype=torch.int16, device=E.dev) for i, p in enumerate(planes_rec)]
    b_rest = [torch.zeros_like(b) for b in b_cdef]
    d_dbl = [T(p) for p in planes_rec]; d_srcp = [T(p) for p in planes_cur]
    for i, p in enumerate(planes_rec):
        b_cdef[i][EXT:EXT + p.shape[0], EXT:EXT + p.shape[1]] = d_dbl[i]
    org = lambda b, i: b[i].data_ptr() + (EXT * xs[i] + EXT) * 2
    n_units = [max((p.shape[1] + US[i] // 2) // US[i], 1) * max((p.shape[0] + US[i] // 2) // US[i], 1) for i, p in enumerate(planes_rec)]
    L.svt_hip_sgr_search_units_scratch_bytes.restype = C.c_size_t
    scr = [L.svt_hip_sgr_search_units_scratch_bytes(p.shape[1], p.shape[0], US[i]) for i, p in enumerate(planes_rec)]
    d_scr = [torch.zeros(n, dtype=torch.uint8, device=E.dev) for n in scr]
    d_uxqd = [torch.zeros((n, 16, 2), dtype=torch.int32, device=E.dev) for n in n_units]; d_uerr = [torch.zeros((n, 16), dtype=torch.int64, device=E.dev) for n in n_units]
    d_ubest = [torch.zeros(n, dtype=torch.uint8, device=E.dev) for n in n_units]; d_ubx = [torch.zeros((n, 2), dtype=torch.int32, device=E.dev) for n in n_units]
    J = (pkg.SgrUnitsPlaneDev * 3)()
    for p in range(3):
        ph_, pw_ = planes_rec[p].shape
        J[p] = pkg.SgrUnitsPlaneDev(org(b_cdef, p), xs[p], d_srcp[p].data_ptr(), pw_, pw_, ph_, US[p], int(p > 0), 0xFFFF, d_uxqd[p].data_ptr(), d_uerr[p].data_ptr(), d_ubest[p].data_ptr(),
                                    d_ubx[p].data_ptr(), d_scr[p].data_ptr(), scr[p])

    def sgr_units():
        for p in range(3):
            ph_, pw_ = planes_rec[p].shape
            ctx.check(L.svt_hip_generate_padding_dev(ctx.h, org(b_cdef, p), 2, xs[p], pw_, ph_, EXT, EXT), "extend")
        ctx.check(L.svt_hip_sgr_search_units_picture_dev(ctx.h, 2, BD, 3, J), "sgr units 10")

    def sgr_apply():
        for p in range(3):
            ph_, pw_ = planes_rec[p].shape
            ctx.check(L.svt_hip_sgr_apply_plane_dev(ctx.h, 2, BD, org(b_cdef, p), xs[p], org(b_rest, p), xs[p], pw_, ph_, US[p], int(p > 0), d_dbl[p].data_ptr(), pw_, d_ubest[p].data_ptr(),
                                                    d_ubx[p].data_ptr()), "sgr apply 10")
    out["stages_ms"]["sgr_units_search"] = timed(torch, stream, sgr_units, reps)
    out["stages_ms"]["sgr_apply"] = timed(torch, stream, sgr_apply, reps)
    unfinished = int(sum(int(d_scr[p][:128].cpu().numpy().view(np.uint32)[2]) for p in range(3)))
    chain = ["hbd_sad_window_64x64", "hbd_sad_variance_64x64_32x32", "txfm_chain_64x64_highbd_quantize_b", "sgr_units_search", "sgr_apply"]
    total = sum(out["stages_ms"][k] for k in chain)
    out.update({"n_sb": n_sb, "chain": chain, "chain_ms": total, "sb_per_s": n_sb / (total * 1e-3), "sgr_walks_unfinished": unfinished,
                "note": "sb_per_s = 2040 / the sum of the chain's stage times (one frame, stages back to back on one stream; the other transform sizes / quantizer are listed, not summed)"})
    print(json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", required=True, choices=["config1", "10bit", "config2"])
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    config1(a.reps) if a.which == "config1" else (config2(a.reps) if a.which == "config2" else hbd(a.reps))
