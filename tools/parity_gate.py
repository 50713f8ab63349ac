"""The parity gate of bench.py ("every output array compared byte-for-byte before any timing is accepted").

bench.py replays frame 0's chain once more, stage by stage, and hands the snapshot of every stage's output here; each stage is then recomputed on the host FROM THE
DEVICE'S OWN INPUT of that stage by the reference's kernels (oracle/_ref/libsvtav1_ref_simd.so through oracle/ref_bench.c: the functions tests/test_ref_bench.py pins to
the oracle, and tests/test_full4k_vs_reference_gpu.py uses for the whole-frame comparison) — the pyramids by the oracle's C (the reference library carries no entry for
them) — and compared with the device's output of the same stage.  One boolean per stage; bench.py's `parity_spot_check` is their AND.

Checker code: only bench.py (after its timed region) and tests/ import this.  `reference_chain()` builds the same snapshot from the reference alone, which is how
tests/test_parity_gate.py checks the gate itself on a box without a GPU (every stage true; one flipped byte in a stage's output turns exactly that stage false)."""
import ctypes as C
import os

import numpy as np

STAGES = ["pyramids", "hme_l0_l1_l2", "me_fullpel_85pu", "subpel_convolve", "fwd_quant_inv_recon", "deblock", "cdef_search", "cdef_strength_select", "cdef_apply",
          "sgr_units_search", "sgr_apply"]
EXT = 3
P3, I3 = C.c_void_p * 3, C.c_int * 3


def _adr(x, keep):
    if x is None: return 0
    if isinstance(x, np.ndarray):
        keep.append(x)
        return x.ctypes.data
    if isinstance(x, int): return x
    keep.append(x)
    return C.addressof(x)


def _par(refb, stage, slots, n, chunk, threads):
    keep = []
    a = (C.c_int64 * len(slots))(*[_adr(v, keep) for v in slots])
    refb.refb_parallel(stage, C.addressof(a), n, chunk, threads, 1)


def setup_refb(refb):
    refb.refb_setup.restype = C.c_uint64; refb.refb_setup.argtypes = [C.c_uint64]
    refb.refb_parallel.restype = C.c_double; refb.refb_parallel.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    refb.refb_setup(0xFFFFFFFFFFFFFFFF)
    return refb


def conv_jobs_from_me(pkg, mv, sb_cols, w, h, frac):
    """numpy restatement of svt_hip_subpel_jobs_from_me_dev (include/svt_hip.h): the 16x16 PU of a block is entry 5 + z-order index of its superblock's table"""
    bw, bh = w >> 4, h >> 4
    k = np.arange(bw * bh)
    bx, by = k % bw, k // bw
    sb = (by >> 2) * sb_cols + (bx >> 2)
    qx, qy = bx & 3, by & 3
    z = ((qy >> 1) * 2 + (qx >> 1)) * 4 + (qy & 1) * 2 + (qx & 1)
    word = mv.reshape(-1, 85)[sb, 5 + z].astype(np.uint32)
    mx = (word & 0xffff).astype(np.uint16).view(np.int16).astype(np.int32) >> 2
    my = (word >> 16).astype(np.uint16).view(np.int16).astype(np.int32) >> 2
    CB = (pkg.ConvBlk * len(k))()
    fr = np.asarray(frac).reshape(-1, 2)
    for i in range(len(k)):
        CB[i] = pkg.ConvBlk(int(bx[i] * 16 + mx[i]), int(by[i] * 16 + my[i]), int(bx[i] * 16), int(by[i] * 16), 16, 16, 0, 0, int(fr[i, 0] & 15), int(fr[i, 1] & 15), 0, 0)
    return CB


def tx_job_list(F):
    """(tx_size, plane, descs) in the order bench.py's Pipeline builds its job lists"""
    out = []
    for (kind, ts), descs in sorted(F.descs.items()):
        for plane in ([0] if kind == 0 else [1, 2]):
            out.append((ts, plane, descs))
    return out


def units(pw, ph, us):
    return max((pw + us // 2) // us, 1) * max((ph + us // 2) // us, 1)


def expected(stage, F, S, refb, orc, pkg, tc, workload, threads, cdef_lambda, unit_size=256):
    """the reference's output of `stage` for the inputs the snapshot S holds (S[...] entries of the earlier stages); -> dict of the stage's output arrays"""
    from conftest import ptr
    W, H, n_sb = F.w, F.h, F.n_sb
    st = F.cur_y_p.shape[1]
    org = F.pad * st + F.pad
    strides = [p.shape[1] for p in F.cur]
    if stage == "pyramids":
        out = {}
        for name, src_p in (("cur", F.cur_y_p), ("ref", F.ref_y_p)):
            for step, pad in ((2, workload.PADQ), (4, workload.PADS)):
                b = np.zeros((H // step + 2 * pad, W // step + 2 * pad), np.uint8)
                orc.orc_downsample_2d(C.c_void_p(src_p.ctypes.data + org), st, W, H, C.c_void_p(b.ctypes.data + pad * b.shape[1] + pad), b.shape[1], step, 1)
                out[f"{name}_{step}"] = b
        vp = np.zeros((F.sb_rows * 64 + 64, F.sb_cols * 64 + 64), np.uint8); vp[:H, :W] = F.cur[0]   # what the stage reads: the picture in a zero frame of whole superblocks
        mean, var = np.zeros((n_sb, 85), np.uint8), np.zeros((n_sb, 85), np.uint16)
        for i in range(n_sb):
            sx, sy = (i % F.sb_cols) * 64, (i // F.sb_cols) * 64
            orc.orc_variance_pyramid_sb(C.c_void_p(vp.ctypes.data + sy * vp.shape[1] + sx), vp.shape[1], 0, C.c_void_p(mean.ctypes.data + 85 * i), C.c_void_p(var.ctypes.data + 170 * i))
        out["ymean"], out["yvar"] = mean, var
        return out
    if stage == "hme_l0_l1_l2":
        out = {}
        planes = {"cur": (S["cur_4"], S["cur_2"], F.cur_y_p), "ref": (S["ref_4"], S["ref_2"], F.ref_y_p)}
        for lvl, J in enumerate(workload.hme_jobs(F)):
            c, r = planes["cur"][lvl], planes["ref"][lvl]
            sad, xy = np.zeros(n_sb, np.uint32), np.zeros((n_sb, 2), np.int16)
            _par(refb, 1, [c, c.shape[1], r, r.shape[1], J, sad, xy], n_sb, 8, threads)
            out[f"hme_sad_{lvl}"], out[f"hme_xy_{lvl}"] = sad, xy
        return out
    if stage == "me_fullpel_85pu":
        sad, mv = np.zeros((n_sb, 85), np.uint32), np.zeros((n_sb, 85), np.uint32)
        _par(refb, 0, [F.cur_y_p, F.ref_y_p, st, F.pad, F.pad, S["sbs"], n_sb, 0, sad, mv], n_sb, 8, threads)
        return {"sad": sad, "mv": mv}
    if stage == "subpel_convolve":
        CB = conv_jobs_from_me(pkg, S["mv"], F.sb_cols, W, H, S["frac"])
        n = (W >> 4) * (H >> 4)
        dst = np.zeros((H, W), np.uint8)
        _par(refb, 2, [F.ref_y_p.ctypes.data + org, st, dst, W, CB], n, 64, threads)
        return {"conv_jobs": np.frombuffer(bytes(CB), np.uint8).copy(), "subpel": dst}
    if stage == "fwd_quant_inv_recon":
        pred = [S["subpel"], F.ref[1], F.ref[2]]
        rec = [np.zeros_like(p) for p in F.ref]   # samples no transform block covers are never written: the device buffers start as zeros
        out = {}
        for k, (ts, plane, descs) in enumerate(tx_job_list(F)):
            nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
            n = len(descs)
            sc, isc = F.scans(ts), F.scan_tables(ts)
            SC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in sc]); ISC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in isc])
            q, eob = np.zeros((n, nk), np.int32), np.zeros(n, np.uint16)
            from concurrent.futures import ThreadPoolExecutor
            nt = max(1, min(threads, 32, n))
            with ThreadPoolExecutor(nt) as ex:
                list(ex.map(lambda be: refb.refb_txfm_chain_8bit(ptr(F.cur[plane]), strides[plane], ptr(pred[plane]), strides[plane], ptr(rec[plane]), strides[plane], ptr(descs),
                                                                 be[0], be[1], ts, ptr(F.qp[plane]), SC, ISC, tc.TX_SCALE[ts], ptr(q), ptr(eob)),
                            [(i * n // nt, (i + 1) * n // nt) for i in range(nt)]))
            out[f"q_{k}"], out[f"eob_{k}"] = q, eob
        for p in range(3): out[f"recon_{p}"] = rec[p]
        return out
    if stage == "deblock":
        out = {}
        import dlf_common   # tests/: the edge planes of the recomputation come from the ORACLE's statement of set_lpf_parameters (pinned to the reference's frame loop),
        for p in range(3):  # not from the product's builder, which made the device's own (F.edges) -- a wrong product builder fails this gate
            ev, eh = dlf_common.build_edges(F.mi, F.mi_cols, F.mi_rows, p, W >> (p > 0), H >> (p > 0))
            img = S[f"recon_{p}"].copy()
            refb.refb_deblock_plane(ptr(img), strides[p], ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 0)
            out[f"dbl_{p}"] = img
        return out
    if stage == "cdef_search":
        mse = np.zeros((2, n_sb, 64), np.uint64)
        d = [S[f"dbl_{p}"] for p in range(3)]
        _par(refb, 5, d + strides + [F.cur[0], F.cur[1], F.cur[2]] + strides + [W, H, F.skip8, F.cdef_damping, mse], n_sb, 4, threads)
        return {"mse": mse}
    if stage == "cdef_strength_select":
        m = np.ascontiguousarray(S["mse"]).reshape(2, n_sb, 64)
        fin, sel = np.zeros(17, np.int32), np.zeros(n_sb, np.int32)
        refb.refb_cdef_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        refb.refb_cdef_finish(m[0].ctypes.data, m[1].ctypes.data, n_sb, cdef_lambda, fin.ctypes.data, sel.ctypes.data)
        return {"cdef_fin": fin, "cdef_sel": sel, "cdef_y": fin[1:9][sel].astype(np.uint8), "cdef_uv": fin[9:17][sel].astype(np.uint8)}
    if stage == "cdef_apply":
        d = [S[f"dbl_{p}"] for p in range(3)]
        outs = [p.copy() for p in d]
        _par(refb, 6, d + strides + outs + [W, H, F.skip8, np.ascontiguousarray(S["cdef_y"]), np.ascontiguousarray(S["cdef_uv"]), F.cdef_damping], n_sb, 4, threads)
        return {f"cdef_{p}": outs[p] for p in range(3)}
    if stage == "sgr_units_search":
        out = {}
        for p in range(3):
            ss = int(p > 0)
            ph, pw = F.cur[p].shape
            ext = np.ascontiguousarray(np.pad(S[f"cdef_{p}"], EXT, mode="edge")); est = ext.shape[1]; eoff = EXT * est + EXT
            nu = units(pw, ph, unit_size)
            lim = np.zeros((nu, 4), np.int32)
            orc.orc_rest_unit_limits(pw, ph, ss, unit_size, ptr(lim))
            res = np.zeros((nu, 3), np.int32)
            _par(refb, 9, [ext.ctypes.data + eoff, est, F.cur[p], strides[p], lim, 64 >> ss, 64 >> ss, res], nu, 1, threads)
            out[f"unit_ep_{p}"], out[f"unit_xqd_{p}"] = res[:, 0].astype(np.uint8), np.ascontiguousarray(res[:, 1:3])
        return out
    if stage == "sgr_apply":
        out = {}
        for p in range(3):
            ph, pw = F.cur[p].shape
            work = np.ascontiguousarray(np.pad(S[f"cdef_{p}"], EXT, mode="edge")); est = work.shape[1]; eoff = EXT * est + EXT
            dbl = S[f"dbl_{p}"].copy()
            dst = np.zeros((ph, pw), np.uint8)
            rc = refb.ref_shim_lr_apply_plane(p, 8, 0, W, H, ptr(dbl), strides[p], C.c_void_p(work.ctypes.data + eoff), est, ptr(dst), pw, unit_size,
                                              ptr(np.ascontiguousarray(S[f"unit_ep_{p}"])), ptr(np.ascontiguousarray(S[f"unit_xqd_{p}"]).astype(np.int32)))
            assert rc == 0
            out[f"rest_{p}"] = dst
        return out
    raise KeyError(stage)


def check_chain(F, S, refb, orc, pkg, tc, workload, threads, cdef_lambda, stages=STAGES, unit_size=256):
    """-> ({stage: bool}, {stage: [names of the arrays that differ]}): every stage's device output against the reference's output for the device's input"""
    ok, bad = {}, {}
    for stage in stages:
        exp = expected(stage, F, S, refb, orc, pkg, tc, workload, threads, cdef_lambda, unit_size)
        diff = []
        for k, v in exp.items():
            g = S.get(k)
            if g is None or g.shape != v.shape:
                diff.append(f"{k}: shape {None if g is None else g.shape} != {v.shape}")
                continue
            g = np.asarray(g).view(v.dtype) if g.dtype.itemsize == v.dtype.itemsize else np.asarray(g)
            if not np.array_equal(g, v):
                w = np.argwhere(g != v)
                diff.append(f"{k}: {len(w)} of {v.size} differ, first at {w[0].tolist()}: device {g[tuple(w[0])]} reference {v[tuple(w[0])]}")
        ok[stage] = not diff
        if diff: bad[stage] = diff[:24]
    return ok, bad


def reference_chain(F, refb, orc, pkg, tc, workload, threads, cdef_lambda, sbs, frac, unit_size=256):
    """the snapshot the reference alone produces (each stage fed by the previous one's reference output): what a correct device leaves behind"""
    S = {"sbs": sbs, "frac": frac}
    for stage in STAGES:
        S.update(expected(stage, F, S, refb, orc, pkg, tc, workload, threads, cdef_lambda, unit_size))
    return S
