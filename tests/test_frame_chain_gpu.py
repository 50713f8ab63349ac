"""GPU parity of the WHOLE bench step on a small frame: ME -> fwd txfm + quant -> inv txfm + recon ->
deblock -> CDEF search -> CDEF apply -> self-guided search -> self-guided apply (stripe-aware), every stage fed by the
previous stage's GPU output and compared with the oracle running the same chain (workload identical to bench.py's, tests/workload.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr
import me_common as mc
import txfm_common as tc
import workload

pytestmark = pytest.mark.gpu
P3, I3 = C.c_void_p * 3, C.c_int * 3


def test_chain_small_frame(hip, pkg, orc):
    W, H = 336, 208     # 6 x 4 SBs, ragged last column (16 px) / row (16 px)
    F = workload.Frame(W, H, seed=5)
    L = hip.L
    # ---------------- oracle chain
    sbs = mc.windows(orc, W, H, 64, 64)
    o_sad, o_mv = mc.oracle_frame(orc, F.cur_y_p, F.ref_y_p, F.cur_y_p.shape[1], F.pad, sbs, 0)
    o_recon = [p.copy() for p in F.ref]
    o_q = {}
    for (kind, ts), descs in sorted(F.descs.items()):
        nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
        scans = F.scans(ts)
        SC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in scans])
        for plane in ([0] if kind == 0 else [1, 2]):
            q = np.zeros((len(descs), nk), np.int32); eob = np.zeros(len(descs), np.uint16)
            orc.orc_txfm_chain_8bit(ptr(F.cur[plane]), F.cur[plane].shape[1], ptr(F.ref[plane]), F.ref[plane].shape[1], ptr(o_recon[plane]),
                                    o_recon[plane].shape[1], ptr(descs), 0, len(descs), ts, 0, ptr(F.qp[plane]), SC, tc.TX_SCALE[ts], ptr(q), ptr(eob))
            o_q[(plane, ts)] = (q, eob)
    o_dlf = [p.copy() for p in o_recon]
    for p in range(3):
        ev, eh = F.edges[p]
        orc.orc_deblock_plane(ptr(o_dlf[p]), 1, o_dlf[p].shape[1], 8, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 0)
    o_mse = np.zeros((2, F.n_sb, 64), np.uint64)
    orc.orc_cdef_search_frame(P3(*[p.ctypes.data for p in o_dlf]), I3(*[p.shape[1] for p in o_dlf]), P3(*[p.ctypes.data for p in F.cur]),
                              I3(*[p.shape[1] for p in F.cur]), 1, W, H, ptr(F.skip8), F.cdef_damping, 8, 0, ptr(o_mse), 0, F.n_sb)
    o_out = [p.copy() for p in o_dlf]
    orc.orc_cdef_apply_frame(P3(*[p.ctypes.data for p in o_dlf]), P3(*[p.ctypes.data for p in o_out]), I3(*[p.shape[1] for p in o_dlf]), 1, W, H,
                             ptr(F.skip8), ptr(F.cdef_y), ptr(F.cdef_uv), F.cdef_damping, 8)
    # ---------------- HIP chain
    g_sad, g_mv = mc.hip_frame(hip, F.cur_y_p, F.ref_y_p, F.cur_y_p.shape[1], F.pad, sbs, 0)
    assert np.array_equal(g_sad, o_sad) and np.array_equal(g_mv, o_mv)
    d_cur = [hip.to_device(p) for p in F.cur]; d_pred = [hip.to_device(p) for p in F.ref]
    d_rec = [hip.to_device(p) for p in F.ref]
    strides = [p.shape[1] for p in F.cur]
    for (kind, ts), descs in sorted(F.descs.items()):
        nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
        d_desc = hip.to_device(descs)
        st = pkg.ScanTables(); keep = []
        for c, s in enumerate(F.scan_tables(ts)):
            if s is not None:
                p = hip.to_device(s); keep.append(p); st.iscan[c] = p.value
        for plane in ([0] if kind == 0 else [1, 2]):
            qs = pkg.QuantParams(); qp = F.qp[plane]
            for name, row in (("zbin", qp[0]), ("round", qp[1]), ("quant", qp[2]), ("quant_shift", qp[3]), ("dequant", qp[4])):
                getattr(qs, name)[0] = int(row[0]); getattr(qs, name)[1] = int(row[1])
            qs.log_scale = tc.TX_SCALE[ts]; qs.variant = 0
            n = len(descs)
            d_q, d_dq, d_eob = hip.empty(n * nk * 4), hip.empty(n * nk * 4), hip.empty(n * 2)
            hip.check(L.svt_hip_fwd_txfm_quant_batch_dev(hip.h, ts, 1, d_cur[plane], strides[plane], d_pred[plane], strides[plane], d_desc, n,
                                                        C.byref(qs), C.byref(st), None, d_q, d_dq, d_eob, None, None))
            hip.check(L.svt_hip_inv_txfm_add_batch_dev(hip.h, ts, 1, 8, d_dq, d_pred[plane], strides[plane], d_rec[plane], strides[plane], d_desc, n))
            q = hip.to_host(d_q, (n, nk), np.int32); eob = hip.to_host(d_eob, (n,), np.uint16)
            assert np.array_equal(q, o_q[(plane, ts)][0]) and np.array_equal(eob, o_q[(plane, ts)][1]), (plane, ts)
            hip.free(d_q, d_dq, d_eob)
        hip.free(d_desc, *keep)
    for p in range(3):
        assert np.array_equal(hip.to_host(d_rec[p], F.ref[p].shape, np.uint8), o_recon[p]), ("recon", p)
        ev, eh = F.edges[p]
        d_ev, d_eh = hip.to_device(ev), hip.to_device(eh)
        hip.check(L.svt_hip_deblock_plane_dev(hip.h, d_rec[p], 1, strides[p], 8, d_ev, d_eh, ev.shape[1], ev.shape[0], 0))
        assert np.array_equal(hip.to_host(d_rec[p], F.ref[p].shape, np.uint8), o_dlf[p]), ("deblock", p)
        hip.free(d_ev, d_eh)
    d_skip = hip.to_device(F.skip8)
    d_mse = hip.to_device(np.zeros((2, F.n_sb, 64), np.uint64)); d_dir = hip.empty(F.n_sb * 64); d_var = hip.empty(F.n_sb * 256)
    hip.check(L.svt_hip_cdef_search_frame_dev(hip.h, 1, P3(*[p.value for p in d_rec]), I3(*strides), P3(*[p.value for p in d_cur]), I3(*strides),
                                             W, H, d_skip, F.cdef_damping, 8, d_mse, d_dir, d_var))
    assert np.array_equal(hip.to_host(d_mse, (2, F.n_sb, 64), np.uint64), o_mse)
    d_out = [hip.to_device(p) for p in o_dlf]
    d_cy, d_cuv = hip.to_device(F.cdef_y), hip.to_device(F.cdef_uv)
    hip.check(L.svt_hip_cdef_apply_frame_dev(hip.h, 1, P3(*[p.value for p in d_rec]), P3(*[p.value for p in d_out]), I3(*strides), W, H, d_skip,
                                            d_cy, d_cuv, F.cdef_damping, 8, d_dir, d_var))
    for p in range(3):
        assert np.array_equal(hip.to_host(d_out[p], F.ref[p].shape, np.uint8), o_out[p]), ("cdef apply", p)
    # ---------------- loop restoration on the CDEF output; stripe context rows come from the deblocked picture (d_rec)
    EXT, US = 3, 64
    rng = np.random.default_rng(77)
    for p in range(3):
        ss = int(p > 0)
        ph, pw = o_out[p].shape
        ext = np.ascontiguousarray(np.pad(o_out[p], EXT, mode="edge")); st = ext.shape[1]; off = EXT * st + EXT
        nu = max((pw + US // 2) // US, 1) * max((ph + US // 2) // US, 1)
        e_sums = np.zeros((nu, 16, 5), np.int64)
        orc.orc_sgr_search_plane(C.c_void_p(ext.ctypes.data + off), 1, st, ptr(F.cur[p]), F.cur[p].shape[1], pw, ph, ss, ss, US, 8, 0xFFFF, ptr(e_sums))
        u_ep = rng.integers(0, 16, nu).astype(np.uint8); u_ep[nu // 2] = 255
        u_xqd = np.stack([rng.integers(-96, 32, nu), rng.integers(-32, 96, nu)], 1).astype(np.int32)
        e_dst = np.zeros((ph, pw), np.uint8)
        work = ext.copy()
        orc.orc_sgr_apply_plane(ptr(o_dlf[p]), o_dlf[p].shape[1], C.c_void_p(work.ctypes.data + off), st, 1, pw, ph, ss, ss, US, 8, ptr(u_ep), ptr(u_xqd), ptr(e_dst), pw)
        d_ext, d_sums, d_dst, d_ep, d_xqd = hip.to_device(ext), hip.to_device(np.zeros_like(e_sums)), hip.to_device(np.zeros_like(e_dst)), hip.to_device(u_ep), hip.to_device(u_xqd)
        hip.check(L.svt_hip_sgr_search_plane_dev(hip.h, 1, 8, d_ext.value + off, st, d_cur[p], strides[p], pw, ph, US, ss, 0xFFFF, d_sums), "sgr search")
        assert np.array_equal(hip.to_host(d_sums, e_sums.shape, np.int64), e_sums), ("sgr search", p)
        hip.check(L.svt_hip_sgr_apply_plane_dev(hip.h, 1, 8, d_ext.value + off, st, d_dst, pw, pw, ph, US, ss, d_rec[p], strides[p], d_ep, d_xqd), "sgr apply")
        assert np.array_equal(hip.to_host(d_dst, e_dst.shape, np.uint8), e_dst), ("sgr apply", p)
        hip.free(d_ext, d_sums, d_dst, d_ep, d_xqd)
    hip.free(*d_cur, *d_pred, *d_rec, *d_out, d_skip, d_mse, d_dir, d_var, d_cy, d_cuv)
