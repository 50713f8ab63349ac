"""GPU parity at BASELINE.json's full size against the REFERENCE ITSELF: one whole 3840x2160 8-bit 4:2:0 frame (the bench workload, 2040 SBs)
through every stage of the step — HME, integer ME, sub-pel prediction, residual/transform/quantise/inverse/recon for all 19 block lists,
deblocking, CDEF search + apply, self-guided search + stripe-aware apply — on the HIP path (through the C ABI), compared bit for bit over the
ENTIRE frame with the reference's own kernels as its x86 build dispatches them (oracle/_ref SIMD flavour: SSE2..AVX2 / AVX-512, driven by
oracle/ref_bench.c on a pthread pool).  Every stage is fed by the previous stage's GPU output.  The library is built in the build container
by oracle/Makefile.ref and travels to the GPU box; nothing here reads /root/reference.  Skipped if the library is absent."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import ROOT, ptr
import me_common as mc
import txfm_common as tc
import workload

pytestmark = pytest.mark.gpu
P3, I3 = C.c_void_p * 3, C.c_int * 3
W, H = 3840, 2160


@pytest.fixture(scope="module")
def refb():
    path = os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libsvtav1_ref_simd.so not built")
    L = C.CDLL(path)
    L.refb_setup.restype = C.c_uint64; L.refb_setup.argtypes = [C.c_uint64]
    L.refb_parallel.restype = C.c_double; L.refb_parallel.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    assert L.refb_setup(0xFFFFFFFFFFFFFFFF) & (1 << 8), "the host has no AVX2?"
    return L


def par(refb, stage, slots, n, chunk):
    keep = []

    def adr(x):
        if x is None: return 0
        if isinstance(x, np.ndarray): keep.append(x); return x.ctypes.data
        if isinstance(x, int): return x
        keep.append(x); return C.addressof(x)
    a = (C.c_int64 * len(slots))(*[adr(v) for v in slots])
    refb.refb_parallel(stage, C.addressof(a), n, chunk, min(len(os.sched_getaffinity(0)), 128), 1)


@pytest.mark.parametrize("size_seed", [(3840, 2160, 11), (1920, 1080, 21), (1280, 720, 22), (704, 400, 23), (352, 288, 24)])
def test_whole_4k_frame_every_stage(hip, pkg, orc, refb, size_seed):
    W, H, seed = size_seed          # the 4K bench frame first, then other picture sizes / seeds (ragged last SB row / column) through the same chain
    F = workload.Frame(W, H, seed=seed)
    L = hip.L
    n_sb, st, org = F.n_sb, F.cur_y_p.shape[1], F.pad * F.cur_y_p.shape[1] + F.pad
    # ---------------------------------------------------------------- pyramids (GPU) + HME on them
    d_cur_p, d_ref_p = hip.to_device(F.cur_y_p), hip.to_device(F.ref_y_p)
    pyr = {}
    for name, d_p in (("cur", d_cur_p), ("ref", d_ref_p)):
        out = []
        for step, pad in ((4, workload.PADS), (2, workload.PADQ)):
            buf = np.zeros((H // step + 2 * pad, W // step + 2 * pad), np.uint8)
            d_b = hip.to_device(buf)
            hip.check(L.svt_hip_downsample_2d_dev(hip.h, d_p.value + org, st, W, H, d_b.value + pad * buf.shape[1] + pad, buf.shape[1], step, 1), "downsample")
            out.append((hip.to_host(d_b, buf.shape, np.uint8), d_b))
        pyr[name] = out + [((F.cur_y_p if name == "cur" else F.ref_y_p), d_p)]
    for lvl, S in enumerate(workload.hme_jobs(F)):
        (c_h, d_c), (r_h, d_r) = pyr["cur"][lvl], pyr["ref"][lvl]
        d_S = hip.to_device(np.frombuffer(bytes(S), np.uint8).copy()); d_sad, d_xy = hip.to_device(np.zeros(n_sb, np.uint32)), hip.to_device(np.zeros((n_sb, 2), np.int16))
        hip.check(L.svt_hip_sad_loop_batch_dev(hip.h, d_c, c_h.shape[1], d_r, r_h.shape[1], d_S, n_sb, d_sad, d_xy), "hme")
        e_sad, e_xy = np.zeros(n_sb, np.uint32), np.zeros((n_sb, 2), np.int16)
        par(refb, 1, [c_h, c_h.shape[1], r_h, r_h.shape[1], S, e_sad, e_xy], n_sb, 4)
        assert np.array_equal(hip.to_host(d_sad, (n_sb,), np.uint32), e_sad) and np.array_equal(hip.to_host(d_xy, (n_sb, 2), np.int16), e_xy), ("hme", lvl)
        hip.free(d_S, d_sad, d_xy)
    # ---------------------------------------------------------------- integer ME, 85 PUs per SB
    sbs = mc.windows(orc, W, H, 64, 64)
    g_sad, g_mv = mc.hip_frame(hip, F.cur_y_p, F.ref_y_p, st, F.pad, sbs, 0)
    e_sad, e_mv = np.zeros((n_sb, 85), np.uint32), np.zeros((n_sb, 85), np.uint32)
    par(refb, 0, [F.cur_y_p, F.ref_y_p, st, F.pad, F.pad, sbs, n_sb, 0, e_sad, e_mv], n_sb, 1)
    assert np.array_equal(g_sad, e_sad) and np.array_equal(g_mv, e_mv), "integer ME"
    # ---------------------------------------------------------------- sub-pel prediction of every 16x16
    CB, nb = workload.conv_jobs(F, 14)
    d_cb, d_sp = hip.to_device(np.frombuffer(bytes(CB), np.uint8)[:nb * C.sizeof(pkg.ConvBlk)].copy()), hip.to_device(np.zeros((H, W), np.uint8))
    hip.check(L.svt_hip_subpel_predict_batch_dev(hip.h, 1, 8, d_ref_p.value + org, st, d_sp, W, d_cb, nb), "subpel")
    e_sp = np.zeros((H, W), np.uint8)
    par(refb, 2, [F.ref_y_p.ctypes.data + org, st, e_sp, W, CB], nb, 64)
    assert np.array_equal(hip.to_host(d_sp, (H, W), np.uint8), e_sp), "sub-pel prediction"
    hip.free(d_cb, d_sp, d_cur_p, d_ref_p, *[d for name in pyr for (_, d) in pyr[name][:2]])
    # ---------------------------------------------------------------- residual -> transform -> quantise -> inverse -> recon, all lists
    d_cur = [hip.to_device(p) for p in F.cur]; d_pred = [hip.to_device(p) for p in F.ref]; d_rec = [hip.to_device(p) for p in F.ref]
    strides = [p.shape[1] for p in F.cur]
    e_rec = [p.copy() for p in F.ref]
    for (kind, ts), descs in sorted(F.descs.items()):
        nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
        d_desc = hip.to_device(descs)
        stt = pkg.ScanTables(); keep = []
        isc, sc = F.scan_tables(ts), F.scans(ts)
        for c, s in enumerate(isc):
            if s is not None:
                p = hip.to_device(s); keep.append(p); stt.iscan[c] = p.value
        for plane in ([0] if kind == 0 else [1, 2]):
            qs = pkg.QuantParams(); qp = F.qp[plane]
            for name, row in (("zbin", qp[0]), ("round", qp[1]), ("quant", qp[2]), ("quant_shift", qp[3]), ("dequant", qp[4])):
                getattr(qs, name)[0] = int(row[0]); getattr(qs, name)[1] = int(row[1])
            qs.log_scale = tc.TX_SCALE[ts]; qs.variant = 0
            n = len(descs)
            d_q, d_dq, d_eob = hip.empty(n * nk * 4), hip.empty(n * nk * 4), hip.empty(n * 2)
            hip.check(L.svt_hip_fwd_txfm_quant_batch_dev(hip.h, ts, 1, d_cur[plane], strides[plane], d_pred[plane], strides[plane], d_desc, n, C.byref(qs), C.byref(stt), None,
                                                        d_q, d_dq, d_eob, None, None))
            hip.check(L.svt_hip_inv_txfm_add_batch_dev(hip.h, ts, 1, 8, d_dq, d_pred[plane], strides[plane], d_rec[plane], strides[plane], d_desc, n))
            # reference: one C call per chunk of the list; each writes the qcoeff / eob / recon of its own blocks
            e_q, e_eob = np.zeros((n, nk), np.int32), np.zeros(n, np.uint16)
            SC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in sc]); ISC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in isc])
            nt = 32
            with ThreadPoolExecutor(nt) as ex:
                list(ex.map(lambda be: refb.refb_txfm_chain_8bit(ptr(F.cur[plane]), strides[plane], ptr(F.ref[plane]), strides[plane], ptr(e_rec[plane]), strides[plane],
                                                                 ptr(descs), be[0], be[1], ts, ptr(qp), SC, ISC, tc.TX_SCALE[ts], ptr(e_q), ptr(e_eob)),
                            [(i * n // nt, (i + 1) * n // nt) for i in range(nt)]))
            assert np.array_equal(hip.to_host(d_q, (n, nk), np.int32), e_q) and np.array_equal(hip.to_host(d_eob, (n,), np.uint16), e_eob), ("quantised coefficients", plane, ts)
            hip.free(d_q, d_dq, d_eob)
        hip.free(d_desc, *keep)
    g_rec = [hip.to_host(d_rec[p], F.ref[p].shape, np.uint8) for p in range(3)]
    for p in range(3):
        assert np.array_equal(g_rec[p], e_rec[p]), ("reconstruction", p)
    # ---------------------------------------------------------------- deblocking (whole planes, normative order)
    e_dlf = [p.copy() for p in g_rec]
    for p in range(3):
        ev, eh = F.edges[p]
        d_ev, d_eh = hip.to_device(ev), hip.to_device(eh)
        hip.check(L.svt_hip_deblock_plane_dev(hip.h, d_rec[p], 1, strides[p], 8, d_ev, d_eh, ev.shape[1], ev.shape[0], 0))
        refb.refb_deblock_plane(ptr(e_dlf[p]), strides[p], ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 0)
        hip.free(d_ev, d_eh)
    g_dlf = [hip.to_host(d_rec[p], F.ref[p].shape, np.uint8) for p in range(3)]
    for p in range(3):
        assert np.array_equal(g_dlf[p], e_dlf[p]) and (g_dlf[p] != g_rec[p]).any(), ("deblocking", p)
    # ---------------------------------------------------------------- CDEF: 64-strength search table, then apply
    d_skip = hip.to_device(F.skip8)
    d_mse = hip.to_device(np.zeros((2, n_sb, 64), np.uint64)); d_dir = hip.empty(n_sb * 64); d_var = hip.empty(n_sb * 256)
    hip.check(L.svt_hip_cdef_search_frame_dev(hip.h, 1, P3(*[p.value for p in d_rec]), I3(*strides), P3(*[p.value for p in d_cur]), I3(*strides), W, H, d_skip,
                                             F.cdef_damping, 8, d_mse, d_dir, d_var))
    e_mse = np.zeros((2, n_sb, 64), np.uint64)
    par(refb, 5, [g_dlf[0], g_dlf[1], g_dlf[2]] + strides + [F.cur[0], F.cur[1], F.cur[2]] + strides + [W, H, F.skip8, F.cdef_damping, e_mse], n_sb, 1)
    assert np.array_equal(hip.to_host(d_mse, (2, n_sb, 64), np.uint64), e_mse), "CDEF search table"
    d_out = [hip.to_device(p) for p in g_dlf]
    d_cy, d_cuv = hip.to_device(F.cdef_y), hip.to_device(F.cdef_uv)
    hip.check(L.svt_hip_cdef_apply_frame_dev(hip.h, 1, P3(*[p.value for p in d_rec]), P3(*[p.value for p in d_out]), I3(*strides), W, H, d_skip, d_cy, d_cuv,
                                            F.cdef_damping, 8, d_dir, d_var))
    e_out = [p.copy() for p in g_dlf]
    par(refb, 6, [g_dlf[0], g_dlf[1], g_dlf[2]] + strides + e_out + [W, H, F.skip8, F.cdef_y, F.cdef_uv, F.cdef_damping], n_sb, 2)
    g_out = [hip.to_host(d_out[p], F.ref[p].shape, np.uint8) for p in range(3)]
    for p in range(3):
        assert np.array_equal(g_out[p], e_out[p]) and (g_out[p] != g_dlf[p]).any(), ("CDEF apply", p)
    # ---------------------------------------------------------------- loop restoration: projection search (all 16 sets) and stripe-aware apply
    EXT, US = 3, 256
    rng = np.random.default_rng(77)
    orc.orc_sgr_solve.restype = None
    for p in range(3):
        ss = int(p > 0)
        ph, pw = g_out[p].shape
        ext = np.ascontiguousarray(np.pad(g_out[p], EXT, mode="edge")); est = ext.shape[1]; off = EXT * est + EXT
        nu = max((pw + US // 2) // US, 1) * max((ph + US // 2) // US, 1)
        lim = np.zeros((nu, 4), np.int32)
        orc.orc_rest_unit_limits(pw, ph, ss, US, ptr(lim))
        d_ext, d_sums = hip.to_device(ext), hip.to_device(np.zeros((nu, 16, 5), np.int64))
        hip.check(L.svt_hip_sgr_search_plane_dev(hip.h, 1, 8, d_ext.value + off, est, d_cur[p], strides[p], pw, ph, US, ss, 0xFFFF, d_sums), "sgr search")
        sums = hip.to_host(d_sums, (nu, 16, 5), np.int64)
        g_xq = np.zeros((nu, 16, 2), np.int32)
        for u in range(nu):        # the host-side solve of svt_get_proj_subspace (EbRestorationPick.c:497-538) on the GPU's integer sums
            size = int((lim[u, 1] - lim[u, 0]) * (lim[u, 3] - lim[u, 2]))
            for ep in range(16):
                orc.orc_sgr_solve(ptr(np.ascontiguousarray(sums[u, ep])), size, ep, C.c_void_p(g_xq.ctypes.data + (u * 16 + ep) * 8))
        e_xq = np.zeros((nu, 16, 2), np.int32)
        par(refb, 7, [ext.ctypes.data + off, est, F.cur[p], strides[p], lim, 64 >> ss, 64 >> ss, 0xFFFF, e_xq], nu, 1)
        assert np.array_equal(g_xq, e_xq) and e_xq.any(), ("self-guided projection coefficients", p)
        u_ep = rng.integers(0, 16, nu).astype(np.uint8); u_ep[nu // 2] = 255
        u_xqd = np.stack([rng.integers(-96, 32, nu), rng.integers(-32, 96, nu)], 1).astype(np.int32)
        d_dst, d_ep, d_xqd = hip.to_device(np.zeros((ph, pw), np.uint8)), hip.to_device(u_ep), hip.to_device(u_xqd)
        hip.check(L.svt_hip_sgr_apply_plane_dev(hip.h, 1, 8, d_ext.value + off, est, d_dst, pw, pw, ph, US, ss, d_rec[p], strides[p], d_ep, d_xqd), "sgr apply")
        e_dst = np.zeros((ph, pw), np.uint8)
        work = ext.copy(); dbl = g_dlf[p].copy()
        assert refb.ref_shim_lr_apply_plane(p, 8, 0, W, H, ptr(dbl), strides[p], C.c_void_p(work.ctypes.data + off), est, ptr(e_dst), pw, US, ptr(u_ep), ptr(u_xqd)) == 0
        assert np.array_equal(hip.to_host(d_dst, (ph, pw), np.uint8), e_dst) and (nu == 1 or (e_dst != g_out[p]).any()), ("loop restoration apply", p)   # nu == 1: the only unit is the RESTORE_NONE one
        hip.free(d_ext, d_sums, d_dst, d_ep, d_xqd)
    hip.free(*d_cur, *d_pred, *d_rec, *d_out, d_skip, d_mse, d_dir, d_var, d_cy, d_cuv)


def test_whole_4k_frame_10bit(hip, pkg, orc, refb):
    """BASELINE.json configs[3] at full size against the reference itself: every 64x64 / 32x32 block pair of a 3840x2160 10-bit plane (sad_16b,
    highbd_10 variance + sse), the 64-point transform chain of every 64x64 block (quantised / dequantised coefficients, EOB, reconstruction) and the
    self-guided search (projection coefficients of all 2040 units x 16 sets) + stripe-aware apply of the whole plane — HIP vs the reference's SIMD kernels."""
    from test_config4_hbd_gpu import frame10
    BD = 10
    cur, ref = frame10(5)
    nt = min(len(os.sched_getaffinity(0)), 64)

    def threads(fn, n):
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(fn, [(i * n // nt, (i + 1) * n // nt) for i in range(nt)]))

    # ---------------------------------------------------------------- block SAD / variance
    rng = np.random.default_rng(6)
    pairs = []
    for by in range(0, H - 63, 64):
        for bx in range(0, W - 63, 64):
            ox, oy = int(rng.integers(0, 3)), int(rng.integers(0, 3))
            rx, ry = min(bx + ox, W - 64), min(by + oy, H - 64)
            pairs.append((bx, by, rx, ry, 64, 64))
            for q in range(4):
                pairs.append((bx + 32 * (q & 1), by + 32 * (q >> 1), rx + 32 * (q & 1), ry + 32 * (q >> 1), 32, 32))
    n = len(pairs)
    P = (pkg.BlkPair * n)(*[pkg.BlkPair(*p) for p in pairs])
    d_a, d_b, d_p = hip.to_device(cur), hip.to_device(ref), hip.to_device(np.frombuffer(bytes(P), np.uint8))
    d_sad, d_var, d_sse = hip.empty(n * 4), hip.empty(n * 4), hip.empty(n * 4)
    hip.check(hip.L.svt_hip_block_sad_batch_dev(hip.h, 2, d_a, W, d_b, W, d_p, n, d_sad), "sad16")
    hip.check(hip.L.svt_hip_block_variance_batch_dev(hip.h, 2, BD, d_a, W, d_b, W, d_p, n, d_var, d_sse), "var10")
    e_sad, e_var, e_sse = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    threads(lambda be: refb.refb_hbd_sad_var_batch(ptr(cur), W, ptr(ref), W, P, be[0], be[1], ptr(e_sad), ptr(e_var), ptr(e_sse)), n)
    assert np.array_equal(hip.to_host(d_sad, (n,), np.uint32), e_sad) and np.array_equal(hip.to_host(d_var, (n,), np.uint32), e_var), "HBD SAD / variance"
    assert np.array_equal(hip.to_host(d_sse, (n,), np.uint32), e_sse), "HBD sse"
    hip.free(d_p, d_sad, d_var, d_sse)
    # ---------------------------------------------------------------- 64-point transform chain on every 64x64 block
    ts = 4
    descs = np.array([pkg.tx_desc(x, y, 0) for y in range(0, H - 63, 64) for x in range(0, W - 63, 64)], np.uint32)
    nb = len(descs); NK = 1024
    g = np.load(os.path.join(ROOT, "tests", "golden", "txfm_tables.npz"))
    qp = np.ascontiguousarray(g["qp/10/60/0"]); scan, iscan = np.ascontiguousarray(g[f"scan/{ts}/0"]), np.ascontiguousarray(g[f"iscan/{ts}/0"])
    qs = pkg.QuantParams()
    for name, row in (("zbin", qp[0]), ("round", qp[1]), ("quant", qp[2]), ("quant_shift", qp[3]), ("dequant", qp[4])):
        getattr(qs, name)[0] = int(row[0]); getattr(qs, name)[1] = int(row[1])
    qs.log_scale = tc.TX_SCALE[ts]; qs.variant = 1
    d_isc = hip.to_device(iscan.astype(np.int16))
    stt = pkg.ScanTables(); stt.iscan[0] = d_isc.value
    d_desc = hip.to_device(descs)
    d_q, d_dq, d_eob, d_rec = hip.empty(nb * NK * 4), hip.empty(nb * NK * 4), hip.empty(nb * 2), hip.to_device(np.zeros_like(cur))
    hip.check(hip.L.svt_hip_fwd_txfm_quant_batch_dev(hip.h, ts, 2, d_a, W, d_b, W, d_desc, nb, C.byref(qs), C.byref(stt), None, d_q, d_dq, d_eob, None, None), "fwd64")
    hip.check(hip.L.svt_hip_inv_txfm_add_batch_dev(hip.h, ts, 2, BD, d_dq, d_b, W, d_rec, W, d_desc, nb), "inv64")
    e_q, e_dq, e_eob, e_rec = np.zeros((nb, NK), np.int32), np.zeros((nb, NK), np.int32), np.zeros(nb, np.uint16), np.zeros_like(cur)
    SC = (C.c_void_p * 3)(scan.ctypes.data, None, None); ISC = (C.c_void_p * 3)(iscan.ctypes.data, None, None)
    threads(lambda be: refb.refb_txfm_chain_hbd(ptr(cur), W, ptr(ref), W, ptr(e_rec), W, ptr(descs), be[0], be[1], ts, BD, ptr(qp), SC, ISC, tc.TX_SCALE[ts],
                                                ptr(e_q), ptr(e_dq), ptr(e_eob)), nb)
    assert np.array_equal(hip.to_host(d_q, (nb, NK), np.int32), e_q) and np.array_equal(hip.to_host(d_dq, (nb, NK), np.int32), e_dq), "64-point quantised coefficients"
    assert np.array_equal(hip.to_host(d_eob, (nb,), np.uint16), e_eob) and e_eob.any()
    assert np.array_equal(hip.to_host(d_rec, cur.shape, np.uint16), e_rec), "64-point reconstruction"
    hip.free(d_desc, d_q, d_dq, d_eob, d_rec, d_isc, d_b)
    # ---------------------------------------------------------------- self-guided restoration, whole luma plane, unit 64
    EXT, US = 3, 64
    dgd = ref
    ext = np.ascontiguousarray(np.pad(dgd, EXT, mode="edge")); st = ext.shape[1]; off = (EXT * st + EXT) * 2
    nu = max((W + 32) // 64, 1) * max((H + 32) // 64, 1)
    lim = np.zeros((nu, 4), np.int32); orc.orc_rest_unit_limits(W, H, 0, US, ptr(lim))
    d_ext, d_sums = hip.to_device(ext), hip.to_device(np.zeros((nu, 16, 5), np.int64))
    hip.check(hip.L.svt_hip_sgr_search_plane_dev(hip.h, 2, BD, d_ext.value + off, st, d_a, W, W, H, US, 0, 0xFFFF, d_sums), "search10")
    sums = hip.to_host(d_sums, (nu, 16, 5), np.int64)
    g_xq = np.zeros((nu, 16, 2), np.int32)
    for u in range(nu):
        size = int((lim[u, 1] - lim[u, 0]) * (lim[u, 3] - lim[u, 2]))
        for ep in range(16):
            orc.orc_sgr_solve(C.c_void_p(sums.ctypes.data + (u * 16 + ep) * 40), size, ep, C.c_void_p(g_xq.ctypes.data + (u * 16 + ep) * 8))
    e_xq = np.zeros((nu, 16, 2), np.int32)
    threads(lambda be: refb.refb_sgr_search_plane_hbd(C.c_void_p(ext.ctypes.data + off), st, ptr(cur), W, ptr(lim), be[0], be[1], 64, 64, 0xFFFF, BD, ptr(e_xq)), nu)
    assert np.array_equal(g_xq, e_xq) and e_xq.any(), "10-bit self-guided projection coefficients"
    rng = np.random.default_rng(10)
    u_ep = rng.integers(0, 16, nu).astype(np.uint8); u_ep[::11] = 255
    u_xqd = np.stack([rng.integers(-96, 32, nu), rng.integers(-32, 96, nu)], 1).astype(np.int32)
    d_ep, d_xqd, d_dst = hip.to_device(u_ep), hip.to_device(u_xqd), hip.to_device(np.zeros_like(cur))
    hip.check(hip.L.svt_hip_sgr_apply_plane_dev(hip.h, 2, BD, d_ext.value + off, st, d_dst, W, W, H, US, 0, d_a, W, d_ep, d_xqd), "apply10")   # stripes see `cur` as the deblocked plane
    e_dst = np.zeros_like(cur)
    work = ext.copy(); dbl = cur.copy()
    assert refb.ref_shim_lr_apply_plane(0, BD, 1, W, H, ptr(dbl), W, C.c_void_p(work.ctypes.data + off), st, ptr(e_dst), W, US, ptr(u_ep), ptr(u_xqd)) == 0
    assert np.array_equal(hip.to_host(d_dst, cur.shape, np.uint16), e_dst) and (e_dst != dgd).any(), "10-bit loop restoration apply"
    hip.free(d_a, d_ext, d_sums, d_ep, d_xqd, d_dst)
