"""BASELINE.json configs[3]: 4K 10-bit 4:2:0 (high-bit-depth path) — HBD block SAD / variance, 64-point transform + quant + inverse
at bit depth 10, self-guided restoration search + apply on the full 3840x2160 10-bit luma plane.
At full size the oracle checks a seeded sample; size-independent properties cover the rest: the SAD / SSE quad-tree sums up
(64x64 = sum of its four 32x32), a unit's search sums do not depend on what the frame holds elsewhere, RESTORE_NONE units are a copy."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ptr
import txfm_common as tc

pytestmark = pytest.mark.gpu
W, H, BD = 3840, 2160, 10


def frame10(seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    base = 480 + 300 * np.sin(xx / 97.0) * np.cos(yy / 61.0) + 60 * (((xx // 16).astype(np.int32) + (yy // 16).astype(np.int32)) % 2)
    cur = np.clip(base + rng.normal(0, 9, (H, W)), 0, 1023).astype(np.uint16)
    ref = np.clip(np.roll(base, (3, -5), (0, 1)) + rng.normal(0, 9, (H, W)), 0, 1023).astype(np.uint16)
    return cur, ref


def test_hbd_sad_variance_4k(hip, pkg, orc):
    cur, ref = frame10(5)
    rng = np.random.default_rng(6)
    pairs = []
    for by in range(0, H - 63, 64):          # every full 64x64 SB and its four 32x32 quadrants, co-located + small offset
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
    sad = hip.to_host(d_sad, (n,), np.uint32).astype(np.int64); var = hip.to_host(d_var, (n,), np.uint32); sse = hip.to_host(d_sse, (n,), np.uint32).astype(np.int64)
    hip.free(d_a, d_b, d_p, d_sad, d_var, d_sse)
    # property over the whole frame: the quad-tree adds up
    g = sad.reshape(-1, 5); assert np.array_equal(g[:, 0], g[:, 1:].sum(1))
    # (highbd_10 sse is rounded to 8-bit scale per block, ROUND_POWER_OF_TWO(sse, 4): only additive up to the rounding)
    g = sse.reshape(-1, 5); assert np.abs(g[:, 0] - g[:, 1:].sum(1)).max() <= 2
    # seeded sample against the oracle (sad_16b_kernel_c / svt_aom_highbd_10_variance*_c restatements)
    orc.orc_sad_16b.restype = C.c_uint32; orc.orc_variance_hbd10.restype = C.c_uint32
    for i in rng.choice(n, 200, replace=False):
        ax, ay, bx, by, w, h = pairs[i]
        pa = C.c_void_p(cur.ctypes.data + (ay * W + ax) * 2); pb = C.c_void_p(ref.ctypes.data + (by * W + bx) * 2)
        assert sad[i] == orc.orc_sad_16b(pa, W, pb, W, h, w), pairs[i]
        s = C.c_uint32()
        assert var[i] == orc.orc_variance_hbd10(pa, W, pb, W, w, h, C.byref(s)) and sse[i] == s.value, pairs[i]


def test_hbd_txfm64_quant_inverse_4k(hip, pkg, orc):
    cur, prd = frame10(7)
    ts = 4   # TX_64X64
    descs = np.array([pkg.tx_desc(x, y, 0) for y in range(0, H - 63, 64) for x in range(0, W - 63, 64)], np.uint32)
    n = len(descs)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "txfm_tables.npz"))   # the reference's own tables (make_golden.py)
    qp = g["qp/10/60/0"]; scan, iscan = g[f"scan/{ts}/0"], g[f"iscan/{ts}/0"]
    qs = pkg.QuantParams()
    for name, row in (("zbin", qp[0]), ("round", qp[1]), ("quant", qp[2]), ("quant_shift", qp[3]), ("dequant", qp[4])):
        getattr(qs, name)[0] = int(row[0]); getattr(qs, name)[1] = int(row[1])
    qs.log_scale = tc.TX_SCALE[ts]; qs.variant = 1     # svt_aom_highbd_quantize_b
    d_isc = hip.to_device(iscan.astype(np.int16))
    st = pkg.ScanTables(); st.iscan[0] = d_isc.value
    d_cur, d_prd, d_desc = hip.to_device(cur), hip.to_device(prd), hip.to_device(descs)
    NK = 32 * 32
    d_q, d_dq, d_eob, d_rec = hip.empty(n * NK * 4), hip.empty(n * NK * 4), hip.empty(n * 2), hip.to_device(np.zeros_like(cur))
    hip.check(hip.L.svt_hip_fwd_txfm_quant_batch_dev(hip.h, ts, 2, d_cur, W, d_prd, W, d_desc, n, C.byref(qs), C.byref(st), None, d_q, d_dq, d_eob, None, None), "fwd64")
    hip.check(hip.L.svt_hip_inv_txfm_add_batch_dev(hip.h, ts, 2, BD, d_dq, d_prd, W, d_rec, W, d_desc, n), "inv64")
    q = hip.to_host(d_q, (n, NK), np.int32); dq = hip.to_host(d_dq, (n, NK), np.int32); eob = hip.to_host(d_eob, (n,), np.uint16)
    rec = hip.to_host(d_rec, cur.shape, np.uint16)
    hip.free(d_cur, d_prd, d_desc, d_q, d_dq, d_eob, d_rec, d_isc)
    # properties at full size: eob bounds the non-zero coefficients in scan order; reconstruction is closer to the source than the prediction
    pos = iscan.astype(np.int64)
    for i in range(0, n, 37):
        nz = np.nonzero(q[i])[0]
        assert (len(nz) == 0 and eob[i] == 0) or (eob[i] == pos[nz].max() + 1)
    e_rec = ((rec[:H - H % 64, :].astype(np.int64) - cur[:H - H % 64, :]) ** 2).sum(); e_prd = ((prd[:H - H % 64, :].astype(np.int64) - cur[:H - H % 64, :]) ** 2).sum()
    assert e_rec < e_prd
    # seeded sample vs the oracle chain (fwd 64x64 -> handle_transform64x64 -> highbd quantize_b -> inverse)
    rng = np.random.default_rng(8)
    for i in rng.choice(n, 12, replace=False):
        x, y = int(descs[i] & 0x3FFF), int((descs[i] >> 14) & 0x3FFF)
        res = (cur[y:y + 64, x:x + 64].astype(np.int32) - prd[y:y + 64, x:x + 64]).astype(np.int16)
        co = tc.orc_fwd(orc, np.ascontiguousarray(res), 64, 0, ts, BD)
        orc.orc_handle_transform.restype = C.c_uint64
        orc.orc_handle_transform(ptr(co), ts)
        eq, edq = np.zeros(NK, np.int32), np.zeros(NK, np.int32); e_eob = C.c_uint16()
        z = [np.array(r, np.int16) for r in qp[:5]]
        orc.orc_quantize(1, ptr(co), NK, ptr(z[0]), ptr(z[1]), ptr(z[2]), ptr(z[3]), ptr(eq), ptr(edq), ptr(z[4]), C.byref(e_eob), ptr(scan.astype(np.int16)), tc.TX_SCALE[ts])
        assert np.array_equal(q[i], eq) and np.array_equal(dq[i], edq) and eob[i] == e_eob.value, i
        er = np.zeros((64, 64), np.uint16)
        orc.orc_inv_txfm2d_add(ptr(edq), ptr(np.ascontiguousarray(prd[y:y + 64, x:x + 64])), 64, ptr(er), 64, 0, ts, BD)
        assert np.array_equal(rec[y:y + 64, x:x + 64], er), i


def test_hbd_selfguided_4k(hip, orc):
    cur, dgd0 = frame10(9)
    EXT, US = 3, 64
    ext = np.ascontiguousarray(np.pad(dgd0, EXT, mode="edge")); st = ext.shape[1]; off = (EXT * st + EXT) * 2
    ux, uy = max((W + 32) // 64, 1), max((H + 32) // 64, 1)
    lim = np.zeros((ux * uy, 4), np.int32); orc.orc_rest_unit_limits(W, H, 0, US, ptr(lim))
    d_ext, d_src = hip.to_device(ext), hip.to_device(cur)
    d_sums = hip.to_device(np.zeros((ux * uy, 16, 5), np.int64))
    hip.check(hip.L.svt_hip_sgr_search_plane_dev(hip.h, 2, BD, d_ext.value + off, st, d_src, W, W, H, US, 0, 0xFFFF, d_sums), "search10")
    sums = hip.to_host(d_sums, (ux * uy, 16, 5), np.int64)
    rng = np.random.default_rng(10)
    u_ep = rng.integers(0, 16, ux * uy).astype(np.uint8); u_ep[::11] = 255
    u_xqd = np.stack([rng.integers(-96, 32, ux * uy), rng.integers(-32, 96, ux * uy)], 1).astype(np.int32)
    d_ep, d_xqd, d_dst = hip.to_device(u_ep), hip.to_device(u_xqd), hip.to_device(np.zeros_like(cur))
    hip.check(hip.L.svt_hip_sgr_apply_plane_dev(hip.h, 2, BD, d_ext.value + off, st, d_dst, W, W, H, US, 0, d_src, W, d_ep, d_xqd), "apply10")   # stripes see `cur` as the deblocked plane
    out = hip.to_host(d_dst, cur.shape, np.uint16)
    hip.free(d_ext, d_src, d_sums, d_ep, d_xqd, d_dst)
    # property: RESTORE_NONE units are copies of the degraded picture
    for u in range(0, ux * uy, 11):
        x0, x1, y0, y1 = lim[u]; assert np.array_equal(out[y0:y1, x0:x1], dgd0[y0:y1, x0:x1])
    # seeded sample of units (corners, bottom row with the ragged last unit, interior) vs the oracle
    prm = np.ctypeslib.as_array((C.c_int32 * 4 * 16).in_dll(orc, "orc_sgr_params"))
    pick = [0, ux - 1, (uy - 1) * ux, ux * uy - 1] + [int(v) for v in rng.choice(ux * uy, 4, replace=False)]
    for u in pick:
        x0, x1, y0, y1 = [int(v) for v in lim[u]]; w, h = x1 - x0, y1 - y0
        fs = ((w + 7) & ~7) + 8
        for ep in (1, 10, 15):
            f0 = np.zeros((h, fs), np.int32); f1 = np.zeros((h, fs), np.int32)
            for i in range(0, h, 64):
                for j in range(0, w, 64):
                    orc.orc_sgr_filter(C.c_void_p(ext.ctypes.data + off + ((y0 + i) * st + x0 + j) * 2), 2, min(64, w - j), min(64, h - i), st,
                                       C.c_void_p(f0.ctypes.data + (i * fs + j) * 4), C.c_void_p(f1.ctypes.data + (i * fs + j) * 4), fs, ep, BD)
            s = (C.c_int64 * 5)()
            orc.orc_sgr_proj_sums(C.c_void_p(cur.ctypes.data + (y0 * W + x0) * 2), W, C.c_void_p(ext.ctypes.data + off + (y0 * st + x0) * 2), st, 2, w, h, ptr(f0), fs, ptr(f1), fs, ep, s)
            assert list(sums[u, ep]) == list(s), (u, ep)
    # apply: the oracle's stripe-aware plane function on a crop that contains whole unit rows (rows 0..183 = units rows 0-2 incl. the 8-row offset)
    CH = 184
    crop_ext = np.ascontiguousarray(ext[:CH + 2 * EXT + 64, :]); crop_dbl = np.ascontiguousarray(cur[:CH + 64, :])
    exp = np.zeros((CH + 64, W), np.uint16)
    cuy = max((CH + 64 + 32) // 64, 1)
    orc.orc_sgr_apply_plane(ptr(crop_dbl), W, C.c_void_p(crop_ext.ctypes.data + off), st, 2, W, CH + 64, 0, 0, US, BD, ptr(u_ep[:ux * cuy].copy()),
                            ptr(u_xqd[:ux * cuy].copy()), ptr(exp), W)
    assert np.array_equal(out[:CH - 64], exp[:CH - 64]), np.argwhere(out[:CH - 64] != exp[:CH - 64])[:5]


def test_hbd_windowed_search(hip, pkg, orc):
    """configs[3] 'HBD SAD, windowed full search': svt_hip_sad_loop16_batch_dev (sad_16b_kernel over a window, svt_sad_loop_kernel's order) vs the
    oracle on a 10-bit frame: 64x64 blocks with 64x64 / clipped windows, small blocks, sub-sampled rows, ties."""
    rng = np.random.default_rng(12)
    w, h = 512, 320
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 480 + 300 * np.sin(xx / 37.0) * np.cos(yy / 29.0)
    cur = np.clip(base + rng.normal(0, 9, (h, w)), 0, 1023).astype(np.uint16)
    ref = np.clip(np.roll(base, (2, -3), (0, 1)) + rng.normal(0, 9, (h, w)), 0, 1023).astype(np.uint16)
    ref[:70, :140] = 500; cur[:64, :64] = 503           # a flat region: every candidate ties, the first one wins
    jobs = []
    for by in range(0, h - 63, 64):
        for bx in range(0, w - 63, 64):
            x0, y0 = max(bx - 32, 0), max(by - 32, 0)
            jobs.append((bx, by, x0, y0, 64, 64, min(64, w - 64 - x0 + 1), min(64, h - 64 - y0 + 1), 1, 0))
    jobs += [(32, 32, 20, 20, 32, 32, 16, 13, 1, 0), (128, 64, 100, 40, 48, 17, 24, 40, 1, 0), (64, 128, 50, 120, 16, 64, 8, 64, 1, 0), (192, 64, 160, 33, 64, 1, 56, 33, 1, 0)]   # the LDS form's other shapes
    jobs += [(33, 35, 21, 22, 32, 32, 16, 13, 1, 0), (129, 65, 101, 40, 64, 20, 24, 30, 1, 0), (65, 129, 51, 121, 16, 40, 8, 20, 1, 0), (97, 3, 70, 0, 48, 16, 40, 8, 1, 0)]   # source blocks at odd columns: their sample pairs do not start on a dword (the LDS form's 16-bit-load path)
    jobs += [(16, 16, 8, 8, 16, 16, 17, 9, 1, 0), (100, 40, 90, 30, 32, 32, 5, 40, 2, 0), (200, 100, 199, 99, 8, 8, 3, 3, 1, 0), (64, 64, 60, 60, 64, 32, 9, 9, 2, 0)]
    n = len(jobs)
    S = (pkg.SadLoop * n)(*[pkg.SadLoop(*j) for j in jobs])
    e_sad, e_xy = np.zeros(n, np.uint32), np.full((n, 2), -7, np.int16)
    orc.orc_sad_loop16_batch(ptr(cur), w, ptr(ref), w, S, 0, n, ptr(e_sad), ptr(e_xy))
    d_c, d_r, d_S = hip.to_device(cur), hip.to_device(ref), hip.to_device(np.frombuffer(bytes(S), np.uint8).copy())
    d_sad, d_xy = hip.to_device(np.zeros(n, np.uint32)), hip.to_device(np.full((n, 2), -7, np.int16))
    hip.check(hip.L.svt_hip_sad_loop16_batch_dev(hip.h, d_c, w, d_r, w, d_S, n, d_sad, d_xy), "sad loop 16")
    g_sad, g_xy = hip.to_host(d_sad, (n,), np.uint32), hip.to_host(d_xy, (n, 2), np.int16)
    hip.free(d_c, d_r, d_S, d_sad, d_xy)
    assert np.array_equal(g_sad, e_sad) and np.array_equal(g_xy, e_xy), np.argwhere(g_sad != e_sad)[:5]
    assert (e_xy[0] == (0, 0)).all() and e_sad.min() < 0xffffff
