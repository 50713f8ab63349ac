#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference C path (oracle/_ref/libsvtav1_ref.so, built by
oracle/Makefile.ref from /root/reference).  Run in the build container only; the fixtures are
committed so the GPU box (no /root/reference) and later rounds can pin the oracle and the HIP path.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref.so"))
ref.setup_common_rtcd_internal(0)
ref.setup_rtcd_internal(0)
u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)


def P(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def ref_fullpel_sb(src, ref_win, xo, yo, saw, sah, sub):
    """Drive the reference kernels exactly like open_loop_me_fullpel_search_sblock
    (Source/Lib/Encoder/Codec/EbMotionEstimation.c:814-850, :462-506, :508-812)."""
    best_sad = np.full(85, 128 * 128 * 255, np.uint32)
    best_mv = np.zeros(85, np.uint32)
    e16 = np.zeros((16, 8), np.uint32); e8 = np.zeros((64, 8), np.uint32); e32 = np.zeros((4, 8), np.uint32)
    o16 = np.zeros(16, np.uint32); o8 = np.zeros(64, np.uint32); o32 = np.zeros(4, np.uint32)
    ss, rs = src.strides[0], ref_win.strides[0]
    b = best_sad.ctypes.data; m = best_mv.ctypes.data
    bs64, bs32, bs16, bs8 = b, b + 4, b + 20, b + 84
    bm64, bm32, bm16, bm8 = m, m + 4, m + 20, m + 84
    z = [0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15]
    w8 = saw & ~7
    for cy in range(sah):
        cx = 0
        while cx < saw:
            mv = ((((cy + yo) & 0xFFFF) << 18) | (((cx + xo) << 2) & 0xFFFF)) & 0xFFFFFFFF
            rp = ref_win.ctypes.data + cy * rs + cx
            if cx < w8:
                ref.svt_ext_all_sad_calculation_8x8_16x16_c(C.c_void_p(src.ctypes.data), ss, C.c_void_p(rp), rs, C.c_uint32(mv),
                    C.c_void_p(bs8), C.c_void_p(bs16), C.c_void_p(bm8), C.c_void_p(bm16), P(e16), P(e8), sub)
                ref.svt_ext_eight_sad_calculation_32x32_64x64_c(P(e16), C.c_void_p(bs32), C.c_void_p(bs64), C.c_void_p(bm32),
                    C.c_void_p(bm64), C.c_uint32(mv), P(e32))
                cx += 8
            else:
                for Y in range(4):
                    for X in range(4):
                        p = z[4 * Y + X]
                        ref.svt_ext_sad_calculation_8x8_16x16_c(C.c_void_p(src.ctypes.data + 16 * Y * ss + 16 * X), ss,
                            C.c_void_p(rp + 16 * Y * rs + 16 * X), rs, C.c_void_p(bs8 + 16 * p), C.c_void_p(bs16 + 4 * p),
                            C.c_void_p(bm8 + 16 * p), C.c_void_p(bm16 + 4 * p), C.c_uint32(mv),
                            C.c_void_p(o16.ctypes.data + 4 * p), C.c_void_p(o8.ctypes.data + 16 * p), sub)
                ref.svt_ext_sad_calculation_32x32_64x64_c(P(o16), C.c_void_p(bs32), C.c_void_p(bs64), C.c_void_p(bm32),
                    C.c_void_p(bm64), C.c_uint32(mv), P(o32))
                cx += 1
    return best_sad, best_mv


def gen_me():
    rng = np.random.default_rng(13596)  # the reference tests' seed (test/random.h:142)
    cases = {}
    specs = [("rand_16x8", 16, 8, 0, "rand"), ("rand_24x5_sub", 24, 5, 1, "rand"), ("tie_16x4", 16, 4, 0, "tie"),
             ("refmax_8x3", 8, 3, 0, "refmax"), ("ragged_5x4", 5, 4, 0, "rand"), ("ragged_3x2_sub", 3, 2, 1, "rand"),
             ("rand_32x6", 32, 6, 1, "rand")]
    for name, saw, sah, sub, pat in specs:
        src = rng.integers(0, 256, (64, 64), dtype=np.uint8)
        win = rng.integers(0, 256, (64 + sah - 1, 64 + saw - 1 + 9), dtype=np.uint8)
        if pat == "tie":
            src[:] = 90; win[:] = 100
        if pat == "refmax":
            win[:] = 255
        xo, yo = int(rng.integers(-40, 10)), int(rng.integers(-40, 10))
        sad, mv = ref_fullpel_sb(src, win, xo, yo, saw, sah, sub)
        cases[name + "/src"] = src; cases[name + "/win"] = win
        cases[name + "/par"] = np.array([xo, yo, saw, sah, sub], np.int32)
        cases[name + "/sad"] = sad; cases[name + "/mv"] = mv
    # svt_sad_loop_kernel_c (HME)
    for i, (bw, bh, saw, sah) in enumerate([(16, 16, 32, 9), (32, 32, 16, 7), (64, 64, 8, 5), (16, 8, 48, 3)]):
        src = rng.integers(0, 256, (bh, bw), dtype=np.uint8)
        rs = bw + saw + 3
        win = rng.integers(0, 256, (bh + sah, rs), dtype=np.uint8)
        best = C.c_uint64(0); xc = C.c_int16(0); yc = C.c_int16(0)
        ref.svt_sad_loop_kernel_c(P(src), bw, P(win), rs, bh, bw, C.byref(best), C.byref(xc), C.byref(yc), rs,
                                  C.c_int16(saw), C.c_int16(sah))
        cases[f"sadloop{i}/src"] = src; cases[f"sadloop{i}/win"] = win
        cases[f"sadloop{i}/par"] = np.array([bw, bh, saw, sah], np.int32)
        cases[f"sadloop{i}/out"] = np.array([best.value, xc.value, yc.value], np.int64)
    np.savez_compressed(os.path.join(HERE, "me_fullpel.npz"), **cases)
    print("me_fullpel.npz:", len(cases), "arrays")


def gen_txfm():
    """Tables (scan orders, quantizer parameters, flips) and fwd-txfm / quant / inv-txfm vectors."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import txfm_common as tc
    tabs = {}
    for ts in range(19):
        for cls, tt in ((0, 0), (1, 10), (2, 11)):
            if max(tc.TXW[ts], tc.TXH[ts]) > 16 and cls:
                continue  # 1-D classes only exist for <= 16-pt transforms
            sc, isc = tc.ref_scan(ref, ts, tt)
            tabs[f"scan/{ts}/{cls}"] = sc; tabs[f"iscan/{ts}/{cls}"] = isc
    for bd in (8, 10):
        for qi in (0, 20, 60, 120, 200, 255):
            for pl in range(3):
                tabs[f"qp/{bd}/{qi}/{pl}"] = tc.ref_qparams(ref, bd, qi, pl)
    fl = np.zeros((16, 2), np.int32)
    for tt in range(16):
        ud, lr = C.c_int(0), C.c_int(0)
        ref.ref_shim_flip(tt, C.byref(ud), C.byref(lr)); fl[tt] = (ud.value, lr.value)
    tabs["flip"] = fl
    np.savez_compressed(os.path.join(HERE, "txfm_tables.npz"), **tabs)
    print("txfm_tables.npz:", len(tabs), "arrays")

    rng = np.random.default_rng(13596)
    vec = {}
    for ts in range(19):
        w, h = tc.TXW[ts], tc.TXH[ts]
        kw, kh = min(w, 32), min(h, 32)
        types = tc.legal_types(ts)
        pick = types if len(types) <= 2 else [0, 3, 6, 9, 10, 13, 15, int(rng.integers(0, 16))]
        for tt in pick:
            for bd in (8, 10):
                lim = (1 << bd) - 1
                x = rng.integers(-lim, lim + 1, (h, w), dtype=np.int16)
                co = tc.ref_fwd(ref, x, w, tt, ts, bd)
                en = tc.ref_handle(ref, co, ts) if max(w, h) == 64 else 0
                co = np.ascontiguousarray(co[:kw * kh])
                scan, iscan = tc.ref_scan(ref, ts, tt)
                qp = tc.ref_qparams(ref, bd, 60, 0)
                v = 0 if bd == 8 else 1
                q, dq, eob = tc.ref_quant(ref, v, co, qp, scan, iscan, tc.TX_SCALE[ts])
                qf, dqf, eobf = tc.ref_quant(ref, v + 2, co, qp, scan, iscan, tc.TX_SCALE[ts])
                pred = rng.integers(0, 1 << bd, (h, w), dtype=np.uint16)
                rec = np.zeros((h, w), np.uint16)
                tc.ref_inv(ref, dq, pred, w, rec, w, tt, ts, bd)
                k = f"{ts}/{tt}/{bd}"
                vec[k + "/x"] = x; vec[k + "/coeff"] = co; vec[k + "/energy"] = np.array([en], np.uint64)
                vec[k + "/q"] = q; vec[k + "/dq"] = dq; vec[k + "/eob"] = np.array([eob, eobf], np.int32)
                vec[k + "/qf"] = qf; vec[k + "/dqf"] = dqf
                vec[k + "/pred"] = pred; vec[k + "/rec"] = rec
    np.savez_compressed(os.path.join(HERE, "txfm_quant.npz"), **vec)
    print("txfm_quant.npz:", len(vec), "arrays")


if __name__ == "__main__":
    gen_me()
    gen_txfm()
