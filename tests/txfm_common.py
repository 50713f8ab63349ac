"""Shared helpers for transform / quant parity tests."""
import ctypes as C

import numpy as np

from conftest import ptr

TXW = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TXH = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
TX_NAMES = [f"{w}x{h}" for w, h in zip(TXW, TXH)]


def legal_types(ts):
    """AV1 legality as used by the reference's tests (/root/reference/test/TxfmCommon.h:178-218):
    64-pt: DCT_DCT only; 32-pt: DCT_DCT and IDTX; <=16: all 16 types."""
    m = max(TXW[ts], TXH[ts])
    return [0] if m == 64 else ([0, 9] if m == 32 else list(range(16)))


def ref_fwd(ref, x, stride, tt, ts, bd):
    w, h = TXW[ts], TXH[ts]
    name = f"svt_av1_transform_two_d_{w}x{h}_c" if w == h else f"svt_av1_fwd_txfm2d_{w}x{h}_c"
    out = np.zeros(w * h, np.int32)
    getattr(ref, name)(ptr(x), ptr(out), stride, tt, C.c_uint8(bd))
    return out


def ref_handle(ref, coeff, ts):
    w, h = TXW[ts], TXH[ts]
    f = getattr(ref, f"svt_handle_transform{w}x{h}_c")
    f.restype = C.c_uint64
    return f(ptr(coeff))


def ref_inv(ref, coeff, pred, stride_r, recon, stride_w, tt, ts, bd):
    w, h = TXW[ts], TXH[ts]
    f = getattr(ref, f"svt_av1_inv_txfm2d_add_{w}x{h}_c")
    if w == h:
        f(ptr(coeff), ptr(pred), stride_r, ptr(recon), stride_w, tt, bd)
    elif min(w, h) == 4:
        f(ptr(coeff), ptr(pred), stride_r, ptr(recon), stride_w, tt, ts, bd)
    else:
        f(ptr(coeff), ptr(pred), stride_r, ptr(recon), stride_w, tt, ts, min(w, 32) * min(h, 32), bd)


def orc_fwd(orc, x, stride, tt, ts, bd):
    out = np.zeros(TXW[ts] * TXH[ts], np.int32)
    orc.orc_fwd_txfm2d(ptr(x), ptr(out), stride, tt, ts, bd)
    return out


# ------------------------------------------------------------------ tables from the real reference
def ref_scan(ref, ts, tt):
    s, i = C.POINTER(C.c_int16)(), C.POINTER(C.c_int16)()
    n = ref.ref_shim_scan(ts, tt, C.byref(s), C.byref(i))
    return np.ctypeslib.as_array(s, (n,)).copy(), np.ctypeslib.as_array(i, (n,)).copy()


def ref_qparams(ref, bd, qindex, plane):
    out = np.zeros((7, 2), np.int16)
    ref.ref_shim_qparams(bd, qindex, plane, ptr(out))
    return out  # rows: zbin, round, quant, quant_shift, dequant, round_fp, quant_fp


TX_SCALE = [0, 0, 0, 1, 2, 0, 0, 0, 0, 1, 1, 2, 2, 0, 0, 0, 0, 1, 1]  # av1_get_tx_scale_tab (EbFullLoop.h:66)
SCAN_CLASS = [0] * 10 + [1, 2, 1, 2, 1, 2]  # per TxType: 0 default zig-zag, 1 mrow, 2 mcol (EbCoefficients.h:2563)


def orc_quant(orc, variant, coeff, qp, scan, log_scale):
    n = len(scan)
    q = np.zeros(n, np.int32); dq = np.zeros(n, np.int32); eob = C.c_uint16(0)
    rnd, qnt = (qp[5], qp[6]) if variant >= 2 else (qp[1], qp[2])
    orc.orc_quantize(variant, ptr(coeff), n, ptr(qp[0]), ptr(np.ascontiguousarray(rnd)), ptr(np.ascontiguousarray(qnt)),
                     ptr(qp[3]), ptr(q), ptr(dq), ptr(qp[4]), C.byref(eob), ptr(scan), log_scale)
    return q, dq, eob.value


def ref_quant(ref, variant, coeff, qp, scan, iscan, log_scale):
    n = len(scan)
    q = np.zeros(n, np.int32); dq = np.zeros(n, np.int32); eob = C.c_uint16(0)
    zb, rnd, qnt, qsh, deq, rfp, qfp = [np.ascontiguousarray(np.repeat(r[[0, 1]], [1, 7])) for r in qp]  # SIMD-width layout
    if variant == 0:
        ref.svt_aom_quantize_b_c_ii(ptr(coeff), C.c_ssize_t(n), ptr(zb), ptr(rnd), ptr(qnt), ptr(qsh), ptr(q), ptr(dq), ptr(deq),
                                    C.byref(eob), ptr(scan), ptr(iscan), None, None, log_scale)
    elif variant == 1:
        ref.svt_aom_highbd_quantize_b_c(ptr(coeff), C.c_ssize_t(n), ptr(zb), ptr(rnd), ptr(qnt), ptr(qsh), ptr(q), ptr(dq), ptr(deq),
                                        C.byref(eob), ptr(scan), ptr(iscan), None, None, log_scale)
    elif variant == 2:
        name = ["svt_av1_quantize_fp_c", "svt_av1_quantize_fp_32x32_c", "svt_av1_quantize_fp_64x64_c"][log_scale]
        getattr(ref, name)(ptr(coeff), C.c_ssize_t(n), ptr(zb), ptr(rfp), ptr(qfp), ptr(qsh), ptr(q), ptr(dq), ptr(deq),
                           C.byref(eob), ptr(scan), ptr(iscan))
    else:
        ref.svt_av1_highbd_quantize_fp_c(ptr(coeff), C.c_ssize_t(n), ptr(zb), ptr(rfp), ptr(qfp), ptr(qsh), ptr(q), ptr(dq), ptr(deq),
                                         C.byref(eob), ptr(scan), ptr(iscan), C.c_int16(log_scale))
    return q, dq, eob.value
