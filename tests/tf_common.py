"""Shared inputs for the temporal-filter tests: the MeContext TF fields of every 64x64 block (OrcTfBlk64 == SvtHipTfBlk64 layout),
synthetic central / motion-compensated pictures, and the reference-test style extremes
(/root/reference/test/TemporalFilterTestPlanewise.cc:200-330: random pixels, random block errors, random noise levels)."""
import ctypes as C

import numpy as np


class TfBlk64(C.Structure):
    _fields_ = [("mv16_x", C.c_int16 * 16), ("mv16_y", C.c_int16 * 16), ("err16", C.c_uint64 * 16),
                ("mv32_x", C.c_int16 * 4), ("mv32_y", C.c_int16 * 4), ("err32", C.c_uint64 * 4), ("split", C.c_int32 * 4)]


class TfRef(C.Structure):
    _fields_ = [("pred", C.c_void_p * 3), ("pred_stride", C.c_int * 3), ("blocks", C.c_void_p)]


BLK_DTYPE = np.dtype([("mv16_x", np.int16, 16), ("mv16_y", np.int16, 16), ("err16", np.uint64, 16),
                      ("mv32_x", np.int16, 4), ("mv32_y", np.int16, 4), ("err32", np.uint64, 4), ("split", np.int32, 4)], align=True)
assert BLK_DTYPE.itemsize == C.sizeof(TfBlk64) == 256


def make_blocks(rng, n, bd, big_mv=False, err_max=60):
    b = np.zeros(n, BLK_DTYPE)
    sc = 16 if bd > 8 else 1
    b["err16"] = rng.integers(0, 256 * err_max + 1, (n, 16)) * sc          # mean squared error up to err_max per pixel
    b["err32"] = rng.integers(0, 1024 * err_max + 1, (n, 4)) * sc
    hi = 700 if big_mv else 40
    for k in ("mv16_x", "mv16_y"): b[k] = rng.integers(-hi, hi + 1, (n, 16))
    for k in ("mv32_x", "mv32_y"): b[k] = rng.integers(-hi, hi + 1, (n, 4))
    b["split"] = rng.integers(0, 2, (n, 4))
    if n > 2:
        b["err16"][1] = 0; b["err32"][1] = 0; b["mv16_x"][1] = 0; b["mv16_y"][1] = 0; b["mv32_x"][1] = 0; b["mv32_y"][1] = 0   # perfect match
        b["err16"][2] = np.uint64(1) << np.uint64(40); b["err32"][2] = np.uint64(1) << np.uint64(40)                          # hopeless block
    return b


def make_pictures(rng, w, h, bd, ss_x, ss_y, n_refs, noise=6.0):
    """Central picture + n_refs 'motion compensated' pictures (central + noise of varying strength, a few saturated / flat regions)."""
    dt = np.uint8 if bd == 8 else np.uint16
    mx = (1 << bd) - 1
    yy, xx = np.mgrid[0:h, 0:w]
    base = (110 + 70 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 25 * (((xx // 16) + (yy // 16)) % 2)) * (1 << (bd - 8))
    planes = [base, base[::1 << ss_y, ::1 << ss_x] * 0.5 + 60 * (1 << (bd - 8)), 200 * (1 << (bd - 8)) - base[::1 << ss_y, ::1 << ss_x] * 0.4]
    src = [np.ascontiguousarray(np.clip(p + rng.normal(0, 2 * (1 << (bd - 8)), p.shape), 0, mx).astype(dt)) for p in planes]
    src[0][:40, :40] = mx; src[0][40:64, :64] = 0
    preds = []
    for f in range(n_refs):
        sg = noise * (0.3 + f) * (1 << (bd - 8))
        pr = [np.ascontiguousarray(np.clip(s.astype(np.float64) + rng.normal(0, sg, s.shape), 0, mx).astype(dt)) for s in src]
        if f == 0: pr[0][:32, :32] = 0                     # maximum squared differences in one 32x32 block
        preds.append(pr)
    return src, preds


# SvtHipTfSubpelBlk (include/svt_hip.h): one (64x64 block, frame) pair of the TF sub-pel stage
SUBPEL_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("dst_x", np.int32), ("dst_y", np.int32), ("blk_index", np.int32),
                         ("mv32", np.uint32, 4), ("mv16", np.uint32, 16)], align=True)
assert SUBPEL_DTYPE.itemsize == 100


def mv_word(mx, my):
    """the ME table's vector word: (y << 16) | x, quarter-pel int16 halves of an integer vector"""
    return ((np.asarray(my, np.int64) * 4 & 0xffff) << 16 | (np.asarray(mx, np.int64) * 4 & 0xffff)).astype(np.uint32)


def make_subpel_case(rng, w, h, bd, pad, max_mv=9, noise=3.0, edge_mv=True):
    """A central picture and a padded reference picture (the central one shifted by a smooth sub-pel field + noise), 4:2:0, and one job per 64x64 block
    whose integer vectors point near the true motion; blocks on the picture border get vectors that push the clamp of clamp_mv_to_umv_border_sb."""
    dt = np.uint8 if bd == 8 else np.uint16
    mx = (1 << bd) - 1
    sc = 1 << (bd - 8)

    def tex(hh, ww, ox, oy, k):
        yy, xx = np.mgrid[0:hh, 0:ww].astype(np.float64)
        xx = xx + ox; yy = yy + oy
        return (120 + 60 * np.sin(xx / (7.0 + k)) * np.cos(yy / (5.0 + k)) + 30 * np.sin((xx + 2 * yy) / 3.1)) * sc

    src, ref = [], []
    for p in range(3):
        s = 0 if p == 0 else 1
        ww, hh, pd = w >> s, h >> s, pad >> s
        src.append(np.ascontiguousarray(np.clip(tex(hh, ww, 0, 0, p) + rng.normal(0, noise * sc, (hh, ww)), 0, mx).astype(dt)))
        full = np.clip(tex(hh + 2 * pd, ww + 2 * pd, -pd + 2.3 / (1 + s), -pd - 1.6 / (1 + s), p) + rng.normal(0, noise * sc, (hh + 2 * pd, ww + 2 * pd)), 0, mx).astype(dt)
        ref.append(np.ascontiguousarray(full))
    bc, br = w // 64, h // 64
    jobs = np.zeros(bc * br, SUBPEL_DTYPE)
    for r in range(br):
        for c in range(bc):
            j = jobs[r * bc + c]
            j["x"], j["y"], j["dst_x"], j["dst_y"], j["blk_index"] = c * 64, r * 64, c * 64, r * 64, r * bc + c
            m32x = rng.integers(-max_mv, max_mv + 1, 4); m32y = rng.integers(-max_mv, max_mv + 1, 4)
            m16x = rng.integers(-max_mv, max_mv + 1, 16); m16y = rng.integers(-max_mv, max_mv + 1, 16)
            if edge_mv and (r == 0 or c == 0 or r == br - 1 or c == bc - 1):   # far outside the picture: the clamp decides
                far = pad - 8
                m32x[:] = -far if c == 0 else (far if c == bc - 1 else m32x); m32y[:] = -far if r == 0 else (far if r == br - 1 else m32y)
                m16x[:8] = m32x[0]; m16y[:8] = m32y[0]
            j["mv32"] = mv_word(m32x, m32y); j["mv16"] = mv_word(m16x, m16y)
    return src, ref, jobs
