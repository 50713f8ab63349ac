"""Shared by the mode-decision precompute tests: the 85 square PUs of a 64x64 superblock in the order of the open-loop ME results (64x64, 32x32 x 4, 16x16 x 16,
8x8 x 64: EbMeTierZeroPu, Encoder/Codec/EbMotionEstimationLcuResults.h) — any order works for the entry point, which takes the list as an argument — and a
synthetic picture with its reference planes and a vector table."""
import ctypes as C

import numpy as np


def square_pus():
    pus = [(0, 0, 64, 64)]
    for s, n in ((32, 2), (16, 4), (8, 8)):
        pus += [(x * s, y * s, s, s) for y in range(n) for x in range(n)]
    return pus


def make_case(rng, w, h, n_refs, pad=40, mv_range=24, frac_none=0.1):
    """source + n_refs padded reference planes of a w x h picture, and [n_sb][85][n_refs] vectors (whole samples) some of which are 'none' and some of which
    point outside the reference's allocation"""
    sb_cols, sb_rows = (w + 63) // 64, (h + 63) // 64
    n_sb = sb_cols * sb_rows
    base = rng.integers(0, 256, (h + 2 * pad + 64, w + 2 * pad + 64)).astype(np.uint8)
    src = np.ascontiguousarray(base[pad:pad + h, pad:pad + w])
    refs = []
    for r in range(n_refs):
        dx, dy = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
        p = base[pad + dy - pad:pad + dy + h + pad, pad + dx - pad:pad + dx + w + pad].astype(np.int16) if min(pad + dy - pad, pad + dx - pad) >= 0 else None
        if p is None:
            p = rng.integers(0, 256, (h + 2 * pad, w + 2 * pad)).astype(np.int16)
        p = np.clip(p + rng.integers(-3, 4, p.shape), 0, 255).astype(np.uint8)
        if r == 1: p[:] = 255   # saturated differences
        refs.append(np.ascontiguousarray(p))
    pus = square_pus()
    mvx = rng.integers(-mv_range, mv_range + 1, (n_sb, len(pus), n_refs)).astype(np.int16)
    mvy = rng.integers(-mv_range, mv_range + 1, (n_sb, len(pus), n_refs)).astype(np.int16)
    far = rng.random(mvx.shape) < 0.03
    mvx[far] = rng.integers(-3 * pad, 3 * pad, int(far.sum())).astype(np.int16)   # some of these leave the allocation
    mvx[rng.random(mvx.shape) < frac_none] = -32768
    mv = (mvx.astype(np.uint16).astype(np.uint32)) | (mvy.astype(np.uint16).astype(np.uint32) << 16)
    return src, refs, pus, np.ascontiguousarray(mv), sb_cols, n_sb, pad


def oracle_table(orc, src, refs, pus, mv, sb_cols, n_sb, pad, pic_w, pic_h):
    n_refs = len(refs)
    pu4 = np.array(pus, np.uint8)
    planes = (C.c_void_p * n_refs)(*[r.ctypes.data + pad * r.shape[1] + pad for r in refs])
    strides = (C.c_int * n_refs)(*[r.shape[1] for r in refs])
    box = np.array([[-pad, -pad, r.shape[1] - pad, r.shape[0] - pad] for r in refs], np.int32)
    out = np.zeros(mv.shape, np.uint32)
    orc.orc_md_fullpel_sad_picture(src.ctypes.data_as(C.c_void_p), src.shape[1], pic_w, pic_h, sb_cols, n_sb, len(pus), pu4.ctypes.data_as(C.c_void_p), n_refs, planes, strides,
                                   box.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def oracle_grid(orc, src, refs, pus, mv, sb_cols, n_sb, pad, pic_w, pic_h, bank):
    """the 7 x 7 sub-pel grid table of svt_hip_md_subpel_grid_picture_dev by the oracle: [n_sb][n_pus][n_refs][49][2]"""
    n_refs = len(refs)
    pu4 = np.array(pus, np.uint8)
    planes = (C.c_void_p * n_refs)(*[r.ctypes.data + pad * r.shape[1] + pad for r in refs])
    strides = (C.c_int * n_refs)(*[r.shape[1] for r in refs])
    box = np.array([[-pad, -pad, r.shape[1] - pad, r.shape[0] - pad] for r in refs], np.int32)
    out = np.zeros(mv.shape + (49, 2), np.uint32)
    orc.orc_md_subpel_grid_picture(src.ctypes.data_as(C.c_void_p), src.shape[1], pic_w, pic_h, sb_cols, n_sb, len(pus), pu4.ctypes.data_as(C.c_void_p), n_refs, planes, strides,
                                   box.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p), bank, out.ctypes.data_as(C.c_void_p))
    return out


def oracle_avg_table(orc, src, refs, pus, mv, sb_cols, n_sb, pad, pic_w, pic_h, pairs):
    """the compound-average table of svt_hip_md_fullpel_avg_sad_picture_dev by the oracle: [n_sb][n_pus][n_pairs]"""
    n_refs = len(refs)
    pu4 = np.array(pus, np.uint8)
    planes = (C.c_void_p * n_refs)(*[r.ctypes.data + pad * r.shape[1] + pad for r in refs])
    strides = (C.c_int * n_refs)(*[r.shape[1] for r in refs])
    box = np.array([[-pad, -pad, r.shape[1] - pad, r.shape[0] - pad] for r in refs], np.int32)
    pr = np.array(pairs, np.uint8)
    out = np.zeros(mv.shape[:2] + (len(pairs),), np.uint32)
    orc.orc_md_fullpel_avg_sad_picture(src.ctypes.data_as(C.c_void_p), src.shape[1], pic_w, pic_h, sb_cols, n_sb, len(pus), pu4.ctypes.data_as(C.c_void_p), n_refs, planes, strides,
                                       box.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p), len(pairs), pr.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out
