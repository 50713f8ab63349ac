"""Shared helpers for the CDEF tests: synthetic frames, skip maps, and a driver that runs the REAL
reference per-filter-block functions (svt_cdef_filter_fb + compute_cdef_dist*) the way cdef_seg_search
does (/root/reference/Source/Lib/Encoder/Codec/EbCdefProcess.c:168-273)."""
import ctypes as C

import numpy as np

from conftest import ptr

BSTRIDE, VB, HB, VERY_LARGE = 144, 3, 8, 16384


class CdefList(C.Structure):
    _fields_ = [("by", C.c_uint8), ("bx", C.c_uint8), ("skip", C.c_uint8)]


def make_frame(w, h, bd, seed, smooth=True):
    rng = np.random.default_rng(seed)
    planes_src, planes_rec = [], []
    for pli in range(3):
        pw, ph = (w, h) if pli == 0 else (w // 2, h // 2)
        yy, xx = np.mgrid[0:ph, 0:pw]
        base = 90 + 50 * np.sin(xx / 9.0 + pli) * np.cos(yy / 7.0) + 0.1 * xx
        edge = 40 * (((xx + 2 * yy) // 12) % 2)
        s = (base + edge + (0 if smooth else rng.normal(0, 20, (ph, pw)))) * (1 << (bd - 8))
        src = np.clip(s, 0, (1 << bd) - 1)
        rec = np.clip(src + rng.normal(0, 6 * (1 << (bd - 8)), (ph, pw)), 0, (1 << bd) - 1)
        dt = np.uint8 if bd == 8 else np.uint16
        planes_src.append(np.ascontiguousarray(src.astype(dt))); planes_rec.append(np.ascontiguousarray(rec.astype(dt)))
    skip8 = (rng.random((h // 8, w // 8)) < 0.3).astype(np.uint8)
    return planes_src, planes_rec, skip8


def ref_search_fb(ref, rec, src, bd, skip8, fbr, fbc, pri_damping, ngi=64):
    """mse[2][64] of one filter block via the reference's own functions."""
    cs = bd - 8
    h, w = rec[0].shape
    nvfb, nhfb = (h + 63) // 64, (w + 63) // 64
    nb_y, nb_x = min(8, h // 8 - 8 * fbr), min(8, w // 8 - 8 * fbc)
    dl = (CdefList * 64)(); count = 0
    for by in range(nb_y):
        for bx in range(nb_x):
            if not skip8[8 * fbr + by, 8 * fbc + bx]:
                dl[count] = CdefList(by, bx, 0); count += 1
    mse = np.zeros((2, 64), np.uint64)
    if count == 0:
        return None
    dirs = ((C.c_int32 * 16) * 16)(); var = ((C.c_int32 * 16) * 16)(); dirinit = C.c_int32(0)
    for pli in range(3):
        dec = 1 if pli else 0
        p = rec[pli].astype(np.uint16); ph, pw = p.shape
        inbuf = np.full(BSTRIDE * (128 + 2 * VB), VERY_LARGE, np.uint16)
        yoff, xoff = VB * (fbr != 0), HB * (fbc != 0)
        ysize = ((nb_y * 8) >> dec) + VB * (fbr + 1 < nvfb) + yoff
        xsize = ((nb_x * 8) >> dec) + HB * (fbc + 1 < nhfb) + xoff
        y0, x0 = ((64 * fbr) >> dec) - yoff, ((64 * fbc) >> dec) - xoff
        view = inbuf.reshape(-1, BSTRIDE)
        view[VB - yoff:VB - yoff + ysize, HB - xoff:HB - xoff + xsize] = p[y0:y0 + ysize, x0:x0 + xsize]
        in_ptr = C.c_void_p(inbuf.ctypes.data + 2 * (VB * BSTRIDE + HB))
        bsize = 0 if dec else 3  # BLOCK_4X4 / BLOCK_8X8
        s = src[pli]
        sp = C.c_void_p(s.ctypes.data + s.itemsize * (((64 * fbr) >> dec) * s.shape[1] + ((64 * fbc) >> dec)))
        for gi in range(ngi):
            pri, sec = gi // 4, gi % 4
            if bd == 8:
                tmp = np.zeros(1 << 14, np.uint8)
                ref.svt_cdef_filter_fb(ptr(tmp), None, BSTRIDE, in_ptr, dec, dec, dirs, C.byref(dirinit), var, pli, dl, count,
                                       pri, sec + (sec == 3), pri_damping, pri_damping, cs)
                f = ref.compute_cdef_dist_8bit_c; f.restype = C.c_uint64
                d = f(sp, s.shape[1], ptr(tmp), dl, count, bsize, cs, pli)
            else:
                tmp = np.zeros(1 << 14, np.uint16)
                ref.svt_cdef_filter_fb(None, ptr(tmp), BSTRIDE, in_ptr, dec, dec, dirs, C.byref(dirinit), var, pli, dl, count,
                                       pri, sec + (sec == 3), pri_damping, pri_damping, cs)
                f = ref.compute_cdef_dist_c; f.restype = C.c_uint64
                d = f(sp, s.shape[1], ptr(tmp), dl, count, bsize, cs, pli)
            mse[0 if pli == 0 else 1, gi] += np.uint64(d)
    return mse


def orc_search(orc, rec, src, bd, skip8, pri_damping, pick=0, fb_begin=0, fb_end=None):
    h, w = rec[0].shape
    nfb = ((h + 63) // 64) * ((w + 63) // 64)
    fb_end = nfb if fb_end is None else fb_end
    mse = np.zeros((2, nfb, 64), np.uint64)
    P3 = C.c_void_p * 3; I3 = C.c_int * 3
    orc.orc_cdef_search_frame(P3(*[r.ctypes.data for r in rec]), I3(*[r.shape[1] for r in rec]), P3(*[s.ctypes.data for s in src]),
                              I3(*[s.shape[1] for s in src]), rec[0].itemsize, w, h, ptr(skip8), pri_damping, bd, pick, ptr(mse),
                              fb_begin, fb_end)
    return mse
