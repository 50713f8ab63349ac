"""GPU parity: self-guided restoration (HIP through the C ABI) vs the oracle (pinned to
svt_av1_selfguided_restoration_c / svt_apply_selfguided_restoration_c / svt_get_proj_subspace_c):
filter planes for all 16 parameter sets, the per-unit projection sums of the search, and the apply
pass; 8- and 10-bit.  Mirrors /root/reference/test/selfguided_filter_test.cc:248-562."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr

pytestmark = pytest.mark.gpu
EXT = 3


def make_planes(w, h, bd, seed):
    rng = np.random.default_rng(seed)
    dt = np.uint8 if bd == 8 else np.uint16
    yy, xx = np.mgrid[0:h, 0:w]
    base = (80 + 60 * np.sin(xx / 11.0) * np.cos(yy / 5.0) + 30 * (((xx // 7) + (yy // 9)) % 2)) * (1 << (bd - 8))
    src = np.clip(base, 0, (1 << bd) - 1)
    dgd = np.clip(src + rng.normal(0, 5 * (1 << (bd - 8)), (h, w)), 0, (1 << bd) - 1)
    dgd[:16, :16] = (1 << bd) - 1; dgd[16:32, :16] = 0
    ext = np.ascontiguousarray(np.pad(dgd.astype(dt), EXT, mode="edge"))
    return np.ascontiguousarray(src.astype(dt)), ext


def units(size, unit):
    return max((size + unit // 2) // unit, 1)


@pytest.mark.parametrize("bd", [8, 10])
def test_filter_search_apply(hip, orc, bd):
    w, h, US = 200, 152, 64          # 3 x 2 restoration units, last ones larger / ragged
    src, ext = make_planes(w, h, bd, 50 + bd)
    st = ext.shape[1]
    off = (EXT * st + EXT) * ext.itemsize
    d_ext, d_src = hip.to_device(ext), hip.to_device(src)
    prm = np.ctypeslib.as_array((C.c_int32 * 4 * 16).in_dll(orc, "orc_sgr_params"))
    ux, uy = units(w, US), units(h, US)
    # oracle: filter the plane in 64x64 processing units like apply_sgr (EbRestorationPick.c:554-581)
    def orc_filter(ep):
        f0 = np.zeros((h, w), np.int32); f1 = np.zeros((h, w), np.int32)
        for y0 in range(0, h, 64):
            for x0 in range(0, w, 64):
                pw, ph = min(64, w - x0), min(64, h - y0)
                p = C.c_void_p(ext.ctypes.data + off + (y0 * st + x0) * ext.itemsize)
                orc.orc_sgr_filter(p, ext.itemsize, pw, ph, st, C.c_void_p(f0.ctypes.data + (y0 * w + x0) * 4), C.c_void_p(f1.ctypes.data + (y0 * w + x0) * 4), w, ep, bd)
        return f0, f1
    e_sums = np.zeros((ux * uy, 16, 5), np.int64)
    d_f0, d_f1 = hip.empty(w * h * 4), hip.empty(w * h * 4)
    for ep in range(16):
        f0, f1 = orc_filter(ep)
        hip.check(hip.L.svt_hip_sgr_filter_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, w, h, ep, d_f0, d_f1, w), "filter")
        if prm[ep][0] > 0: assert np.array_equal(hip.to_host(d_f0, (h, w), np.int32), f0), ("flt0", bd, ep)
        if prm[ep][1] > 0: assert np.array_equal(hip.to_host(d_f1, (h, w), np.int32), f1), ("flt1", bd, ep)
        for uyi in range(uy):
            for uxi in range(ux):
                x0, y0 = uxi * US, uyi * US
                x1 = w if uxi == ux - 1 else x0 + US; y1 = h if uyi == uy - 1 else y0 + US
                s = (C.c_int64 * 5)()
                orc.orc_sgr_proj_sums(C.c_void_p(src.ctypes.data + (y0 * w + x0) * src.itemsize), w, C.c_void_p(ext.ctypes.data + off + (y0 * st + x0) * ext.itemsize), st,
                                      ext.itemsize, x1 - x0, y1 - y0, C.c_void_p(f0.ctypes.data + (y0 * w + x0) * 4), w, C.c_void_p(f1.ctypes.data + (y0 * w + x0) * 4), w, ep, s)
                e_sums[uyi * ux + uxi, ep] = list(s)
    d_sums = hip.to_device(np.zeros_like(e_sums))
    hip.check(hip.L.svt_hip_sgr_search_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, 0xFFFF, d_sums), "search")
    assert np.array_equal(hip.to_host(d_sums, e_sums.shape, np.int64), e_sums)
    # apply: per-unit parameter set + xqd, one unit left unrestored
    rng = np.random.default_rng(3)
    u_ep = rng.integers(0, 16, ux * uy).astype(np.uint8); u_ep[2] = 255
    u_xqd = np.stack([rng.integers(-96, 32, ux * uy), rng.integers(-32, 96, ux * uy)], 1).astype(np.int32)
    exp = ext[EXT:EXT + h, EXT:EXT + w].copy()
    for uyi in range(uy):
        for uxi in range(ux):
            u = uyi * ux + uxi
            if u_ep[u] > 15: continue
            x0, y0 = uxi * US, uyi * US
            x1 = w if uxi == ux - 1 else x0 + US; y1 = h if uyi == uy - 1 else y0 + US
            orc.orc_sgr_apply(C.c_void_p(ext.ctypes.data + off + (y0 * st + x0) * ext.itemsize), ext.itemsize, x1 - x0, y1 - y0, st, int(u_ep[u]),
                              ptr(np.ascontiguousarray(u_xqd[u])), C.c_void_p(exp.ctypes.data + (y0 * w + x0) * exp.itemsize), w, bd)
    d_dst = hip.to_device(ext[EXT:EXT + h, EXT:EXT + w].copy()); d_ep, d_xqd = hip.to_device(u_ep), hip.to_device(u_xqd)
    hip.check(hip.L.svt_hip_sgr_apply_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_dst, w, w, h, US, d_ep, d_xqd), "apply")
    got = hip.to_host(d_dst, (h, w), ext.dtype)
    assert (exp != ext[EXT:EXT + h, EXT:EXT + w]).any()
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:5]
    hip.free(d_ext, d_src, d_f0, d_f1, d_sums, d_dst, d_ep, d_xqd)
