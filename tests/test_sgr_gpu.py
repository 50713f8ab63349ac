"""GPU parity: self-guided restoration (HIP through the C ABI) vs the oracle (pinned to
svt_av1_selfguided_restoration_c / svt_apply_selfguided_restoration_c / svt_get_proj_subspace_c):
filter planes for all 16 parameter sets, the per-unit projection sums of the search, and the apply
pass; 8- and 10-bit.  Mirrors /root/reference/test/selfguided_filter_test.cc:248-562."""
import ctypes as C
import os

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
    d_f0, d_f1 = hip.empty(w * h * 4), hip.empty(w * h * 4)
    for ep in range(16):
        f0, f1 = orc_filter(ep)
        hip.check(hip.L.svt_hip_sgr_filter_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, w, h, ep, d_f0, d_f1, w), "filter")
        if prm[ep][0] > 0: assert np.array_equal(hip.to_host(d_f0, (h, w), np.int32), f0), ("flt0", bd, ep)
        if prm[ep][1] > 0: assert np.array_equal(hip.to_host(d_f1, (h, w), np.int32), f1), ("flt1", bd, ep)
    hip.free(d_ext, d_src, d_f0, d_f1)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("ss", [0, 1])
def test_search_and_stripe_apply(hip, orc, bd, ss):
    """Frame level: projection sums per restoration unit and the stripe-aware apply vs the oracle functions
    that tests/test_oracle_vs_ref.py pins to av1_foreach_rest_unit_in_frame / svt_av1_loop_restoration_filter_unit:
    unit rows offset by 8 >> ss_y, stripes of 64 >> ss_y rows seeing the deblocked picture across their boundaries."""
    for (w, h, US) in ((200, 152, 64), (328, 264, 128)):
        src, ext = make_planes(w, h, bd, 50 + bd + ss)
        st = ext.shape[1]
        off = (EXT * st + EXT) * ext.itemsize
        rng = np.random.default_rng(3 + ss)
        dbl = np.clip(ext[EXT:EXT + h, EXT:EXT + w].astype(np.int32) + rng.integers(-9, 10, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(ext.dtype)
        ux, uy = units(w, US), units(h, US)
        d_ext, d_src, d_dbl = hip.to_device(ext), hip.to_device(src), hip.to_device(dbl)
        # --- search sums, all 16 sets and a sparse mask (sets 11-13 alias 2/5/8 inside the kernel)
        for mask in (0xFFFF, 0x3801, 0x0124):
            e_sums = np.zeros((ux * uy, 16, 5), np.int64)
            orc.orc_sgr_search_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(src), w, w, h, ss, ss, US, bd, mask, ptr(e_sums))
            d_sums = hip.to_device(np.zeros_like(e_sums))
            hip.check(hip.L.svt_hip_sgr_search_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, ss, mask, d_sums), "search")
            got = hip.to_host(d_sums, e_sums.shape, np.int64)
            hip.free(d_sums)
            assert np.array_equal(got, e_sums), (bd, ss, w, h, US, hex(mask), np.argwhere(got != e_sums)[:5])
        # --- apply: per-unit parameter set + xqd, one unit RESTORE_NONE; with and without stripe boundaries
        u_ep = rng.integers(0, 16, ux * uy).astype(np.uint8); u_ep[2] = 255
        u_xqd = np.stack([rng.integers(-96, 32, ux * uy), rng.integers(-32, 96, ux * uy)], 1).astype(np.int32)
        d_ep, d_xqd = hip.to_device(u_ep), hip.to_device(u_xqd)
        work = ext.copy()
        exp = np.zeros((h, w), ext.dtype)
        orc.orc_sgr_apply_plane(ptr(dbl), w, C.c_void_p(work.ctypes.data + off), st, ext.itemsize, w, h, ss, ss, US, bd, ptr(u_ep), ptr(u_xqd), ptr(exp), w)
        assert np.array_equal(work, ext)
        d_dst = hip.to_device(np.zeros((h, w), ext.dtype))
        hip.check(hip.L.svt_hip_sgr_apply_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_dst, w, w, h, US, ss, d_dbl, w, d_ep, d_xqd), "apply")
        got = hip.to_host(d_dst, (h, w), ext.dtype)
        assert (exp != ext[EXT:EXT + h, EXT:EXT + w]).any()
        assert np.array_equal(got, exp), (bd, ss, w, h, US, np.argwhere(got != exp)[:5])
        # without the deblocked plane every unit is filtered from the extended CDEF picture as is (the search's view of the filter)
        exp2 = np.zeros((h, w), ext.dtype)
        lim = np.zeros((ux * uy, 4), np.int32)
        orc.orc_rest_unit_limits(w, h, ss, US, ptr(lim))
        for u, (x0, x1, y0, y1) in enumerate(lim):
            if u_ep[u] > 15:
                exp2[y0:y1, x0:x1] = ext[EXT + y0:EXT + y1, EXT + x0:EXT + x1]
                continue
            orc.orc_sgr_apply(C.c_void_p(ext.ctypes.data + off + (int(y0) * st + int(x0)) * ext.itemsize), ext.itemsize, int(x1 - x0), int(y1 - y0), st, int(u_ep[u]),
                              ptr(np.ascontiguousarray(u_xqd[u])), C.c_void_p(exp2.ctypes.data + (int(y0) * w + int(x0)) * exp2.itemsize), w, bd)
        hip.check(hip.L.svt_hip_sgr_apply_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_dst, w, w, h, US, ss, None, 0, d_ep, d_xqd), "apply")
        got2 = hip.to_host(d_dst, (h, w), ext.dtype)
        assert np.array_equal(got2, exp2), (bd, ss, "no stripes", np.argwhere(got2 != exp2)[:5])
        assert (exp2 != exp).any(), "stripe boundaries must matter on this content"
        hip.free(d_ext, d_src, d_dbl, d_dst, d_ep, d_xqd)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("ss", [0, 1])
def test_mixed_wiener_sgr_apply(hip, orc, bd, ss):
    """svt_hip_lr_apply_plane_dev: a plane whose units are a mix of RESTORE_NONE / RESTORE_WIENER / RESTORE_SGRPROJ, stripe boundaries
    from the deblocked plane, vs orc_lr_apply_plane (pinned to svt_av1_loop_restoration_filter_unit incl. wiener_filter_stripe)."""
    for (w, h, US) in ((200, 152, 64), (328, 264, 128)):
        src, ext = make_planes(w, h, bd, 90 + bd + ss)
        st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
        rng = np.random.default_rng(13 + ss)
        dbl = np.clip(ext[EXT:EXT + h, EXT:EXT + w].astype(np.int32) + rng.integers(-9, 10, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(ext.dtype)
        nu = units(w, US) * units(h, US)
        u_ep = rng.integers(0, 16, nu).astype(np.uint8); u_ep[::3] = 254; u_ep[1] = 255
        u_xqd = np.stack([rng.integers(-96, 32, nu), rng.integers(-32, 96, nu)], 1).astype(np.int32)
        u_wn = np.zeros((nu, 2, 8), np.int16)
        for u in range(nu):
            for d in range(2):
                t = [int(rng.integers(-5, 11)), int(rng.integers(-23, 9)), int(rng.integers(-17, 47))]
                if u == 0: t = [10, 8, 46] if d else [-5, -23, -17]          # extreme taps
                if ss: t[0] = 0
                u_wn[u, d, :7] = [t[0], t[1], t[2], -2 * sum(t), t[2], t[1], t[0]]
        work = ext.copy(); exp = np.zeros((h, w), ext.dtype)
        orc.orc_lr_apply_plane(ptr(dbl), w, C.c_void_p(work.ctypes.data + off), st, ext.itemsize, w, h, ss, ss, US, bd, ptr(u_ep), ptr(u_xqd), ptr(u_wn), ptr(exp), w)
        d_ext, d_dbl, d_ep, d_xqd, d_wn, d_dst = hip.to_device(ext), hip.to_device(dbl), hip.to_device(u_ep), hip.to_device(u_xqd), hip.to_device(u_wn), hip.to_device(np.zeros_like(exp))
        hip.check(hip.L.svt_hip_lr_apply_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_dst, w, w, h, US, ss, d_dbl, w, d_ep, d_xqd, d_wn), "lr apply")
        got = hip.to_host(d_dst, (h, w), ext.dtype)
        assert np.array_equal(got, exp), (bd, ss, w, h, US, np.argwhere(got != exp)[:5])
        # svt_hip_lr_try_unit_dev = try_restoration_unit_seg: ONE unit filtered (only its tiles are launched: the rest of the destination keeps the
        # marker) and its SSE against the source, for every unit of the plane incl. the over-sized last row / column
        d_src2, d_sse = hip.to_device(src), hip.to_device(np.zeros(1, np.uint64))
        voff = 8 >> ss; ux, uy = units(w, US), units(h, US)
        for u in range(nu):
            d_one = hip.to_device(np.full_like(exp, 7))
            hip.check(hip.L.svt_hip_lr_try_unit_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_one, w, w, h, US, ss, d_dbl, w, d_ep, d_xqd, d_wn, d_src2, w, u, d_sse), "try unit")
            one = hip.to_host(d_one, (h, w), ext.dtype); sse = int(hip.to_host(d_sse, (1,), np.uint64)[0])
            uj, ui = u % ux, u // ux
            x0 = uj * US; x1 = w if uj == ux - 1 else x0 + US
            y0 = ui * US; y1 = h if ui == uy - 1 else y0 + US
            v0 = max(y0 - voff, 0); v1 = y1 - voff if y1 < h else y1
            ref_rect = exp[v0:v1, x0:x1].astype(np.int64); src_rect = src[v0:v1, x0:x1].astype(np.int64)
            assert np.array_equal(one[v0:v1, x0:x1], exp[v0:v1, x0:x1]), ("try unit pixels", bd, ss, US, u)
            mask = np.ones((h, w), bool); mask[v0:v1, x0:x1] = False
            assert (one[mask] == 7).all(), ("try unit touched samples outside its unit", bd, ss, US, u)
            assert sse == int(((ref_rect - src_rect) ** 2).sum()), ("try unit sse", bd, ss, US, u)
            hip.free(d_one)
        hip.free(d_ext, d_dbl, d_ep, d_xqd, d_wn, d_dst, d_src2, d_sse)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("ss", [0, 1])
def test_wiener_walk_units(hip, bd, ss):
    """svt_hip_wiener_walk_units_dev (finer_tile_search_wiener_seg of every unit on the device: the coordinate descent AND all of its probes in one launch) against the
    CPU test double's restatement of the same walk on the oracle's restoration filter (oracle/hip_mock.c): the same refined taps, the same error and the same number
    of probes for every unit; inactive units untouched; windows 7 / 5 / 3; unit sizes 64 / 128 with over-sized last rows and columns; starting filters from identity to
    the corners of the tap ranges."""
    import shard_common as sc
    if not os.path.exists(sc.MOCK_LIB):
        pytest.skip("oracle/_ref/mock/libsvtav1_hip.so not built")
    M = C.CDLL(sc.MOCK_LIB)
    sig = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    M.svt_hip_wiener_walk_units_dev.argtypes = sig
    for (w, h, US, win) in ((200, 152, 64, 7), (328, 264, 128, 7), (200, 152, 64, 5), (136, 72, 64, 3)):
        if ss and win == 7: win = 5     # chroma planes search the 5-tap window at most (search_wiener_seg :1352-1358)
        src, ext = make_planes(w, h, bd, 190 + bd + ss + US)
        st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
        rng = np.random.default_rng(31 + ss + win)
        dbl = np.clip(ext[EXT:EXT + h, EXT:EXT + w].astype(np.int32) + rng.integers(-9, 10, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(ext.dtype)
        nu = units(w, US) * units(h, US)
        act = (rng.random(nu) < 0.8).astype(np.uint8); act[0] = 1
        o = (7 - win) >> 1
        wn = np.zeros((nu, 2, 8), np.int16)
        for u in range(nu):
            for d in range(2):
                t = [int(rng.integers(-5, 11)), int(rng.integers(-23, 9)), int(rng.integers(-17, 47))]
                if u % 4 == 0: t = [0, 0, 0]                                 # identity
                if u == 1: t = [10, 8, 46] if d else [-5, -23, -17]          # the corners of the ranges
                for k in range(o): t[k] = 0
                wn[u, d, :7] = [t[0], t[1], t[2], -2 * sum(t), t[2], t[1], t[0]]
        # expected: the CPU test double (host memory is its "device")
        e_wn = wn.copy(); e_err = np.zeros(nu, np.int64); e_pr = np.zeros(nu, np.uint32)
        work = ext.copy()
        assert M.svt_hip_wiener_walk_units_dev(None, ext.itemsize, bd, C.c_void_p(work.ctypes.data + off), st, w, h, US, ss, ptr(dbl), w, ptr(src), w, ptr(e_wn), ptr(act), win, ptr(e_err), ptr(e_pr)) == 0
        d_ext, d_dbl, d_src, d_wn, d_act, d_err, d_pr = hip.to_device(ext), hip.to_device(dbl), hip.to_device(src), hip.to_device(wn), hip.to_device(act), hip.to_device(np.full(nu, -1, np.int64)), hip.to_device(np.zeros(nu, np.uint32))
        hip.check(hip.L.svt_hip_wiener_walk_units_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, w, h, US, ss, d_dbl, w, d_src, w, d_wn, d_act, win, d_err, d_pr), "wiener walk")
        g_wn, g_err, g_pr = hip.to_host(d_wn, wn.shape, np.int16), hip.to_host(d_err, (nu,), np.int64), hip.to_host(d_pr, (nu,), np.uint32)
        hip.free(d_ext, d_dbl, d_src, d_wn, d_act, d_err, d_pr)
        on = act.astype(bool)
        assert np.array_equal(g_wn, e_wn), (bd, ss, w, h, US, win, np.argwhere(g_wn != e_wn)[:4])
        assert np.array_equal(g_err[on], e_err[on]) and (g_err[~on] == -1).all(), (bd, ss, US, win)
        assert np.array_equal(g_pr[on], e_pr[on]) and e_pr[on].min() >= 7, (bd, ss, US, win, g_pr, e_pr)
        assert (e_wn[on] != wn[on]).any(), "no walk moved a tap: the content does not exercise the search"


@pytest.mark.parametrize("bd", [8, 10])
def test_wiener_walk_units_picture(hip, pkg, bd):
    """svt_hip_wiener_walk_units_picture_dev: the walks of three planes (luma 7-tap, two chroma planes 5-tap, different sizes and unit counts) in one launch == the three
    per-plane launches (which test_wiener_walk_units pins to the reference's walk); a plane count outside 1..3 is refused."""
    rng = np.random.default_rng(77 + bd)
    planes, keep, exp = [], [], []
    for i, (w, h, US, ss, win) in enumerate(((328, 264, 128, 0, 7), (168, 136, 64, 1, 5), (200, 96, 64, 1, 5))):
        src, ext = make_planes(w, h, bd, 500 + bd + i)
        st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
        dbl = np.clip(ext[EXT:EXT + h, EXT:EXT + w].astype(np.int32) + rng.integers(-9, 10, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(ext.dtype)
        nu = units(w, US) * units(h, US)
        act = (rng.random(nu) < 0.8).astype(np.uint8); act[0] = 1
        o = (7 - win) >> 1
        wn = np.zeros((nu, 2, 8), np.int16)
        for u in range(nu):
            for d in range(2):
                t = [int(rng.integers(-5, 11)), int(rng.integers(-23, 9)), int(rng.integers(-17, 47))]
                for k in range(o): t[k] = 0
                wn[u, d, :7] = [t[0], t[1], t[2], -2 * sum(t), t[2], t[1], t[0]]
        d_ext, d_dbl, d_src, d_act = hip.to_device(ext), hip.to_device(dbl), hip.to_device(src), hip.to_device(act)
        # expected: the per-plane entry point
        d_wn, d_err, d_pr = hip.to_device(wn), hip.to_device(np.full(nu, -1, np.int64)), hip.to_device(np.zeros(nu, np.uint32))
        hip.check(hip.L.svt_hip_wiener_walk_units_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, w, h, US, ss, d_dbl, w, d_src, w, d_wn, d_act, win, d_err, d_pr), "wiener walk")
        exp.append((hip.to_host(d_wn, wn.shape, np.int16), hip.to_host(d_err, (nu,), np.int64), hip.to_host(d_pr, (nu,), np.uint32)))
        hip.free(d_wn, d_err, d_pr)
        d_wn, d_err, d_pr = hip.to_device(wn), hip.to_device(np.full(nu, -1, np.int64)), hip.to_device(np.zeros(nu, np.uint32))
        planes.append(pkg.WienerWalkPlane(d_ext.value + off, st, w, h, US, ss, d_dbl.value, w, d_src.value, w, d_wn.value, d_act.value, win, d_err.value, d_pr.value))
        keep.append((d_ext, d_dbl, d_src, d_act, d_wn, d_err, d_pr, wn.shape, nu))
    arr = (pkg.WienerWalkPlane * 3)(*planes)
    hip.check(hip.L.svt_hip_wiener_walk_units_picture_dev(hip.h, 1 if bd == 8 else 2, bd, 3, arr), "wiener walk (picture)")
    for (d_ext, d_dbl, d_src, d_act, d_wn, d_err, d_pr, shape, nu), (e_wn, e_err, e_pr) in zip(keep, exp):
        assert np.array_equal(hip.to_host(d_wn, shape, np.int16), e_wn) and np.array_equal(hip.to_host(d_err, (nu,), np.int64), e_err) and np.array_equal(hip.to_host(d_pr, (nu,), np.uint32), e_pr)
    assert hip.L.svt_hip_wiener_walk_units_picture_dev(hip.h, 1 if bd == 8 else 2, bd, 4, arr) != 0 and hip.L.svt_hip_wiener_walk_units_picture_dev(hip.h, 1 if bd == 8 else 2, bd, 0, arr) != 0
    for k in keep: hip.free(*k[:7])


@pytest.mark.parametrize("bd", [8, 10])
def test_search_extreme_content(hip, orc, bd):
    """Bound proofs of the on-chip search (packed A'/B' fields, 24-bit multiplies, int32 partial sums): binary 0 / max content in flat
    areas, single-pixel and 2x2 checkerboards, random binary noise and isolated spikes, against a source that is the complement."""
    w, h, US = 264, 200, 64
    mx = (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    rng = np.random.default_rng(77 + bd)
    yy, xx = np.mgrid[0:h, 0:w]
    dgd = np.zeros((h, w), np.int64)
    dgd[:, :66] = mx * ((xx[:, :66] + yy[:, :66]) & 1)                       # 1-px checkerboard
    dgd[:, 66:132] = mx * (((xx[:, 66:132] >> 1) + (yy[:, 66:132] >> 1)) & 1)  # 2x2 checkerboard
    dgd[:, 132:198] = mx * rng.integers(0, 2, (h, 66))                       # binary noise
    dgd[:, 198:] = mx; dgd[::5, 198::7] = 0                                  # flat max with isolated zero spikes
    dgd[100:, 198:] = 0; dgd[100::6, 200::5] = mx                            # flat zero with isolated max spikes
    for comp in (True, False):
        src = (mx - dgd) if comp else np.full((h, w), mx, np.int64)
        ext = np.ascontiguousarray(np.pad(dgd.astype(dt), EXT, mode="edge")); st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
        srcp = np.ascontiguousarray(src.astype(dt))
        ux, uy = units(w, US), units(h, US)
        e_sums = np.zeros((ux * uy, 16, 5), np.int64)
        orc.orc_sgr_search_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(srcp), w, w, h, 0, 0, US, bd, 0xFFFF, ptr(e_sums))
        d_ext, d_src, d_sums = hip.to_device(ext), hip.to_device(srcp), hip.to_device(np.zeros_like(e_sums))
        hip.check(hip.L.svt_hip_sgr_search_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, 0, 0xFFFF, d_sums), "search")
        got = hip.to_host(d_sums, e_sums.shape, np.int64)
        hip.free(d_ext, d_src, d_sums)
        if bd == 10 and comp: assert np.abs(e_sums).max() > (1 << 31), "content must push the sums past 32 bits"
        assert np.array_equal(got, e_sums), (bd, comp, np.argwhere(got != e_sums)[:5])


def _smooth_noisy(w, h, bd, seed, sigma):
    """Source with textured regions; degraded picture = coarse quantisation of it (coding-like artefacts, strength per 64x64 region) plus noise
    in some regions: different parameter sets win and the projections land inside as well as on the clamps of the tap range."""
    rng = np.random.default_rng(seed)
    dt = np.uint8 if bd == 8 else np.uint16
    sc = 1 << (bd - 8)
    yy, xx = np.mgrid[0:h, 0:w]
    clean = (100 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 25 * (((xx + yy) // 11) % 2) + 0.15 * xx) * sc
    region = (xx // 64 + 2 * (yy // 64)) % 4
    src = np.clip(clean + rng.normal(0, 1, (h, w)) * 6 * sc * (region == 1), 0, (1 << bd) - 1)
    q = np.array([2, 6, 12, 24])[region] * sc
    dgd = np.clip((src // q) * q + q // 2 + rng.normal(0, 1, (h, w)) * sigma * sc * (region == 3), 0, (1 << bd) - 1)
    return np.ascontiguousarray(src.astype(dt)), np.ascontiguousarray(np.pad(dgd.astype(dt), EXT, mode="edge"))


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("ss", [0, 1])
def test_proj_error_candidates(hip, orc, bd, ss):
    """svt_hip_sgr_proj_error_plane_dev vs get_pixel_proj_error of the oracle (pinned to svt_av1_{lowbd,highbd}_pixel_proj_error): random and
    extreme xqd pairs per (unit, set), ragged units, all 16 sets and a sparse mask, ncand 1 / 5 / 12."""
    w, h, US = 200, 152, 64
    src, ext = _smooth_noisy(w, h, bd, 300 + bd + ss, 4)
    st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
    ux, uy = units(w, US), units(h, US); nu = ux * uy
    lim = np.zeros((nu, 4), np.int32); orc.orc_rest_unit_limits(w, h, ss, US, ptr(lim))
    d_ext, d_src = hip.to_device(ext), hip.to_device(src)
    rng = np.random.default_rng(5 + ss)
    orc.orc_sgr_proj_error.restype = C.c_int64
    pu = 64 >> ss
    for mask, nc in ((0xFFFF, 5), (0x4401, 24), (0x8020, 1)):
        xqd = np.stack([rng.integers(-96, 32, (nu, 16, nc)), rng.integers(-32, 96, (nu, 16, nc))], -1).astype(np.int32)
        xqd[:, :, 0] = (-96, -32); xqd[0, :, nc - 1] = (31, 95)
        d_xqd = hip.to_device(np.ascontiguousarray(xqd)); d_err = hip.to_device(np.full((nu, 16, nc), -1, np.int64))
        hip.check(hip.L.svt_hip_sgr_proj_error_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, ss, mask, nc, d_xqd, d_err), "proj error")
        got = hip.to_host(d_err, (nu, 16, nc), np.int64)
        hip.free(d_xqd, d_err)
        for u in range(nu):
            x0, x1, y0, y1 = [int(v) for v in lim[u]]; uw, uh = x1 - x0, y1 - y0
            fs = ((uw + 7) & ~7) + 8
            for ep in range(16):
                if not (mask >> ep) & 1:
                    assert not got[u, ep].any()
                    continue
                f0 = np.zeros((uh, fs), np.int32); f1 = np.zeros((uh, fs), np.int32)
                for i in range(0, uh, pu):
                    for j in range(0, uw, pu):
                        orc.orc_sgr_filter(C.c_void_p(ext.ctypes.data + off + ((y0 + i) * st + x0 + j) * ext.itemsize), ext.itemsize, min(pu, uw - j), min(pu, uh - i), st,
                                           C.c_void_p(f0.ctypes.data + (i * fs + j) * 4), C.c_void_p(f1.ctypes.data + (i * fs + j) * 4), fs, ep, bd)
                for k in range(nc):
                    xq = (C.c_int32 * 2)()
                    orc.orc_sgr_decode_xq(ptr(np.ascontiguousarray(xqd[u, ep, k])), xq, ep)
                    e = orc.orc_sgr_proj_error(C.c_void_p(src.ctypes.data + (y0 * w + x0) * src.itemsize), w, C.c_void_p(ext.ctypes.data + off + (y0 * st + x0) * ext.itemsize), st,
                                               ext.itemsize, uw, uh, ptr(f0), fs, ptr(f1), fs, xq, ep)
                    assert got[u, ep, k] == e, (bd, ss, hex(mask), u, ep, k)
    assert hip.L.svt_hip_sgr_proj_error_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, ss, 1, 25, d_ext, d_ext) != 0
    hip.free(d_ext, d_src)


@pytest.mark.parametrize("bd", [8, 10])
def test_search_units_plane(hip, orc, bd):
    """svt_hip_sgr_search_units_plane (sums -> solve -> encode_xq -> finer search, all on the device) vs the oracle's search_selfguided_restoration
    restatement (pinned to the reference's static functions through oracle/ref_shim_restpick.c): xqd, error and best set of every unit."""
    for (w, h, US, ss, mask, sigma) in ((264, 200, 64, 0, 0xFFFF, 5), (168, 120, 64, 1, 0xFFFF, 9), (328, 264, 128, 0, 0x0F38, 3), (1000, 584, 256, 0, 0x4221, 4)):
        src, ext = _smooth_noisy(w, h, bd, 400 + bd + ss, sigma)
        st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
        nu = units(w, US) * units(h, US)
        e_xqd = np.zeros((nu, 16, 2), np.int32); e_err = np.zeros((nu, 16), np.int64); e_best = np.zeros(nu, np.uint8)
        orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(src), w, w, h, ss, ss, US, bd, mask, ptr(e_xqd), ptr(e_err), ptr(e_best))
        d_ext, d_src = hip.to_device(ext), hip.to_device(src)
        g_xqd = np.zeros_like(e_xqd); g_err = np.zeros_like(e_err); g_best = np.zeros_like(e_best); rounds = C.c_int(0)
        hip.check(hip.L.svt_hip_sgr_search_units_plane(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, ss, mask, ptr(g_xqd), ptr(g_err), ptr(g_best),
                                                       C.byref(rounds)), "search units")
        hip.free(d_ext, d_src)
        assert np.array_equal(g_err, e_err), (bd, w, h, np.argwhere(g_err != e_err)[:5])
        assert np.array_equal(g_xqd, e_xqd) and np.array_equal(g_best, e_best)
        assert rounds.value == 0   # no host rounds: the whole search runs on the device
        if mask == 0xFFFF: assert len(set(int(v) for v in e_best)) > 1, "content should make different sets win"


@pytest.mark.parametrize("bd", [8, 10])
def test_search_units_largest_unit_extreme_content(hip, orc, bd):
    """The walk's error accumulators (sgr_walk.hip, "Accumulator ranges"): ONE restoration unit of the largest size a picture can have (376 x 376 with unit size 256:
    383 is the limit, 1.5 x 256 rounds to two units), binary 0 / max content against its complement, an unrelated binary source and a flat extreme -- the per-sample
    error is as large as content can make it and a data thread sees its maximum number of chunks.  xqd, error and best set of all 16 sets against the oracle."""
    w = h = 376
    US, mx = 256, (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    rng = np.random.default_rng(900 + bd)
    yy, xx = np.mgrid[0:h, 0:w]
    dgd = np.zeros((h, w), np.int64)
    dgd[:, :128] = mx * ((xx[:, :128] + yy[:, :128]) & 1)                          # 1-px checkerboard
    dgd[:, 128:256] = mx * (((xx[:, 128:256] >> 1) + (yy[:, 128:256] >> 2)) & 1)   # 2 x 4 blocks
    dgd[:, 256:] = mx * rng.integers(0, 2, (h, w - 256))                           # binary noise
    assert units(w, US) == 1 and units(h, US) == 1
    for kind in ("complement", "binary", "flat"):
        src = {"complement": mx - dgd, "binary": mx * rng.integers(0, 2, (h, w)), "flat": np.full((h, w), mx, np.int64)}[kind]
        ext = np.ascontiguousarray(np.pad(dgd.astype(dt), EXT, mode="edge")); st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
        srcp = np.ascontiguousarray(src.astype(dt))
        e_xqd = np.zeros((1, 16, 2), np.int32); e_err = np.zeros((1, 16), np.int64); e_best = np.zeros(1, np.uint8)
        orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(srcp), w, w, h, 0, 0, US, bd, 0xFFFF, ptr(e_xqd), ptr(e_err), ptr(e_best))
        d_ext, d_src = hip.to_device(ext), hip.to_device(srcp)
        g_xqd = np.zeros_like(e_xqd); g_err = np.zeros_like(e_err); g_best = np.zeros_like(e_best); rounds = C.c_int(0)
        hip.check(hip.L.svt_hip_sgr_search_units_plane(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, 0, 0xFFFF, ptr(g_xqd), ptr(g_err), ptr(g_best),
                                                       C.byref(rounds)), "search units")
        hip.free(d_ext, d_src)
        if bd == 10 and kind != "flat": assert e_err.max() > (1 << 34), "the unit's error must be far past 32 bits"
        assert np.array_equal(g_err, e_err), (bd, kind, np.argwhere(g_err != e_err)[:5], g_err, e_err)
        assert np.array_equal(g_xqd, e_xqd) and np.array_equal(g_best, e_best), (bd, kind)


def test_search_units_picture(hip, pkg, orc):
    """Three planes in one call (svt_hip_sgr_search_units_picture) = the per-plane results."""
    w, h, bd = 264, 200, 8
    planes, keep, exp = (pkg.SgrSearchPlane * 3)(), [], []
    for p in range(3):
        ss = int(p > 0); pw, ph = w >> ss, h >> ss
        src, ext = _smooth_noisy(pw, ph, bd, 700 + p, 5)
        st = ext.shape[1]; off = (EXT * st + EXT)
        nu = units(pw, 64) * units(ph, 64)
        mask = (0xFFFF, 0x03C0, 0x8001)[p]
        e_xqd = np.zeros((nu, 16, 2), np.int32); e_err = np.zeros((nu, 16), np.int64); e_best = np.zeros(nu, np.uint8)
        orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), 1, st, ptr(src), pw, pw, ph, ss, ss, 64, bd, mask, ptr(e_xqd), ptr(e_err), ptr(e_best))
        d_ext, d_src = hip.to_device(ext), hip.to_device(src)
        g = (np.zeros_like(e_xqd), np.zeros_like(e_err), np.zeros_like(e_best))
        planes[p] = pkg.SgrSearchPlane(d_ext.value + off, st, d_src.value, pw, pw, ph, 64, ss, mask, g[0].ctypes.data, g[1].ctypes.data, g[2].ctypes.data)
        keep += [d_ext, d_src]; exp.append(((e_xqd, e_err, e_best), g))
    rounds = C.c_int(0)
    hip.check(hip.L.svt_hip_sgr_search_units_picture(hip.h, 1, bd, 3, planes, C.byref(rounds)), "picture search")
    for p, (e, g) in enumerate(exp):
        assert np.array_equal(g[1], e[1]) and np.array_equal(g[0], e[0]) and np.array_equal(g[2], e[2]), p
    assert hip.L.svt_hip_sgr_search_units_picture(hip.h, 1, bd, 4, planes, None) != 0
    hip.free(*keep)


@pytest.mark.parametrize("bd", [8, 10])
def test_search_units_plane_dev_chain(hip, orc, bd):
    """The device-output form: results stay in HBM, d_best_ep / d_best_xqd are exactly the per-unit arrays svt_hip_sgr_apply_plane_dev takes, so the
    search -> trial filter chain needs no host round trip.  Checked: every (unit, set) result, the best set, and the plane filtered with the winners."""
    w, h, US, ss, mask = 328, 264, 128, 0, 0xFFFF
    src, ext = _smooth_noisy(w, h, bd, 900 + bd, 6)
    st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
    nu = units(w, US) * units(h, US)
    e_xqd = np.zeros((nu, 16, 2), np.int32); e_err = np.zeros((nu, 16), np.int64); e_best = np.zeros(nu, np.uint8)
    orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(src), w, w, h, ss, ss, US, bd, mask, ptr(e_xqd), ptr(e_err), ptr(e_best))
    L = hip.L
    L.svt_hip_sgr_search_units_scratch_bytes.restype = C.c_size_t
    nbytes = L.svt_hip_sgr_search_units_scratch_bytes(w, h, US)
    assert nbytes > 33 * 2 * w * h
    d_ext, d_src = hip.to_device(ext), hip.to_device(src)
    d_scr = hip.empty(nbytes); d_xqd = hip.empty(nu * 16 * 8); d_err = hip.empty(nu * 16 * 8); d_best = hip.empty(nu); d_bx = hip.empty(nu * 8)
    hip.check(L.svt_hip_sgr_search_units_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, ss, mask, d_xqd, d_err, d_best, d_bx, d_scr, nbytes), "units dev")
    # chained apply with the winners, still on the device (stripe context rows taken from the same plane)
    d_out = hip.empty(h * w * ext.itemsize)
    hip.check(L.svt_hip_sgr_apply_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_out, w, w, h, US, ss, d_ext.value + off, st, d_best, d_bx), "apply")
    g_xqd = hip.to_host(d_xqd, e_xqd.shape, np.int32); g_err = hip.to_host(d_err, e_err.shape, np.int64); g_best = hip.to_host(d_best, e_best.shape, np.uint8)
    g_bx = hip.to_host(d_bx, (nu, 2), np.int32); out = hip.to_host(d_out, (h, w), ext.dtype)
    assert np.array_equal(g_err, e_err) and np.array_equal(g_xqd, e_xqd) and np.array_equal(g_best, e_best)
    assert np.array_equal(g_bx, e_xqd[np.arange(nu), e_best])
    exp = np.zeros((h, w), ext.dtype)
    work = ext.copy(); dbl = ext.copy()
    orc.orc_sgr_apply_plane(C.c_void_p(dbl.ctypes.data + off), st, C.c_void_p(work.ctypes.data + off), st, ext.itemsize, w, h, ss, ss, US, bd, ptr(e_best), ptr(np.ascontiguousarray(g_bx)), ptr(exp), w)
    assert np.array_equal(out, exp)
    # a scratch that is too small is rejected
    assert L.svt_hip_sgr_search_units_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, ss, mask, d_xqd, d_err, d_best, d_bx, d_scr, nbytes - 1) != 0
    hip.free(d_ext, d_src, d_scr, d_xqd, d_err, d_best, d_bx, d_out)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("window", [None, 160, 24])
def test_search_units_histogram_evaluation_and_its_fallback(hip, orc, bd, window):
    """The one-filter sets (10 .. 15) of a unit are evaluated on a histogram over flt - u (sgr_walk.hip "HISTOGRAM EVALUATION"): samples outside its window go to an exact
    list, a unit with more than 256 of them is evaluated sample by sample.  Content cannot push |flt - u| past the window's 3071 (the filter passes high-variance samples
    through), so the other cases narrow the window (SVT_HIP_SGR_WALK_HIST_W, read per launch): 160 leaves a few listed samples, 24 sends part of the units down the
    sample-by-sample path.  Results of all 16 sets against the oracle; the walk's own counter says how many walks took the histogram."""
    w, h, US, ss, mask = 384, 192, 64, 0, 0xFFFF
    mx = (1 << bd) - 1
    src, ext = _smooth_noisy(w, h, bd, 1200 + bd, 4)
    rng = np.random.default_rng(1300 + bd)
    ext[EXT:EXT + h, EXT + w // 2:EXT + w] = (mx * rng.integers(0, 2, (h, w // 2))).astype(ext.dtype)
    ext[:, EXT + w:] = ext[:, EXT + w - 1:EXT + w]; ext[:EXT, :] = ext[EXT:EXT + 1, :]; ext[EXT + h:, :] = ext[EXT + h - 1:EXT + h, :]
    st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
    nu = units(w, US) * units(h, US)
    e_xqd = np.zeros((nu, 16, 2), np.int32); e_err = np.zeros((nu, 16), np.int64); e_best = np.zeros(nu, np.uint8)
    orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(src), w, w, h, ss, ss, US, bd, mask, ptr(e_xqd), ptr(e_err), ptr(e_best))
    L = hip.L
    L.svt_hip_sgr_search_units_scratch_bytes.restype = C.c_size_t
    nbytes = L.svt_hip_sgr_search_units_scratch_bytes(w, h, US)
    d_ext, d_src = hip.to_device(ext), hip.to_device(src)
    d_scr = hip.empty(nbytes); d_xqd = hip.empty(nu * 16 * 8); d_err = hip.empty(nu * 16 * 8); d_best = hip.empty(nu); d_bx = hip.empty(nu * 8)
    if window is not None: os.environ["SVT_HIP_SGR_WALK_HIST_W"] = str(window << (bd - 8))
    try:
        hip.check(L.svt_hip_sgr_search_units_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, ss, mask, d_xqd, d_err, d_best, d_bx, d_scr, nbytes), "units dev")
    finally:
        os.environ.pop("SVT_HIP_SGR_WALK_HIST_W", None)
    g_xqd = hip.to_host(d_xqd, e_xqd.shape, np.int32); g_err = hip.to_host(d_err, e_err.shape, np.int64); g_best = hip.to_host(d_best, e_best.shape, np.uint8)
    stats = hip.to_host(d_scr, (32,), np.uint32)   # the scratch starts with the walk's counters: [2] unfinished walks, [3] walks evaluated on the histogram
    hip.free(d_ext, d_src, d_scr, d_xqd, d_err, d_best, d_bx)
    assert np.array_equal(g_err, e_err), (bd, np.argwhere(g_err != e_err)[:8])
    assert np.array_equal(g_xqd, e_xqd) and np.array_equal(g_best, e_best)
    assert stats[2] == 0
    if window != 24: assert stats[3] == nu * 6, stats[:4]       # every one-filter walk on the histogram (window 160: with listed samples)
    else: assert 0 < stats[3] < nu * 6, stats[:4]               # some of them, the others sample by sample


@pytest.mark.parametrize("lim", [None, 48, 3])
@pytest.mark.parametrize("ss", [0, 1])
def test_search_units_packed_words_and_escape_lists(hip, orc, lim, ss):
    """SVT_HIP_SGR_PACKED=1 (an opt-in experiment, bit depth 8) runs the unit search on PACKED difference words (sgr.hip STORE == 2, sgr_walk_packed_kernel): d0 / d1 as 11-bit fields, dat - src as 10 bits, a sample
    whose |flt - u| does not fit is a zero word in the plane and an exact entry of the (unit, set)'s escape list.  The right half of the picture is binary 0 / 255 (which
    escapes at the real limit of 1024 as well); SVT_HIP_SGR_ESC_LIM (read per launch) narrows the range so that ordinary samples are listed too: 48 lists a few per cent,
    3 nearly everything -- the walk then is the sample-by-sample sum of the lists.  All 16 sets against the oracle, for this form and for the default 6-byte form."""
    bd = 8
    w, h, US, mask = 424, 328, 128, 0xFFFF   # 3 x 3 units, last column 168 wide (not a multiple of 64), last row 72 + 128
    mx = 255
    src, ext = _smooth_noisy(w, h, bd, 1500 + ss, 5)
    rng = np.random.default_rng(1600 + ss)
    ext[EXT:EXT + h, EXT + w // 2:EXT + w] = (mx * rng.integers(0, 2, (h, w - w // 2))).astype(ext.dtype)
    ext[:, EXT + w:] = ext[:, EXT + w - 1:EXT + w]; ext[:EXT, :] = ext[EXT:EXT + 1, :]; ext[EXT + h:, :] = ext[EXT + h - 1:EXT + h, :]
    st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
    nu = units(w, US) * units(h, US)
    e_xqd = np.zeros((nu, 16, 2), np.int32); e_err = np.zeros((nu, 16), np.int64); e_best = np.zeros(nu, np.uint8)
    orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(src), w, w, h, ss, ss, US, bd, mask, ptr(e_xqd), ptr(e_err), ptr(e_best))
    L = hip.L
    L.svt_hip_sgr_search_units_scratch_bytes.restype = C.c_size_t
    os.environ["SVT_HIP_SGR_PACKED"] = "1"   # the size with the experiment's escape lists
    try: nbytes = L.svt_hip_sgr_search_units_scratch_bytes(w, h, US)
    finally: os.environ.pop("SVT_HIP_SGR_PACKED", None)
    assert nbytes > L.svt_hip_sgr_search_units_scratch_bytes(w, h, US)
    d_ext, d_src = hip.to_device(ext), hip.to_device(src)
    d_scr = hip.empty(nbytes); d_xqd = hip.empty(nu * 16 * 8); d_err = hip.empty(nu * 16 * 8); d_best = hip.empty(nu); d_bx = hip.empty(nu * 8)
    listed = {}
    for form in ("packed", "six_byte"):
        if lim is not None: os.environ["SVT_HIP_SGR_ESC_LIM"] = str(lim)
        if form == "packed": os.environ["SVT_HIP_SGR_PACKED"] = "1"
        try:
            hip.check(L.svt_hip_sgr_search_units_plane_dev(hip.h, ext.itemsize, bd, d_ext.value + off, st, d_src, w, w, h, US, ss, mask, d_xqd, d_err, d_best, d_bx, d_scr, nbytes), "units dev")
        finally:
            os.environ.pop("SVT_HIP_SGR_ESC_LIM", None); os.environ.pop("SVT_HIP_SGR_PACKED", None)
        g_xqd = hip.to_host(d_xqd, e_xqd.shape, np.int32); g_err = hip.to_host(d_err, e_err.shape, np.int64); g_best = hip.to_host(d_best, e_best.shape, np.uint8)
        stats = hip.to_host(d_scr, (32,), np.uint32)   # [2] unfinished walks, [4] listed samples the sample-by-sample walks added
        assert np.array_equal(g_err, e_err), (form, lim, np.argwhere(g_err != e_err)[:8])
        assert np.array_equal(g_xqd, e_xqd) and np.array_equal(g_best, e_best), (form, lim)
        assert stats[2] == 0
        listed[form] = int(stats[4])
    hip.free(d_ext, d_src, d_scr, d_xqd, d_err, d_best, d_bx)
    assert listed["six_byte"] == 0
    if lim is not None: assert listed["packed"] > (w * h if lim == 3 else 1000), listed   # lim 3: most samples of most two-filter sets are listed


@pytest.mark.parametrize("bd", [8, 10])
def test_search_units_picture_dev(hip, pkg, orc, bd):
    """svt_hip_sgr_search_units_picture_dev: three planes of different sizes / unit sizes / set masks in one call (one walk launch for the picture:
    grid.z = plane, a plane with fewer units than the widest one leaves workgroups without work) against the oracle's per-plane search."""
    planes = [(328, 264, 128, 0, 0xFFFF), (168, 136, 64, 1, 0x0F3C), (200, 96, 64, 1, 0x8001)]
    L = hip.L
    L.svt_hip_sgr_search_units_scratch_bytes.restype = C.c_size_t
    jobs = (pkg.SgrUnitsPlaneDev * 3)()
    keep, expect = [], []
    for i, (w, h, US, ss, mask) in enumerate(planes):
        src, ext = _smooth_noisy(w, h, bd, 1500 + 10 * i + bd, 5 + i)
        st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
        nu = units(w, US) * units(h, US)
        e_xqd = np.zeros((nu, 16, 2), np.int32); e_err = np.zeros((nu, 16), np.int64); e_best = np.zeros(nu, np.uint8)
        orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(src), w, w, h, ss, ss, US, bd, mask, ptr(e_xqd), ptr(e_err), ptr(e_best))
        nbytes = L.svt_hip_sgr_search_units_scratch_bytes(w, h, US)
        d = dict(ext=hip.to_device(ext), src=hip.to_device(src), scr=hip.empty(nbytes), xqd=hip.to_device(np.zeros_like(e_xqd)), err=hip.to_device(np.zeros_like(e_err)),
                 best=hip.empty(nu), bx=hip.empty(nu * 8))
        keep.append(d); expect.append((nu, mask, e_xqd, e_err, e_best))
        jobs[i] = pkg.SgrUnitsPlaneDev(d["ext"].value + off, st, d["src"].value, w, w, h, US, ss, mask, d["xqd"].value, d["err"].value, d["best"].value, d["bx"].value,
                                       d["scr"].value, nbytes)
    hip.check(L.svt_hip_sgr_search_units_picture_dev(hip.h, 1 if bd == 8 else 2, bd, 3, jobs), "units picture dev")
    for d, (nu, mask, e_xqd, e_err, e_best) in zip(keep, expect):
        g_xqd = hip.to_host(d["xqd"], e_xqd.shape, np.int32); g_err = hip.to_host(d["err"], e_err.shape, np.int64); g_best = hip.to_host(d["best"], e_best.shape, np.uint8)
        on = np.array([(mask >> e) & 1 for e in range(16)], bool)
        assert np.array_equal(g_err[:, on], e_err[:, on]) and np.array_equal(g_xqd[:, on], e_xqd[:, on]) and np.array_equal(g_best, e_best), (bd, nu, hex(mask))
        hip.free(*d.values())


@pytest.mark.parametrize("n_pics", [2, 4])
def test_search_units_several_pictures_one_call(hip, pkg, orc, n_pics):
    """svt_hip_sgr_search_units_picture_dev with the planes of SEVERAL pictures (n_planes = 3 x pictures <= SVT_HIP_SGR_MAX_PLANES = 12): one sums / difference-plane
    launch over every plane's tiles and one walk launch for every (plane, unit, set) -- the north-star's "many concurrent frames in one launch per kernel class" for the
    restoration unit search.  Pictures of different sizes, unit sizes and set masks; every plane against the oracle's per-plane search; a thirteenth plane is refused."""
    bd = 8
    shapes = [(328, 264, 128, 0, 0xFFFF), (168, 136, 64, 1, 0x0F3C), (200, 96, 64, 1, 0x8001), (264, 200, 64, 0, 0x4421), (136, 104, 64, 1, 0xFFFF), (136, 104, 64, 1, 0x03C0)]
    planes = [shapes[(3 * q + p) % len(shapes)] for q in range(n_pics) for p in range(3)]
    L = hip.L
    L.svt_hip_sgr_search_units_scratch_bytes.restype = C.c_size_t
    jobs = (pkg.SgrUnitsPlaneDev * len(planes))()
    keep, expect = [], []
    for i, (w, h, US, ss, mask) in enumerate(planes):
        src, ext = _smooth_noisy(w, h, bd, 2500 + 10 * i, 5 + i % 4)
        st = ext.shape[1]; off = (EXT * st + EXT) * ext.itemsize
        nu = units(w, US) * units(h, US)
        e_xqd = np.zeros((nu, 16, 2), np.int32); e_err = np.zeros((nu, 16), np.int64); e_best = np.zeros(nu, np.uint8)
        orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), ext.itemsize, st, ptr(src), w, w, h, ss, ss, US, bd, mask, ptr(e_xqd), ptr(e_err), ptr(e_best))
        nbytes = L.svt_hip_sgr_search_units_scratch_bytes(w, h, US)
        d = dict(ext=hip.to_device(ext), src=hip.to_device(src), scr=hip.empty(nbytes), xqd=hip.to_device(np.zeros_like(e_xqd)), err=hip.to_device(np.zeros_like(e_err)),
                 best=hip.empty(nu), bx=hip.empty(nu * 8))
        keep.append(d); expect.append((nu, mask, e_xqd, e_err, e_best))
        jobs[i] = pkg.SgrUnitsPlaneDev(d["ext"].value + off, st, d["src"].value, w, w, h, US, ss, mask, d["xqd"].value, d["err"].value, d["best"].value, d["bx"].value,
                                       d["scr"].value, nbytes)
    hip.check(L.svt_hip_sgr_search_units_picture_dev(hip.h, 1, bd, len(planes), jobs), "units of several pictures")
    for d, (nu, mask, e_xqd, e_err, e_best) in zip(keep, expect):
        g_xqd = hip.to_host(d["xqd"], e_xqd.shape, np.int32); g_err = hip.to_host(d["err"], e_err.shape, np.int64); g_best = hip.to_host(d["best"], e_best.shape, np.uint8)
        on = np.array([(mask >> e) & 1 for e in range(16)], bool)
        assert np.array_equal(g_err[:, on], e_err[:, on]) and np.array_equal(g_xqd[:, on], e_xqd[:, on]) and np.array_equal(g_best, e_best), (n_pics, nu, hex(mask))
    assert L.svt_hip_sgr_search_units_picture_dev(hip.h, 1, bd, 13, jobs) != 0
    for d in keep: hip.free(*d.values())
