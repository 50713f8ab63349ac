"""GPU parity: whole-plane deblocking (HIP, through the C ABI) vs the oracle (which is pinned to the
reference's 16 edge kernels), 8-bit and 10-bit, luma (4/8/14) and chroma (4/6) edges, several
sharpness values, random and smooth content (mirrors /root/reference/test/DeblockTest.cc:210-306
at frame level)."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr
import dlf_common as dc

pytestmark = pytest.mark.gpu


def content(rng, h, w, bd, smooth):
    if smooth:
        yy, xx = np.mgrid[0:h, 0:w]
        base = 60 + 0.3 * xx + 0.2 * yy + 6 * ((xx // 8 + yy // 8) % 3)
        v = base * (1 << (bd - 8)) + rng.integers(-2, 3, (h, w))
    else:
        v = rng.integers(0, 1 << bd, (h, w))
    return np.clip(v, 0, (1 << bd) - 1)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("smooth", [0, 1])
def test_deblock_planes(hip, orc, bd, smooth):
    w, h = 328, 200   # not multiples of 64; last SB partial
    rng = np.random.default_rng(7 + bd + smooth)
    dt = np.uint8 if bd == 8 else np.uint16
    for seed, varied, sharp in ((15, False, 0), (16, True, 3), (17, True, 6)):
        mi, cols, rows = dc.make_mode_info(w, h, seed=seed, varied=varied)
        for plane, (pw, ph) in enumerate(((w, h), (w // 2, h // 2), (w // 2, h // 2))):
            ev, eh = dc.build_edges(mi, cols, rows, plane, pw, ph)
            pad = 16
            img = np.zeros((ph + 2 * pad + 4, pw + 2 * pad + 4), dt)
            img[:] = content(rng, *img.shape, bd, smooth).astype(dt)
            exp = img.copy()
            off = (pad * img.shape[1] + pad) * img.itemsize
            orc.orc_deblock_plane(C.c_void_p(exp.ctypes.data + off), img.itemsize, img.shape[1], bd, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], sharp)
            d_img, d_ev, d_eh = hip.to_device(img), hip.to_device(ev), hip.to_device(eh)
            hip.check(hip.L.svt_hip_deblock_plane_dev(hip.h, d_img.value + off, img.itemsize, img.shape[1], bd, d_ev, d_eh,
                                                     ev.shape[1], ev.shape[0], sharp), "deblock")
            got = hip.to_host(d_img, img.shape, dt)
            hip.free(d_img, d_ev, d_eh)
            if smooth:
                assert (exp != img).any(), "test content never triggers a filter"
            assert np.array_equal(got, exp), (bd, smooth, plane, seed, np.argwhere(got != exp)[:4])


def test_single_direction_and_4k_property(hip, orc):
    """Vertical-only launch == oracle vertical-only; and on a full 4K luma plane the GPU result is
    idempotent w.r.t. re-running with all levels zero (no edges -> untouched)."""
    w, h = 3840, 2160
    rng = np.random.default_rng(5)
    img = content(rng, h, w, 8, 1).astype(np.uint8)
    mi, cols, rows = dc.make_mode_info(w, h, seed=15)
    ev, eh = dc.build_edges(mi, cols, rows, 0, w, h)
    band = slice(64 * 8, 64 * 10)
    d_img, d_ev, d_eh = hip.to_device(img), hip.to_device(ev), hip.to_device(eh)
    hip.check(hip.L.svt_hip_deblock_plane_dev(hip.h, d_img, 1, w, 8, d_ev, None, ev.shape[1], ev.shape[0], 0))
    got_v = hip.to_host(d_img, img.shape, np.uint8)
    exp = img.copy()
    orc.orc_deblock_plane(ptr(exp), 1, w, 8, ptr(ev), ptr(np.zeros_like(eh)), ev.shape[1], ev.shape[0], 0)
    assert np.array_equal(got_v[band], exp[band])
    hip.check(hip.L.svt_hip_deblock_plane_dev(hip.h, d_img, 1, w, 8, None, d_eh, ev.shape[1], ev.shape[0], 0))
    got = hip.to_host(d_img, img.shape, np.uint8)
    exp2 = img.copy()
    orc.orc_deblock_plane(ptr(exp2), 1, w, 8, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 0)
    assert np.array_equal(got, exp2)
    hip.free(d_img, d_ev, d_eh)


@pytest.mark.parametrize("bd", [8, 10])
def test_plane_sse(hip, orc, bd):
    """picture_sse_calculations' kernel: ragged width / height, offset origins, full-range content."""
    rng = np.random.default_rng(31 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    orc.orc_plane_sse.restype = C.c_uint64
    for (w, h) in ((1, 1), (7, 3), (333, 77), (1920, 1080)):
        a = rng.integers(0, 1 << bd, (h + 5, w + 9)).astype(dt)
        b = rng.integers(0, 1 << bd, (h + 3, w + 20)).astype(dt)
        exp = orc.orc_plane_sse(a.itemsize, C.c_void_p(a.ctypes.data + (2 * a.shape[1] + 3) * a.itemsize), a.shape[1],
                                C.c_void_p(b.ctypes.data + (1 * b.shape[1] + 5) * b.itemsize), b.shape[1], w, h)
        d_a, d_b, d_s = hip.to_device(a), hip.to_device(b), hip.empty(8)
        hip.check(hip.L.svt_hip_plane_sse_dev(hip.h, a.itemsize, C.c_void_p(d_a.value + (2 * a.shape[1] + 3) * a.itemsize), a.shape[1],
                                              C.c_void_p(d_b.value + (1 * b.shape[1] + 5) * b.itemsize), b.shape[1], w, h, d_s), "sse")
        got = int(hip.to_host(d_s, (1,), np.uint64)[0])
        hip.free(d_a, d_b, d_s)
        assert got == exp, (bd, w, h, got, exp)


@pytest.mark.parametrize("bd,mode", [(8, 1), (8, 3), (10, 3)])
def test_filter_level_search(hip, orc, pkg, bd, mode):
    """svt_av1_pick_filter_level's per-plane search on the device vs the oracle's restatement of
    search_filter_level / try_filter_frame: same best level, same error, for luma (both directions) and chroma."""
    w, h = 328, 200
    dt = np.uint8 if bd == 8 else np.uint16
    rng = np.random.default_rng(91 + bd + mode)
    mi, cols, rows = dc.make_mode_info(w, h, seed=21, varied=False)
    for plane, (pw, ph) in enumerate(((w, h), (w // 2, h // 2), (w // 2, h // 2))):
        ev, eh = dc.build_edges(mi, cols, rows, plane, pw, ph)
        src = content(rng, ph, pw, bd, 1).astype(dt)
        # "recon" = source with blocking artefacts on the transform grid + noise, so that some filtering helps
        yy, xx = np.mgrid[0:ph, 0:pw]
        blk = (((xx // 8) * 5 + (yy // 8) * 3) % 7 - 3) * (2 << (bd - 8))
        rec = np.clip(src.astype(np.int32) + blk + rng.integers(-1, 2, src.shape), 0, (1 << bd) - 1).astype(dt)
        for dirn, start in (((0, 8), (1, 30)) if plane == 0 else ((0, 12),)):
            tmp = np.zeros_like(rec)
            best_err = C.c_int64()
            probes = (C.c_int64 * 64)()
            orc.orc_dlf_search_level.restype = C.c_int
            exp_lvl = orc.orc_dlf_search_level(ptr(rec), ptr(tmp), rec.itemsize, pw, bd, pw, ph, ptr(src), pw, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0],
                                               2, plane, dirn, 9, start, mode, 0, C.byref(best_err), probes)
            P = pkg.DlfSearch(plane, dirn, 9, start, mode, 0, 2)
            d_rec, d_tmp, d_src, d_ev, d_eh, d_s = hip.to_device(rec), hip.empty(rec.nbytes), hip.to_device(src), hip.to_device(ev), hip.to_device(eh), hip.empty(8)
            lvl, err = C.c_int(), C.c_int64()
            hip.check(hip.L.svt_hip_dlf_search_level_dev(hip.h, C.byref(P), d_rec, d_tmp, rec.itemsize, pw, bd, pw, ph, d_src, pw, d_ev, d_eh,
                                                        ev.shape[1], ev.shape[0], d_s, C.byref(lvl), C.byref(err)), "dlf search")
            after = hip.to_host(d_rec, rec.shape, dt)
            hip.free(d_rec, d_tmp, d_src, d_ev, d_eh, d_s)
            assert np.array_equal(after, rec), "the unfiltered plane must stay untouched"
            assert (lvl.value, err.value) == (exp_lvl, best_err.value), (bd, mode, plane, dirn, lvl.value, exp_lvl, err.value, best_err.value)
            assert sum(1 for v in probes if v >= 0) >= 2


@pytest.mark.parametrize("bd,mode", [(8, 1), (8, 3), (10, 3)])
def test_filter_level_search_of_a_picture_in_lockstep(hip, orc, pkg, bd, mode):
    """svt_hip_dlf_search_levels_picture_dev: the three planes' searches advanced together (the one or two levels each walk needs next are measured in one round trip) give what the
    oracle's restatement of search_filter_level gives plane by plane: same best level, same error; the unfiltered planes stay untouched."""
    w, h = 328, 200
    dt = np.uint8 if bd == 8 else np.uint16
    rng = np.random.default_rng(191 + bd + mode)
    mi, cols, rows = dc.make_mode_info(w, h, seed=23, varied=False)
    planes = (pkg.DlfSearchPlane * 3)()
    keep, exp, recs = [], [], []
    for plane, (pw, ph) in enumerate(((w, h), (w // 2, h // 2), (w // 2, h // 2))):
        ev, eh = dc.build_edges(mi, cols, rows, plane, pw, ph)
        src = content(rng, ph, pw, bd, 1).astype(dt)
        yy, xx = np.mgrid[0:ph, 0:pw]
        blk = (((xx // 8) * 5 + (yy // 8) * 3) % 7 - 3) * ((2 + plane) << (bd - 8))
        rec = np.clip(src.astype(np.int32) + blk + rng.integers(-1, 2, src.shape), 0, (1 << bd) - 1).astype(dt)
        start = (8, 30, 3)[plane]
        tmp = np.zeros_like(rec)
        best_err = C.c_int64()
        probes = (C.c_int64 * 64)()
        orc.orc_dlf_search_level.restype = C.c_int
        lvl = orc.orc_dlf_search_level(ptr(rec), ptr(tmp), rec.itemsize, pw, bd, pw, ph, ptr(src), pw, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 0, plane, 2, 0, start, mode, 0,
                                       C.byref(best_err), probes)
        exp.append((lvl, best_err.value)); recs.append(rec)
        d = [hip.to_device(rec), hip.empty(rec.nbytes), hip.empty(rec.nbytes), hip.to_device(src), hip.to_device(ev), hip.to_device(eh)]
        keep.append(d)
        planes[plane] = pkg.DlfSearchPlane(pkg.DlfSearch(plane, 2, 0, start, mode, 0, 0), d[0].value, (C.c_void_p * 2)(d[1].value, d[2].value), pw, pw, ph, d[3].value, pw,
                                           d[4].value, d[5].value, ev.shape[1], ev.shape[0])
    d_s = hip.empty(8 * 6)
    best, err = (C.c_int * 3)(), (C.c_int64 * 3)()
    hip.check(hip.L.svt_hip_dlf_search_levels_picture_dev(hip.h, 3, planes, recs[0].itemsize, bd, d_s, best, err), "dlf picture search")
    for plane in range(3):
        assert np.array_equal(hip.to_host(keep[plane][0], recs[plane].shape, dt), recs[plane]), "the unfiltered plane must stay untouched"
        assert (best[plane], err[plane]) == exp[plane], (bd, mode, plane, best[plane], err[plane], exp[plane])
    assert hip.L.svt_hip_dlf_search_levels_picture_dev(hip.h, 4, planes, recs[0].itemsize, bd, d_s, best, err) != 0
    hip.free(d_s, *[x for d in keep for x in d])


@pytest.mark.parametrize("bd", [8, 10])
def test_deblock_frame_all_planes(hip, orc, bd):
    """svt_hip_deblock_frame_dev (all three planes, one launch per direction) == three oracle plane calls; a NULL plane is skipped."""
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    w, h = 328, 200
    rng = np.random.default_rng(70 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    mi, cols, rows = dc.make_mode_info(w, h, seed=21, varied=True)
    planes, exp, edges = [], [], []
    for plane, (pw, ph) in enumerate(((w, h), (w // 2, h // 2), (w // 2, h // 2))):
        ev, eh = dc.build_edges(mi, cols, rows, plane, pw, ph)
        img = np.ascontiguousarray(content(rng, ph, pw + 8, bd, plane == 1).astype(dt))
        e = img.copy()
        orc.orc_deblock_plane(ptr(e), img.itemsize, img.shape[1], bd, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 2)
        planes.append(img); exp.append(e); edges.append((ev, eh))
    for skip in (None, 1):
        d_p = [hip.to_device(p) for p in planes]; d_ev = [hip.to_device(e[0]) for e in edges]; d_eh = [hip.to_device(e[1]) for e in edges]
        pp = [d_p[i].value if i != skip else None for i in range(3)]
        hip.check(hip.L.svt_hip_deblock_frame_dev(hip.h, P3(*pp), planes[0].itemsize, I3(*[p.shape[1] for p in planes]), bd, P3(*[d.value for d in d_ev]),
                                                 P3(*[d.value for d in d_eh]), I3(*[e[0].shape[1] for e in edges]), I3(*[e[0].shape[0] for e in edges]), 2), "deblock frame")
        for i in range(3):
            got = hip.to_host(d_p[i], planes[i].shape, dt)
            assert np.array_equal(got, planes[i] if i == skip else exp[i]), (bd, skip, i)
            assert (exp[i] != planes[i]).any()
        hip.free(*d_p, *d_ev, *d_eh)


@pytest.mark.parametrize("bd,size", [(bd, size) for size in [(328, 200), (1928, 1088), (136, 72), (3840, 2160)] for bd in (8, 10)
                                     if not (size == (3840, 2160) and bd == 10)])   # the 4K case runs once (8-bit)
def test_deblock_frame_fused(hip, orc, bd, size):
    """svt_hip_deblock_frame_fused_dev (both directions of all planes in one out-of-place launch, tiles of 128 x 64 with a 7-sample halo) == the oracle's two passes per
    plane: sizes whose last tile is partial in both directions, varied transform sizes (4 / 8 / 14-tap luma, 4 / 6-tap chroma), three sharpness values; the source planes
    stay untouched, the destination's samples outside the plane extent too, a NULL plane is skipped."""
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    w, h = size
    rng = np.random.default_rng(170 + bd + w)
    dt = np.uint8 if bd == 8 else np.uint16
    for seed, varied, sharp in ((21, True, 0), (22, True, 3), (23, False, 6)):
        mi, cols, rows = dc.make_mode_info(w, h, seed=seed, varied=varied)
        planes, exp, edges = [], [], []
        for plane, (pw, ph) in enumerate(((w, h), (w // 2, h // 2), (w // 2, h // 2))):
            ev, eh = dc.build_edges(mi, cols, rows, plane, pw, ph)
            img = np.ascontiguousarray(content(rng, ph + 3, pw + 8, bd, plane != 2).astype(dt))   # stride > width, rows below the plane
            e = img.copy()
            orc.orc_deblock_plane(ptr(e), img.itemsize, img.shape[1], bd, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], sharp)
            planes.append(img); exp.append(e); edges.append((ev, eh))
        for skip in ((None, 2) if seed == 21 else (None,)):
            d_p = [hip.to_device(p) for p in planes]
            fill = [np.full_like(p, 37) for p in planes]
            d_o = [hip.to_device(f) for f in fill]
            d_ev = [hip.to_device(e[0]) for e in edges]; d_eh = [hip.to_device(e[1]) for e in edges]
            pp = [d_p[i].value if i != skip else None for i in range(3)]
            dims = ((w, h), (w // 2, h // 2), (w // 2, h // 2))
            hip.check(hip.L.svt_hip_deblock_frame_fused_dev(hip.h, P3(*pp), P3(*[d.value for d in d_o]), planes[0].itemsize, I3(*[p.shape[1] for p in planes]), bd,
                                                           I3(*[d[0] for d in dims]), I3(*[d[1] for d in dims]), P3(*[d.value for d in d_ev]), P3(*[d.value for d in d_eh]),
                                                           I3(*[e[0].shape[1] for e in edges]), I3(*[e[0].shape[0] for e in edges]), sharp), "fused deblock")
            for i in range(3):
                pw, ph = dims[i]
                got = hip.to_host(d_o[i], planes[i].shape, dt)
                src_after = hip.to_host(d_p[i], planes[i].shape, dt)
                assert np.array_equal(src_after, planes[i]), "the source plane must stay untouched"
                if i == skip:
                    assert (got == 37).all()
                    continue
                assert np.array_equal(got[:ph, :pw], exp[i][:ph, :pw]), (bd, size, seed, i, np.argwhere(got[:ph, :pw] != exp[i][:ph, :pw])[:4])
                assert (got[ph:] == 37).all() and (got[:, pw:] == 37).all(), "samples outside the plane extent were written"
                assert i == 2 or (exp[i][:ph, :pw] != planes[i][:ph, :pw]).any()   # plane 2 is noise: nothing may be flat enough to filter
            hip.free(*d_p, *d_o, *d_ev, *d_eh)


def host_edges_crop(L, raw, cols, rows, plane, pw, ph, fw, fh):
    """the expectation: the ORACLE's statement of set_lpf_parameters (pinned to the reference's frame loop), not the product's host builder"""
    return dc.build_edges(raw, cols, rows, plane, pw, ph, fw, fh)


@pytest.mark.parametrize("w,h,pad,sb", [(200, 136, (0, 0), 64), (264, 136, (2, 4), 128), (1920, 1080, (0, 0), 64)])
def test_edge_planes_of_random_partitions_on_the_device(hip, w, h, pad, sb):
    """the device builder on the grids of the E2 pin: random AV1 partitions (every block size incl. 4:1 and 128-wide blocks), transform depths, intra / inter / skip, levels
    from per-reference and per-mode deltas -- against the oracle (tests/dlf_common.py: oracle_edges), which test_deblocking_edges_of_a_frame pins to the reference's loop"""
    L = hip.L
    rng = np.random.default_rng(w + h)
    hh = (h + 7) // 8 * 8
    f = dc.make_reference_mode_info(rng, w, hh, sb)
    lf = [int(rng.integers(1, 64)), int(rng.integers(1, 64)), int(rng.integers(1, 64)), int(rng.integers(1, 64)), 3, 1] + [int(v) for v in rng.integers(-20, 21, 10)]
    summ, edges, _ = dc.oracle_edges(dc.oracle(), f, w, hh, lf, pad[0], pad[1], sb)
    cols, rows = w // 4, hh // 4
    pw = [w, w // 2, w // 2]; ph = [hh, hh // 2, hh // 2]
    fw = [L.svt_hip_dlf_filtered_units(w, pad[0], sb, int(p > 0)) for p in range(3)]
    fh = [L.svt_hip_dlf_filtered_units(hh, pad[1], sb, int(p > 0)) for p in range(3)]
    d_mi = hip.to_device(summ)
    I3, P3 = C.c_int * 3, C.c_void_p * 3
    outs = [[hip.empty(2 * ((pw[p] + 3) // 4) * ((ph[p] + 3) // 4)) for p in range(3)] for _ in range(2)]
    hip.check(L.svt_hip_dlf_build_edges_picture_dev(hip.h, d_mi, cols, rows, 1, 1, I3(*pw), I3(*ph), I3(*fw), I3(*fh), None, P3(*[o.value for o in outs[0]]), P3(*[o.value for o in outs[1]])), "edges")
    for p in range(3):
        gv = hip.to_host(outs[0][p], edges[p][0].shape, np.uint16); gh = hip.to_host(outs[1][p], edges[p][1].shape, np.uint16)
        assert np.array_equal(gv, edges[p][0]) and np.array_equal(gh, edges[p][1]), (p, np.argwhere(gv != edges[p][0])[:4], np.argwhere(gh != edges[p][1])[:4])
        assert (gv != 0).any() and (gh != 0).any()
    hip.free(d_mi, *[o for pair in outs for o in pair])


@pytest.mark.parametrize("w,h,pad", [(328, 200, (0, 0)), (192, 128, (6, 2)), (3840, 2160, (0, 0))])
def test_edge_planes_built_on_the_device(hip, w, h, pad):
    """svt_hip_dlf_build_edges_picture_dev == the oracle's statement of set_lpf_parameters (oracle/dlf_oracle.c, pinned to svt_av1_loop_filter_frame by
    tests/test_oracle_vs_ref.py::test_deblocking_edges_of_a_frame), for the records' own levels (varied per block, zeros included), for frame-uniform stand-in levels — a
    zero among them, and one plane left out — and for a padded picture's filtered extent (the extent itself: svt_hip_dlf_filtered_units == orc_dlf_filtered_units)."""
    L = hip.L
    mi, cols, rows = dc.make_mode_info(w, h, seed=31 + w, varied=True)
    pw = [w, w // 2, w // 2]; ph = [h, h // 2, h // 2]
    fw = [L.svt_hip_dlf_filtered_units(w, pad[0], 64, int(p > 0)) for p in range(3)]
    fh = [L.svt_hip_dlf_filtered_units(h, pad[1], 64, int(p > 0)) for p in range(3)]
    assert min(fw + fh) >= 0
    assert fw == [dc.oracle().orc_dlf_filtered_units(w, pad[0], 64, int(p > 0)) for p in range(3)] and fh == [dc.oracle().orc_dlf_filtered_units(h, pad[1], 64, int(p > 0)) for p in range(3)]
    rec = C.sizeof(mi) // (cols * rows)
    assert rec == 13
    raw = np.frombuffer(mi, np.uint8).reshape(-1, rec).copy()
    d_mi = hip.to_device(raw)
    I3, P3 = C.c_int * 3, C.c_void_p * 3
    for levels, mask in ((None, 7), (((23, 9), (0, 0), (17, 17)), 7), (((0, 31), (5, 5), (63, 63)), 5)):
        outs = [[hip.empty(2 * ((pw[p] + 3) // 4) * ((ph[p] + 3) // 4)) if mask & (1 << p) else None for p in range(3)] for _ in range(2)]
        lv = (C.c_int * 6)(*[levels[p][d] for p in range(3) for d in range(2)]) if levels else None
        hip.check(L.svt_hip_dlf_build_edges_picture_dev(hip.h, d_mi, cols, rows, 1, 1, I3(*pw), I3(*ph), I3(*fw), I3(*fh), C.cast(lv, C.c_void_p) if lv else None,
                                                        P3(*[o.value if o else None for o in outs[0]]), P3(*[o.value if o else None for o in outs[1]])), "edges")
        ref_raw = raw.copy()
        if levels:
            ref_raw[:, 7:13] = np.array(levels, np.uint8).reshape(6)   # level[3][2] follows the seven geometry bytes
        for p in range(3):
            if not mask & (1 << p):
                continue
            ev, eh = host_edges_crop(L, ref_raw, cols, rows, p, pw[p], ph[p], fw[p], fh[p])
            gv = hip.to_host(outs[0][p], ev.shape, np.uint16); gh = hip.to_host(outs[1][p], eh.shape, np.uint16)
            if not levels or levels[p][0]:
                assert (ev != 0).any() and (eh != 0).any()
            assert np.array_equal(gv, ev) and np.array_equal(gh, eh), (levels, p, np.argwhere(gv != ev)[:4], np.argwhere(gh != eh)[:4])
        hip.free(*[o for pair in outs for o in pair if o])
    hip.free(d_mi)
