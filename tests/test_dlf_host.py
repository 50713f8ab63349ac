"""CPU: host-side edge-descriptor builder (restatement of set_lpf_parameters,
/root/reference/Source/Lib/Encoder/Codec/EbDeblockingFilter.c:168-319) — structural properties."""
import ctypes as C
import os

import numpy as np
import pytest

import dlf_common as dc


def test_edge_builder_properties():
    w, h = 200, 136
    mi, cols, rows = dc.make_mode_info(w, h, seed=3)
    for plane, (pw, ph) in enumerate(((w, h), (w // 2, h // 2), (w // 2, h // 2))):
        ev, eh = dc.build_edges(mi, cols, rows, plane, pw, ph)
        assert not ev[:, 0].any() and not eh[0, :].any()          # no filtering on the picture boundary
        lens = set(np.unique(ev & 0xff)) | set(np.unique(eh & 0xff))
        assert lens <= ({0, 4, 8, 14} if plane == 0 else {0, 4, 6})
        assert (ev & 0xff).any() and (eh & 0xff).any()
        # an edge exists only where the unit starts a transform block in that direction
        ss = 0 if plane == 0 else 1
        for uy in range(ev.shape[0]):
            for ux in range(ev.shape[1]):
                m = mi[min(ss | ((4 * uy << ss) >> 2), rows - 1) * cols + min(ss | ((4 * ux << ss) >> 2), cols - 1)]
                tw = m.tx_w_log2 if plane == 0 else m.uv_tx_w_log2
                th = m.tx_h_log2 if plane == 0 else m.uv_tx_h_log2
                if (4 * ux) & ((1 << tw) - 1): assert ev[uy, ux] == 0
                if (4 * uy) & ((1 << th) - 1): assert eh[uy, ux] == 0
                for e in (ev[uy, ux], eh[uy, ux]):
                    if e: assert (e >> 8) in (20, 12)


def test_zero_level_means_no_edges():
    mi, cols, rows = dc.make_mode_info(128, 128, seed=1, levels=(0, 0, 0, 0))
    ev, eh = dc.build_edges(mi, cols, rows, 0, 128, 128)
    assert not ev.any() and not eh.any()


def test_product_host_builder_equals_the_oracle():
    """The product's host builder (svt_hip_dlf_build_edges_crop, svt-av1_amd/csrc/svt_hip_host.cpp -- also what the CPU test double of the library links) against the oracle's
    statement of set_lpf_parameters, which tests/test_oracle_vs_ref.py::test_deblocking_edges_of_a_frame pins to the reference's own frame loop: the synthetic grids of the
    other deblocking tests, and grids of random AV1 partitions with every block size / transform depth / skip / level combination, full and cropped extents."""
    L = dc.pkg.lib()
    orc = dc.oracle()
    rng = np.random.default_rng(77)
    n = 0
    for (w, h, pad, sb) in ((200, 136, (0, 0), 64), (136, 72, (6, 6), 64), (192, 128, (6, 6), 64), (384, 256, (0, 0), 128), (264, 136, (2, 4), 128), (72, 72, (8, 8), 64)):
        for it in range(3):
            f = dc.make_reference_mode_info(rng, w, h, sb, p_skip=(0.2, 0.6, 0.95)[it], p_inter=(0.7, 0.9, 0.3)[it])
            lf = [int(rng.integers(0, 64)) for _ in range(4)] + [2, it % 2] + [int(v) for v in rng.integers(-20, 21, 10)]
            summ, edges, _ = dc.oracle_edges(orc, f, w, h, lf, pad[0], pad[1], sb)
            for plane in range(3):
                ss = int(plane > 0)
                fw, fh = L.svt_hip_dlf_filtered_units(w, pad[0], sb, ss), L.svt_hip_dlf_filtered_units(h, pad[1], sb, ss)
                assert (fw, fh) == (orc.orc_dlf_filtered_units(w, pad[0], sb, ss), orc.orc_dlf_filtered_units(h, pad[1], sb, ss))
                ev, eh = dc.product_host_edges(summ, w // 4, h // 4, plane, w >> ss, h >> ss, fw, fh)
                assert np.array_equal(ev, edges[plane][0]) and np.array_equal(eh, edges[plane][1]), (w, h, it, plane)
                n += int((ev != 0).sum() + (eh != 0).sum())
    assert n > 10000
    for (w, h) in ((200, 136), (328, 200)):
        mi, cols, rows = dc.make_mode_info(w, h, seed=w, varied=True)
        for plane in range(3):
            ss = int(plane > 0)
            a = dc.product_host_edges(mi, cols, rows, plane, w >> ss, h >> ss); b = dc.build_edges(mi, cols, rows, plane, w >> ss, h >> ss)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_filtered_units_sweep():
    """svt_hip_dlf_filtered_units == orc_dlf_filtered_units (the unit ranges of svt_av1_filter_block_plane_vert / _horz) over coded sizes, paddings, both superblock sizes"""
    L = dc.pkg.lib()
    orc = dc.oracle()
    for sb in (64, 128):
        for coded in range(8, 520, 8):
            for pad in range(0, 8):
                for ss in (0, 1):
                    assert L.svt_hip_dlf_filtered_units(coded, pad, sb, ss) == orc.orc_dlf_filtered_units(coded, pad, sb, ss), (coded, pad, sb, ss)


def _reference_level_walk(err, start, mode, only_4x4):
    """search_filter_level (Encoder/Codec/EbDeblockingFilter.c:1026-1187) restated in Python on an error table err[level]; returns (best level, its error, levels measured in order)"""
    measured = []
    ss = {}
    def probe(l):
        if l not in ss:
            ss[l] = err[l]; measured.append(l)
        return ss[l]
    direction = 0
    mid = min(max(start, 0), 63)
    step = 4 if mid < 16 else mid // 4
    best_err = probe(mid); best = mid
    single = mode <= 2
    if single: step = 2
    while step > 0:
        high, low = min(mid + step, 63), max(mid - step, 0)
        bias = (best_err >> (15 - (mid // 8))) * step
        if not only_4x4: bias >>= 1
        if direction <= 0 and low != mid:
            e = probe(low)
            if e < best_err + bias:
                if e < best_err: best_err = e
                best = low
        if direction >= 0 and high != mid:
            e = probe(high)
            if e < best_err - bias:
                if not single: best_err = e
                best = high
        if single: break
        if best == mid:
            step //= 2; direction = 0
        else:
            direction = -1 if best < mid else 1
            mid = best
    return best, ss[best], measured


def test_level_search_plan_equals_the_reference_walk():
    """svt_hip_dlf_search_plan / svt_hip_dlf_search_levels_host (svt-av1_amd/csrc/svt_hip_host.cpp: the walk replayed as a plan over the errors known so far, one or two levels per
    round) == the reference's search_filter_level on random error tables: convex, noisy, flat (ties), monotone; every start level, both modes, both bias rules.  The host logic is
    loaded from the CPU test double, which links the product's own svt_hip_host.cpp."""
    import e2e_common as E
    path = os.path.join(E.MOCK_DIR, "libsvtav1_hip.so")
    if not os.path.exists(path):
        pytest.skip("CPU test double not built")
    L = C.CDLL(path)

    class Search(C.Structure):
        _fields_ = [("plane", C.c_int), ("dir", C.c_int), ("other_level", C.c_int), ("start_level", C.c_int), ("loop_filter_mode", C.c_int), ("tx_mode_only_4x4", C.c_int), ("sharpness", C.c_int)]
    TRY = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_int, C.c_int)
    rng = np.random.default_rng(77)
    two_at_once = 0
    for it in range(400):
        kind = it % 4
        centre = int(rng.integers(0, 64))
        lv = np.arange(64)
        if kind == 0: err = 1000000 + (lv - centre) ** 2 * int(rng.integers(50, 5000))
        elif kind == 1: err = 1000000 + (lv - centre) ** 2 * 300 + rng.integers(0, 40000, 64)
        elif kind == 2: err = np.full(64, 123456) + (rng.random(64) < 0.1) * 7
        else: err = 500000 + lv * int(rng.integers(-3000, 3000)) + 200000
        err = [int(max(v, 0)) for v in err]
        start, mode, only4 = int(rng.integers(-2, 70)), int(rng.choice([1, 2, 3])), int(rng.integers(0, 2))
        exp_level, exp_err, exp_order = _reference_level_walk(err, start, mode, only4)
        q = Search(0, 2, 0, start, mode, only4, 0)
        order = []
        def try_level(user, lv_v, lv_h):
            assert lv_v == lv_h
            order.append(lv_v)
            return err[lv_v]
        best, best_err = C.c_int(), C.c_int64()
        assert L.svt_hip_dlf_search_levels_host(C.byref(q), TRY(try_level), None, C.byref(best), C.byref(best_err)) == 0
        assert (best.value, best_err.value) == (exp_level, exp_err), (it, start, mode, only4)
        assert sorted(order) == sorted(exp_order) and len(order) == len(set(order)), (order, exp_order)   # the same levels, each measured once
        # the plan itself: known errors in, the next one or two levels out
        ss = (C.c_int64 * 64)(*([-1] * 64))
        need = (C.c_int * 2)()
        rounds = 0
        while True:
            n = L.svt_hip_dlf_search_plan(C.byref(q), ss, need, C.byref(best), C.byref(best_err))
            if n == 0: break
            assert n in (1, 2)
            two_at_once += n == 2
            for k in range(n): ss[need[k]] = err[need[k]]
            rounds += 1
        assert (best.value, best_err.value) == (exp_level, exp_err) and rounds <= len(exp_order)
    assert two_at_once > 100   # the low and the high neighbour of an iteration do come back together
