"""GPU parity: Wiener restoration statistics on the matrix cores (svt_hip_wiener_stats_plane_dev) vs the oracle
(orc_wiener_stats_plane, pinned to svt_av1_compute_stats_c + the reference's unit geometry): windows 7 / 5 / 3, unit sizes 64 / 128 / 256,
ragged planes, luma and chroma unit offsets, extreme content (all-max / all-min / checkerboard units that stress the int8 bias)."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr

pytestmark = pytest.mark.gpu
EXT = 3


def units(size, unit):
    return max((size + unit // 2) // unit, 1)


@pytest.mark.parametrize("win", [7, 5, 3])
def test_wiener_stats(hip, orc, win):
    rng = np.random.default_rng(40 + win)
    for (w, h, US, ss) in ((200, 152, 64, 0), (328, 264, 128, 1), (520, 300, 256, 0), (96, 72, 64, 1)):
        yy, xx = np.mgrid[0:h, 0:w]
        dgd = np.clip(110 + 70 * np.sin(xx / 13.0) * np.cos(yy / 7.0) + rng.normal(0, 12, (h, w)), 0, 255).astype(np.uint8)
        dgd[:40, :70] = 255; dgd[40:80, :70] = 0; dgd[80:120, :64] = ((xx[80:120, :64] + yy[80:120, :64]) & 1) * 255
        src = np.clip(dgd.astype(np.int32) + rng.integers(-25, 26, (h, w)), 0, 255).astype(np.uint8)
        src[:20, :30] = 0
        ext = np.ascontiguousarray(np.pad(dgd, EXT, mode="edge")); st = ext.shape[1]; off = EXT * st + EXT
        nu = units(w, US) * units(h, US); w2 = win * win
        Me, He = np.zeros((nu, w2), np.int64), np.zeros((nu, w2 * w2), np.int64)
        orc.orc_wiener_stats_plane(win, C.c_void_p(ext.ctypes.data + off), st, ptr(src), w, 1, 8, w, h, ss, US, ptr(Me), ptr(He))
        d_ext, d_src, d_M, d_H = hip.to_device(ext), hip.to_device(src), hip.empty(Me.nbytes), hip.empty(He.nbytes)
        hip.check(hip.L.svt_hip_wiener_stats_plane_dev(hip.h, 1, 8, win, d_ext.value + off, st, d_src, w, w, h, US, ss, d_M, d_H), "wiener stats")
        Mg, Hg = hip.to_host(d_M, Me.shape, np.int64), hip.to_host(d_H, He.shape, np.int64)
        hip.free(d_ext, d_src, d_M, d_H)
        assert np.array_equal(Mg, Me), (win, w, h, US, ss, np.argwhere(Mg != Me)[:5])
        assert np.array_equal(Hg, He), (win, w, h, US, ss, np.argwhere(Hg != He)[:5])


@pytest.mark.parametrize("bd", [10, 12])
@pytest.mark.parametrize("win", [7, 5, 3])
def test_wiener_stats_16bit(hip, orc, bd, win):
    """svt_av1_compute_stats_highbd: the 16-bit path runs three 8-bit component planes (v = 32 h + l; H, L, H + L) through the int8 MFMA Gram
    kernel and recombines exactly; bit_depth_divider applied with C truncation.  Extremes: all-max / all-zero / checkerboard units."""
    rng = np.random.default_rng(60 + win + bd)
    mx = (1 << bd) - 1
    for (w, h, US, ss) in ((200, 152, 64, 0), (328, 264, 128, 1), (520, 300, 256, 0)):
        yy, xx = np.mgrid[0:h, 0:w]
        dgd = np.clip((110 + 70 * np.sin(xx / 13.0) * np.cos(yy / 7.0)) * (1 << (bd - 8)) + rng.normal(0, 12 << (bd - 8), (h, w)), 0, mx).astype(np.uint16)
        dgd[:40, :70] = mx; dgd[40:80, :70] = 0; dgd[80:120, :64] = ((xx[80:120, :64] + yy[80:120, :64]) & 1) * mx
        src = np.clip(dgd.astype(np.int32) + rng.integers(-25 << (bd - 8), (25 << (bd - 8)) + 1, (h, w)), 0, mx).astype(np.uint16)
        src[:20, :30] = 0; src[20:40, :30] = mx
        src[64:128, 128:192] = mx - dgd[64:128, 128:192]          # an anti-correlated unit: negative sums, so the divider's truncation toward zero matters
        ext = np.ascontiguousarray(np.pad(dgd, EXT, mode="edge")); st = ext.shape[1]; off = (EXT * st + EXT) * 2
        nu = units(w, US) * units(h, US); w2 = win * win
        Me, He = np.zeros((nu, w2), np.int64), np.zeros((nu, w2 * w2), np.int64)
        orc.orc_wiener_stats_plane(win, C.c_void_p(ext.ctypes.data + off), st, ptr(src), w, 2, bd, w, h, ss, US, ptr(Me), ptr(He))
        d_ext, d_src, d_M, d_H = hip.to_device(ext), hip.to_device(src), hip.empty(Me.nbytes), hip.empty(He.nbytes)
        hip.check(hip.L.svt_hip_wiener_stats_plane_dev(hip.h, 2, bd, win, d_ext.value + off, st, d_src, w, w, h, US, ss, d_M, d_H), "wiener stats 16")
        Mg, Hg = hip.to_host(d_M, Me.shape, np.int64), hip.to_host(d_H, He.shape, np.int64)
        hip.free(d_ext, d_src, d_M, d_H)
        assert np.array_equal(Mg, Me), (bd, win, w, h, US, ss, np.argwhere(Mg != Me)[:5], Mg[Mg != Me][:4], Me[Mg != Me][:4])
        assert np.array_equal(Hg, He), (bd, win, w, h, US, ss, np.argwhere(Hg != He)[:5])
        assert US != 64 or (Me < 0).any()


@pytest.mark.parametrize("win", [7, 5, 3])
def test_wiener_initial_filters_on_the_device(hip, orc, win):
    """svt_hip_wiener_init_units_dev == orc_wiener_unit_init (pinned to the reference's wiener_decompose_sep_sym / finalize_sym_filter / compute_score by
    tests/test_oracle_vs_ref.py::test_wiener_initial_filter) for a batch of units: statistics of blurred, noisy, identical, flat and unrelated picture pairs at 8 and 10 bits,
    and raw random statistics (singular systems, clamped taps); both outcomes occur."""
    from wiener_common import wiener_unit_stats
    rng = np.random.default_rng(900 + win)
    w2 = win * win
    Ms, Hs = [], []
    for bd in (8, 10):
        for kind in range(5):
            for _ in range(3):
                M, H = wiener_unit_stats(orc, rng, win, bd, kind)
                Ms.append(M); Hs.append(H)
    for t in range(8):
        G = rng.integers(-300, 300, (w2, 3 * w2)).astype(np.int64)
        Hs.append((G @ G.T).reshape(-1).copy() if t < 4 else rng.integers(-(1 << 30), 1 << 30, w2 * w2).astype(np.int64))
        Ms.append(rng.integers(-(1 << 26), 1 << 26, w2).astype(np.int64))
    n = len(Ms)
    M = np.ascontiguousarray(np.stack(Ms)); H = np.ascontiguousarray(np.stack(Hs))
    exp_wn = np.zeros((n, 16), np.int16); exp_st = np.zeros(n, np.int8)
    for u in range(n):
        exp_st[u] = orc.orc_wiener_unit_init(win, ptr(M[u]), ptr(H[u]), C.c_void_p(exp_wn[u].ctypes.data), C.c_void_p(exp_wn[u].ctypes.data + 16))
    d_M, d_H = hip.to_device(M), hip.to_device(H)
    d_wn, d_act, d_st = hip.to_device(np.full((n, 16), 77, np.int16)), hip.to_device(np.full(n, 9, np.uint8)), hip.to_device(np.full(n, 9, np.int8))
    hip.check(hip.L.svt_hip_wiener_init_units_dev(hip.h, win, n, d_M, d_H, d_wn, d_act, d_st), "wiener init")
    got_wn, got_act, got_st = hip.to_host(d_wn, (n, 16), np.int16), hip.to_host(d_act, (n,), np.uint8), hip.to_host(d_st, (n,), np.int8)
    hip.free(d_M, d_H, d_wn, d_act, d_st)
    assert np.array_equal(got_st, exp_st), (got_st, exp_st)
    assert np.array_equal(got_act, (exp_st == 1).astype(np.uint8))
    assert np.array_equal(got_wn, exp_wn), np.argwhere(got_wn != exp_wn)[:6]
    assert set(exp_st.tolist()) == {1, 2}
