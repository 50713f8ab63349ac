"""GPU parity: batched 85-PU integer full search (HIP, via the C ABI) vs the oracle restatement.
Mirrors the reference's Allsad_CalculationTest / Extsad_CalculationTest matrices
(/root/reference/test/SadTest.cc:838-1222): random / all-max / all-equal (ties) inputs,
FULL and SUB sad, plus picture-edge windows (ragged widths < 8)."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr
import me_common as mc

pytestmark = pytest.mark.gpu


def _run(hip, orc, cur, refp, w, h, sa_w, sa_h, sub, centers=None):
    pad = mc.synth.PAD
    cur_p, ref_p = mc.synth.pad_plane(cur), mc.synth.pad_plane(refp)
    stride = cur_p.shape[1]
    sbs = mc.windows(orc, w, h, sa_w, sa_h, centers)
    o_sad, o_mv = mc.oracle_frame(orc, cur_p, ref_p, stride, pad, sbs, sub)
    g_sad, g_mv = mc.hip_frame(hip, cur_p, ref_p, stride, pad, sbs, sub)
    assert np.array_equal(g_sad, o_sad), f"SAD mismatch at {np.argwhere(g_sad != o_sad)[:5]}"
    assert np.array_equal(g_mv, o_mv), f"MV mismatch at {np.argwhere(g_mv != o_mv)[:5]}"


@pytest.mark.parametrize("sub", [0, 1])
@pytest.mark.parametrize("sa", [(64, 64), (16, 16), (24, 9), (8, 1), (72, 40), (136, 70)])
def test_small_frame_random(hip, orc, sub, sa):
    w, h = 256, 192
    cur, refp = mc.synth.make_luma_pair(w, h, seed=5)
    _run(hip, orc, cur, refp, w, h, sa[0], sa[1], sub)


@pytest.mark.parametrize("pattern", ["ref_max", "src_max", "equal"])
def test_extreme_patterns(hip, orc, pattern):
    w, h = 192, 128
    rng = np.random.default_rng(3)
    cur = rng.integers(0, 256, (h, w), dtype=np.uint8)
    refp = rng.integers(0, 256, (h, w), dtype=np.uint8)
    if pattern == "ref_max":
        refp[:] = 255
    elif pattern == "src_max":
        cur[:] = 255
    else:  # every candidate ties: the first in raster order must win for all 85 PUs
        cur[:] = 77
        refp[:] = 80
    _run(hip, orc, cur, refp, w, h, 64, 64, 0)


def test_edge_windows_and_ragged_width(hip, orc):
    """Non-multiple-of-64 picture, large centred offsets -> windows clamped at the picture edge,
    including widths 1..7 that take the narrow kernel."""
    w, h = 200, 136
    cur, refp = mc.synth.make_luma_pair(w, h, seed=9)
    rng = np.random.default_rng(11)
    n = len(mc.synth.sb_grid(w, h))
    centers = [(int(rng.integers(-90, 260)), int(rng.integers(-90, 180))) for _ in range(n)]
    _run(hip, orc, cur, refp, w, h, 32, 24, 0, centers)
    _run(hip, orc, cur, refp, w, h, 64, 64, 1, centers)


def test_waves_per_sb_variants(hip, orc):
    w, h = 128, 128
    cur, refp = mc.synth.make_luma_pair(w, h, seed=21)
    for waves in (1, 4, 2):
        hip.check(hip.L.svt_hip_me_set_waves_per_sb(hip.h, waves))
        _run(hip, orc, cur, refp, w, h, 64, 64, 0)


def test_1080p_row_band_vs_oracle_and_properties(hip, orc):
    """Full 1080p frame on the GPU; the oracle checks one SB row band bit-exactly (it needs
    ~18 ms/SB), the rest through size-independent properties: SAD(64x64) equals the sum of its four
    32x32 SADs *at the 64x64 MV* is not available without recomputation, so we verify instead that
    every reported (SAD, MV) pair is consistent: recomputing the SAD at the reported MV with numpy
    gives the reported SAD, and no candidate in a sampled set beats it."""
    w, h = 1920, 1080
    pad = mc.synth.PAD
    cur, refp = mc.synth.make_luma_pair(w, h, seed=1)
    cur_p, ref_p = mc.synth.pad_plane(cur), mc.synth.pad_plane(refp)
    stride = cur_p.shape[1]
    sbs = mc.windows(orc, w, h, 64, 64)
    g_sad, g_mv = mc.hip_frame(hip, cur_p, ref_p, stride, pad, sbs, 0)
    n = len(sbs)
    band = range(30 * 8, 30 * 8 + 30)  # SB row 8
    o_sad, o_mv = mc.oracle_frame(orc, cur_p, ref_p, stride, pad, sbs, 0, band.start, band.stop)
    assert np.array_equal(g_sad[band.start:band.stop], o_sad[band.start:band.stop])
    assert np.array_equal(g_mv[band.start:band.stop], o_mv[band.start:band.stop])
    # consistency of every SB's 64x64 and 32x32 results
    rng = np.random.default_rng(0)
    for i in rng.choice(n, 60, replace=False):
        d = sbs[i]
        for pu, (ox, oy, sz) in {0: (0, 0, 64), 1: (0, 0, 32), 4: (32, 32, 32)}.items():
            mvw = int(g_mv[i, pu])
            mx = int(np.array([mvw & 0xFFFF], np.uint16).view(np.int16)[0]) // 4
            my = int(np.array([mvw >> 16], np.uint16).view(np.int16)[0]) // 4
            assert d.x_origin <= mx < d.x_origin + d.width and d.y_origin <= my < d.y_origin + d.height
            s = cur_p[pad + d.sb_y + oy: pad + d.sb_y + oy + sz, pad + d.sb_x + ox: pad + d.sb_x + ox + sz].astype(np.int32)
            r = ref_p[pad + d.sb_y + oy + my: pad + d.sb_y + oy + my + sz, pad + d.sb_x + ox + mx: pad + d.sb_x + ox + mx + sz].astype(np.int32)
            assert int(np.abs(s - r).sum()) == int(g_sad[i, pu])
            for _ in range(8):
                cx = int(rng.integers(d.x_origin, d.x_origin + d.width))
                cy = int(rng.integers(d.y_origin, d.y_origin + d.height))
                r2 = ref_p[pad + d.sb_y + oy + cy: pad + d.sb_y + oy + cy + sz, pad + d.sb_x + ox + cx: pad + d.sb_x + ox + cx + sz].astype(np.int32)
                assert int(np.abs(s - r2).sum()) >= int(g_sad[i, pu])


def test_maximum_search_area(hip, orc):
    """256 x 256 candidates per SB -- the reference's largest search area (MAX_SEARCH_AREA, 16 tiles of 64 x 64 per SB, raster index
    up to 65535 = the full 16-bit key field) -- on a frame whose inner SBs get the whole window and whose border SBs get cropped ones;
    flat regions make exact SAD ties common, so the raster-order tie-break is exercised across tile boundaries."""
    w, h = 320, 256
    cur, refp = mc.synth.make_luma_pair(w, h, seed=17)
    cur[64:128, 64:192] = 90; refp[:, :96] = 90; refp[40:200, 150:300] = 90   # large flat areas -> ties
    _run(hip, orc, cur, refp, w, h, 256, 256, 0)
    _run(hip, orc, cur, refp, w, h, 256, 256, 1)


@pytest.mark.parametrize("sub", [0, 1])
def test_search_area_above_65536_candidates(hip, orc, sub):
    """Search areas larger than the 16-bit raster index of the packed keys (the reference configures up to 750 x 750,
    EbMotionEstimationProcess.c:124-137) take the strip-walking instance of the kernel: strips of whole candidate rows merged with the strict
    '<' of the reference.  336 x 200 = 67 200 candidates -> a 195-row strip and a 5-row strip; flat content makes exact ties across the
    strip boundary common.  Mixed with ordinary 64 x 64 windows in the same call (the small ones must be left to the ordinary instance)."""
    w, h = 384, 320
    cur, refp = mc.synth.make_luma_pair(w, h, seed=23)
    cur[100:180, 60:300] = 77; refp[:, 40:340] = 77   # ties everywhere in the middle band, across the strip boundary
    pad = mc.synth.PAD
    cur_p, ref_p = mc.synth.pad_plane(cur), mc.synth.pad_plane(refp)
    stride = cur_p.shape[1]
    big = mc.windows(orc, w, h, 336, 200)
    small = mc.windows(orc, w, h, 64, 64)
    pick = [7, 8, 14, 15]                                  # inner SBs: the clamp leaves the whole 336 x 200 window
    sbs = (type(big[0]) * 8)(*([big[i] for i in pick] + [small[i] for i in pick]))
    assert all(sbs[i].width * sbs[i].height > 65536 for i in range(4))
    o_sad, o_mv = mc.oracle_frame(orc, cur_p, ref_p, stride, pad, sbs, sub)
    g_sad, g_mv = mc.hip_frame(hip, cur_p, ref_p, stride, pad, sbs, sub)
    assert np.array_equal(g_sad, o_sad) and np.array_equal(g_mv, o_mv)


def test_256x256_search_area_on_a_4k_frame(hip, orc):
    """SURVEY 8(c) row B6 at the reference's high-resolution search area: 3840 x 2160, 256 x 256 window (65 536 candidates, the largest the packed-key
    instance takes), SBs in the picture corners (clamped windows), on the edges and in the interior, a non-zero HME centre on some.  The oracle
    runs only the chosen SBs (a whole 4K frame at this area is minutes of scalar C)."""
    w, h = 3840, 2160
    cur, refp = mc.synth.make_luma_pair(w, h, seed=31)
    pad = mc.synth.PAD
    cur_p, ref_p = mc.synth.pad_plane(cur), mc.synth.pad_plane(refp)
    stride = cur_p.shape[1]
    cols, rows = (w + 63) // 64, (h + 63) // 64
    n = cols * rows
    centers = [(0, 0)] * n
    pick = [0, cols - 1, (rows - 1) * cols, n - 1, 5 * cols, 7 * cols + 23, 16 * cols + 30, rows // 2 * cols + cols - 1, 20 * cols + 31]
    for k, i in enumerate(pick[5:]):
        centers[i] = ((-37, 22), (64, -40), (15, 9), (-120, 80))[k]
    allw = mc.windows(orc, w, h, 256, 256, centers)
    sbs = (type(allw[0]) * len(pick))(*[allw[i] for i in pick])
    assert max(s.width * s.height for s in sbs) == 65536
    o_sad, o_mv = mc.oracle_frame(orc, cur_p, ref_p, stride, pad, sbs, 0)
    g_sad, g_mv = mc.hip_frame(hip, cur_p, ref_p, stride, pad, sbs, 0)
    assert np.array_equal(g_sad, o_sad) and np.array_equal(g_mv, o_mv)
