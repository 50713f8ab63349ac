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
