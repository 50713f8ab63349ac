"""GPU parity: HME pyramids (decimation / 2x2 down-sampling), per-SB variance pyramid and the HME
exhaustive search (svt_sad_loop_kernel semantics incl. first-minimum tie break and sub-SAD rows),
HIP through the C ABI vs the oracle.  Mirrors /root/reference/test/SadTest.cc:611-785 (sad_LoopTest)
and compute_mean_test.cc."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr

pytestmark = pytest.mark.gpu


def test_downsample_and_variance_pyramid(hip, orc):
    rng = np.random.default_rng(2)
    W, H = 448, 256
    img = rng.integers(0, 256, (H + 64, W + 64), dtype=np.uint8)
    img[:64, :64] = 255; img[64:128, :64] = 0
    d_img = hip.to_device(img)
    for step in (2, 4):
        for filt in (0, 1):
            exp = np.zeros((H // step, W // step + 8), np.uint8)
            orc.orc_downsample_2d(ptr(img), img.shape[1], W, H, ptr(exp), exp.shape[1], step, filt)
            d_out = hip.to_device(np.zeros_like(exp))
            hip.check(hip.L.svt_hip_downsample_2d_dev(hip.h, d_img, img.shape[1], W, H, d_out, exp.shape[1], step, filt))
            assert np.array_equal(hip.to_host(d_out, exp.shape, np.uint8), exp), (step, filt)
            hip.free(d_out)
    cols, n = W // 64, (W // 64) * (H // 64)
    for full in (0, 1):
        e_m = np.zeros((n, 85), np.uint8); e_v = np.zeros((n, 85), np.uint16)
        for sb in range(n):
            p = C.c_void_p(img.ctypes.data + (sb // cols) * 64 * img.shape[1] + (sb % cols) * 64)
            orc.orc_variance_pyramid_sb(p, img.shape[1], full, C.c_void_p(e_m.ctypes.data + sb * 85), C.c_void_p(e_v.ctypes.data + sb * 170))
        d_m, d_v = hip.empty(n * 85), hip.empty(n * 170)
        hip.check(hip.L.svt_hip_variance_pyramid_dev(hip.h, d_img, img.shape[1], cols, n, full, d_m, d_v))
        assert np.array_equal(hip.to_host(d_m, (n, 85), np.uint8), e_m) and np.array_equal(hip.to_host(d_v, (n, 85), np.uint16), e_v), full
        hip.free(d_m, d_v)
    hip.free(d_img)


def test_sad_loop_batch(hip, pkg, orc):
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (160, 256), dtype=np.uint8); ref = rng.integers(0, 256, (400, 600), dtype=np.uint8)
    ref[0:120, 0:200] = 7; src[0:64, 0:64] = 9            # all candidates tie -> first wins
    cases = [(16, 16, 32, 9, 1), (32, 32, 16, 7, 1), (64, 64, 8, 5, 1), (16, 16, 48, 3, 2), (16, 16, 240, 60, 1), (32, 32, 16, 16, 2),
             (64, 64, 16, 16, 2), (16, 8, 64, 32, 1), (8, 8, 7, 3, 1)]
    n = len(cases) * 4
    S = (pkg.SadLoop * n)()
    e_sad = np.zeros(n, np.uint32); e_xy = np.zeros((n, 2), np.int16)
    for i in range(n):
        bw, bh, saw, sah, rs = cases[i % len(cases)]
        tie = i < 3
        sx, sy = (0, 0) if tie else (int(rng.integers(0, 256 - bw)), int(rng.integers(0, 160 - bh)))
        rx, ry = (0, 0) if tie else (int(rng.integers(0, 600 - bw - saw)), int(rng.integers(0, 400 - bh - sah)))
        S[i] = pkg.SadLoop(sx, sy, rx, ry, bw, bh, saw, sah, rs, 0)
        best = C.c_uint64(0); xc = C.c_int16(-3); yc = C.c_int16(-4)
        orc.orc_sad_loop(C.c_void_p(src.ctypes.data + sy * 256 + sx), 256 * rs, C.c_void_p(ref.ctypes.data + ry * 600 + rx), 600 * rs, bh // rs, bw,
                         C.byref(best), C.byref(xc), C.byref(yc), 600, C.c_int16(saw), C.c_int16(sah))
        e_sad[i] = best.value; e_xy[i] = (xc.value, yc.value)
    d_src, d_ref, d_S = hip.to_device(src), hip.to_device(ref), hip.to_device(np.frombuffer(bytes(S), np.uint8))
    d_sad, d_xy = hip.empty(n * 4), hip.to_device(np.tile(np.array([-3, -4], np.int16), (n, 1)))
    hip.check(hip.L.svt_hip_sad_loop_batch_dev(hip.h, d_src, 256, d_ref, 600, d_S, n, d_sad, d_xy))
    assert np.array_equal(hip.to_host(d_sad, (n,), np.uint32), e_sad)
    assert np.array_equal(hip.to_host(d_xy, (n, 2), np.int16), e_xy)
    hip.free(d_src, d_ref, d_S, d_sad, d_xy)
