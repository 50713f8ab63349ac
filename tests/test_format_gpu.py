"""GPU parity: the 8+2-bit <-> 16-bit picture-format conversions around the high-bit-depth path (svt_hip_picture_format_dev) vs the oracle,
which tests/test_oracle_vs_ref.py pins to Common/C_DEFAULT/EbPackUnPack_C.c.  Includes a 4K 10-bit round trip (unpack -> c_pack -> compressed
pack == identity) as the size-independent property."""
import numpy as np
import pytest

from conftest import ptr

pytestmark = pytest.mark.gpu


def call(hip, orc, mode, i0, i1, o0_shape, o0_dt, w, h, want_o1=False):
    e0 = np.zeros(o0_shape, o0_dt); e1 = np.zeros((h, w + 4), np.uint8) if want_o1 else None
    orc.orc_picture_format(mode, ptr(i0), i0.shape[1], ptr(i1) if i1 is not None else None, i1.shape[1] if i1 is not None else 0, ptr(e0), e0.shape[1],
                           ptr(e1) if want_o1 else None, e1.shape[1] if want_o1 else 0, w, h)
    d0 = hip.to_device(i0); d1 = hip.to_device(i1) if i1 is not None else None
    g0 = hip.to_device(np.zeros(o0_shape, o0_dt)); g1 = hip.to_device(np.zeros((h, w + 4), np.uint8)) if want_o1 else None
    hip.check(hip.L.svt_hip_picture_format_dev(hip.h, mode, d0, i0.shape[1], d1, i1.shape[1] if i1 is not None else 0, g0, o0_shape[1], g1, (w + 4) if want_o1 else 0, w, h), "format")
    r0 = hip.to_host(g0, o0_shape, o0_dt); r1 = hip.to_host(g1, (h, w + 4), np.uint8) if want_o1 else None
    hip.free(*[d for d in (d0, d1, g0, g1) if d is not None])
    assert np.array_equal(r0, e0), mode
    if want_o1: assert np.array_equal(r1, e1), mode
    return r0, r1


def test_all_modes(hip, orc):
    rng = np.random.default_rng(5)
    for (w, h) in ((64, 16), (200, 37), (8, 3), (1924, 5), (3, 2)):
        w4 = w & ~3
        in8 = rng.integers(0, 256, (h, w + 5)).astype(np.uint8); inn = rng.integers(0, 256, (h, w + 3)).astype(np.uint8)
        comp = rng.integers(0, 256, (h, w // 4 + 2)).astype(np.uint8)
        a16 = rng.integers(0, 65536, (h, w + 7)).astype(np.uint16); b16 = rng.integers(0, 1024, (h, w + 1)).astype(np.uint16)
        call(hip, orc, 0, in8, inn, (h, w + 2), np.uint16, w, h)
        if w4: call(hip, orc, 1, in8, comp, (h, w + 2), np.uint16, w4, h)
        call(hip, orc, 2, a16, None, (h, w + 2), np.uint8, w, h, want_o1=True)
        call(hip, orc, 3, in8, None, (h, w + 2), np.uint16, w, h)
        call(hip, orc, 4, a16, None, (h, w + 2), np.uint8, w, h)
        if w4: call(hip, orc, 5, inn, None, (h, w // 4 + 2), np.uint8, w4, h)
        call(hip, orc, 6, a16, b16, (h, w + 2), np.uint8, w, h)
    assert hip.L.svt_hip_picture_format_dev(hip.h, 1, None, 0, None, 0, None, 0, None, 0, 6, 4) != 0     # width not a multiple of 4
    assert hip.L.svt_hip_picture_format_dev(hip.h, 3, None, 0, None, 0, None, 0, None, 0, 0, 4) == 0     # empty


def test_4k_10bit_round_trip(hip):
    W, H = 3840, 2160
    rng = np.random.default_rng(6)
    pic = rng.integers(0, 1024, (H, W)).astype(np.uint16)
    d_pic, d_8, d_n, d_c, d_back = hip.to_device(pic), hip.empty(W * H), hip.empty(W * H), hip.empty(W * H // 4), hip.empty(W * H * 2)
    hip.check(hip.L.svt_hip_picture_format_dev(hip.h, 2, d_pic, W, None, 0, d_8, W, d_n, W, W, H), "unpack")
    hip.check(hip.L.svt_hip_picture_format_dev(hip.h, 5, d_n, W, None, 0, d_c, W // 4, None, 0, W, H), "c_pack")
    hip.check(hip.L.svt_hip_picture_format_dev(hip.h, 1, d_8, W, d_c, W // 4, d_back, W, None, 0, W, H), "compressed pack")
    assert np.array_equal(hip.to_host(d_back, (H, W), np.uint16), pic)
    hip.check(hip.L.svt_hip_picture_format_dev(hip.h, 0, d_8, W, d_n, W, d_back, W, None, 0, W, H), "pack")
    assert np.array_equal(hip.to_host(d_back, (H, W), np.uint16), pic)
    assert np.array_equal(hip.to_host(d_8, (H, W), np.uint8), (pic >> 2).astype(np.uint8))
    hip.free(d_pic, d_8, d_n, d_c, d_back)


@pytest.mark.parametrize("dt", [np.uint8, np.uint16])
def test_generate_padding(hip, orc, dt):
    import ctypes as C
    rng = np.random.default_rng(8)
    for (w, h, pw, ph) in ((64, 48, 16, 8), (200, 37, 68, 68), (8, 3, 4, 2), (3840, 2160, 160, 160), (33, 17, 0, 5), (33, 17, 7, 0)):
        buf = rng.integers(0, 250, (h + 2 * ph + 1, w + 2 * pw + 3)).astype(dt)
        exp = buf.copy()
        off = (ph * buf.shape[1] + pw) * buf.itemsize
        orc.orc_generate_padding(C.c_void_p(exp.ctypes.data + off), buf.itemsize, buf.shape[1], w, h, pw, ph)
        d = hip.to_device(buf)
        hip.check(hip.L.svt_hip_generate_padding_dev(hip.h, d.value + off, buf.itemsize, buf.shape[1], w, h, pw, ph), "padding")
        got = hip.to_host(d, buf.shape, dt)
        hip.free(d)
        assert np.array_equal(got, exp), (w, h, pw, ph)
        assert np.array_equal(got[:h + 2 * ph, :w + 2 * pw], np.pad(buf[ph:ph + h, pw:pw + w], ((ph, ph), (pw, pw)), mode="edge"))
