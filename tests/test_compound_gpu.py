"""GPU parity: compound (two-reference) prediction (HIP through the C ABI) vs the oracle, which tests/test_oracle_vs_ref.py pins to two
svt_av1_[highbd_]jnt_convolve_*_c calls, svt_av1_build_compound_diffwtd_mask_d16_c and svt_aom_{lowbd,highbd}_blend_a64_d16_mask_c.
All 22 AV1 block sizes, the four convolve flavours per reference, average / distance / diff-weighted / supplied-mask (also sub-sampled)."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr
import comp_common as cmc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_compound_blocks(hip, orc, bd):
    rng = np.random.default_rng(500 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    W, H = 1536, 1152
    ref0 = rng.integers(0, 1 << bd, (H, W)).astype(dt); ref1 = rng.integers(0, 1 << bd, (H, W)).astype(dt)
    ref0[:200, :200] = (1 << bd) - 1; ref1[:200, :200] = 0
    n = 80
    blks, masks = cmc.make_blocks(rng, W - 256, H - 128, n, 1 << 21)
    for b in blks:
        b.src0_x += 64; b.src0_y += 64; b.src1_x += 64; b.src1_y += 64; b.dst_x += 8; b.dst_y += 8
    exp = np.zeros((H, W), dt); m_exp = masks.copy()
    orc.orc_compound_predict_batch(ref0.itemsize, bd, ptr(ref0), W, ptr(ref1), W, ptr(exp), W, ptr(m_exp), blks, 0, n)
    d_r0, d_r1, d_dst, d_m = hip.to_device(ref0), hip.to_device(ref1), hip.to_device(np.zeros((H, W), dt)), hip.to_device(masks)
    d_b = hip.to_device(np.frombuffer(bytes(blks), np.uint8).copy())
    hip.check(hip.L.svt_hip_compound_predict_batch_dev(hip.h, ref0.itemsize, bd, d_r0, W, d_r1, W, d_dst, W, d_m, d_b, n), "compound")
    got = hip.to_host(d_dst, (H, W), dt); m_got = hip.to_host(d_m, masks.shape, np.uint8)
    hip.free(d_r0, d_r1, d_dst, d_m, d_b)
    for i, b in enumerate(blks):
        g, e = got[b.dst_y:b.dst_y + b.h, b.dst_x:b.dst_x + b.w], exp[b.dst_y:b.dst_y + b.h, b.dst_x:b.dst_x + b.w]
        assert np.array_equal(g, e), (bd, i, b.type, b.w, b.h, (b.subpel0_x, b.subpel0_y, b.subpel1_x, b.subpel1_y), np.argwhere(g != e)[:4])
    assert np.array_equal(got, exp) and np.array_equal(m_got, m_exp)
    assert (m_exp != masks).any() and exp.any()


def test_compound_empty_and_bad_args(hip):
    assert hip.L.svt_hip_compound_predict_batch_dev(hip.h, 1, 8, None, 0, None, 0, None, 0, None, None, 0) == 0
    assert hip.L.svt_hip_compound_predict_batch_dev(hip.h, 1, 10, None, 0, None, 0, None, 0, None, None, 0) != 0
    assert hip.L.svt_hip_compound_predict_batch_dev(hip.h, 1, 8, None, 0, None, 0, None, 0, None, None, 3) != 0


class ObmcBlk(C.Structure):
    _fields_ = [("pre_x", C.c_int32), ("pre_y", C.c_int32), ("w", C.c_uint8), ("h", C.c_uint8), ("xoffset", C.c_uint8), ("yoffset", C.c_uint8), ("wm_off", C.c_int32)]


def test_obmc_costs(hip, orc):
    """svt_hip_obmc_cost_batch_dev vs the oracle (pinned to svt_aom_obmc_sad / obmc_variance / obmc_sub_pixel_variance for all block sizes)."""
    rng = np.random.default_rng(9)
    W, H = 640, 480
    pre = rng.integers(0, 256, (H, W)).astype(np.uint8); pre[:140, :140] = 255
    n = 4 * len(cmc.SIZES)
    blks = (ObmcBlk * n)()
    off = 0
    for i in range(n):
        w, h = cmc.SIZES[i % len(cmc.SIZES)]
        blks[i] = ObmcBlk(int(rng.integers(0, W - 130)) if i else 0, int(rng.integers(0, H - 130)) if i else 0, w, h,
                          0 if i % 3 == 0 else int(rng.integers(0, 8)), 0 if i % 3 == 0 else int(rng.integers(0, 8)), off)
        off += w * h
    wsrc = rng.integers(0, 255 * 4096 + 1, off).astype(np.int32); mask = rng.integers(0, 4097, off).astype(np.int32)
    wsrc[:128 * 128] = 0; mask[:128 * 128] = 4096
    exp = np.zeros((n, 3), np.uint32)
    orc.orc_obmc_batch(ptr(pre), W, ptr(wsrc), ptr(mask), blks, n, ptr(exp))
    d_pre, d_w, d_m, d_b, d_o = hip.to_device(pre), hip.to_device(wsrc), hip.to_device(mask), hip.to_device(np.frombuffer(bytes(blks), np.uint8).copy()), hip.empty(exp.nbytes)
    hip.check(hip.L.svt_hip_obmc_cost_batch_dev(hip.h, d_pre, W, d_w, d_m, d_b, n, d_o), "obmc")
    got = hip.to_host(d_o, exp.shape, np.uint32)
    hip.free(d_pre, d_w, d_m, d_b, d_o)
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:6]
    assert exp.any() and hip.L.svt_hip_obmc_cost_batch_dev(hip.h, None, 0, None, None, None, 0, None) == 0


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("ss", [0, 1])
def test_warp_blocks(hip, orc, bd, ss):
    """svt_hip_warp_predict_batch_dev vs the oracle (pinned to svt_av1_[highbd_]warp_affine_c): block sizes 8..128, random and extreme shear,
    models that move the block outside the plane (edge clamping), luma and 4:2:0 chroma sub-sampling."""
    rng = np.random.default_rng(700 + bd + ss)
    dt = np.uint8 if bd == 8 else np.uint16
    W, H = 1152, 640
    plane = rng.integers(0, 1 << bd, (H, W)).astype(dt)
    n = 40
    blks = cmc.warp_blocks(rng, W, H, n)
    exp = np.zeros((H, W), dt)
    orc.orc_warp_predict_batch(plane.itemsize, bd, ptr(plane), W, H, W, ptr(exp), W, ss, ss, blks, n)
    d_p, d_o, d_b = hip.to_device(plane), hip.to_device(np.zeros((H, W), dt)), hip.to_device(np.frombuffer(bytes(blks), np.uint8).copy())
    hip.check(hip.L.svt_hip_warp_predict_batch_dev(hip.h, plane.itemsize, bd, d_p, W, H, W, d_o, W, ss, ss, d_b, n), "warp")
    got = hip.to_host(d_o, (H, W), dt)
    hip.free(d_p, d_o, d_b)
    for i, b in enumerate(blks):
        g, e = got[b.p_row:b.p_row + b.p_height, b.p_col:b.p_col + b.p_width], exp[b.p_row:b.p_row + b.p_height, b.p_col:b.p_col + b.p_width]
        assert np.array_equal(g, e), (bd, ss, i, b.p_width, b.p_height, np.argwhere(g != e)[:4])
    assert np.array_equal(got, exp) and exp.any()


@pytest.mark.parametrize("bd", [8, 10])
def test_blend_a64(hip, orc, bd):
    """svt_hip_blend_a64_batch_dev vs the oracle (pinned to svt_aom_[highbd_]blend_a64_{mask,hmask,vmask}_c), incl. blocks blended in place."""
    rng = np.random.default_rng(900 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    W, H = 1024, 768
    a = rng.integers(0, 1 << bd, (H, W)).astype(dt); b1 = rng.integers(0, 1 << bd, (H, W)).astype(dt)
    n = 48
    blks, masks = cmc.blend_blocks(rng, W, H, n, 1 << 19)
    # in-place blocks read their own destination; keep every other block's sources away from all destinations so that block order cannot matter
    for b in blks:
        if (b.src0_x, b.src0_y) != (b.dst_x, b.dst_y): b.src0_x, b.src0_y = b.dst_x, b.dst_y
    exp = a.copy()
    orc.orc_blend_a64_batch(a.itemsize, ptr(exp), W, ptr(b1), W, ptr(exp), W, ptr(masks), blks, n)
    d_a, d_b, d_m, d_k = hip.to_device(a), hip.to_device(b1), hip.to_device(masks), hip.to_device(np.frombuffer(bytes(blks), np.uint8).copy())
    hip.check(hip.L.svt_hip_blend_a64_batch_dev(hip.h, a.itemsize, d_a, W, d_b, W, d_a, W, d_m, d_k, n), "blend")
    got = hip.to_host(d_a, (H, W), dt)
    hip.free(d_a, d_b, d_m, d_k)
    assert np.array_equal(got, exp) and (exp != a).any()
