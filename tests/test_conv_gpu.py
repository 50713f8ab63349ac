"""GPU parity: batched sub-pel prediction (AV1 sr convolve + upsampled_pred), block SAD and block
variance (HIP, through the C ABI) vs the oracle (pinned to svt_av1_[highbd_]convolve_*_sr_c,
svt_aom_upsampled_pred_c, svt_aom_[highbd_10_]variance*_c, svt_fast_loop_nxm_sad_kernel).
Mirrors /root/reference/test/convolve_2d_test.cc:785-1040, VarianceTest.cc, SadTest.cc:430-539."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10])
def test_subpel_predict(hip, pkg, orc, bd):
    rng = np.random.default_rng(40 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    PADV = 24
    refp = rng.integers(0, 1 << bd, (300 + 2 * PADV, 420 + 2 * PADV)).astype(dt)
    refp[PADV:PADV + 64, PADV:PADV + 64] = (1 << bd) - 1
    n = 160
    blks = (pkg.ConvBlk * n)()
    dst_w, dst_h = 2048, 2048
    exp = np.zeros((dst_h, dst_w), dt)
    cx = cy = rowh = 0
    for i in range(n):
        w = int(rng.choice([4, 8, 16, 32, 64, 128])); h = int(rng.choice([4, 8, 16, 32, 64, 128]))
        if cx + w > dst_w: cx = 0; cy += rowh; rowh = 0
        rowh = max(rowh, h)
        mode = 1 if (bd == 8 and i % 3 == 0) else 0
        sx, sy = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        if mode: sx &= ~1; sy &= ~1
        if i % 5 == 0: sx = 0
        if i % 7 == 0: sy = 0
        bx = int(rng.integers(0, 6)); by_ = int(rng.integers(0, 6))
        if mode: bx = by_ = int(rng.choice([3, 4, 0]))
        srx, sry = int(rng.integers(0, 420 - w)), int(rng.integers(0, 300 - h))
        blks[i] = pkg.ConvBlk(srx, sry, cx, cy, w, h, bx, by_, sx, sy, mode, 0)
        sp = C.c_void_p(refp.ctypes.data + ((sry + PADV) * refp.shape[1] + srx + PADV) * refp.itemsize)
        if mode:
            tmp = np.zeros(w * h, np.uint8)
            orc.orc_upsampled_pred(sp, refp.shape[1], ptr(tmp), w, h, sx >> 1, sy >> 1, bx)
            exp[cy:cy + h, cx:cx + w] = tmp.reshape(h, w)
        else:
            dp = C.c_void_p(exp.ctypes.data + (cy * dst_w + cx) * exp.itemsize)
            orc.orc_convolve_sr(sp, refp.shape[1], dp, dst_w, refp.itemsize, w, h, bx, by_, sx, sy, bd)
        cx += w
    d_ref, d_dst, d_b = hip.to_device(refp), hip.to_device(np.zeros_like(exp)), hip.to_device(np.frombuffer(bytes(blks), np.uint8))
    off = (PADV * refp.shape[1] + PADV) * refp.itemsize
    hip.check(hip.L.svt_hip_subpel_predict_batch_dev(hip.h, refp.itemsize, bd, d_ref.value + off, refp.shape[1], d_dst, dst_w, d_b, n), "predict")
    got = hip.to_host(d_dst, exp.shape, dt)
    hip.free(d_ref, d_dst, d_b)
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:5]


@pytest.mark.parametrize("bd", [8, 10])
def test_block_sad_and_variance(hip, pkg, orc, bd):
    rng = np.random.default_rng(60 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    a = rng.integers(0, 1 << bd, (256, 320)).astype(dt); b = rng.integers(0, 1 << bd, (256, 352)).astype(dt)
    a[:64, :64] = (1 << bd) - 1; b[:64, :64] = 0; b[64:128, :64] = a[64:128, :64]
    sizes = [(4, 4), (8, 8), (16, 16), (16, 32), (32, 16), (64, 64), (8, 32), (64, 16), (128, 128), (4, 16)]
    n = 120
    pairs = (pkg.BlkPair * n)(); e_sad = np.zeros(n, np.uint32); e_var = np.zeros(n, np.uint32); e_sse = np.zeros(n, np.uint32)
    orc.orc_nxm_sad.restype = C.c_uint32; orc.orc_sad_16b.restype = C.c_uint32
    orc.orc_variance.restype = C.c_uint32; orc.orc_variance_hbd10.restype = C.c_uint32
    for i in range(n):
        w, h = sizes[i % len(sizes)]
        ax, ay = (0, 0) if i < 10 else (int(rng.integers(0, 320 - w)), int(rng.integers(0, 256 - h)))
        bx, by_ = (0, 0 if i % 2 == 0 else 64) if i < 10 else (int(rng.integers(0, 352 - w)), int(rng.integers(0, 256 - h)))
        pairs[i] = pkg.BlkPair(ax, ay, bx, by_, w, h)
        pa = C.c_void_p(a.ctypes.data + (ay * 320 + ax) * a.itemsize); pb = C.c_void_p(b.ctypes.data + (by_ * 352 + bx) * b.itemsize)
        s = C.c_uint32(0)
        if bd == 8:
            e_sad[i] = orc.orc_nxm_sad(pa, 320, pb, 352, h, w)
            e_var[i] = orc.orc_variance(pa, 320, pb, 352, w, h, C.byref(s))
        else:
            e_sad[i] = orc.orc_sad_16b(pa, 320, pb, 352, h, w)
            e_var[i] = orc.orc_variance_hbd10(pa, 320, pb, 352, w, h, C.byref(s))
        e_sse[i] = s.value
    d_a, d_b, d_p = hip.to_device(a), hip.to_device(b), hip.to_device(np.frombuffer(bytes(pairs), np.uint8))
    d_s, d_v, d_e = hip.empty(n * 4), hip.empty(n * 4), hip.empty(n * 4)
    hip.check(hip.L.svt_hip_block_sad_batch_dev(hip.h, a.itemsize, d_a, 320, d_b, 352, d_p, n, d_s), "sad")
    hip.check(hip.L.svt_hip_block_variance_batch_dev(hip.h, a.itemsize, bd, d_a, 320, d_b, 352, d_p, n, d_v, d_e), "var")
    assert np.array_equal(hip.to_host(d_s, (n,), np.uint32), e_sad)
    assert np.array_equal(hip.to_host(d_v, (n,), np.uint32), e_var)
    assert np.array_equal(hip.to_host(d_e, (n,), np.uint32), e_sse)
    hip.free(d_a, d_b, d_p, d_s, d_v, d_e)


@pytest.mark.gpu
def test_subpel_jobs_from_me_table(hip, pkg):
    """svt_hip_subpel_jobs_from_me_dev: the sub-pel job list of every whole 16x16 block from the [n_sb][85] ME table (16x16 PUs at 5 + z-order index,
    EbMeTierZeroPu; MV word y << 16 | x in quarter-pel) plus q4 phases, against the same arithmetic spelled out here.  A picture whose last superblock
    row / column is partial (200 x 152: 13 x 10 blocks... 12 x 9 whole)."""
    rng = np.random.default_rng(77)
    w, h = 200, 152
    sb_cols, sb_rows = (w + 63) // 64, (h + 63) // 64
    n_sb = sb_cols * sb_rows
    mvx = rng.integers(-32, 32, (n_sb, 85)).astype(np.int16) * 4; mvy = rng.integers(-32, 32, (n_sb, 85)).astype(np.int16) * 4
    word = (mvy.astype(np.uint16).astype(np.uint32) << 16) | mvx.astype(np.uint16).astype(np.uint32)
    bw, bh = w // 16, h // 16
    frac = rng.integers(0, 16, (bw * bh, 2)).astype(np.uint8)
    d_mv, d_frac = hip.to_device(word), hip.to_device(frac)
    d_out = hip.empty(C.sizeof(pkg.ConvBlk) * bw * bh)
    hip.check(hip.L.svt_hip_subpel_jobs_from_me_dev(hip.h, d_mv, sb_cols, w, h, d_frac, d_out), "jobs from me")
    raw = hip.to_host(d_out, (C.sizeof(pkg.ConvBlk) * bw * bh,), np.uint8)
    got = (pkg.ConvBlk * (bw * bh)).from_buffer_copy(raw.tobytes())
    for k in range(bw * bh):
        bx, by = k % bw, k // bw
        sb = (by // 4) * sb_cols + bx // 4
        qx, qy = bx % 4, by % 4
        z = ((qy // 2) * 2 + qx // 2) * 4 + (qy % 2) * 2 + qx % 2
        g = got[k]
        assert (g.src_x, g.src_y, g.dst_x, g.dst_y, g.w, g.h, g.bank_x, g.bank_y, g.subpel_x, g.subpel_y, g.mode) == \
            (bx * 16 + int(mvx[sb, 5 + z]) // 4, by * 16 + int(mvy[sb, 5 + z]) // 4, bx * 16, by * 16, 16, 16, 0, 0, int(frac[k, 0]), int(frac[k, 1]), 0), k
    hip.free(d_mv, d_frac, d_out)
