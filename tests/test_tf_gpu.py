"""GPU parity: alt-ref temporal filtering (HIP, whole picture x whole window in one launch through the C ABI) vs the oracle, which
tests/test_oracle_vs_ref.py pins to svt_av1_apply_temporal_filter_planewise(_hbd)_c / estimate_noise(_highbd).  Bit-exact, including the
float weight (glibc expf is reproduced on the device).  Mirrors /root/reference/test/TemporalFilterTestPlanewise.cc at frame level."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr
import tf_common as tfc

pytestmark = pytest.mark.gpu
P3, I3 = C.c_void_p * 3, C.c_int * 3


def run_case(hip, orc, pkg, w, h, bd, ss_x, ss_y, n_refs, center, tf_chroma, seed, noise, decay, mfs, big_mv=False, err_max=20):
    rng = np.random.default_rng(seed)
    src, preds = tfc.make_pictures(rng, w, h, bd, ss_x, ss_y, n_refs, noise=2.5)
    nb = (w // 64) * (h // 64)
    blocks = [tfc.make_blocks(rng, nb, bd, big_mv=big_mv and f == 1, err_max=err_max) for f in range(n_refs)]
    # ---------------- oracle
    o_refs = (tfc.TfRef * (n_refs + 1))()
    order = []          # window order: refs before the centre, the centre, refs after
    k = 0
    for f in range(n_refs + 1):
        if f == center:
            o_refs[f].blocks = None
            order.append(None)
        else:
            for p in range(3):
                o_refs[f].pred[p] = preds[k][p].ctypes.data; o_refs[f].pred_stride[p] = preds[k][p].shape[1]
            o_refs[f].blocks = blocks[k].ctypes.data
            order.append(k); k += 1
    exp = [np.zeros_like(p) for p in src]
    e_sse = np.zeros(2, np.uint64)
    nl = np.asarray(noise, np.float64)
    orc.orc_tf_filter_frame(src[0].itemsize, bd, P3(*[p.ctypes.data for p in src]), I3(*[p.shape[1] for p in src]), P3(*[p.ctypes.data for p in exp]),
                            I3(*[p.shape[1] for p in exp]), w, h, ss_x, ss_y, tf_chroma, o_refs, n_refs + 1, ptr(nl), decay, mfs, ptr(e_sse))
    # ---------------- HIP
    d_src = [hip.to_device(p) for p in src]; d_dst = [hip.to_device(np.zeros_like(p)) for p in src]
    d_pred = [[hip.to_device(p) for p in pr] for pr in preds]; d_blk = [hip.to_device(b) for b in blocks]
    g_refs = (pkg.TfRef * (n_refs + 1))()
    for f, k in enumerate(order):
        if k is None: continue
        for p in range(3):
            g_refs[f].pred[p] = d_pred[k][p].value; g_refs[f].pred_stride[p] = preds[k][p].shape[1]
        g_refs[f].blocks = d_blk[k].value
    d_sse = hip.empty(16)
    hip.check(hip.L.svt_hip_tf_filter_frame_dev(hip.h, src[0].itemsize, bd, P3(*[p.value for p in d_src]), I3(*[p.shape[1] for p in src]),
                                               P3(*[p.value for p in d_dst]), I3(*[p.shape[1] for p in src]), w, h, ss_x, ss_y, tf_chroma, g_refs,
                                               n_refs + 1, nl.ctypes.data_as(C.POINTER(C.c_double)), decay, mfs, d_sse), "tf filter")
    got = [hip.to_host(d, p.shape, p.dtype) for d, p in zip(d_dst, src)]
    g_sse = hip.to_host(d_sse, (2,), np.uint64)
    for p in range(3 if tf_chroma else 1):
        assert np.array_equal(got[p], exp[p]), (bd, ss_x, ss_y, p, np.argwhere(got[p] != exp[p])[:5], got[p][got[p] != exp[p]][:5], exp[p][got[p] != exp[p]][:5])
    assert np.array_equal(g_sse, e_sse), (g_sse, e_sse)
    # in place (the reference overwrites the central picture)
    hip.check(hip.L.svt_hip_tf_filter_frame_dev(hip.h, src[0].itemsize, bd, P3(*[p.value for p in d_src]), I3(*[p.shape[1] for p in src]),
                                               P3(*[p.value for p in d_src]), I3(*[p.shape[1] for p in src]), w, h, ss_x, ss_y, tf_chroma, g_refs,
                                               n_refs + 1, nl.ctypes.data_as(C.POINTER(C.c_double)), decay, mfs, d_sse), "tf filter in place")
    for p in range(3 if tf_chroma else 1):
        assert np.array_equal(hip.to_host(d_src[p], src[p].shape, src[p].dtype), exp[p]), ("in place", p)
    hip.free(*d_src, *d_dst, *[x for pr in d_pred for x in pr], *d_blk, d_sse)
    return exp, src, e_sse


@pytest.mark.parametrize("bd", [8, 10])
def test_filter_frame_420(hip, orc, pkg, bd):
    exp, src, sse = run_case(hip, orc, pkg, 192, 128, bd, 1, 1, n_refs=4, center=2, tf_chroma=1, seed=5 + bd, noise=(1.7, 0.9, 2.4), decay=4, mfs=128)
    assert sse[0] > 0 and sse[1] > 0 and (exp[0] != src[0]).any()
    run_case(hip, orc, pkg, 128, 64, bd, 1, 1, n_refs=6, center=0, tf_chroma=1, seed=9 + bd, noise=(0.2, 3.0, 0.0), decay=3, mfs=2160, big_mv=True)
    run_case(hip, orc, pkg, 64, 128, bd, 1, 1, n_refs=2, center=2, tf_chroma=0, seed=19 + bd, noise=(6.0, 6.0, 6.0), decay=2, mfs=64, err_max=3)
    run_case(hip, orc, pkg, 64, 64, bd, 1, 1, n_refs=15, center=7, tf_chroma=1, seed=29 + bd, noise=(1.0, 1.0, 1.0), decay=4, mfs=720, err_max=1)   # full window


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("ss", [(0, 0), (1, 0)])
def test_filter_frame_444_422(hip, orc, pkg, bd, ss):
    run_case(hip, orc, pkg, 128, 128, bd, ss[0], ss[1], n_refs=3, center=1, tf_chroma=1, seed=40 + bd + ss[0], noise=(2.0, 1.0, 0.5), decay=4, mfs=128)


def test_only_central_is_identity(hip, orc, pkg):
    exp, src, sse = run_case(hip, orc, pkg, 128, 64, 8, 1, 1, n_refs=0, center=0, tf_chroma=1, seed=3, noise=(1.0, 1.0, 1.0), decay=4, mfs=64)
    assert all(np.array_equal(a, b) for a, b in zip(exp, src)) and not sse.any()


@pytest.mark.parametrize("bd", [8, 10])
def test_estimate_noise(hip, orc, pkg, bd):
    orc.orc_tf_estimate_noise.restype = C.c_double
    rng = np.random.default_rng(8 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    for it, (w, h) in enumerate(((640, 360), (203, 77), (64, 3), (1920, 1080))):
        img = np.clip(120 + 40 * np.sin(np.arange(w) / 9.0)[None, :] + rng.normal(0, 1 + 2 * it, (h, w)), 0, 255)
        img = (img * (1 << (bd - 8))).astype(dt)
        if it == 1: img[:] = rng.integers(0, 1 << bd, (h, w))
        stride = w + 16
        buf = np.zeros((h, stride), dt); buf[:, :w] = img
        e_out = np.zeros(2, np.int64)
        e = orc.orc_tf_estimate_noise(ptr(buf), buf.itemsize, bd, w, h, stride, ptr(e_out))
        d_img, d_out = hip.to_device(buf), hip.empty(16)
        hip.check(hip.L.svt_hip_tf_estimate_noise_dev(hip.h, d_img, buf.itemsize, bd, w, h, stride, d_out), "noise")
        got = hip.to_host(d_out, (2,), np.int64)
        hip.free(d_img, d_out)
        assert np.array_equal(got, e_out), (bd, w, h, got, e_out)
        assert hip.L.svt_hip_tf_noise_sigma(int(got[0]), int(got[1])) == e


def test_bad_arguments(hip, pkg):
    refs = (pkg.TfRef * 1)()
    nl = (C.c_double * 3)(1, 1, 1)
    d = hip.empty(64 * 64 * 2)
    P = P3(d.value, d.value, d.value); S = I3(64, 32, 32)
    assert hip.L.svt_hip_tf_filter_frame_dev(hip.h, 1, 8, P, S, P, S, 60, 64, 1, 1, 1, refs, 1, nl, 4, 64, d) != 0          # width not a multiple of 64
    assert hip.L.svt_hip_tf_filter_frame_dev(hip.h, 1, 8, P, S, P, S, 64, 64, 1, 1, 1, refs, 17, nl, 4, 64, d) != 0         # window too long
    assert hip.L.svt_hip_tf_filter_frame_dev(hip.h, 1, 10, P, S, P, S, 64, 64, 1, 1, 1, refs, 1, nl, 4, 64, d) != 0         # 8-bit samples, bd 10
    assert hip.L.svt_hip_tf_filter_frame_dev(hip.h, 1, 8, P, S, P, S, 64, 64, 0, 1, 1, refs, 1, nl, 4, 64, d) != 0          # 4:4:0 is not a format
    hip.free(d)


def test_filter_4k_window7(hip, orc, pkg):
    """BASELINE-size picture (3840 x 2176 = the 64-aligned extent of 2160p), 7-frame window: one 64-row band bit-exact vs the oracle (64x64 blocks
    are independent), plus size-independent properties: window order does not matter, identical predictors with zero error leave the picture
    unchanged (weight 1000 everywhere), the SSE output equals the squared change of the picture."""
    w, h, bd, n_refs = 3840, 2176, 8, 6
    rng = np.random.default_rng(4)
    src, preds = tfc.make_pictures(rng, w, h, bd, 1, 1, n_refs, noise=1.5)
    nb = (w // 64) * (h // 64)
    blocks = [tfc.make_blocks(rng, nb, bd, err_max=12) for _ in range(n_refs)]
    nl = np.asarray((1.2, 0.8, 0.9), np.float64)
    d_src = [hip.to_device(p) for p in src]
    d_pred = [[hip.to_device(p) for p in pr] for pr in preds]; d_blk = [hip.to_device(b) for b in blocks]
    strides = I3(*[p.shape[1] for p in src])

    def run(order, center_at, pred_src=None, blk_src=None):
        refs = (pkg.TfRef * (n_refs + 1))()
        slots = [f for f in range(n_refs + 1) if f != center_at]
        for f, k in zip(slots, order):
            for p in range(3):
                refs[f].pred[p] = (pred_src or d_pred)[k][p].value; refs[f].pred_stride[p] = src[p].shape[1]
            refs[f].blocks = (blk_src or d_blk)[k].value
        d_dst = [hip.to_device(np.zeros_like(p)) for p in src]; d_sse = hip.empty(16)
        hip.check(hip.L.svt_hip_tf_filter_frame_dev(hip.h, 1, bd, P3(*[p.value for p in d_src]), strides, P3(*[p.value for p in d_dst]), strides, w, h, 1, 1, 1,
                                                   refs, n_refs + 1, nl.ctypes.data_as(C.POINTER(C.c_double)), 4, 2160, d_sse), "tf 4k")
        out = [hip.to_host(d, p.shape, p.dtype) for d, p in zip(d_dst, src)]
        sse = hip.to_host(d_sse, (2,), np.uint64)
        hip.free(*d_dst, d_sse)
        return out, sse

    out, sse = run(list(range(n_refs)), 3)
    assert int(sse[0]) == int(((out[0].astype(np.int64) - src[0]) ** 2).sum())
    assert int(sse[1]) == int(((out[1].astype(np.int64) - src[1]) ** 2).sum() + ((out[2].astype(np.int64) - src[2]) ** 2).sum())
    assert (out[0] != src[0]).mean() > 0.2
    out2, sse2 = run([4, 2, 0, 5, 1, 3], 0)
    assert all(np.array_equal(a, b) for a, b in zip(out, out2)) and np.array_equal(sse, sse2)
    # one band against the oracle
    band = 17
    o_refs = (tfc.TfRef * (n_refs + 1))()
    keep = []
    for f in range(n_refs + 1):
        if f == 3: o_refs[f].blocks = None; continue
        k = f if f < 3 else f - 1
        crop = [np.ascontiguousarray(preds[k][p][band * (64 >> (p > 0)):(band + 1) * (64 >> (p > 0))]) for p in range(3)]
        bb = np.ascontiguousarray(blocks[k][band * (w // 64):(band + 1) * (w // 64)])
        keep += crop + [bb]
        for p in range(3):
            o_refs[f].pred[p] = crop[p].ctypes.data; o_refs[f].pred_stride[p] = crop[p].shape[1]
        o_refs[f].blocks = bb.ctypes.data
    s_crop = [np.ascontiguousarray(src[p][band * (64 >> (p > 0)):(band + 1) * (64 >> (p > 0))]) for p in range(3)]
    exp = [np.zeros_like(p) for p in s_crop]; e_sse = np.zeros(2, np.uint64)
    orc.orc_tf_filter_frame(1, bd, P3(*[p.ctypes.data for p in s_crop]), I3(*[p.shape[1] for p in s_crop]), P3(*[p.ctypes.data for p in exp]),
                            I3(*[p.shape[1] for p in exp]), w, 64, 1, 1, 1, o_refs, n_refs + 1, ptr(nl), 4, 2160, ptr(e_sse))
    for p in range(3):
        assert np.array_equal(out[p][band * (64 >> (p > 0)):(band + 1) * (64 >> (p > 0))], exp[p]), ("band", p)
    # identical predictors, zero block error, zero motion: every weight is 1000 and the picture is unchanged
    zero = tfc.make_blocks(rng, nb, bd); zero[:] = np.zeros(1, tfc.BLK_DTYPE)[0]
    d_zero = hip.to_device(zero)
    out3, sse3 = run([0] * n_refs, 2, pred_src=[d_src], blk_src=[d_zero])
    assert all(np.array_equal(a, b) for a, b in zip(out3, src)) and not sse3.any()
    hip.free(*d_src, *[x for pr in d_pred for x in pr], *d_blk, d_zero)


# ------------------------------------------------------------------------------------------------ the sub-pel stage (hook "tf_subpel")
def run_subpel(hip, orc, pkg, w, h, bd, th16, tf_hp, tf_chroma, seed, pad=80, max_mv=9):
    rng = np.random.default_rng(seed)
    src, ref, jobs = tfc.make_subpel_case(rng, w, h, bd, pad, max_mv=max_mv)
    pb = src[0].itemsize
    nb = len(jobs)
    pads = [pad, pad >> 1, pad >> 1]
    # the reference planes are handed over with the pointer at picture sample (0, 0)
    r_off = [(pads[p] * ref[p].shape[1] + pads[p]) * pb for p in range(3)]
    e_pred = [np.zeros_like(p) for p in src]; e_blk = np.zeros(nb, tfc.BLK_DTYPE)
    orc.orc_tf_subpel_frame.argtypes = [C.c_int, C.c_int, P3, I3, P3, I3, P3, I3, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    orc.orc_tf_subpel_frame(pb, bd, P3(*[p.ctypes.data for p in src]), I3(*[p.shape[1] for p in src]), P3(*[ref[p].ctypes.data + r_off[p] for p in range(3)]),
                            I3(*[p.shape[1] for p in ref]), P3(*[p.ctypes.data for p in e_pred]), I3(*[p.shape[1] for p in e_pred]), w // 4, h // 4, th16, tf_hp,
                            tf_chroma, jobs.ctypes.data, nb, e_blk.ctypes.data)
    d_src = [hip.to_device(p) for p in src]; d_ref = [hip.to_device(p) for p in ref]; d_pred = [hip.to_device(np.zeros_like(p)) for p in src]
    d_jobs = hip.to_device(jobs); d_blk = hip.to_device(np.zeros(nb, tfc.BLK_DTYPE))
    hip.check(hip.L.svt_hip_tf_subpel_frame_dev(hip.h, pb, bd, P3(*[p.value for p in d_src]), I3(*[p.shape[1] for p in src]),
                                               P3(*[d_ref[p].value + r_off[p] for p in range(3)]), I3(*[p.shape[1] for p in ref]), P3(*[p.value for p in d_pred]),
                                               I3(*[p.shape[1] for p in src]), w // 4, h // 4, th16, tf_hp, tf_chroma, d_jobs, nb, d_blk), "tf subpel")
    g_blk = hip.to_host(d_blk, (nb,), tfc.BLK_DTYPE)
    for k in ("mv32_x", "mv32_y", "err32", "split", "mv16_x", "mv16_y", "err16"):
        assert np.array_equal(g_blk[k], e_blk[k]), (k, np.argwhere(g_blk[k] != e_blk[k])[:4], g_blk[k][g_blk[k] != e_blk[k]][:4], e_blk[k][g_blk[k] != e_blk[k]][:4])
    for p in range(3 if tf_chroma else 1):
        got = hip.to_host(d_pred[p], src[p].shape, src[p].dtype)
        assert np.array_equal(got, e_pred[p]), (p, np.argwhere(got != e_pred[p])[:4])
    hip.free(*d_src, *d_ref, *d_pred, d_jobs, d_blk)
    return e_blk, e_pred, src


@pytest.mark.parametrize("bd", [8, 10])
def test_subpel_search_and_prediction(hip, orc, pkg, bd):
    """tf_32x32 / tf_16x16_sub_pel_search + split decision + tf_inter_prediction, every (block, frame) pair in one launch, vs the oracle; the threshold
    is set so that both kinds of 32x32 block occur (with and without the 16x16 rounds), vectors of border blocks run into the clamp"""
    blk, _, _ = run_subpel(hip, orc, pkg, 256, 192, bd, th16=1 << 40, tf_hp=1, tf_chroma=1, seed=5)     # no 16x16 rounds: the 32x32 errors of this content
    th = int(np.median(blk["err32"]))
    blk, _, _ = run_subpel(hip, orc, pkg, 256, 192, bd, th16=th, tf_hp=1, tf_chroma=1, seed=5)
    assert 0 < np.count_nonzero(blk["err32"] >= th) < blk["err32"].size
    assert blk["split"].any() and not blk["split"].all()
    run_subpel(hip, orc, pkg, 128, 128, bd, th16=0, tf_hp=0, tf_chroma=0, seed=6)              # every block searched at 16x16, no eighth-pel round, luma only
    run_subpel(hip, orc, pkg, 128, 64, bd, th16=1 << 40, tf_hp=1, tf_chroma=1, seed=7)         # no 16x16 rounds at all


def test_subpel_finds_the_motion(hip, orc, pkg):
    """size-independent property: with the reference = the central picture shifted by (+2.3, -1.6) samples the chosen 32x32 vectors of interior blocks land
    within an eighth of a sample of that shift when the integer vector starts at the nearest full-pel position"""
    rng = np.random.default_rng(11)
    blk, _, _ = run_subpel(hip, orc, pkg, 256, 256, 8, th16=1 << 40, tf_hp=1, tf_chroma=1, seed=11, max_mv=0)
    # interior blocks only (border vectors are pushed out on purpose)
    inner = [r * 4 + c for r in range(1, 3) for c in range(1, 3)]
    # the reference texture is offset by (-2.3, +1.6): block at x in the central picture sits at x - 2.3 in the reference -> vector ~ (-2.3, +1.6) * 8
    mvx = blk["mv32_x"][inner].astype(np.int32); mvy = blk["mv32_y"][inner].astype(np.int32)
    assert np.all(np.abs(mvx - round(-2.3 * 8)) <= 12) and np.all(np.abs(mvy - round(1.6 * 8)) <= 12), (mvx, mvy)
