"""GPU parity: the picture-level mode-decision precompute (svt_hip_md_fullpel_sad_picture_dev, svt-av1_amd/csrc/md_pre.hip) vs the oracle (oracle/md_oracle.c, pinned to
the reference's svt_av1_convolve_2d_copy_sr_c + svt_nxm_sad_kernel_helper_c by tests/test_oracle_vs_ref.py::test_md_fullpel_candidate).  What fast_loop_core
(EbProductCodingLoop.c:907) computes for a full-pel single-reference candidate, for every (superblock, PU, reference) of a picture in one launch."""
import ctypes as C

import numpy as np
import pytest

import md_common as M

pytestmark = pytest.mark.gpu


def run_hip(hip, pkg, src, refs, pus, mv, sb_cols, n_sb, pad, pic_w, pic_h):
    n_refs = len(refs)
    d_src, d_mv = hip.to_device(src), hip.to_device(mv)
    d_refs = [hip.to_device(r) for r in refs]
    d_sad = hip.empty(mv.size * 4)
    pu_arr = (pkg.MdPu * len(pus))(*[pkg.MdPu(*p) for p in pus])
    planes = (pkg.MdRefPlane * n_refs)()
    for r in range(n_refs):
        planes[r] = pkg.MdRefPlane(d_refs[r].value + pad * refs[r].shape[1] + pad, refs[r].shape[1], -pad, -pad, refs[r].shape[1] - pad, refs[r].shape[0] - pad)
    hip.check(hip.L.svt_hip_md_fullpel_sad_picture_dev(hip.h, d_src, src.shape[1], pic_w, pic_h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, d_sad), "md pre")
    got = hip.to_host(d_sad, mv.shape, np.uint32)
    hip.free(d_src, d_mv, d_sad, *d_refs)
    return got


@pytest.mark.parametrize("w,h,n_refs", [(336, 208, 3), (64, 64, 1), (200, 152, 7), (1280, 720, 2)])
def test_md_fullpel_sad_picture(hip, pkg, orc, w, h, n_refs):
    """whole pictures incl. partial last superblock rows / columns, 1..7 references, missing vectors, vectors that leave the allocation, an odd source stride"""
    rng = np.random.default_rng(w * 7 + h + n_refs)
    src, refs, pus, mv, sb_cols, n_sb, pad = M.make_case(rng, w, h, n_refs)
    exp = M.oracle_table(orc, src, refs, pus, mv, sb_cols, n_sb, pad, w, h)
    got = run_hip(hip, pkg, src, refs, pus, mv, sb_cols, n_sb, pad, w, h)
    assert (exp == 0xffffffff).any() and (exp != 0xffffffff).any()
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:5]


@pytest.mark.parametrize("w,h,n_refs,pairs", [(336, 208, 3, [(0, 1), (2, 0)]), (64, 64, 2, [(0, 1)]), (200, 152, 4, [(0, 2), (1, 3), (3, 3)])])
def test_md_tables_on_16_bit_planes(hip, pkg, orc, w, h, n_refs, pairs):
    """svt_hip_md_fullpel_sad_picture_hbd_dev / svt_hip_md_fullpel_avg_sad_picture_hbd_dev (the planes a 10-bit encode's mode decision decides on): == the oracle's 16-bit
    restatements (sad_16b_kernel; the high-bit-depth compound copy), which tests/test_oracle_vs_ref.py pins to the reference; odd sample addresses (2-byte aligned rows),
    values at the top of the 10-bit range."""
    rng = np.random.default_rng(w * 3 + h + n_refs)
    src8, refs8, pus, mv, sb_cols, n_sb, pad = M.make_case(rng, w, h, n_refs)
    up = lambda a: np.ascontiguousarray((a.astype(np.uint16) << 2) | rng.integers(0, 4, a.shape).astype(np.uint16))
    src, refs = up(src8), [up(r) for r in refs8]
    refs[0][:, :] = np.where(rng.random(refs[0].shape) < 0.5, 1022, 1023).astype(np.uint16)
    pu4 = np.array(pus, np.uint8)
    planes_o = (C.c_void_p * n_refs)(*[r.ctypes.data + 2 * (pad * r.shape[1] + pad) for r in refs])
    strides = (C.c_int * n_refs)(*[r.shape[1] for r in refs])
    box = np.array([[-pad, -pad, r.shape[1] - pad, r.shape[0] - pad] for r in refs], np.int32)
    pr = np.array(pairs, np.uint8)
    exp = np.zeros(mv.shape, np.uint32); exp2 = np.zeros(mv.shape[:2] + (len(pairs),), np.uint32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    orc.orc_md_fullpel_sad_picture16(vp(src), src.shape[1], w, h, sb_cols, n_sb, len(pus), vp(pu4), n_refs, planes_o, strides, vp(box), vp(mv), vp(exp))
    orc.orc_md_fullpel_avg_sad_picture16(vp(src), src.shape[1], w, h, sb_cols, n_sb, len(pus), vp(pu4), n_refs, planes_o, strides, vp(box), vp(mv), len(pairs), vp(pr), 10, vp(exp2))
    d_src, d_mv = hip.to_device(src), hip.to_device(mv)
    d_refs = [hip.to_device(r) for r in refs]
    d_out, d_out2 = hip.empty(exp.size * 4), hip.empty(exp2.size * 4)
    pu_arr = (pkg.MdPu * len(pus))(*[pkg.MdPu(*p) for p in pus])
    planes = (pkg.MdRefPlane * n_refs)()
    for r in range(n_refs):
        planes[r] = pkg.MdRefPlane(d_refs[r].value + 2 * (pad * refs[r].shape[1] + pad), refs[r].shape[1], -pad, -pad, refs[r].shape[1] - pad, refs[r].shape[0] - pad)
    hip.check(hip.L.svt_hip_md_fullpel_sad_picture_hbd_dev(hip.h, d_src, src.shape[1], w, h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, d_out), "md sad 16")
    hip.check(hip.L.svt_hip_md_fullpel_avg_sad_picture_hbd_dev(hip.h, d_src, src.shape[1], w, h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, len(pairs), vp(pr), d_out2), "md avg sad 16")
    got, got2 = hip.to_host(d_out, exp.shape, np.uint32), hip.to_host(d_out2, exp2.shape, np.uint32)
    hip.free(d_src, d_mv, d_out, d_out2, *d_refs)
    assert (exp != 0xffffffff).any() and (exp2 != 0xffffffff).any()
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:5]
    assert np.array_equal(got2, exp2), np.argwhere(got2 != exp2)[:5]


def test_md_fullpel_sad_other_pu_lists(hip, pkg, orc):
    """the PU list is an argument: rectangular PUs (64x32, 16x64, 4x16 ...) and a list of one"""
    rng = np.random.default_rng(5)
    src, refs, _, _, sb_cols, n_sb, pad = M.make_case(rng, 256, 128, 2)
    for pus in ([(0, 0, 64, 32), (0, 32, 64, 32), (16, 0, 16, 64), (4, 8, 4, 16), (60, 60, 4, 4), (8, 0, 48, 1)], [(32, 32, 32, 32)]):
        mvx = rng.integers(-20, 21, (n_sb, len(pus), 2)).astype(np.int16); mvy = rng.integers(-20, 21, (n_sb, len(pus), 2)).astype(np.int16)
        mv = np.ascontiguousarray(mvx.astype(np.uint16).astype(np.uint32) | (mvy.astype(np.uint16).astype(np.uint32) << 16))
        exp = M.oracle_table(orc, src, refs, pus, mv, sb_cols, n_sb, pad, 256, 128)
        got = run_hip(hip, pkg, src, refs, pus, mv, sb_cols, n_sb, pad, 256, 128)
        assert np.array_equal(got, exp)


def test_md_fullpel_sad_bad_arguments(hip, pkg):
    buf = hip.empty(4096)
    pu = (pkg.MdPu * 1)(pkg.MdPu(0, 0, 6, 8))   # width not a multiple of 4
    pl = (pkg.MdRefPlane * 1)(pkg.MdRefPlane(buf.value, 64, 0, 0, 64, 64))
    assert hip.L.svt_hip_md_fullpel_sad_picture_dev(hip.h, buf, 64, 64, 64, 1, 1, 1, pu, 1, pl, buf, buf) != 0
    pu[0] = pkg.MdPu(0, 0, 8, 8)
    assert hip.L.svt_hip_md_fullpel_sad_picture_dev(hip.h, buf, 64, 64, 64, 1, 1, 1, pu, 8, pl, buf, buf) != 0   # more references than SVT_HIP_MD_MAX_REFS
    assert hip.L.svt_hip_md_fullpel_sad_picture_dev(hip.h, buf, 64, 64, 64, 1, 0, 1, pu, 1, pl, buf, buf) == 0    # no superblocks: nothing to do
    hip.free(buf)


@pytest.mark.parametrize("w,h,n_refs,pairs", [(200, 152, 2, [(0, 1)]), (336, 208, 4, [(0, 2), (1, 3), (0, 1), (3, 0)]), (128, 64, 3, [(2, 0), (1, 1)]), (64, 64, 2, [(0, 1), (1, 0)])])
def test_md_fullpel_avg_sad_picture(hip, pkg, orc, w, h, n_refs, pairs):
    """svt_hip_md_fullpel_avg_sad_picture_dev: the compound-average candidates (two ME vectors, prediction (a + b + 1) >> 1) of every (superblock, square PU, pair of table
    columns) == the oracle's restatement of the reference's two jnt_convolve_2d_copy calls + SAD (pinned to them in tests/test_oracle_vs_ref.py); missing vectors, PUs outside
    the picture and blocks that leave either allocation say "not computed"; content at the top of the range pins the rounding."""
    rng = np.random.default_rng(w + 7 * h + n_refs)
    src, refs, pus, mv, sb_cols, n_sb, pad = M.make_case(rng, w, h, n_refs)
    refs[0][:, :] = np.where(rng.random(refs[0].shape) < 0.5, 254, 255).astype(np.uint8)
    exp = M.oracle_avg_table(orc, src, refs, pus, mv, sb_cols, n_sb, pad, w, h, pairs)
    d_src, d_mv = hip.to_device(src), hip.to_device(mv)
    d_refs = [hip.to_device(r) for r in refs]
    d_out = hip.empty(exp.size * 4)
    pu_arr = (pkg.MdPu * len(pus))(*[pkg.MdPu(*p) for p in pus])
    planes = (pkg.MdRefPlane * n_refs)()
    for r in range(n_refs):
        planes[r] = pkg.MdRefPlane(d_refs[r].value + pad * refs[r].shape[1] + pad, refs[r].shape[1], -pad, -pad, refs[r].shape[1] - pad, refs[r].shape[0] - pad)
    pr = np.array(pairs, np.uint8)
    hip.check(hip.L.svt_hip_md_fullpel_avg_sad_picture_dev(hip.h, d_src, src.shape[1], w, h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, len(pairs), pr.ctypes.data_as(C.c_void_p), d_out), "md avg sad")
    got = hip.to_host(d_out, exp.shape, np.uint32)
    assert hip.L.svt_hip_md_fullpel_avg_sad_picture_dev(hip.h, d_src, src.shape[1], w, h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, 17, pr.ctypes.data_as(C.c_void_p), d_out) != 0   # too many pairs
    bad = np.array([(n_refs, 0)], np.uint8)
    assert hip.L.svt_hip_md_fullpel_avg_sad_picture_dev(hip.h, d_src, src.shape[1], w, h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, 1, bad.ctypes.data_as(C.c_void_p), d_out) != 0   # a column that does not exist
    hip.free(d_src, d_mv, d_out, *d_refs)
    assert (exp != 0xffffffff).any()
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:6]


@pytest.mark.parametrize("w,h,n_refs,bank", [(200, 152, 2, 4), (128, 64, 1, 0), (336, 208, 3, 4), (64, 64, 1, 3)])
def test_md_subpel_grid_picture(hip, pkg, orc, w, h, n_refs, bank):
    """svt_hip_md_subpel_grid_picture_dev: (variance, sse) of all 49 quarter-pel positions around every (superblock, square PU, reference)'s full-pel vector == the oracle's
    svt_upsampled_pref_error restatement (orc_upsampled_pred + orc_variance, each pinned to the reference's C functions) position by position; USE_4_TAPS / USE_8_TAPS /
    USE_2_TAPS kernels; PUs outside the picture, missing vectors and windows that leave the allocation say "not computed"."""
    rng = np.random.default_rng(w + 3 * h + bank)
    src, refs, pus, mv, sb_cols, n_sb, pad = M.make_case(rng, w, h, n_refs, pad=48, mv_range=20)
    if bank == 0:   # saturating content: the 8-bit clips of both passes
        refs[0][:, :] = np.where(rng.random(refs[0].shape) < 0.5, 0, 255).astype(np.uint8)
    exp = M.oracle_grid(orc, src, refs, pus, mv, sb_cols, n_sb, pad, w, h, bank)
    d_src, d_mv = hip.to_device(src), hip.to_device(mv)
    d_refs = [hip.to_device(r) for r in refs]
    d_out = hip.empty(exp.size * 4)
    pu_arr = (pkg.MdPu * len(pus))(*[pkg.MdPu(*p) for p in pus])
    planes = (pkg.MdRefPlane * n_refs)()
    for r in range(n_refs):
        planes[r] = pkg.MdRefPlane(d_refs[r].value + pad * refs[r].shape[1] + pad, refs[r].shape[1], -pad, -pad, refs[r].shape[1] - pad, refs[r].shape[0] - pad)
    hip.check(hip.L.svt_hip_md_subpel_grid_picture_dev(hip.h, d_src, src.shape[1], w, h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, bank, d_out), "md grid")
    got = hip.to_host(d_out, exp.shape, np.uint32)
    # the half-pel round alone (svt_hip_md_halfpel_grid_picture_dev): positions (1, 3, 5) x (1, 3, 5) of the same table
    exp9 = np.ascontiguousarray(exp.reshape(exp.shape[:-2] + (7, 7, 2))[..., 1::2, 1::2, :]).reshape(exp.shape[:-2] + (9, 2))
    d_out9 = hip.empty(exp9.size * 4)
    hip.check(hip.L.svt_hip_md_halfpel_grid_picture_dev(hip.h, d_src, src.shape[1], w, h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, bank, d_out9), "md half-pel grid")
    got9 = hip.to_host(d_out9, exp9.shape, np.uint32)
    hip.free(d_src, d_mv, d_out, d_out9, *d_refs)
    assert (exp == 0xffffffff).any() and (exp != 0xffffffff).any()
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:6]
    assert np.array_equal(got9, exp9), np.argwhere(got9 != exp9)[:6]
