"""GPU parity: CDEF strength search (full 64-entry distortion table, incl. the FP64 luma metric) and
frame application (HIP, through the C ABI) vs the oracle (pinned to svt_cdef_find_dir_c /
svt_cdef_filter_block_c / svt_cdef_filter_fb / compute_cdef_dist*), 8- and 10-bit.
Mirrors /root/reference/test/CdefTest.cc:342-761 at frame level."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr
import cdef_common as cc

pytestmark = pytest.mark.gpu
P3, I3 = C.c_void_p * 3, C.c_int * 3


def gpu_search(hip, rec, src, bd, skip8, pri_damping):
    h, w = rec[0].shape
    nfb = ((h + 63) // 64) * ((w + 63) // 64)
    d_rec = [hip.to_device(p) for p in rec]; d_src = [hip.to_device(p) for p in src]
    d_skip = hip.to_device(skip8)
    d_mse = hip.to_device(np.zeros((2, nfb, 64), np.uint64)); d_dir = hip.empty(nfb * 64); d_var = hip.empty(nfb * 64 * 4)
    hip.check(hip.L.svt_hip_cdef_search_frame_dev(hip.h, rec[0].itemsize, P3(*[p.value for p in d_rec]), I3(*[p.shape[1] for p in rec]),
                                                 P3(*[p.value for p in d_src]), I3(*[p.shape[1] for p in src]), w, h, d_skip, pri_damping, bd,
                                                 d_mse, d_dir, d_var), "cdef search")
    mse = hip.to_host(d_mse, (2, nfb, 64), np.uint64)
    hip.free(*d_rec, *d_src, d_skip, d_mse, d_dir, d_var)
    return mse


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("smooth", [True, False])
@pytest.mark.parametrize("size", [(208, 144), (200, 136)])
def test_search_table(hip, orc, bd, smooth, size):
    # 4 x 3 filter blocks; 208 x 144: the last ones 16 samples, 200 x 136: 8 luma / 4 chroma samples — narrower than the 8-sample halo the
    # reference stages for their left / upper neighbours (EbCdefProcess.c:210-226: the halo then extends into the picture padding, but no filter
    # tap reaches further than 2 samples, so the padding is never read)
    src, rec, skip8 = cc.make_frame(size[0], size[1], bd, seed=3 + bd, smooth=smooth)
    skip8[0:8, 8:16] = 1          # one all-skip filter block: its entries stay untouched (zero)
    for damping in (3, 5, 6):
        exp = cc.orc_search(orc, rec, src, bd, skip8, damping)
        got = gpu_search(hip, rec, src, bd, skip8, damping)
        assert np.array_equal(got[0], exp[0]), ("Y", bd, damping, np.argwhere(got[0] != exp[0])[:5])
        assert np.array_equal(got[1], exp[1]), ("UV", bd, damping, np.argwhere(got[1] != exp[1])[:5])
        assert not got[:, 1].any()


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("size", [(208, 144), (200, 136)])
def test_apply_frame(hip, orc, bd, size):
    src, rec, skip8 = cc.make_frame(size[0], size[1], bd, seed=30 + bd)
    h, w = rec[0].shape
    nfb = 12
    rng = np.random.default_rng(1)
    ys = rng.integers(0, 64, nfb).astype(np.uint8); uvs = rng.integers(0, 64, nfb).astype(np.uint8)
    ys[3] = 0; uvs[3] = 0      # unfiltered fb
    ys[5] = 0                  # luma off, chroma on
    exp = [p.copy() for p in rec]
    orc.orc_cdef_apply_frame(P3(*[p.ctypes.data for p in rec]), P3(*[p.ctypes.data for p in exp]), I3(*[p.shape[1] for p in rec]),
                             rec[0].itemsize, w, h, ptr(skip8), ptr(ys), ptr(uvs), 5, bd)
    d_in = [hip.to_device(p) for p in rec]; d_out = [hip.to_device(np.full_like(p, 77)) for p in rec]   # every sample must be written: no initial copy
    d_skip, d_ys, d_uvs, d_dir = hip.to_device(skip8), hip.to_device(ys), hip.to_device(uvs), hip.empty(nfb * 64)
    hip.check(hip.L.svt_hip_cdef_apply_frame_dev(hip.h, rec[0].itemsize, P3(*[p.value for p in d_in]), P3(*[p.value for p in d_out]),
                                                I3(*[p.shape[1] for p in rec]), w, h, d_skip, d_ys, d_uvs, 5, bd, d_dir, None), "cdef apply")
    for pli in range(3):
        got = hip.to_host(d_out[pli], rec[pli].shape, rec[pli].dtype)
        assert (exp[pli] != rec[pli]).any()
        assert np.array_equal(got, exp[pli]), (bd, pli, np.argwhere(got != exp[pli])[:5])
    # second form: direction / variance handed over from the strength search on the same picture instead of being recomputed
    dir1 = hip.to_host(d_dir, (nfb * 64,), np.uint8)
    d_src = [hip.to_device(p) for p in src]
    d_mse, d_dir2, d_var = hip.to_device(np.zeros((2, nfb, 64), np.uint64)), hip.empty(nfb * 64), hip.empty(nfb * 64 * 4)
    hip.check(hip.L.svt_hip_cdef_search_frame_dev(hip.h, rec[0].itemsize, P3(*[p.value for p in d_in]), I3(*[p.shape[1] for p in rec]),
                                                 P3(*[p.value for p in d_src]), I3(*[p.shape[1] for p in src]), w, h, d_skip, 5, bd,
                                                 d_mse, d_dir2, d_var), "cdef search")
    d_out2 = [hip.to_device(p) for p in rec]
    hip.check(hip.L.svt_hip_cdef_apply_frame_dev(hip.h, rec[0].itemsize, P3(*[p.value for p in d_in]), P3(*[p.value for p in d_out2]),
                                                I3(*[p.shape[1] for p in rec]), w, h, d_skip, d_ys, d_uvs, 5, bd, d_dir2, d_var), "cdef apply (reuse)")
    for pli in range(3):
        got = hip.to_host(d_out2[pli], rec[pli].shape, rec[pli].dtype)
        assert np.array_equal(got, exp[pli]), ("reuse", bd, pli, np.argwhere(got != exp[pli])[:5])
    filt = (ys.repeat(64) != 0) | (uvs.repeat(64) != 0)
    dir2 = hip.to_host(d_dir2, (nfb * 64,), np.uint8)
    assert dir1[filt].any()
    hip.free(*d_in, *d_out, *d_out2, *d_src, d_skip, d_ys, d_uvs, d_dir, d_dir2, d_var, d_mse)


def test_search_1080p_band(hip, orc):
    """Full 1080p frame on the GPU; the oracle checks one row of filter blocks bit-exactly."""
    src, rec, skip8 = cc.make_frame(1920, 1080, 8, seed=77)
    got = gpu_search(hip, rec, src, 8, skip8, 4)
    nh = 30
    exp = cc.orc_search(orc, rec, src, 8, skip8, 4, fb_begin=7 * nh, fb_end=8 * nh)
    assert np.array_equal(got[:, 7 * nh:8 * nh], exp[:, 7 * nh:8 * nh])
    assert got[0].any() and got[1].any()
