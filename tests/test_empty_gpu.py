"""GPU: empty batches are legal and are no-ops (the reference's per-SB loops simply do not run): every batched entry point
returns 0 for a zero-length work list and leaves its outputs untouched."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_batches(hip, pkg):
    L = hip.L
    buf = hip.to_device(np.full(4096, 7, np.uint8))
    qs = pkg.QuantParams(); st = pkg.ScanTables(); st.iscan[0] = buf.value
    assert L.svt_hip_me_fullpel_frame_dev(hip.h, buf, buf, 64, 0, 0, buf, 0, 0, buf, buf) == 0
    for ts in (0, 4, 11, 18):
        assert L.svt_hip_fwd_txfm_quant_batch_dev(hip.h, ts, 1, buf, 64, buf, 64, buf, 0, C.byref(qs), C.byref(st), None, buf, buf, buf, buf, None) == 0
        assert L.svt_hip_inv_txfm_add_batch_dev(hip.h, ts, 1, 8, buf, buf, 64, buf, 64, buf, 0) == 0
    assert L.svt_hip_subpel_predict_batch_dev(hip.h, 1, 8, buf, 64, buf, 64, buf, 0) == 0
    assert L.svt_hip_block_sad_batch_dev(hip.h, 1, buf, 64, buf, 64, buf, 0, buf) == 0
    assert L.svt_hip_block_variance_batch_dev(hip.h, 1, 8, buf, 64, buf, 64, buf, 0, buf, buf) == 0
    assert L.svt_hip_sad_loop_batch_dev(hip.h, buf, 64, buf, 64, buf, 0, buf, buf) == 0
    assert L.svt_hip_variance_pyramid_dev(hip.h, buf, 64, 1, 0, 0, buf, buf) == 0
    assert L.svt_hip_deblock_plane_dev(hip.h, buf, 1, 64, 8, buf, buf, 0, 0, 0) == 0
    hip.check(L.svt_hip_sync(hip.h))
    assert (hip.to_host(buf, (4096,), np.uint8) == 7).all()
    hip.free(buf)


def test_memcpy_d2d(hip):
    a = np.arange(5000, dtype=np.uint8)
    d_a, d_b = hip.to_device(a), hip.empty(a.nbytes)
    hip.check(hip.L.svt_hip_memcpy_d2d(hip.h, d_b, d_a, a.nbytes), "d2d")
    assert np.array_equal(hip.to_host(d_b, a.shape, np.uint8), a)
    assert hip.L.svt_hip_memcpy_d2d(hip.h, None, None, 0) == 0 and hip.L.svt_hip_memcpy_d2d(hip.h, None, d_a, 4) != 0
    hip.free(d_a, d_b)
