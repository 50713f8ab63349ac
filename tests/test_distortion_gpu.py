"""GPU parity of the RD-side distortion reductions: coefficient-domain residual / prediction distortion and SATD for
lists of transform blocks, pixel-domain SSE for lists of block pairs (8- and 16-bit) vs the oracle (pinned to
svt_full_distortion_kernel32_bits_c, svt_av1_block_error_c, svt_aom_satd_c, svt_aom_sse_c in tests/test_oracle_vs_ref.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import ptr

pytestmark = pytest.mark.gpu


def test_coeff_distortion(hip, orc):
    rng = np.random.default_rng(21)
    for n, nblk in ((16, 1000), (64, 333), (1024, 77), (100, 5)):
        c = rng.integers(-(1 << 20), 1 << 20, (nblk, n)).astype(np.int32)
        r = (c + rng.integers(-5000, 5000, (nblk, n))).astype(np.int32)
        c[0] = np.iinfo(np.int32).max // 2; r[0] = -(np.iinfo(np.int32).max // 2)     # 64-bit range of the squares
        exp = np.zeros((nblk, 3), np.uint64); exp0 = np.zeros((nblk, 3), np.uint64)
        for i in range(nblk):
            orc.orc_coeff_distortion(ptr(c[i]), ptr(r[i]), n, ptr(exp[i]))
            orc.orc_coeff_distortion(ptr(c[i]), None, n, ptr(exp0[i]))
        d_c, d_r, d_o = hip.to_device(c), hip.to_device(r), hip.empty(nblk * 24)
        hip.check(hip.L.svt_hip_coeff_distortion_batch_dev(hip.h, d_c, d_r, n, nblk, d_o))
        assert np.array_equal(hip.to_host(d_o, (nblk, 3), np.uint64), exp)
        hip.check(hip.L.svt_hip_coeff_distortion_batch_dev(hip.h, d_c, None, n, nblk, d_o))
        assert np.array_equal(hip.to_host(d_o, (nblk, 3), np.uint64), exp0)
        hip.free(d_c, d_r, d_o)


@pytest.mark.parametrize("bd", [8, 10])
def test_block_sse(hip, pkg, orc, bd):
    rng = np.random.default_rng(22 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    a = rng.integers(0, 1 << bd, (300, 420)).astype(dt); b = rng.integers(0, 1 << bd, (310, 400)).astype(dt)
    n = 300
    pairs = []
    for i in range(n):
        w = int(rng.choice([4, 8, 16, 32, 64, 128, 7, 33])); h = int(rng.choice([4, 8, 16, 32, 64, 128, 5]))
        pairs.append((int(rng.integers(0, 420 - w)), int(rng.integers(0, 300 - h)), int(rng.integers(0, 400 - w)), int(rng.integers(0, 310 - h)), w, h))
    P = (pkg.BlkPair * n)(*[pkg.BlkPair(*p) for p in pairs])
    orc.orc_plane_sse.restype = C.c_uint64
    exp = np.array([orc.orc_plane_sse(a.itemsize, C.c_void_p(a.ctypes.data + (ay * 420 + ax) * a.itemsize), 420,
                                      C.c_void_p(b.ctypes.data + (by * 400 + bx) * b.itemsize), 400, w, h) for (ax, ay, bx, by, w, h) in pairs], np.uint64)
    d_a, d_b, d_p, d_o = hip.to_device(a), hip.to_device(b), hip.to_device(np.frombuffer(bytes(P), np.uint8)), hip.empty(n * 8)
    hip.check(hip.L.svt_hip_block_sse_batch_dev(hip.h, a.itemsize, d_a, 420, d_b, 400, d_p, n, d_o))
    assert np.array_equal(hip.to_host(d_o, (n,), np.uint64), exp)
    hip.free(d_a, d_b, d_p, d_o)
