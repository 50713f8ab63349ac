"""CPU: oracle restatement vs the REAL reference C functions (oracle/_ref, only where it was built —
i.e. in the build container).  Input distributions mirror /root/reference/test/SadTest.cc:
REF_MAX / SRC_MAX / RANDOM / ties (:194-246), Allsad/Extsad kernels (:838-1222), sad_LoopTest (:611-785)."""
import ctypes as C

import numpy as np

from conftest import ptr


def _pair(rng, it):
    src = rng.integers(0, 256, (64, 80), dtype=np.uint8)
    refb = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    if it % 5 == 1: src[:] = 255
    if it % 5 == 2: refb[:] = 255
    if it % 5 == 3: refb[:] = src[:, :1]
    return src, refb


def test_ext_all_sad_and_32x32_64x64(orc, ref):
    rng = np.random.default_rng(13596)
    for it in range(120):
        sub = it & 1
        src, refb = _pair(rng, it)
        mv = int(rng.integers(0, 2 ** 32))
        outs = []
        for lib, n8, n32 in ((ref, "svt_ext_all_sad_calculation_8x8_16x16_c", "svt_ext_eight_sad_calculation_32x32_64x64_c"),
                             (orc, "orc_ext_all_sad_8x8_16x16", "orc_ext_eight_sad_32x32_64x64")):
            r2 = np.random.default_rng(it)
            bs8 = r2.integers(0, 20000, 64, dtype=np.uint32); bs16 = r2.integers(0, 70000, 16, dtype=np.uint32)
            bm8 = np.zeros(64, np.uint32); bm16 = np.zeros(16, np.uint32)
            e16 = np.zeros((16, 8), np.uint32); e8 = np.zeros((64, 8), np.uint32)
            getattr(lib, n8)(ptr(src), 80, ptr(refb), 96, C.c_uint32(mv), ptr(bs8), ptr(bs16), ptr(bm8), ptr(bm16), ptr(e16), ptr(e8), sub)
            bs32 = r2.integers(0, 270000, 4, dtype=np.uint32); bs64 = r2.integers(0, 1100000, 1, dtype=np.uint32)
            bm32 = np.zeros(4, np.uint32); bm64 = np.zeros(1, np.uint32); e32 = np.zeros((4, 8), np.uint32)
            getattr(lib, n32)(ptr(e16), ptr(bs32), ptr(bs64), ptr(bm32), ptr(bm64), C.c_uint32(mv), ptr(e32))
            outs.append([bs8, bs16, bm8, bm16, e16, e8, bs32, bs64, bm32, bm64, e32])
        for a, b in zip(*outs):
            assert np.array_equal(a, b)


def test_single_candidate_kernels(orc, ref):
    rng = np.random.default_rng(7)
    for it in range(60):
        sub = it & 1
        src, refb = _pair(rng, it)
        mv = int(rng.integers(0, 2 ** 32))
        outs = []
        for lib, n8, n32 in ((ref, "svt_ext_sad_calculation_8x8_16x16_c", "svt_ext_sad_calculation_32x32_64x64_c"),
                             (orc, "orc_ext_sad_8x8_16x16", "orc_ext_sad_32x32_64x64")):
            r2 = np.random.default_rng(it)
            bs8 = r2.integers(0, 20000, 4, dtype=np.uint32); bs16 = r2.integers(0, 70000, 1, dtype=np.uint32)
            bm8 = np.zeros(4, np.uint32); bm16 = np.zeros(1, np.uint32); s16 = np.zeros(1, np.uint32); s8 = np.zeros(4, np.uint32)
            getattr(lib, n8)(ptr(src), 80, ptr(refb), 96, ptr(bs8), ptr(bs16), ptr(bm8), ptr(bm16), C.c_uint32(mv), ptr(s16), ptr(s8), sub)
            sad16 = r2.integers(0, 60000, 16, dtype=np.uint32)
            bs32 = r2.integers(0, 200000, 4, dtype=np.uint32); bs64 = r2.integers(0, 900000, 1, dtype=np.uint32)
            bm32 = np.zeros(4, np.uint32); bm64 = np.zeros(1, np.uint32); s32 = np.zeros(4, np.uint32)
            getattr(lib, n32)(ptr(sad16), ptr(bs32), ptr(bs64), ptr(bm32), ptr(bm64), C.c_uint32(mv), ptr(s32))
            outs.append([bs8, bs16, bm8, bm16, s16, s8, bs32, bs64, bm32, bm64, s32])
        for a, b in zip(*outs):
            assert np.array_equal(a, b)


def test_sad_loop_and_nxm(orc, ref):
    rng = np.random.default_rng(1)
    orc.orc_nxm_sad.restype = C.c_uint32
    ref.svt_fast_loop_nxm_sad_kernel.restype = C.c_uint32
    for it in range(80):
        bw = int(rng.choice([4, 8, 16, 24, 32, 48, 64])); bh = int(rng.choice([4, 8, 16, 32, 64]))
        saw = int(rng.integers(1, 40)); sah = int(rng.integers(1, 20))
        src = rng.integers(0, 256, (bh, bw), dtype=np.uint8); rs = bw + saw + 5
        refb = rng.integers(0, 256, (bh + sah + 2, rs), dtype=np.uint8)
        if it % 4 == 0: refb[:] = 7; src[:] = 9
        res = []
        for lib, name in ((ref, "svt_sad_loop_kernel_c"), (orc, "orc_sad_loop")):
            best = C.c_uint64(0); xc = C.c_int16(-5); yc = C.c_int16(-7)
            getattr(lib, name)(ptr(src), bw, ptr(refb), rs, bh, bw, C.byref(best), C.byref(xc), C.byref(yc), rs, C.c_int16(saw), C.c_int16(sah))
            res.append((best.value, xc.value, yc.value))
        assert res[0] == res[1]
        assert ref.svt_fast_loop_nxm_sad_kernel(ptr(src), bw, ptr(refb), rs, bh, bw) == orc.orc_nxm_sad(ptr(src), bw, ptr(refb), rs, bh, bw)
