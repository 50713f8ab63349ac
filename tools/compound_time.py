"""Times svt_hip_compound_predict_batch_dev (every 16x16 block of a 2160p luma plane, a mix of the four compound types and of phases) and
svt_hip_obmc_cost_batch_dev (every 16x16 block) with HIP events."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package  # noqa: E402
import comp_common as cmc  # noqa: E402
from test_compound_gpu import ObmcBlk  # noqa: E402

pkg = load_package()
hip = pkg.Context(0)
W, H, PAD = 3840, 2160, 64
rng = np.random.default_rng(0)
for bd in (8, 10):
    dt = np.uint8 if bd == 8 else np.uint16
    ref0 = rng.integers(0, 1 << bd, (H + 2 * PAD, W + 2 * PAD)).astype(dt); ref1 = rng.integers(0, 1 << bd, ref0.shape).astype(dt)
    n = (W // 16) * (H // 16)
    blks = (cmc.CompBlk * n)()
    masks = rng.integers(0, 65, n * 256).astype(np.uint8)
    i = 0
    for by in range(0, H, 16):
        for bx in range(0, W, 16):
            b = blks[i]
            b.dst_x, b.dst_y, b.w, b.h = bx, by, 16, 16
            b.src0_x, b.src0_y = PAD + bx + int(rng.integers(-8, 9)), PAD + by + int(rng.integers(-8, 9))
            b.src1_x, b.src1_y = PAD + bx + int(rng.integers(-8, 9)), PAD + by + int(rng.integers(-8, 9))
            b.subpel0_x, b.subpel0_y, b.subpel1_x, b.subpel1_y = [int(v) for v in rng.integers(1, 16, 4)]
            b.type = i % 4; b.fwd_offset, b.bck_offset = 9, 7
            b.mask_off, b.mask_stride = i * 256, 16
            i += 1
    d_r0, d_r1, d_dst, d_m = hip.to_device(ref0), hip.to_device(ref1), hip.to_device(np.zeros((H, W), dt)), hip.to_device(masks)
    d_b = hip.to_device(np.frombuffer(bytes(blks), np.uint8).copy())
    st = ref0.shape[1]
    call = lambda: hip.check(hip.L.svt_hip_compound_predict_batch_dev(hip.h, ref0.itemsize, bd, d_r0, st, d_r1, st, d_dst, W, d_m, d_b, n), "compound")
    for _ in range(3): call()
    ms = C.c_float(); hip.L.svt_hip_timer_start(hip.h)
    for _ in range(20): call()
    hip.L.svt_hip_timer_stop_ms(hip.h, C.byref(ms))
    t = ms.value / 20
    print(f"compound predict, {n} 16x16 blocks ({W}x{H}), bd {bd}, 2-D phases on both references: {t:.3f} ms  "
          f"{3 * W * H * ref0.itemsize / t / 1e6:.0f} GB/s algorithmic (2 reads + 1 write per sample)")
    hip.free(d_r0, d_r1, d_dst, d_m, d_b)
pre = rng.integers(0, 256, (H + 8, W + 8)).astype(np.uint8)
n = (W // 16) * (H // 16)
ob = (ObmcBlk * n)()
for i in range(n):
    ob[i] = ObmcBlk((i % (W // 16)) * 16, (i // (W // 16)) * 16, 16, 16, int(rng.integers(0, 8)), int(rng.integers(0, 8)), i * 256)
wsrc = rng.integers(0, 255 * 4096, n * 256).astype(np.int32); mask = rng.integers(0, 4097, n * 256).astype(np.int32)
d_pre, d_w, d_mk, d_ob, d_o = hip.to_device(pre), hip.to_device(wsrc), hip.to_device(mask), hip.to_device(np.frombuffer(bytes(ob), np.uint8).copy()), hip.empty(n * 12)
call = lambda: hip.check(hip.L.svt_hip_obmc_cost_batch_dev(hip.h, d_pre, pre.shape[1], d_w, d_mk, d_ob, n, d_o), "obmc")
for _ in range(3): call()
ms = C.c_float(); hip.L.svt_hip_timer_start(hip.h)
for _ in range(20): call()
hip.L.svt_hip_timer_stop_ms(hip.h, C.byref(ms))
t = ms.value / 20
print(f"OBMC sad + sub-pixel variance, {n} 16x16 blocks: {t:.3f} ms  {W * H * 9 / t / 1e6:.0f} GB/s algorithmic (1 + 4 + 4 bytes per sample)")
