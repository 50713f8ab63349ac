"""Shared inputs for the temporal-filter tests: the MeContext TF fields of every 64x64 block (OrcTfBlk64 == SvtHipTfBlk64 layout),
synthetic central / motion-compensated pictures, and the reference-test style extremes
(/root/reference/test/TemporalFilterTestPlanewise.cc:200-330: random pixels, random block errors, random noise levels)."""
import ctypes as C

import numpy as np


class TfBlk64(C.Structure):
    _fields_ = [("mv16_x", C.c_int16 * 16), ("mv16_y", C.c_int16 * 16), ("err16", C.c_uint64 * 16),
                ("mv32_x", C.c_int16 * 4), ("mv32_y", C.c_int16 * 4), ("err32", C.c_uint64 * 4), ("split", C.c_int32 * 4)]


class TfRef(C.Structure):
    _fields_ = [("pred", C.c_void_p * 3), ("pred_stride", C.c_int * 3), ("blocks", C.c_void_p)]


BLK_DTYPE = np.dtype([("mv16_x", np.int16, 16), ("mv16_y", np.int16, 16), ("err16", np.uint64, 16),
                      ("mv32_x", np.int16, 4), ("mv32_y", np.int16, 4), ("err32", np.uint64, 4), ("split", np.int32, 4)], align=True)
assert BLK_DTYPE.itemsize == C.sizeof(TfBlk64) == 256


def make_blocks(rng, n, bd, big_mv=False, err_max=60):
    b = np.zeros(n, BLK_DTYPE)
    sc = 16 if bd > 8 else 1
    b["err16"] = rng.integers(0, 256 * err_max + 1, (n, 16)) * sc          # mean squared error up to err_max per pixel
    b["err32"] = rng.integers(0, 1024 * err_max + 1, (n, 4)) * sc
    hi = 700 if big_mv else 40
    for k in ("mv16_x", "mv16_y"): b[k] = rng.integers(-hi, hi + 1, (n, 16))
    for k in ("mv32_x", "mv32_y"): b[k] = rng.integers(-hi, hi + 1, (n, 4))
    b["split"] = rng.integers(0, 2, (n, 4))
    if n > 2:
        b["err16"][1] = 0; b["err32"][1] = 0; b["mv16_x"][1] = 0; b["mv16_y"][1] = 0; b["mv32_x"][1] = 0; b["mv32_y"][1] = 0   # perfect match
        b["err16"][2] = np.uint64(1) << np.uint64(40); b["err32"][2] = np.uint64(1) << np.uint64(40)                          # hopeless block
    return b


def make_pictures(rng, w, h, bd, ss_x, ss_y, n_refs, noise=6.0):
    """Central picture + n_refs 'motion compensated' pictures (central + noise of varying strength, a few saturated / flat regions)."""
    dt = np.uint8 if bd == 8 else np.uint16
    mx = (1 << bd) - 1
    yy, xx = np.mgrid[0:h, 0:w]
    base = (110 + 70 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 25 * (((xx // 16) + (yy // 16)) % 2)) * (1 << (bd - 8))
    planes = [base, base[::1 << ss_y, ::1 << ss_x] * 0.5 + 60 * (1 << (bd - 8)), 200 * (1 << (bd - 8)) - base[::1 << ss_y, ::1 << ss_x] * 0.4]
    src = [np.ascontiguousarray(np.clip(p + rng.normal(0, 2 * (1 << (bd - 8)), p.shape), 0, mx).astype(dt)) for p in planes]
    src[0][:40, :40] = mx; src[0][40:64, :64] = 0
    preds = []
    for f in range(n_refs):
        sg = noise * (0.3 + f) * (1 << (bd - 8))
        pr = [np.ascontiguousarray(np.clip(s.astype(np.float64) + rng.normal(0, sg, s.shape), 0, mx).astype(dt)) for s in src]
        if f == 0: pr[0][:32, :32] = 0                     # maximum squared differences in one 32x32 block
        preds.append(pr)
    return src, preds
