"""Shared definitions for the compound-prediction tests: the SvtHipCompBlk / OrcCompBlk record and random block lists."""
import ctypes as C

import numpy as np


class CompBlk(C.Structure):
    _fields_ = [("src0_x", C.c_int32), ("src0_y", C.c_int32), ("src1_x", C.c_int32), ("src1_y", C.c_int32), ("dst_x", C.c_int32), ("dst_y", C.c_int32),
                ("w", C.c_uint8), ("h", C.c_uint8), ("bank_x", C.c_uint8), ("bank_y", C.c_uint8),
                ("subpel0_x", C.c_uint8), ("subpel0_y", C.c_uint8), ("subpel1_x", C.c_uint8), ("subpel1_y", C.c_uint8),
                ("type", C.c_uint8), ("fwd_offset", C.c_uint8), ("bck_offset", C.c_uint8), ("mask_type", C.c_uint8),
                ("mask_sub", C.c_uint8), ("reserved", C.c_uint8 * 3), ("mask_off", C.c_int32), ("mask_stride", C.c_int32)]


assert C.sizeof(CompBlk) == 48
SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (4, 8), (8, 4), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (128, 64), (64, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
# quant_dist_lookup_table[order_idx][..] pairs the encoder uses for COMPOUND_DISTANCE (fwd + bck = 16)
DIST_PAIRS = [(9, 7), (11, 5), (12, 4), (13, 3), (7, 9), (5, 11), (4, 12), (3, 13)]


def make_blocks(rng, W, H, n, mask_bytes):
    """n random compound blocks tiling nothing in particular: destinations are disjoint cells of a 128-pixel grid.  Returns (array, mask buffer)."""
    cols = W // 128
    assert n <= cols * (H // 128)
    blks = (CompBlk * n)()
    masks = rng.integers(0, 65, mask_bytes).astype(np.uint8)
    off = 0
    for i in range(n):
        w, h = SIZES[i % len(SIZES)]
        cx, cy = (i % cols) * 128, (i // cols) * 128
        b = blks[i]
        b.dst_x, b.dst_y, b.w, b.h = cx, cy, w, h
        b.src0_x, b.src0_y = cx + int(rng.integers(-6, 7)), cy + int(rng.integers(-6, 7))
        b.src1_x, b.src1_y = cx + int(rng.integers(-6, 7)), cy + int(rng.integers(-6, 7))
        b.bank_x = b.bank_y = int(rng.integers(0, 4)) if w > 4 and h > 4 else int(rng.integers(4, 6))
        if i % 7 == 3: b.bank_y = int(rng.integers(0, 3))          # dual filters
        sp = [int(rng.integers(0, 16)) for _ in range(4)]
        if i % 5 == 0: sp[0] = 0
        if i % 5 == 1: sp[1] = 0
        if i % 11 == 2: sp = [0, 0, sp[2], 0]
        b.subpel0_x, b.subpel0_y, b.subpel1_x, b.subpel1_y = sp
        b.type = i % 4
        b.fwd_offset, b.bck_offset = DIST_PAIRS[i % len(DIST_PAIRS)]
        b.mask_type = (i // 4) & 1
        b.mask_sub = 0
        b.mask_off, b.mask_stride = -1, 0
        if b.type == 2 and i % 8 != 2:
            b.mask_off = off; off += w * h
        if b.type == 3:
            b.mask_sub = (i // 4) % 2 if max(w, h) <= 64 else 0
            ms = (2 * w if b.mask_sub else w) + int(rng.integers(0, 9))
            b.mask_off, b.mask_stride = off, ms
            off += ms * (2 * h if b.mask_sub else h)
        assert off <= mask_bytes
    return blks, masks


# ---------------------------------------------------------------- warped prediction
class WarpBlk(C.Structure):
    _fields_ = [("mat", C.c_int32 * 6), ("alpha", C.c_int16), ("beta", C.c_int16), ("gamma", C.c_int16), ("delta", C.c_int16),
                ("p_col", C.c_int32), ("p_row", C.c_int32), ("p_width", C.c_uint8), ("p_height", C.c_uint8), ("reserved", C.c_uint8 * 2)]


assert C.sizeof(WarpBlk) == 44
WARP_SIZES = [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (8, 16), (16, 8), (32, 16), (16, 32), (64, 32), (32, 64), (128, 64), (64, 128), (8, 32), (32, 8), (16, 64), (64, 16)]


def warp_model(rng, extreme=False):
    """A model in the style of /root/reference/test/warp_filter_test_util.cc:47-110: random matrix around identity, shear parameters rounded to
    multiples of 64 and inside is_affine_shear_allowed's bounds (4|a| + 7|b| < 2^16, 4|g| + 4|d| < 2^16)."""
    while True:
        lim = (1 << 13) - 64 if not extreme else 1 << 13
        a, b, g, d = [int(rng.integers(-lim, lim + 1)) // 64 * 64 for _ in range(4)]
        if extreme: a, b, g, d = [int(v) for v in rng.choice([-8192, 8192, -4096, 0, 4032], 4)]
        if 4 * abs(a) + 7 * abs(b) >= (1 << 16) or 4 * abs(g) + 4 * abs(d) >= (1 << 16): continue
        m2 = (1 << 16) + a; m3 = b
        m4 = (g * m2) >> 16
        m5 = (1 << 16) + d + (m3 * m4) // m2
        m0, m1 = int(rng.integers(-(1 << 21), 1 << 21)), int(rng.integers(-(1 << 21), 1 << 21))
        return [m0, m1, m2, m3, m4, m5], a, b, g, d


def warp_blocks(rng, W, H, n):
    cols = W // 128
    assert n <= cols * (H // 128)
    blks = (WarpBlk * n)()
    for i in range(n):
        mat, a, b, g, d = warp_model(rng, extreme=(i % 9 == 4))
        if i % 7 == 0: mat[0] += 500 << 16        # far outside the plane: every sample clamps to the right edge
        if i % 7 == 1: mat[1] -= 400 << 16        # ... to the top edge
        w, h = WARP_SIZES[i % len(WARP_SIZES)]
        bl = blks[i]
        for k in range(6): bl.mat[k] = mat[k]
        bl.alpha, bl.beta, bl.gamma, bl.delta = a, b, g, d
        bl.p_col, bl.p_row, bl.p_width, bl.p_height = (i % cols) * 128, (i // cols) * 128, w, h
    return blks


# ---------------------------------------------------------------- pixel-domain mask blends
class BlendBlk(C.Structure):
    _fields_ = [("src0_x", C.c_int32), ("src0_y", C.c_int32), ("src1_x", C.c_int32), ("src1_y", C.c_int32), ("dst_x", C.c_int32), ("dst_y", C.c_int32),
                ("w", C.c_uint8), ("h", C.c_uint8), ("mode", C.c_uint8), ("subw", C.c_uint8), ("subh", C.c_uint8), ("reserved", C.c_uint8 * 3),
                ("mask_off", C.c_int32), ("mask_stride", C.c_int32)]


assert C.sizeof(BlendBlk) == 40
BLEND_SIZES = [(1, 1), (2, 2), (4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (2, 8), (8, 2), (4, 16), (16, 4), (64, 8), (8, 64), (128, 32), (32, 128), (1, 16), (16, 1)]


def blend_blocks(rng, W, H, n, mask_bytes):
    cols = W // 128
    assert n <= cols * (H // 128)
    blks = (BlendBlk * n)()
    masks = rng.integers(0, 65, mask_bytes).astype(np.uint8)
    masks[:300] = 64; masks[300:600] = 0
    off = 0
    for i in range(n):
        w, h = BLEND_SIZES[i % len(BLEND_SIZES)]
        b = blks[i]
        cx, cy = (i % cols) * 128, (i // cols) * 128
        b.dst_x, b.dst_y, b.w, b.h = cx, cy, w, h
        b.src0_x, b.src0_y = (cx, cy) if i % 4 == 0 else (int(rng.integers(0, W - 128)), int(rng.integers(0, H - 128)))    # i % 4 == 0: in place (dst == src0)
        b.src1_x, b.src1_y = int(rng.integers(0, W - 128)), int(rng.integers(0, H - 128))
        b.mode = i % 3
        b.subw, b.subh = ((i // 3) & 1, (i // 6) & 1) if b.mode == 0 and max(w, h) <= 64 else (0, 0)
        if b.mode == 0:
            b.mask_stride = (w << b.subw) + int(rng.integers(0, 5))
            need = b.mask_stride * (h << b.subh)
        else:
            b.mask_stride, need = 0, (w if b.mode == 1 else h)
        b.mask_off = off; off += need
        assert off <= mask_bytes
    return blks, masks
