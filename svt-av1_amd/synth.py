"""Deterministic synthetic 4:2:0 frames for parity tests and bench.py.

cur  = smooth gradient + gaussian-ish noise; ref = cur displaced by a per-64x64-region motion
field + noise, so SAD surfaces have clear minima plus some ties.  Planes are padded by edge
replication exactly like the reference pads its ME input pictures (pad = sb_size + ME_FILTER_TAP
= 68, Source/Lib/Encoder/Globals/EbEncHandle.c:1030-1033).
"""
import numpy as np

PAD = 68


def make_luma_pair(width, height, seed=1, max_mv=24, noise=12.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    base = (64 + 96 * (np.sin(xx / 37.0) * np.cos(yy / 53.0) + 1) / 2 + 0.02 * xx + 0.03 * yy)
    tex = rng.normal(0, noise, (height, width))
    cur = np.clip(base + tex, 0, 255).astype(np.uint8)
    ref = np.empty_like(cur)
    big = np.pad(cur, max_mv + 1, mode="edge")
    for by in range(0, height, 64):
        for bx in range(0, width, 64):
            mvx, mvy = rng.integers(-max_mv, max_mv + 1, 2)
            h = min(64, height - by)
            w = min(64, width - bx)
            y0, x0 = by + max_mv + 1 + mvy, bx + max_mv + 1 + mvx
            ref[by:by + h, bx:bx + w] = big[y0:y0 + h, x0:x0 + w]
    ref = np.clip(ref.astype(np.int16) + rng.integers(-3, 4, ref.shape), 0, 255).astype(np.uint8)
    return cur, ref


def pad_plane(p, pad=PAD):
    """Edge-replicate; returned stride is width + 2*pad (a multiple of 4 for even widths)."""
    return np.ascontiguousarray(np.pad(p, pad, mode="edge"))


def sb_grid(width, height):
    return [(x, y) for y in range(0, height, 64) for x in range(0, width, 64)]
