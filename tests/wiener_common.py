"""Restoration-unit statistics for the Wiener initial-filter tests (CPU: oracle vs the reference, GPU: device vs oracle)."""
import numpy as np

from conftest import ptr


def wiener_unit_stats(orc, rng, win, bd, kind, size=64):
    """M, H of one unit: the degraded picture is the source blurred / sharpened / noised in different ways, so that the solved taps cover their ranges (clamps included)"""
    dt = np.uint8 if bd == 8 else np.uint16
    top = (1 << bd) - 1
    yy, xx = np.mgrid[0:size + 16, 0:size + 16]
    base = (0.5 + 0.25 * np.sin(xx / rng.uniform(2, 9)) + 0.25 * np.cos(yy / rng.uniform(2, 9))) * top * rng.uniform(0.3, 1.0) + rng.normal(0, top / 40.0, xx.shape)
    if kind == 0:      # degraded = blurred source: the filter sharpens (large positive outer taps clamp)
        k = np.array([1, 2, 4, 2, 1], np.float64); k /= k.sum()
        deg = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, base); deg = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, deg)
    elif kind == 1:    # degraded = source + noise: the filter smooths
        deg = base + rng.normal(0, top / 12.0, base.shape)
    elif kind == 2:    # degraded = source: the solved filter is the identity
        deg = base.copy()
    elif kind == 3:    # flat unit: singular systems
        base = np.full(base.shape, top // 3, np.float64); deg = base.copy()
    else:              # unrelated pictures
        deg = rng.integers(0, top + 1, base.shape).astype(np.float64)
    src = np.clip(np.rint(base), 0, top).astype(dt); dgd = np.clip(np.rint(deg), 0, top).astype(dt)
    w2 = win * win
    M, H = np.zeros(w2, np.int64), np.zeros(w2 * w2, np.int64)
    orc.orc_wiener_compute_stats(win, ptr(dgd), ptr(src), dgd.itemsize, bd, 8, 8 + size, 8, 8 + size, dgd.shape[1], src.shape[1], ptr(M), ptr(H))
    return M, H
