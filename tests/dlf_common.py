"""Synthetic mode-info grids and edge descriptors for the deblocking tests / bench (SURVEY.md 8(d),
config 3 (iii)): per 64x64 SB a square transform size from {4,8,16,32,64}, skip flag p = 0.3,
filter levels (Y v/h, U, V) = (20, 20, 12, 12)."""
import ctypes as C

import numpy as np

from conftest import load_package, ptr

pkg = load_package()


def make_mode_info(width, height, seed=15, levels=(20, 20, 12, 12), varied=False):
    rng = np.random.default_rng(seed)
    cols, rows = (width + 3) // 4, (height + 3) // 4
    mi = (pkg.DlfModeInfo * (cols * rows))()
    for sy in range(0, rows, 16):
        for sx in range(0, cols, 16):
            tl = int(rng.integers(2, 7))              # tx log2 for this SB
            bl = max(tl, int(rng.integers(2, 7)))     # prediction block >= tx block
            for by in range(sy, min(sy + 16, rows), 1 << (bl - 2)):
                for bx in range(sx, min(sx + 16, cols), 1 << (bl - 2)):
                    skip = 1 if rng.random() < 0.3 else 0
                    lv = levels if not varied else tuple(int(v) for v in rng.integers(0, 64, 4))
                    for y in range(by, min(by + (1 << (bl - 2)), rows)):
                        for x in range(bx, min(bx + (1 << (bl - 2)), cols)):
                            m = mi[y * cols + x]
                            m.tx_w_log2 = m.tx_h_log2 = tl
                            m.uv_tx_w_log2 = m.uv_tx_h_log2 = min(max(tl - 1, 2), 5)
                            m.bw_log2 = m.bh_log2 = bl
                            m.skip_inter = skip
                            m.level[0][0], m.level[0][1] = lv[0], lv[1]
                            m.level[1][0] = m.level[1][1] = lv[2]
                            m.level[2][0] = m.level[2][1] = lv[3]
    return mi, cols, rows


def build_edges(mi, cols, rows, plane, pw, ph):
    ss = 0 if plane == 0 else 1
    uw, uh = (pw + 3) // 4, (ph + 3) // 4
    ev = np.zeros((uh, uw), np.uint16); eh = np.zeros((uh, uw), np.uint16)
    rc = pkg.lib().svt_hip_dlf_build_edges(C.cast(mi, C.c_void_p), cols, rows, plane, ss, ss, pw, ph, ptr(ev), ptr(eh))
    assert rc == 0
    return ev, eh
