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


_orc = None


def oracle():
    """oracle/liboracle.so — the checker; the expectations of the deblocking tests come from its restatement of set_lpf_parameters (pinned to the reference's frame loop by
    tests/test_oracle_vs_ref.py::test_deblocking_edges_of_a_frame), never from the product's own builders"""
    global _orc
    if _orc is None:
        import os
        import subprocess
        from conftest import ROOT
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
        _orc = C.CDLL(path)
    return _orc


def records(mi, cols, rows):
    """the grid as the 13-byte records both the product's builders and the oracle read"""
    raw = np.frombuffer(mi, np.uint8) if not isinstance(mi, np.ndarray) else mi
    assert raw.size == 13 * cols * rows
    return np.ascontiguousarray(raw.reshape(-1, 13))


def build_edges(mi, cols, rows, plane, pw, ph, filt_w=None, filt_h=None):
    """edge planes of one plane by the ORACLE (orc_dlf_build_edges); filt_w / filt_h: the units the frame loop visits (default: all)"""
    ss = 0 if plane == 0 else 1
    uw, uh = (pw + 3) // 4, (ph + 3) // 4
    ev = np.zeros((uh, uw), np.uint16); eh = np.zeros((uh, uw), np.uint16)
    raw = records(mi, cols, rows)
    oracle().orc_dlf_build_edges(ptr(raw), cols, rows, plane, ss, ss, pw, ph, uw if filt_w is None else filt_w, uh if filt_h is None else filt_h, ptr(ev), ptr(eh))
    return ev, eh


def product_host_edges(mi, cols, rows, plane, pw, ph, filt_w=None, filt_h=None):
    """the same planes by the PRODUCT's host builder (svt_hip_dlf_build_edges[_crop], svt-av1_amd/csrc/svt_hip_host.cpp): what the tests compare with the oracle"""
    ss = 0 if plane == 0 else 1
    uw, uh = (pw + 3) // 4, (ph + 3) // 4
    ev = np.zeros((uh, uw), np.uint16); eh = np.zeros((uh, uw), np.uint16)
    raw = records(mi, cols, rows)
    L = pkg.lib()
    L.svt_hip_dlf_build_edges_crop.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.c_void_p] * 2
    assert L.svt_hip_dlf_build_edges_crop(ptr(raw), cols, rows, plane, ss, ss, pw, ph, uw if filt_w is None else filt_w, uh if filt_h is None else filt_h, ptr(ev), ptr(eh)) == 0
    return ev, eh


# ------------------------------------------------------------------------------------------------------------------------------------------------------
# The reference's own mode-info fields per 4x4 unit (E2 pin): random AV1 partitions of every superblock, a transform depth, inter / intra, skip and a
# prediction mode per block.  BlockSize enum order (Common/Codec/EbDefinitions.h).
BS_DIMS = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64), (128, 128),
           (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
BS_INDEX = {d: i for i, d in enumerate(BS_DIMS)}


def _partition(rng, x, y, n, out, depth_bias):
    """blocks (x, y, w, h) of a square node of size n, AV1 partition types"""
    kinds = ["none", "split", "horz", "vert"] + (["horz4", "vert4"] if 16 <= n <= 64 else []) + (["horz_a", "vert_a"] if n >= 16 else [])
    if n == 128: kinds = ["none", "split", "split", "horz", "vert"]
    k = kinds[int(rng.integers(0, len(kinds)))] if rng.random() > depth_bias or n <= 8 else "split"
    h2 = n // 2
    if k == "none": out.append((x, y, n, n))
    elif k == "horz": out += [(x, y, n, h2), (x, y + h2, n, h2)]
    elif k == "vert": out += [(x, y, h2, n), (x + h2, y, h2, n)]
    elif k == "horz4": out += [(x, y + i * n // 4, n, n // 4) for i in range(4)]
    elif k == "vert4": out += [(x + i * n // 4, y, n // 4, n) for i in range(4)]
    elif k == "horz_a": out += [(x, y, h2, h2), (x + h2, y, h2, h2), (x, y + h2, n, h2)]
    elif k == "vert_a": out += [(x, y, h2, h2), (x, y + h2, h2, h2), (x + h2, y, h2, n)]
    else:
        if n == 8: out += [(x + dx, y + dy, 4, 4) for dy in (0, 4) for dx in (0, 4)]
        else:
            for dy in (0, h2):
                for dx in (0, h2): _partition(rng, x + dx, y + dy, h2, out, depth_bias * 0.6)


def make_reference_mode_info(rng, w, h, sb_size, p_skip=0.4, p_inter=0.7):
    """-> dict of [h / 4][w / 4] uint8 arrays: sb_type, tx_depth, ref_frame0 (0 = intra, 1..7), skip, mode (intra 0..12, inter 13..24, a few INTRA_MODE_4x4 = 25)"""
    cols, rows = w // 4, h // 4
    f = {k: np.zeros((rows, cols), np.uint8) for k in ("sb_type", "tx_depth", "ref_frame0", "skip", "mode")}
    for sy in range(0, h, sb_size):
        for sx in range(0, w, sb_size):
            blocks = []
            _partition(rng, sx, sy, sb_size, blocks, 0.5)
            for (x, y, bw, bh) in blocks:
                if x >= w or y >= h: continue
                inter = rng.random() < p_inter
                vals = dict(sb_type=BS_INDEX[(bw, bh)], tx_depth=int(rng.integers(0, 3)), ref_frame0=int(rng.integers(1, 8)) if inter else 0,
                            skip=int(rng.random() < p_skip), mode=int(rng.integers(13, 25)) if inter else (25 if rng.random() < 0.05 else int(rng.integers(0, 13))))
                for k, v in vals.items():
                    f[k][y // 4:min(y + bh, h) // 4, x // 4:min(x + bw, w) // 4] = v
    return f


def oracle_edges(orc, f, w, h, lf, pad_right=0, pad_bottom=0, sb_size=64):
    """summary grid (the product's SvtHipDlfModeInfo records, 13 bytes each), the oracle's edge planes [plane] = (edges_v, edges_h) and the level table"""
    cols, rows = w // 4, h // 4
    lfa = np.array(lf, np.int32)
    lvl = np.zeros((3, 2, 8, 2), np.uint8)
    orc.orc_dlf_level_table(ptr(lfa), ptr(lvl))
    summ = np.zeros((rows * cols, 13), np.uint8)
    orc.orc_dlf_mode_info_summary(rows * cols, ptr(f["sb_type"]), ptr(f["tx_depth"]), ptr(f["ref_frame0"]), ptr(f["skip"]), ptr(f["mode"]), ptr(lvl), ptr(summ))
    edges = []
    for plane in range(3):
        ss = int(plane > 0)
        pw, ph = w >> ss, h >> ss
        fw = orc.orc_dlf_filtered_units(w, pad_right, sb_size, ss); fh = orc.orc_dlf_filtered_units(h, pad_bottom, sb_size, ss)
        ev = np.zeros(((ph + 3) // 4, (pw + 3) // 4), np.uint16); eh = np.zeros_like(ev)
        orc.orc_dlf_build_edges(ptr(summ), cols, rows, plane, ss, ss, pw, ph, fw, fh, ptr(ev), ptr(eh))
        edges.append((ev, eh))
    return summ, edges, lvl
