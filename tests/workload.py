"""Synthetic per-frame workload of the hot path: one 8-bit 4:2:0 frame,
its reference frame, a per-SB transform tiling, deblocking mode info, CDEF skip map / strengths.
Shared by bench.py (device side through the C ABI) and by the frame-level parity tests (oracle side).
Everything is deterministic in (width, height, seed)."""
import ctypes as C
import os

import numpy as np

from conftest import load_package, ptr
import txfm_common as tc
import dlf_common as dc

pkg = load_package()
synth = __import__("importlib").import_module("svt_av1_amd.synth")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TABLES = np.load(os.path.join(GOLD, "txfm_tables.npz"))

TX_SQ = {2: 0, 3: 1, 4: 2, 5: 3, 6: 4}  # log2 size -> TxSize of the square transform


class Frame:
    def __init__(self, width, height, seed=11, qindex=60):
        self.w, self.h, self.seed = width, height, seed
        rng = np.random.default_rng(seed + 1000)
        cur, ref = synth.make_luma_pair(width, height, seed=seed)
        self.pad = synth.PAD
        self.cur_y_p, self.ref_y_p = synth.pad_plane(cur), synth.pad_plane(ref)     # padded (ME)
        self.cur = [cur, None, None]
        self.ref = [ref, None, None]
        for i, (a, b) in enumerate(((1, 37), (2, 91))):
            self.cur[i + 1] = np.ascontiguousarray(np.clip(128 + (cur[::2, ::2].astype(np.int16) - 128) // (a + 1) + b % 7, 0, 255).astype(np.uint8))
            self.ref[i + 1] = np.ascontiguousarray(np.clip(128 + (ref[::2, ::2].astype(np.int16) - 128) // (a + 1) + b % 7, 0, 255).astype(np.uint8))
        self.sb_cols, self.sb_rows = (width + 63) // 64, (height + 63) // 64
        self.n_sb = self.sb_cols * self.sb_rows
        # ---- transform tiling: one square size per SB (luma log2 2..6, chroma one smaller, min 4x4)
        self.sb_tx = rng.integers(2, 7, self.n_sb)
        # a superblock the picture cuts (4K: the last row is 48 rows high, 1080p: 56) gets the largest size that still tiles its extent, so that EVERY sample of the
        # picture is coded: a step then rewrites the whole reconstruction and is idempotent (samples no block covered used to keep the previous step's deblocked values)
        for sb in range(self.n_sb):
            ew, eh = min(64, width - (sb % self.sb_cols) * 64), min(64, height - (sb // self.sb_cols) * 64)
            while (ew | eh) & ((1 << int(self.sb_tx[sb])) - 1):
                self.sb_tx[sb] -= 1
        self.sb_skip = rng.random(self.n_sb) < 0.0
        self.descs = {}  # (plane_kind, tx_size) -> uint32 descriptor array ; plane_kind 0 luma, 1 chroma (U and V share it)
        lists = {}
        for sb in range(self.n_sb):
            sx, sy = (sb % self.sb_cols) * 64, (sb // self.sb_cols) * 64
            for kind in (0, 1):
                l2 = int(self.sb_tx[sb]) if kind == 0 else max(2, int(self.sb_tx[sb]) - 1)
                ts = TX_SQ[l2]
                n = 1 << l2
                types = tc.legal_types(ts)
                x0, y0 = sx >> kind, sy >> kind
                pw, ph = width >> kind, height >> kind
                k = 0
                for y in range(y0, min(y0 + (64 >> kind), ph), n):
                    for x in range(x0, min(x0 + (64 >> kind), pw), n):
                        if x + n <= pw and y + n <= ph:
                            lists.setdefault((kind, ts), []).append(pkg.tx_desc(x, y, types[(sb + k) % len(types)]))
                            k += 1
        self.descs = {k: np.asarray(v, np.uint32) for k, v in lists.items()}
        # ---- quantizer tables (reference-generated fixtures) and scan tables
        self.qp = [np.ascontiguousarray(TABLES[f"qp/8/{qindex}/{p}"]) for p in range(3)]
        # ---- deblocking mode info consistent with the tiling
        cols, rows = (width + 3) // 4, (height + 3) // 4
        self.mi = (pkg.DlfModeInfo * (cols * rows))()
        self.mi_cols, self.mi_rows = cols, rows
        tx_grid = np.repeat(np.repeat(self.sb_tx.reshape(self.sb_rows, self.sb_cols), 16, 0), 16, 1)[:rows, :cols]
        skip_blk = rng.random(((height + 7) // 8, (width + 7) // 8)) < 0.3
        self.skip8 = np.ascontiguousarray(skip_blk[:height // 8, :width // 8].astype(np.uint8))
        skip_grid = np.repeat(np.repeat(skip_blk, 2, 0), 2, 1)[:rows, :cols]
        arr = np.frombuffer(self.mi, dtype=np.uint8).reshape(rows, cols, C.sizeof(pkg.DlfModeInfo))
        arr[:, :, 0] = tx_grid; arr[:, :, 1] = tx_grid
        arr[:, :, 2] = np.clip(tx_grid - 1, 2, 5); arr[:, :, 3] = np.clip(tx_grid - 1, 2, 5)
        arr[:, :, 4] = np.maximum(tx_grid, 3); arr[:, :, 5] = np.maximum(tx_grid, 3)
        arr[:, :, 6] = skip_grid
        arr[:, :, 7] = 20; arr[:, :, 8] = 20; arr[:, :, 9] = 12; arr[:, :, 10] = 12; arr[:, :, 11] = 12; arr[:, :, 12] = 12
        self.edges = [dc.product_host_edges(self.mi, cols, rows, p, width >> (p > 0), height >> (p > 0)) for p in range(3)]   # the PRODUCT builds its own inputs (the bench never prepares them with the checker); the parity gate compares them with the oracle afterwards
        # ---- CDEF: per-fb strengths for the apply stage (strength *selection* is host logic in the reference)
        self.cdef_y = rng.integers(0, 64, self.n_sb).astype(np.uint8)
        self.cdef_uv = rng.integers(0, 64, self.n_sb).astype(np.uint8)
        self.cdef_damping = 3 + (120 >> 6)   # 3 + (base_q_idx >> 6), EbCdefProcess.c:121
        self._orc_windows = None

    def scan_tables(self, ts):
        out = []
        for cls in range(3):
            key = f"iscan/{ts}/{cls}"
            out.append(np.ascontiguousarray(TABLES[key]) if key in TABLES.files else None)
        return out

    def scans(self, ts):
        out = []
        for cls in range(3):
            key = f"scan/{ts}/{cls}"
            out.append(np.ascontiguousarray(TABLES[key]) if key in TABLES.files else None)
        return out


PADQ, PADS = 32, 16   # padding of the 1/4- and 1/16-resolution HME planes


def hme_jobs(F):
    """The three SvtHipSadLoop job lists of the HME stage (level 0 on the 1/16 planes: 16x16 block, 64x32 window; levels 1 / 2: 32x32 and
    64x64 blocks, 16x16 windows), one search per SB, windows clipped to the padded planes."""
    out = []
    for lvl, (bsz, saw, sah) in enumerate(((16, 64, 32), (32, 16, 16), (64, 16, 16))):
        S = (pkg.SadLoop * F.n_sb)()
        sc = (4, 2, 1)[lvl]
        pad = (PADS, PADQ, F.pad)[lvl]
        for i in range(F.n_sb):
            sx, sy = (i % F.sb_cols) * 64 // sc, (i // F.sb_cols) * 64 // sc
            pw_, ph_ = F.w // sc, F.h // sc
            x0 = min(max(sx - saw // 2, -pad + 1), pw_ - 1); y0 = min(max(sy - sah // 2, -pad + 1), ph_ - 1)
            S[i] = pkg.SadLoop(sx + pad, sy + pad, x0 + pad, y0 + pad, bsz, bsz, min(saw, pw_ + pad - 1 - bsz - x0), min(sah, ph_ + pad - 1 - bsz - y0), 1, 0)
        out.append(S)
    return out


def conv_jobs(F, seed):
    """Every whole 16x16 luma block predicted at a random eighth-pel MV (EIGHTTAP_REGULAR both ways): SvtHipConvBlk list + count."""
    rng = np.random.default_rng(seed)
    n = (F.w // 16) * (F.h // 16)
    CB = (pkg.ConvBlk * n)()
    k = 0
    for by in range(0, F.h, 16):
        for bx in range(0, F.w, 16):
            if bx + 16 <= F.w and by + 16 <= F.h:
                CB[k] = pkg.ConvBlk(bx + int(rng.integers(-8, 9)), by + int(rng.integers(-8, 9)), bx, by, 16, 16, 0, 0, int(rng.integers(0, 16)), int(rng.integers(0, 16)), 0, 0)
                k += 1
    return CB, k
