"""CPU-only: the bench step's chain up to the CDEF output on the ORACLE (test infrastructure), for one synthetic frame, so that the restoration
walk's speculation policies can be studied without a GPU (tools/sgr_walk_sim.c).  Writes the three CDEF-output planes (with the 3-sample border)
and the three source planes to an .npz.

    python tools/sgr_walk_frame.py 1920 1080 /tmp/sgr_frame_1080.npz

Chain (bench.py's, workload identical): open-loop integer ME -> every 16x16 luma block predicted at its ME vector + an eighth-pel phase ->
residual -> transform + quantisation + inverse -> deblock -> CDEF search -> strengths = per-filter-block argmin (the joint selection's restriction to
eight pairs is skipped here: this is a content generator, not a parity check) -> CDEF apply."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package, ptr  # noqa: E402
import me_common as mc  # noqa: E402
import txfm_common as tc  # noqa: E402
import workload  # noqa: E402

P3, I3 = C.c_void_p * 3, C.c_int * 3


def main():
    W, H, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 11
    pkg = load_package()
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    t0 = time.time()
    F = workload.Frame(W, H, seed=seed)
    sbs = mc.windows(orc, W, H, 64, 64)
    sad, mv = mc.oracle_frame(orc, F.cur_y_p, F.ref_y_p, F.cur_y_p.shape[1], F.pad, sbs, 0)
    print("me", time.time() - t0, flush=True)
    # sub-pel prediction at the ME vectors (svt_hip_subpel_jobs_from_me_dev's rule) + the bench's phase pattern
    nblk = (W // 16) * (H // 16)
    frac = np.random.default_rng(14 + 1000 * F.seed).integers(0, 16, (nblk, 2)).astype(np.uint8)
    CB = (pkg.ConvBlk * nblk)()
    bw = W // 16
    for k in range(nblk):
        bx, by = k % bw, k // bw
        sb = (by >> 2) * F.sb_cols + (bx >> 2); qx, qy = bx & 3, by & 3
        z = ((qy >> 1) * 2 + (qx >> 1)) * 4 + (qy & 1) * 2 + (qx & 1)
        word = int(mv[sb, 5 + z])
        s16 = lambda v: v - 65536 if v >= 32768 else v
        mx = s16(word & 0xffff) >> 2; my = s16(word >> 16) >> 2
        CB[k] = pkg.ConvBlk(bx * 16 + int(mx), by * 16 + int(my), bx * 16, by * 16, 16, 16, 0, 0, int(frac[k, 0]), int(frac[k, 1]), 0, 0)
    pred_y = F.ref[0].copy()   # columns / rows past the last whole 16x16 block keep the co-located reference
    st = F.ref_y_p.shape[1]
    orc.orc_subpel_predict_batch(1, 8, C.c_void_p(F.ref_y_p.ctypes.data + F.pad * st + F.pad), st, ptr(pred_y), W, C.cast(CB, C.c_void_p), 0, nblk)
    pred = [pred_y, F.ref[1], F.ref[2]]
    recon = [p.copy() for p in pred]
    for (kind, ts), descs in sorted(F.descs.items()):
        nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
        scans = F.scans(ts)
        SC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in scans])
        for plane in ([0] if kind == 0 else [1, 2]):
            q = np.zeros((len(descs), nk), np.int32); eob = np.zeros(len(descs), np.uint16)
            orc.orc_txfm_chain_8bit(ptr(F.cur[plane]), F.cur[plane].shape[1], ptr(pred[plane]), pred[plane].shape[1], ptr(recon[plane]),
                                    recon[plane].shape[1], ptr(descs), 0, len(descs), ts, 0, ptr(F.qp[plane]), SC, tc.TX_SCALE[ts], ptr(q), ptr(eob))
    print("txfm", time.time() - t0, flush=True)
    dlf = [p.copy() for p in recon]
    for p in range(3):
        ev, eh = F.edges[p]
        orc.orc_deblock_plane(ptr(dlf[p]), 1, dlf[p].shape[1], 8, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 0)
    mse = np.zeros((2, F.n_sb, 64), np.uint64)
    orc.orc_cdef_search_frame(P3(*[p.ctypes.data for p in dlf]), I3(*[p.shape[1] for p in dlf]), P3(*[p.ctypes.data for p in F.cur]),
                              I3(*[p.shape[1] for p in F.cur]), 1, W, H, ptr(F.skip8), F.cdef_damping, 8, 0, ptr(mse), 0, F.n_sb)
    print("cdef search", time.time() - t0, flush=True)
    cy = np.argmin(mse[0], axis=1).astype(np.uint8); cuv = np.argmin(mse[1], axis=1).astype(np.uint8)
    outp = [p.copy() for p in dlf]
    orc.orc_cdef_apply_frame(P3(*[p.ctypes.data for p in dlf]), P3(*[p.ctypes.data for p in outp]), I3(*[p.shape[1] for p in dlf]), 1, W, H,
                             ptr(F.skip8), ptr(cy), ptr(cuv), F.cdef_damping, 8)
    ext = [np.ascontiguousarray(np.pad(p, 3, mode="edge")) for p in outp]
    np.savez(out, e0=ext[0], e1=ext[1], e2=ext[2], s0=F.cur[0], s1=F.cur[1], s2=F.cur[2])
    for p in range(3):
        d = outp[p].astype(np.int32) - F.cur[p].astype(np.int32)
        print("plane", p, outp[p].shape, "mse vs source", float((d * d).mean()))
    print("done", time.time() - t0)


if __name__ == "__main__":
    main()
