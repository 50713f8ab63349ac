"""CPU study of the restoration walk's request policies (development tool): builds tools/sgr_walk_sim.c against liboracle.so and runs the policies
on the planes tools/sgr_walk_frame.py wrote.

    python tools/sgr_walk_frame.py 1920 1080 /tmp/f.npz && python tools/sgr_walk_sim.py /tmp/f.npz
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Policy(C.Structure):
    _fields_ = [("mode", C.c_int), ("cap", C.c_int), ("sigma_k", C.c_double), ("min_p", C.c_double), ("stop_known", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("walks", C.c_long), ("passes", C.c_long), ("points", C.c_long), ("replays", C.c_long), ("hist", C.c_long * 16), ("cost", C.c_double), ("ref_points", C.c_long)]


def main():
    npz = np.load(sys.argv[1])
    max_units = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
    so = "/tmp/sgr_walk_sim.so"
    subprocess.check_call(["gcc", "-O3", "-march=native", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "sgr_walk_sim.c"),
                           "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    L = C.CDLL(so)
    # (mode, points per pass, sigma_k, min_p / run length): 0 = the device's policy, 1 = best-first over the decision tree, 2 = model path + one-step alternatives,
    # 4 = model path that assumes a run of accepted step-2 probes goes on
    pols = [(0, 8, 0, 0), (0, 12, 0, 0), (0, 16, 0, 0), (2, 8, 0, 0), (2, 16, 0, 0), (1, 8, 0.5, 0.1), (1, 8, 1.0, 0.1), (1, 12, 1.0, 0.1), (1, 16, 1.0, 0.1), (1, 16, 1.0, 0.02),
            (4, 8, 0.5, 1), (4, 8, 1.0, 1), (4, 8, 1.0, 2)]
    P = (Policy * len(pols))(*[Policy(m, c, s, p, k) for k, (m, c, s, p) in enumerate(pols)])   # stop_known doubles as the policy's index
    A, B = 44000.0, 2100.0   # shader cycles per pass (36 k of evaluation phase: streaming the non-resident 62 % of a unit, barriers; + 8 k of replay) and per evaluated point: least squares on profiles/r03/sgr_walk_cand_sweep.txt (MI355X, 4 / 6 / 8 points per pass)
    tot = None
    for pl in range(3):
        ext, src = npz[f"e{pl}"], npz[f"s{pl}"]
        ph, pw = src.shape
        S = (Stats * len(pols))()
        noise = (C.c_double * 4)()
        st = ext.shape[1]
        L.sim_plane(C.c_void_p(ext.ctypes.data + 3 * st + 3), st, C.c_void_p(src.ctypes.data), src.shape[1], pw, ph, int(pl > 0), 256, 0xFFFF, P, len(pols), S,
                    C.c_double(A), C.c_double(B), noise, max_units)
        print(f"plane {pl}: {S[0].walks} walks, reference walk {S[0].ref_points / S[0].walks:.2f} points;  noise z rms {np.sqrt(noise[1] / max(noise[0], 1)):.3f}"
              f"  mean |d exact| {noise[2] / max(noise[0], 1):.0f} |d model| {noise[3] / max(noise[0], 1):.0f}")
        for k, (m, c, s, p) in enumerate(pols):
            w = S[k].walks
            print(f"  mode {m} cap {c:2d} sigma {s:3.1f} minp {p:4.2f}: passes {S[k].passes / w:5.2f} points {S[k].points / w:6.2f} cost {S[k].cost / w / 1000:6.1f}k   hist {list(S[k].hist)[:9]}")
        if tot is None:
            tot = [[0, 0, 0, 0.0] for _ in pols]
        for k in range(len(pols)):
            tot[k][0] += S[k].walks; tot[k][1] += S[k].passes; tot[k][2] += S[k].points; tot[k][3] += S[k].cost
    off = np.ctypeslib.as_array((C.c_long * 33 * 33 * 3).in_dll(L, "g_off")).reshape(3, 33, 33)
    cw = np.ctypeslib.as_array((C.c_long * 3).in_dll(L, "g_cls_walks"))
    for cls in range(3):
        if not cw[cls]: continue
        flat = [(off[cls, dy + 16, dx + 16] / cw[cls], dx, dy) for dy in range(-16, 17) for dx in range(-16, 17) if off[cls, dy + 16, dx + 16]]
        flat.sort(reverse=True)
        print(f"class {cls}: {cw[cls]} walks; offsets (dx, dy) by frequency:", " ".join(f"({dx},{dy}):{f:.2f}" for f, dx, dy in flat[:28]))
    cp = np.ctypeslib.as_array((C.c_long * 3 * 32).in_dll(L, "g_cls_pass")).reshape(32, 3); cq = np.ctypeslib.as_array((C.c_long * 3 * 32).in_dll(L, "g_cls_pts")).reshape(32, 3)
    cr = np.ctypeslib.as_array((C.c_long * 3).in_dll(L, "g_cls_ref")); cm = np.ctypeslib.as_array((C.c_long * 3).in_dll(L, "g_cls_minpass"))
    for cls in range(3):
        if not cw[cls]: continue
        print(f"class {cls}: reference {cr[cls] / cw[cls]:.2f} points, lower bound {cm[cls] / cw[cls]:.2f} passes at 8 per pass; " +
              "; ".join(f"pol {k}: {cp[k, cls] / cw[cls]:.2f} / {cq[k, cls] / cw[cls]:.2f}" for k in range(min(len(pols), 4))))
    print("picture:")
    for k, (m, c, s, p) in enumerate(pols):
        w = tot[k][0]
        print(f"  mode {m} cap {c:2d} sigma {s:3.1f} minp {p:4.2f}: passes {tot[k][1] / w:5.2f} points {tot[k][2] / w:6.2f} cost {tot[k][3] / w / 1000:6.1f}k")


if __name__ == "__main__":
    main()
