"""Runs bench.py's CPU baseline legs alone (no GPU work): the reference's SIMD kernels (oracle/_ref SIMD flavour) and, with --port,
the oracle's scalar C, on the bench frame.   python tools/cpu_ref_baseline.py [--w 3840 --h 2160] [--port] [--threads-check]"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import me_common as mc  # noqa: E402
import txfm_common as tc  # noqa: E402
import workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=3840); ap.add_argument("--h", type=int, default=2160)
ap.add_argument("--port", action="store_true"); ap.add_argument("--threads-check", action="store_true")
args = ap.parse_args()
orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
F = workload.Frame(args.w, args.h, seed=11)
sbs = mc.windows(orc, args.w, args.h, 64, 64)
CB, nb = workload.conv_jobs(F, 14)
stages = [dict(key=k) for k in "pyr hme me subpel txfm inv dlf cdef_search cdef_apply sgr_units sgr_apply".split()]
jobs = dict(hme=workload.hme_jobs(F), conv=(CB, nb), unit=256)
refb = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so"))
if args.threads_check:   # does a ctypes call scale over Python threads on this host?
    from conftest import ptr
    import numpy as np
    refb.refb_setup.restype = C.c_uint64; refb.refb_setup.argtypes = [C.c_uint64]; refb.refb_setup(2 ** 64 - 1)
    st = F.cur_y_p.shape[1]
    sad = np.zeros((F.n_sb, 85), np.uint32); mv = np.zeros((F.n_sb, 85), np.uint32)
    for nt in (1, 2, 8, 32, 64):
        ths = [threading.Thread(target=lambda i=i: refb.refb_me_fullpel_frame(ptr(F.cur_y_p), ptr(F.ref_y_p), st, F.pad, F.pad, sbs, F.n_sb, 0, ptr(sad), ptr(mv),
                                                                              i * F.n_sb // nt, (i + 1) * F.n_sb // nt)) for i in range(nt)]
        c0 = time.process_time(); t = time.perf_counter(); [x.start() for x in ths]; [x.join() for x in ths]
        print(f"ME whole frame, {nt} threads: wall {1e3 * (time.perf_counter() - t):.1f} ms, cpu {1e3 * (time.process_time() - c0):.1f} ms")
print(json.dumps(bench.cpu_baseline_reference(refb, orc, F, sbs, mc, tc, stages, jobs), indent=1))
if args.port:
    print(json.dumps(bench.cpu_baseline(orc, F, sbs, mc, tc, stages, jobs), indent=1))
