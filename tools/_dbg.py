import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, "tests")
from conftest import load_package
import md_common as M
pkg = load_package(); hip = pkg.Context(0); orc = C.CDLL("oracle/liboracle.so")
rng = np.random.default_rng(1)
w, h, n_refs, bank = 64, 64, 1, 4
for kind in ("const", "xgrad", "ygrad"):
    src, refs, pus, mv, sb_cols, n_sb, pad = M.make_case(rng, w, h, n_refs, pad=48, mv_range=4, frac_none=0.0)
    mv[:] = 0; src[:] = 0
    yy, xx = np.mgrid[0:refs[0].shape[0], 0:refs[0].shape[1]]
    refs[0][:] = {"const": 100 + 0 * xx, "xgrad": (xx * 2) % 256, "ygrad": (yy * 2) % 256}[kind].astype(np.uint8)
    exp = M.oracle_grid(orc, src, refs, pus, mv, sb_cols, n_sb, pad, w, h, bank)
    d_src, d_mv = hip.to_device(src), hip.to_device(mv)
    d_refs = [hip.to_device(r) for r in refs]
    d_out = hip.empty(exp.size * 4)
    pu_arr = (pkg.MdPu * len(pus))(*[pkg.MdPu(*p) for p in pus])
    planes = (pkg.MdRefPlane * n_refs)()
    for r in range(n_refs):
        planes[r] = pkg.MdRefPlane(d_refs[r].value + pad * refs[r].shape[1] + pad, refs[r].shape[1], -pad, -pad, refs[r].shape[1] - pad, refs[r].shape[0] - pad)
    hip.check(hip.L.svt_hip_md_subpel_grid_picture_dev(hip.h, d_src, src.shape[1], w, h, sb_cols, n_sb, len(pus), pu_arr, n_refs, planes, d_mv, bank, d_out), "md grid")
    got = hip.to_host(d_out, exp.shape, np.uint32)
    pu = 21
    print(kind, "pu", pu, pus[pu], "sum p (sqrt sse/64 for const):")
    print(" exp sse", exp[0, pu, 0, :, 1].reshape(7, 7)[[0, 3]])
    print(" got sse", got[0, pu, 0, :, 1].reshape(7, 7)[[0, 3]])
