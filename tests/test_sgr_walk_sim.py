"""CPU: the restoration-walk policy simulator (tools/sgr_walk_sim.c, a development tool) restates the device replay; this pins it — its reference walk must end
where the oracle's search_selfguided_restoration restatement ends for every (unit, set), and every request policy must reach exactly that result (a policy can
only change how many passes / points a walk takes)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conftest import ptr


def test_simulated_walks_end_where_the_oracle_search_ends(orc, tmp_path):
    import sgr_walk_sim as sim
    so = str(tmp_path / "sgr_walk_sim.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "sgr_walk_sim.c"), "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    L = C.CDLL(so)
    rng = np.random.default_rng(3)
    w, h, US = 200, 136, 64
    yy, xx = np.mgrid[0:h, 0:w]
    src = np.clip(90 + 60 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)
    dgd = np.clip(src.astype(np.int32) + rng.integers(-9, 10, (h, w)) + (xx // 8 % 2) * 3, 0, 255).astype(np.uint8)   # a noisy, blocky copy
    ext = np.ascontiguousarray(np.pad(dgd, 3, mode="edge")); st = ext.shape[1]; off = 3 * st + 3
    nu = max((w + US // 2) // US, 1) * max((h + US // 2) // US, 1)
    e_xqd = np.zeros((nu, 16, 2), np.int32); e_err = np.zeros((nu, 16), np.int64); e_best = np.zeros(nu, np.uint8)
    orc.orc_sgr_search_units_plane(C.c_void_p(ext.ctypes.data + off), 1, st, ptr(src), w, w, h, 0, 0, US, 8, 0xFFFF, ptr(e_xqd), ptr(e_err), ptr(e_best))
    pols = [(0, 8, 0, 0), (0, 16, 0, 0), (2, 8, 0, 0), (2, 16, 0, 0), (1, 8, 1.0, 0.1), (1, 16, 1.0, 0.02), (4, 8, 1.0, 1)]
    P = (sim.Policy * len(pols))(*[sim.Policy(m, c, s, p, k) for k, (m, c, s, p) in enumerate(pols)])
    S = (sim.Stats * len(pols))()
    g_xqd = np.zeros_like(e_xqd); g_err = np.zeros_like(e_err)
    C.c_void_p.in_dll(L, "g_ref_xqd").value = g_xqd.ctypes.data
    C.c_void_p.in_dll(L, "g_ref_err").value = g_err.ctypes.data
    noise = (C.c_double * 4)()
    assert L.sim_plane(C.c_void_p(ext.ctypes.data + off), st, ptr(src), w, w, h, 0, US, 0xFFFF, P, len(pols), S, C.c_double(44000.0), C.c_double(2100.0), noise, 1 << 30) == nu
    assert np.array_equal(g_xqd, e_xqd) and np.array_equal(g_err, e_err)
    assert C.c_long.in_dll(L, "g_policy_mismatch").value == 0
    assert all(S[k].walks == nu * 16 and S[k].passes >= S[k].walks for k in range(len(pols)))
    assert S[0].points >= S[0].ref_points > 0          # the device's policy evaluates at least what the reference's walk evaluates
