"""CPU: the oracle restatement vs the committed golden vectors (generated from the real reference C
path by tests/golden/make_golden.py) — this is what pins the oracle on boxes without /root/reference."""
import ctypes as C
import os

import numpy as np

from conftest import ptr

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "me_fullpel.npz"))


def test_me_fullpel_sb_golden(orc):
    names = sorted({k.split("/")[0] for k in G.files if not k.startswith("sadloop")})
    assert len(names) >= 7
    for n in names:
        src, win = np.ascontiguousarray(G[n + "/src"]), np.ascontiguousarray(G[n + "/win"])
        xo, yo, saw, sah, sub = [int(v) for v in G[n + "/par"]]
        sad = np.zeros(85, np.uint32); mv = np.zeros(85, np.uint32)
        orc.orc_me_fullpel_sb(ptr(src), src.strides[0], ptr(win), win.strides[0], xo, yo, saw, sah, sub, ptr(sad), ptr(mv))
        assert np.array_equal(sad, G[n + "/sad"]), n
        assert np.array_equal(mv, G[n + "/mv"]), n


def test_sad_loop_golden(orc):
    for i in range(4):
        src, win = np.ascontiguousarray(G[f"sadloop{i}/src"]), np.ascontiguousarray(G[f"sadloop{i}/win"])
        bw, bh, saw, sah = [int(v) for v in G[f"sadloop{i}/par"]]
        best = C.c_uint64(0); xc = C.c_int16(0); yc = C.c_int16(0)
        orc.orc_sad_loop(ptr(src), bw, ptr(win), win.strides[0], bh, bw, C.byref(best), C.byref(xc), C.byref(yc),
                         win.strides[0], C.c_int16(saw), C.c_int16(sah))
        assert [best.value, xc.value, yc.value] == [int(v) for v in G[f"sadloop{i}/out"]]
