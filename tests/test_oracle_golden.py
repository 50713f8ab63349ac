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


# ------------------------------------------------------------------- transforms + quantisation
import txfm_common as tc

T = np.load(os.path.join(os.path.dirname(__file__), "golden", "txfm_tables.npz"))
V = np.load(os.path.join(os.path.dirname(__file__), "golden", "txfm_quant.npz"))


def golden_scan(ts, tt):
    cls = tc.SCAN_CLASS[tt] if max(tc.TXW[ts], tc.TXH[ts]) <= 16 else 0
    return np.ascontiguousarray(T[f"scan/{ts}/{cls}"]), np.ascontiguousarray(T[f"iscan/{ts}/{cls}"])


def golden_cases():
    return sorted({tuple(int(v) for v in k.split("/")[:3]) for k in V.files})


def test_txfm_quant_golden(orc):
    orc.orc_handle_transform.restype = C.c_uint64
    cases = golden_cases()
    assert len(cases) > 150
    for ts, tt, bd in cases:
        k = f"{ts}/{tt}/{bd}"
        w, h = tc.TXW[ts], tc.TXH[ts]
        kw, kh = min(w, 32), min(h, 32)
        x = np.ascontiguousarray(V[k + "/x"])
        co = tc.orc_fwd(orc, x, w, tt, ts, bd)
        en = orc.orc_handle_transform(ptr(co), ts)
        co = np.ascontiguousarray(co[:kw * kh])
        assert np.array_equal(co, V[k + "/coeff"]) and en == int(V[k + "/energy"][0]), k
        scan, _ = golden_scan(ts, tt)
        qp = np.ascontiguousarray(T[f"qp/{bd}/60/0"])
        v = 0 if bd == 8 else 1
        q, dq, eob = tc.orc_quant(orc, v, co, qp, scan, tc.TX_SCALE[ts])
        qf, dqf, eobf = tc.orc_quant(orc, v + 2, co, qp, scan, tc.TX_SCALE[ts])
        assert np.array_equal(q, V[k + "/q"]) and np.array_equal(dq, V[k + "/dq"]), k
        assert np.array_equal(qf, V[k + "/qf"]) and np.array_equal(dqf, V[k + "/dqf"]), k
        assert [eob, eobf] == [int(e) for e in V[k + "/eob"]], k
        pred = np.ascontiguousarray(V[k + "/pred"]); rec = np.zeros((h, w), np.uint16)
        orc.orc_inv_txfm2d_add(ptr(dq), ptr(pred), w, ptr(rec), w, tt, ts, bd)
        assert np.array_equal(rec, V[k + "/rec"]), k
