"""GPU parity: batched residual + forward transform + quantize, and inverse transform + recon (HIP,
through the C ABI) vs the oracle restatement and the committed golden vectors.
Matrix mirrors /root/reference/test/FwdTxfm2dAsmTest.cc:153-454, InvTxfm2dAsmTest.cc:691-760,
QuantAsmTest.cc:88-335, quantize_func_test.cc:276-658: all 19 sizes x legal types x bd 8/10,
inputs +-(2^bd-1) incl. all-max / all-min blocks, several q-indices, all four quantizers."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ptr
import txfm_common as tc
from test_oracle_golden import T, V, golden_scan, golden_cases

pytestmark = pytest.mark.gpu


def qparams_struct(pkg, qp, variant, log_scale):
    s = pkg.QuantParams()
    rnd, qnt = (qp[5], qp[6]) if variant >= 2 else (qp[1], qp[2])
    for name, row in (("zbin", qp[0]), ("round", rnd), ("quant", qnt), ("quant_shift", qp[3]), ("dequant", qp[4])):
        getattr(s, name)[0] = int(row[0]); getattr(s, name)[1] = int(row[1])
    s.log_scale = log_scale; s.variant = variant
    return s


def run_fwd(hip, pkg, ts, bd, src, pred, descs, qp, variant, want_coeff=True, shape=0):
    """src/pred: 2-D pixel planes (u8 or u16).  Returns dict of host arrays."""
    w, h = tc.TXW[ts], tc.TXH[ts]
    nk = min(w, 32) * min(h, 32)
    n = len(descs)
    pix = src.dtype.itemsize
    d_src, d_pred = hip.to_device(src), hip.to_device(pred)
    d_desc = hip.to_device(np.asarray(descs, np.uint32))
    scans = pkg.ScanTables(); keep = []
    for cls in range(3):
        key = f"iscan/{ts}/{cls}"
        if key in T.files:
            p = hip.to_device(np.ascontiguousarray(T[key])); keep.append(p); scans.iscan[cls] = p.value
    d_co, d_q, d_dq = hip.empty(n * nk * 4), hip.empty(n * nk * 4), hip.empty(n * nk * 4)
    d_eob, d_cul, d_en = hip.empty(n * 2), hip.empty(n * 4), hip.empty(n * 8)
    qs = qparams_struct(pkg, qp, variant, tc.TX_SCALE[ts]); qs.coeff_shape = shape
    hip.check(hip.L.svt_hip_fwd_txfm_quant_batch_dev(hip.h, ts, pix, d_src, src.shape[1], d_pred, pred.shape[1], d_desc, n,
                                                    C.byref(qs), C.byref(scans), d_co if want_coeff else None, d_q, d_dq, d_eob, d_cul, d_en), "fwd")
    out = dict(coeff=hip.to_host(d_co, (n, nk), np.int32), q=hip.to_host(d_q, (n, nk), np.int32), dq=hip.to_host(d_dq, (n, nk), np.int32),
               eob=hip.to_host(d_eob, (n,), np.uint16), cul=hip.to_host(d_cul, (n,), np.int32), energy=hip.to_host(d_en, (n,), np.uint64))
    hip.free(d_src, d_pred, d_desc, d_co, d_q, d_dq, d_eob, d_cul, d_en, *keep)
    return out


def run_inv(hip, ts, bd, dq, pred, descs):
    n = len(descs)
    pix = pred.dtype.itemsize
    d_dq, d_pred, d_desc = hip.to_device(dq), hip.to_device(pred), hip.to_device(np.asarray(descs, np.uint32))
    d_rec = hip.to_device(np.zeros_like(pred))
    hip.check(hip.L.svt_hip_inv_txfm_add_batch_dev(hip.h, ts, pix, bd, d_dq, d_pred, pred.shape[1], d_rec, pred.shape[1], d_desc, n), "inv")
    rec = hip.to_host(d_rec, pred.shape, pred.dtype)
    hip.free(d_dq, d_pred, d_desc, d_rec)
    return rec


def oracle_block(orc, ts, tt, bd, src, pred, x, y, qp, variant, scan, shape=0):
    w, h = tc.TXW[ts], tc.TXH[ts]
    kw, kh = min(w, 32), min(h, 32)
    res = (src[y:y + h, x:x + w].astype(np.int32) - pred[y:y + h, x:x + w].astype(np.int32)).astype(np.int16)
    res = np.ascontiguousarray(res)
    if shape:
        orc.orc_estimate_transform.restype = C.c_uint64
        co = np.zeros(kw * kh, np.int32)
        en = orc.orc_estimate_transform(ptr(res), w, ptr(co), tt, ts, bd, shape)
    else:
        co = tc.orc_fwd(orc, res, w, tt, ts, bd)
        orc.orc_handle_transform.restype = C.c_uint64
        en = orc.orc_handle_transform(ptr(co), ts)
        co = np.ascontiguousarray(co[:kw * kh])
    q, dq, eob = tc.orc_quant(orc, variant, co, qp, scan, tc.TX_SCALE[ts])
    orc.orc_cul_level.restype = C.c_int32
    cul = orc.orc_cul_level(ptr(q), ptr(scan), eob)
    return co, en, q, dq, eob, cul


@pytest.mark.parametrize("ts", range(19))
def test_fwd_quant_inv_all_types(hip, pkg, orc, ts):
    w, h = tc.TXW[ts], tc.TXH[ts]
    rng = np.random.default_rng(100 + ts)
    for bd in (8, 10):
        dt = np.uint8 if bd == 8 else np.uint16
        types = tc.legal_types(ts)
        PW, PH = 4 * 64 + 8, 3 * 64  # plane with an odd-ish stride
        src = rng.integers(0, 1 << bd, (PH, PW)).astype(dt)
        pred = rng.integers(0, 1 << bd, (PH, PW)).astype(dt)
        src[0:64, 0:64] = (1 << bd) - 1; pred[0:64, 0:64] = 0           # residual = +max
        src[0:64, 64:128] = 0; pred[0:64, 64:128] = (1 << bd) - 1       # residual = -max
        pos = [(x, y) for y in range(0, PH - h + 1, h) for x in range(0, 256 - w + 1, w)]
        rng.shuffle(pos)
        pos = [(0, 0), (64, 0)] + pos[:min(len(pos), 70)]
        descs, tts = [], []
        for i, (x, y) in enumerate(pos):
            tt = types[i % len(types)]
            descs.append(pkg.tx_desc(x, y, tt)); tts.append(tt)
        for qi, variant in ((60, 0 if bd == 8 else 1), (200, 2 if bd == 8 else 3), (20, 0 if bd == 8 else 1)):
            qp = np.ascontiguousarray(T[f"qp/{bd}/{qi}/{qi % 3}"])
            g = run_fwd(hip, pkg, ts, bd, src, pred, descs, qp, variant)
            for i, ((x, y), tt) in enumerate(zip(pos, tts)):
                scan, _ = golden_scan(ts, tt)
                co, en, q, dq, eob, cul = oracle_block(orc, ts, tt, bd, src, pred, x, y, qp, variant, scan)
                assert np.array_equal(g["coeff"][i], co), ("coeff", ts, tt, bd, i)
                assert int(g["energy"][i]) == en, ("energy", ts, tt, bd)
                assert np.array_equal(g["q"][i], q) and np.array_equal(g["dq"][i], dq), ("quant", ts, tt, bd, qi, variant)
                assert int(g["eob"][i]) == eob and int(g["cul"][i]) == cul, ("eob/cul", ts, tt, bd, qi, variant)
            # inverse on the GPU's own dequantized coefficients, non-overlapping blocks only
            seen, keep = set(), []
            for i, (x, y) in enumerate(pos):
                if (x, y) not in seen:
                    seen.add((x, y)); keep.append(i)
            rec = run_inv(hip, ts, bd, np.ascontiguousarray(g["dq"][keep]), pred, [descs[i] for i in keep])
            p16 = pred.astype(np.uint16)
            for i in keep:
                x, y = pos[i]
                exp = np.zeros((h, w), np.uint16)
                orc.orc_inv_txfm2d_add(ptr(np.ascontiguousarray(g["dq"][i])), ptr(np.ascontiguousarray(p16[y:y + h, x:x + w])), w, ptr(exp), w, tts[i], ts, bd)
                assert np.array_equal(rec[y:y + h, x:x + w].astype(np.uint16), exp), ("inv", ts, tts[i], bd, qi)


def test_golden_vectors(hip, pkg):
    """The committed reference-generated vectors through the HIP path (no oracle involved)."""
    by_ts = {}
    for ts, tt, bd in golden_cases():
        by_ts.setdefault((ts, bd), []).append(tt)
    for (ts, bd), tts in sorted(by_ts.items()):
        w, h = tc.TXW[ts], tc.TXH[ts]
        dt = np.uint8 if bd == 8 else np.uint16
        n = len(tts)
        src = np.zeros((h, n * w), np.int32); pred = np.zeros((h, n * w), np.int32)
        for i, tt in enumerate(tts):
            x = V[f"{ts}/{tt}/{bd}/x"].astype(np.int32)
            # realise the residual x as src - pred with both inside the pixel range
            pred[:, i * w:(i + 1) * w] = np.where(x < 0, -x, 0)
            src[:, i * w:(i + 1) * w] = np.where(x < 0, 0, x)
        descs = [pkg.tx_desc(i * w, 0, tt) for i, tt in enumerate(tts)]
        qp = np.ascontiguousarray(T[f"qp/{bd}/60/0"])
        for variant, kq, kdq, ke in ((0 if bd == 8 else 1, "q", "dq", 0), (2 if bd == 8 else 3, "qf", "dqf", 1)):
            g = run_fwd(hip, pkg, ts, bd, src.astype(dt), pred.astype(dt), descs, qp, variant)
            for i, tt in enumerate(tts):
                k = f"{ts}/{tt}/{bd}"
                assert np.array_equal(g["coeff"][i], V[k + "/coeff"]), k
                assert int(g["energy"][i]) == int(V[k + "/energy"][0]), k
                assert np.array_equal(g["q"][i], V[k + "/" + kq]) and np.array_equal(g["dq"][i], V[k + "/" + kdq]), (k, variant)
                assert int(g["eob"][i]) == int(V[k + "/eob"][ke]), (k, variant)
        gp = np.concatenate([V[f"{ts}/{tt}/{bd}/pred"] for tt in tts], axis=1).astype(dt)
        dq = np.stack([V[f"{ts}/{tt}/{bd}/dq"] for tt in tts])
        rec = run_inv(hip, ts, bd, np.ascontiguousarray(dq), np.ascontiguousarray(gp), descs)
        for i, tt in enumerate(tts):
            assert np.array_equal(rec[:, i * w:(i + 1) * w].astype(np.uint16), V[f"{ts}/{tt}/{bd}/rec"]), (ts, tt, bd)


@pytest.mark.parametrize("bd", [8, 10])
def test_mixed_size_launches(hip, pkg, bd):
    """svt_hip_fwd_txfm_quant_multi_dev / svt_hip_inv_txfm_add_multi_dev: all 19 sizes as 19 jobs of one call (two launches of <= 16
    jobs) must reproduce the single-size entry points bit for bit (those are checked against the oracle above), including ragged
    last workgroups and an empty job."""
    rng = np.random.default_rng(70 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    Wp, Hp = 1024, 512
    src = rng.integers(0, 1 << bd, (Hp, Wp)).astype(dt); pred = rng.integers(0, 1 << bd, (Hp, Wp)).astype(dt)
    qp = T[f"qp/{bd}/60/0"]
    variant = 0 if bd == 8 else 1
    d_src, d_pred = hip.to_device(src), hip.to_device(pred)
    d_rec_multi = hip.to_device(np.zeros_like(pred))
    fj = (pkg.FwdTxJob * 20)(); ij = (pkg.InvTxJob * 20)()
    keep, expect, outs = [], [], []
    y0 = 0
    for ts in range(19):
        w, h = tc.TXW[ts], tc.TXH[ts]; nk = min(w, 32) * min(h, 32)
        n = int(rng.integers(3, 40))
        tts = tc.legal_types(ts)
        descs = np.array([pkg.tx_desc((k * w) % (Wp - w + 1) // w * w, y0, tts[k % len(tts)]) for k in range(n)], np.uint32)
        exp = run_fwd(hip, pkg, ts, bd, src, pred, descs, qp, variant)
        exp["rec"] = run_inv(hip, ts, bd, exp["dq"], pred, descs)
        expect.append((ts, descs, exp))
        d_desc = hip.to_device(descs)
        scans = pkg.ScanTables()
        for cls in range(3):
            key = f"iscan/{ts}/{cls}"
            if key in T.files:
                p = hip.to_device(np.ascontiguousarray(T[key])); keep.append(p); scans.iscan[cls] = p.value
        o = dict(co=hip.empty(n * nk * 4), q=hip.empty(n * nk * 4), dq=hip.empty(n * nk * 4), eob=hip.empty(n * 2), cul=hip.empty(n * 4), en=hip.empty(n * 8))
        outs.append(o); keep += [d_desc]
        j = ts if ts < 7 else ts + 1      # job 7 stays empty (nblk = 0)
        fj[j] = pkg.FwdTxJob(ts, n, d_src.value, Wp, d_pred.value, Wp, d_desc.value, qparams_struct(pkg, qp, variant, tc.TX_SCALE[ts]), scans,
                             o["co"].value, o["q"].value, o["dq"].value, o["eob"].value, o["cul"].value, o["en"].value)
        ij[j] = pkg.InvTxJob(ts, n, o["dq"].value, d_pred.value, Wp, d_rec_multi.value, Wp, d_desc.value)
        y0 += h
    hip.check(hip.L.svt_hip_fwd_txfm_quant_multi_dev(hip.h, src.itemsize, fj, 20), "fwd multi")
    hip.check(hip.L.svt_hip_inv_txfm_add_multi_dev(hip.h, src.itemsize, bd, ij, 20), "inv multi")
    rec = hip.to_host(d_rec_multi, pred.shape, dt)
    for (ts, descs, exp), o in zip(expect, outs):
        n = len(descs); nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
        assert np.array_equal(hip.to_host(o["co"], (n, nk), np.int32), exp["coeff"]), ("coeff", ts)
        assert np.array_equal(hip.to_host(o["q"], (n, nk), np.int32), exp["q"]), ("q", ts)
        assert np.array_equal(hip.to_host(o["dq"], (n, nk), np.int32), exp["dq"]), ("dq", ts)
        assert np.array_equal(hip.to_host(o["eob"], (n,), np.uint16), exp["eob"]), ("eob", ts)
        assert np.array_equal(hip.to_host(o["cul"], (n,), np.int32), exp["cul"]), ("cul", ts)
        assert np.array_equal(hip.to_host(o["en"], (n,), np.uint64), exp["energy"]), ("energy", ts)
        w, h = tc.TXW[ts], tc.TXH[ts]
        for d in descs:
            x, y = int(d & 0x3FFF), int((d >> 14) & 0x3FFF)
            assert np.array_equal(rec[y:y + h, x:x + w], exp["rec"][y:y + h, x:x + w]), ("rec", ts)
    # svt_hip_enc_txfm_multi_dev: the same jobs with the reconstruction fused in (dequantised coefficients stay in registers) — identical
    # levels, eobs and reconstruction, with the dq output (pass 0) and without it (pass 1)
    for drop_dq in (0, 1):
        ej = (pkg.EncTxJob * 20)()
        d_rec_f = hip.to_device(np.zeros_like(pred))
        outs2 = []
        for j in range(20):
            if not fj[j].nblk: continue
            n = fj[j].nblk; nk = min(tc.TXW[fj[j].tx_size], 32) * min(tc.TXH[fj[j].tx_size], 32)
            o = dict(q=hip.empty(n * nk * 4), dq=hip.empty(n * nk * 4), eob=hip.empty(n * 2), cul=hip.empty(n * 4))
            outs2.append((j, o))
            f = fj[j]
            ej[j].fwd = pkg.FwdTxJob(f.tx_size, f.nblk, f.d_src, f.src_stride, f.d_pred, f.pred_stride, f.d_descs, f.qp, f.scans, None, o["q"].value,
                                     None if drop_dq else o["dq"].value, o["eob"].value, o["cul"].value, None)
            ej[j].d_recon = d_rec_f.value; ej[j].recon_stride = Wp
        hip.check(hip.L.svt_hip_enc_txfm_multi_dev(hip.h, src.itemsize, bd, ej, 20), "enc multi")
        rec_f = hip.to_host(d_rec_f, pred.shape, dt)
        by_ts = {ts: (descs, exp) for ts, descs, exp in expect}
        for j, o in outs2:
            ts = fj[j].tx_size; descs, exp = by_ts[ts]
            n = len(descs); nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32); w, h = tc.TXW[ts], tc.TXH[ts]
            assert np.array_equal(hip.to_host(o["q"], (n, nk), np.int32), exp["q"]), ("fused q", ts, drop_dq)
            if not drop_dq: assert np.array_equal(hip.to_host(o["dq"], (n, nk), np.int32), exp["dq"]), ("fused dq", ts)
            assert np.array_equal(hip.to_host(o["eob"], (n,), np.uint16), exp["eob"]) and np.array_equal(hip.to_host(o["cul"], (n,), np.int32), exp["cul"]), ("fused eob", ts)
            for d in descs:
                x, y = int(d & 0x3FFF), int((d >> 14) & 0x3FFF)
                assert np.array_equal(rec_f[y:y + h, x:x + w], exp["rec"][y:y + h, x:x + w]), ("fused rec", ts, drop_dq)
        hip.free(d_rec_f, *[v for _, o in outs2 for v in o.values()])
    hip.free(d_src, d_pred, d_rec_multi, *keep, *[v for o in outs for v in o.values()])


@pytest.mark.parametrize("shape", [1, 2, 3])
def test_coeff_shapes(hip, pkg, orc, shape):
    """EB_TRANS_COEFF_SHAPE N2 / N4 / ONLY_DC of av1_estimate_transform (qp.coeff_shape) for every size and legal type: coefficients,
    energy (0), quantized levels, eob and cul_level vs the oracle (pinned to the reference's N2 / N4 transform families in
    tests/test_oracle_vs_ref.py::test_estimate_transform_coeff_shapes and, AVX2 / AVX-512, tests/test_ref_bench.py)."""
    for ts in range(19):
        w, h = tc.TXW[ts], tc.TXH[ts]
        rng = np.random.default_rng(500 + 19 * shape + ts)
        for bd in (8, 10):
            dt = np.uint8 if bd == 8 else np.uint16
            types = tc.legal_types(ts)
            PW, PH = 2 * 64 + 8, 2 * 64
            src = rng.integers(0, 1 << bd, (PH, PW)).astype(dt); pred = rng.integers(0, 1 << bd, (PH, PW)).astype(dt)
            pos = [(x, y) for y in range(0, PH - h + 1, h) for x in range(0, 128 - w + 1, w)]
            rng.shuffle(pos); pos = pos[:max(len(types), 4)]
            tts = [types[i % len(types)] for i in range(len(pos))]
            descs = [pkg.tx_desc(x, y, tt) for (x, y), tt in zip(pos, tts)]
            variant = 0 if bd == 8 else 1
            qp = np.ascontiguousarray(T[f"qp/{bd}/60/0"])
            g = run_fwd(hip, pkg, ts, bd, src, pred, descs, qp, variant, shape=shape)
            for i, ((x, y), tt) in enumerate(zip(pos, tts)):
                scan, _ = golden_scan(ts, tt)
                co, en, q, dq, eob, cul = oracle_block(orc, ts, tt, bd, src, pred, x, y, qp, variant, scan, shape)
                assert en == 0 and int(g["energy"][i]) == 0
                assert np.array_equal(g["coeff"][i], co), ("coeff", ts, tt, bd, shape)
                assert np.array_equal(g["q"][i], q) and np.array_equal(g["dq"][i], dq), ("quant", ts, tt, bd, shape)
                assert int(g["eob"][i]) == eob and int(g["cul"][i]) == cul, ("eob/cul", ts, tt, bd, shape)
            assert any(g["coeff"][i].any() for i in range(len(pos)))


def test_coeff_shape_multi_and_bad_arg(hip, pkg, orc):
    """The mixed-size launch honours the per-job shape; shape 4 is rejected."""
    rng = np.random.default_rng(9)
    src = rng.integers(0, 256, (64, 64)).astype(np.uint8); pred = rng.integers(0, 256, (64, 64)).astype(np.uint8)
    d_src, d_pred = hip.to_device(src), hip.to_device(pred)
    jobs = (pkg.FwdTxJob * 2)()
    keep = []
    for j, (ts, shape) in enumerate(((1, 1), (4, 2))):
        w = tc.TXW[ts]
        descs = np.asarray([pkg.tx_desc(x, y, 0) for y in range(0, 64, w) for x in range(0, 64, w)], np.uint32)
        nk = min(w, 32) ** 2
        d_desc, d_co = hip.to_device(descs), hip.empty(len(descs) * nk * 4)
        keep += [d_desc, d_co]
        J = jobs[j]
        J.tx_size, J.nblk, J.d_src, J.src_stride, J.d_pred, J.pred_stride, J.d_descs = ts, len(descs), d_src.value, 64, d_pred.value, 64, d_desc.value
        J.qp.coeff_shape = shape; J.d_coeff = d_co.value
    hip.check(hip.L.svt_hip_fwd_txfm_quant_multi_dev(hip.h, 1, jobs, 2), "multi")
    orc.orc_estimate_transform.restype = C.c_uint64
    for j, (ts, shape) in enumerate(((1, 1), (4, 2))):
        w = tc.TXW[ts]; nk = min(w, 32) ** 2
        got = hip.to_host(keep[2 * j + 1], (jobs[j].nblk, nk), np.int32)
        k = 0
        for y in range(0, 64, w):
            for x in range(0, 64, w):
                res = np.ascontiguousarray((src[y:y + w, x:x + w].astype(np.int32) - pred[y:y + w, x:x + w]).astype(np.int16))
                co = np.zeros(nk, np.int32)
                orc.orc_estimate_transform(ptr(res), w, ptr(co), 0, ts, 8, shape)
                assert np.array_equal(got[k], co), (ts, shape, k)
                k += 1
    jobs[0].qp.coeff_shape = 4
    assert hip.L.svt_hip_fwd_txfm_quant_multi_dev(hip.h, 1, jobs, 2) != 0
    qs = pkg.QuantParams(); qs.coeff_shape = -1
    assert hip.L.svt_hip_fwd_txfm_quant_batch_dev(hip.h, 1, 1, d_src, 64, d_pred, 64, keep[0], 1, C.byref(qs), None, keep[1], None, None, None, None, None) != 0
    hip.free(d_src, d_pred, *keep)


@pytest.mark.parametrize("ts", range(19), ids=tc.TX_NAMES)
def test_inverse_full_range_vs_reference_c(hip, pkg, ref, ts):
    """The input domain of the reference's own inverse-transform test (test/InvTxfm2dAsmTest.cc:654-675: coefficients = the C forward transform of
    a residual drawn from the full +-(2^bd - 1) range, re-packed for the 64-point sizes), plus saturated residuals (constant, checkerboard, stripes)
    and coefficient buffers saturated at the +-2^(bd+8) input clamp, against the reference's C inverse itself.  This is where a 32-bit butterfly sum
    would leave the reference's 64-bit one (EbInvTransforms.c half_btf) if it ever did."""
    w, h = tc.TXW[ts], tc.TXH[ts]; kw, kh = min(w, 32), min(h, 32)
    rng = np.random.default_rng(500 + ts)
    for bd in (8, 10):
        top = (1 << bd) - 1
        yy, xx = np.mgrid[0:h, 0:w]
        residuals = [rng.integers(-top, top + 1, (h, w)) for _ in range(3)] + [np.full((h, w), top), np.full((h, w), -top), np.where((xx + yy) & 1, top, -top),
                                                                              np.where(xx & 1, top, -top), np.where(yy & 2, top, -top),
                                                                              rng.choice([-top, top], (h, w))]
        lim = (1 << (bd + 8)) - 1
        for tt in tc.legal_types(ts):
            blocks = []
            for r in residuals:
                co = tc.ref_fwd(ref, np.ascontiguousarray(r.astype(np.int16)), w, tt, ts, bd)
                if max(w, h) == 64: tc.ref_handle(ref, co, ts)
                blocks.append(co[:kw * kh].copy())
            sat = rng.choice([-lim, lim], kw * kh).astype(np.int32); blocks.append(sat)
            blocks.append(np.full(kw * kh, lim, np.int32)); blocks.append((rng.integers(-lim, lim + 1, kw * kh)).astype(np.int32))
            one = np.zeros(kw * kh, np.int32); one[0] = lim; blocks.append(one)
            n = len(blocks)
            dt = np.uint8 if bd == 8 else np.uint16
            pred = rng.integers(0, top + 1, (h, w * n)).astype(dt)
            descs = [pkg.tx_desc(i * w, 0, tt) for i in range(n)]
            rec = run_inv(hip, ts, bd, np.ascontiguousarray(np.stack(blocks)), pred, descs)
            for i, co in enumerate(blocks):
                cfull = np.zeros(w * h, np.int32); cfull[:kw * kh] = co
                p16 = np.ascontiguousarray(pred[:, i * w:(i + 1) * w].astype(np.uint16)); exp = np.zeros((h, w), np.uint16)
                tc.ref_inv(ref, cfull, p16, w, exp, w, tt, ts, bd)
                assert np.array_equal(rec[:, i * w:(i + 1) * w].astype(np.uint16), exp), ("inverse full range", ts, tt, bd, i)


@pytest.mark.parametrize("bd", [8, 10])
def test_lossless_iwht4x4_vs_reference_c(hip, pkg, ref, bd):
    """svt_hip_iwht4x4_add_batch_dev against svt_av1_highbd_iwht4x4_16_add_c / svt_av1_highbd_iwht4x4_1_add_c (EbInvTransforms.c:2771-2857), the pair highbd_iwht4x4_add
    chooses between by eob: random coefficients over the full range a 4x4 lossless block can carry, saturated ones (the sums then clip at both ends of the pixel range),
    DC-only blocks down both branches, 8-bit planes at bd 8 as well as 16-bit ones."""
    rng = np.random.default_rng(910 + bd)
    top = (1 << bd) - 1
    n = 96
    lim = (top + 1) * 16 * 4   # forward WHT of a +-top residual, scaled by UNIT_QUANT_FACTOR
    co = rng.integers(-lim, lim + 1, (n, 16)).astype(np.int32)
    co[0] = lim; co[1] = -lim; co[2] = 0; co[3, 1:] = 0; co[4, 1:] = 0; co[4, 0] = -lim; co[5] = rng.choice([-lim, lim], 16)
    eob = np.full(n, 16, np.uint16); eob[3] = 1; eob[4] = 1; eob[6] = 0; eob[7] = 1   # 6 / 7: the DC form on blocks that DO have other coefficients (the reference reads ip[0] only)
    eob[8:] = rng.integers(0, 17, n - 8)
    for pix in ((1, 2) if bd == 8 else (2,)):
        dt = np.uint8 if pix == 1 else np.uint16
        pred = rng.integers(0, top + 1, (4, 4 * n + 3)).astype(dt)
        pred[:, :8] = top; pred[:, 8:16] = 0
        descs = np.asarray([pkg.tx_desc(4 * i, 0, 0) for i in range(n)], np.uint32)
        d_c, d_e, d_p, d_d = hip.to_device(co), hip.to_device(eob), hip.to_device(pred), hip.to_device(descs)
        d_r = hip.to_device(np.zeros_like(pred))
        hip.check(hip.L.svt_hip_iwht4x4_add_batch_dev(hip.h, pix, bd, d_c, d_e, d_p, pred.shape[1], d_r, pred.shape[1], d_d, n), "iwht")
        rec = hip.to_host(d_r, pred.shape, dt)
        # the same list with d_eob NULL: every block takes the 16-coefficient form
        hip.check(hip.L.svt_hip_iwht4x4_add_batch_dev(hip.h, pix, bd, d_c, None, d_p, pred.shape[1], d_r, pred.shape[1], d_d, n), "iwht (no eob)")
        rec16 = hip.to_host(d_r, pred.shape, dt)
        hip.free(d_c, d_e, d_p, d_d, d_r)
        for i in range(n):
            p16 = np.ascontiguousarray(pred[:, 4 * i:4 * i + 4].astype(np.uint16))
            for form, got in ((int(eob[i]) > 1, rec), (True, rec16)):
                exp = np.zeros((4, 4), np.uint16)
                f = ref.svt_av1_highbd_iwht4x4_16_add_c if form else ref.svt_av1_highbd_iwht4x4_1_add_c
                f(C.c_void_p(co[i].ctypes.data), C.c_void_p(p16.ctypes.data >> 1), 4, C.c_void_p(exp.ctypes.data >> 1), 4, bd)   # CONVERT_TO_BYTEPTR
                assert np.array_equal(got[:, 4 * i:4 * i + 4].astype(np.uint16), exp), ("iwht4x4", bd, pix, i, int(eob[i]), form)
