"""CPU: oracle/ref_bench.c — the loops that drive the REAL reference kernels for bench.py's cpu_baseline (kind "reference") — against the
oracle batch functions on the same job lists, for both flavours of oracle/_ref: the plain C build and the SIMD build (SSE2..AVX-512
kernels dispatched by the reference's own RTCD setup).  Bit-exact everywhere, so (a) the timed reference work is exactly the work the
HIP stages do and (b) the reference's SIMD kernels agree with the C path the oracle is pinned to.  Skipped where oracle/_ref was not
built (it is built in the build container and travels to the GPU box)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, ptr
import me_common as mc
import txfm_common as tc
import workload

P3, I3 = C.c_void_p * 3, C.c_int * 3
W, H = 256, 176            # 4 x 3 SBs, last row partial (48 rows)


def _lib(name):
    path = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not built (oracle/Makefile.ref)")
    L = C.CDLL(path)
    L.refb_setup.restype = C.c_uint64; L.refb_setup.argtypes = [C.c_uint64]
    return L


@pytest.fixture(scope="module", params=["libsvtav1_ref.so", "libsvtav1_ref_simd.so"])
def refb(request):
    L = _lib(request.param)
    flags = L.refb_setup(0xFFFFFFFFFFFFFFFF)
    if "simd" in request.param:
        assert flags & (1 << 2), hex(flags)        # at least SSE2: the RTCD tables really point at SIMD kernels
    return L


@pytest.fixture(scope="module")
def F():
    return workload.Frame(W, H, seed=5)


def test_me_and_hme(orc, refb, F):
    st = F.cur_y_p.shape[1]
    sbs = mc.windows(orc, W, H, 64, 64)
    for sub in (0, 1):
        e_sad, e_mv = mc.oracle_frame(orc, F.cur_y_p, F.ref_y_p, st, F.pad, sbs, sub)
        sad = np.zeros((F.n_sb, 85), np.uint32); mv = np.zeros((F.n_sb, 85), np.uint32)
        refb.refb_me_fullpel_frame(ptr(F.cur_y_p), ptr(F.ref_y_p), st, F.pad, F.pad, sbs, F.n_sb, sub, ptr(sad), ptr(mv), 0, F.n_sb)
        assert np.array_equal(sad, e_sad) and np.array_equal(mv, e_mv), sub
    # a window whose width is not a multiple of 8 exercises the single-point kernels
    odd = mc.windows(orc, W, H, 13, 9)
    e_sad, e_mv = mc.oracle_frame(orc, F.cur_y_p, F.ref_y_p, st, F.pad, odd, 0)
    sad = np.zeros((F.n_sb, 85), np.uint32); mv = np.zeros((F.n_sb, 85), np.uint32)
    refb.refb_me_fullpel_frame(ptr(F.cur_y_p), ptr(F.ref_y_p), st, F.pad, F.pad, odd, F.n_sb, 0, ptr(sad), ptr(mv), 0, F.n_sb)
    assert np.array_equal(sad, e_sad) and np.array_equal(mv, e_mv)
    # HME: the three levels on decimated planes
    org = F.pad * st + F.pad
    planes = {}
    for name, src_p in (("cur", F.cur_y_p), ("ref", F.ref_y_p)):
        q = np.zeros((H // 2 + 64, W // 2 + 64), np.uint8); s_ = np.zeros((H // 4 + 32, W // 4 + 32), np.uint8)
        orc.orc_downsample_2d(C.c_void_p(src_p.ctypes.data + org), st, W, H, C.c_void_p(q.ctypes.data + 32 * q.shape[1] + 32), q.shape[1], 2, 1)
        orc.orc_downsample_2d(C.c_void_p(src_p.ctypes.data + org), st, W, H, C.c_void_p(s_.ctypes.data + 16 * s_.shape[1] + 16), s_.shape[1], 4, 1)
        planes[name] = (s_, q, src_p)
    for lvl, S in enumerate(workload.hme_jobs(F)):
        c, r = planes["cur"][lvl], planes["ref"][lvl]
        out = []
        for fn in (orc.orc_sad_loop_batch, refb.refb_sad_loop_batch):
            sad = np.zeros(F.n_sb, np.uint32); xy = np.zeros((F.n_sb, 2), np.int16)
            fn(ptr(c), c.shape[1], ptr(r), r.shape[1], S, 0, F.n_sb, ptr(sad), ptr(xy))
            out.append((sad, xy))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), lvl


def test_subpel_and_txfm_chain(orc, refb, F):
    CB, nb = workload.conv_jobs(F, 3)
    st = F.ref_y_p.shape[1]
    org = C.c_void_p(F.ref_y_p.ctypes.data + F.pad * st + F.pad)
    a = np.zeros((H, W), np.uint8); b = np.zeros((H, W), np.uint8)
    orc.orc_subpel_predict_batch(1, 8, org, st, ptr(a), W, CB, 0, nb)
    refb.refb_subpel_predict_batch(org, st, ptr(b), W, CB, 0, nb)
    assert np.array_equal(a, b) and a.any()
    for (kind, ts), descs in sorted(F.descs.items()):
        scans, iscans = F.scans(ts), F.scan_tables(ts)
        SC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in scans])
        ISC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in iscans])
        nk = min(tc.TXW[ts], 32) * min(tc.TXH[ts], 32)
        for plane in ([0] if kind == 0 else [1, 2]):
            qp = F.qp[plane]
            n = len(descs)
            outs = []
            for which in (0, 1):
                recon = np.zeros_like(F.ref[plane]); q = np.zeros((n, nk), np.int32); eob = np.zeros(n, np.uint16)
                args = [ptr(F.cur[plane]), F.cur[plane].shape[1], ptr(F.ref[plane]), F.ref[plane].shape[1], ptr(recon), recon.shape[1], ptr(descs), 0, n, ts]
                if which == 0:
                    orc.orc_txfm_chain_8bit(*args, 0, ptr(qp), SC, tc.TX_SCALE[ts], ptr(q), ptr(eob))
                else:
                    refb.refb_txfm_chain_8bit(*args, ptr(qp), SC, ISC, tc.TX_SCALE[ts], ptr(q), ptr(eob))
                outs.append((recon, q, eob))
            for x, y in zip(*outs):
                assert np.array_equal(x, y), (kind, ts, plane)
            assert outs[0][2].any()


def test_deblock_cdef_sgr(orc, refb, F):
    # deblocking
    for p in range(3):
        ev, eh = F.edges[p]
        a = F.ref[p].copy(); b = F.ref[p].copy()
        orc.orc_deblock_plane(ptr(a), 1, a.shape[1], 8, ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 0)
        refb.refb_deblock_plane(ptr(b), b.shape[1], ptr(ev), ptr(eh), ev.shape[1], ev.shape[0], 0)
        assert np.array_equal(a, b) and (a != F.ref[p]).any(), p
    # CDEF search + apply
    skip8 = F.skip8.copy(); skip8[0:8, 8:16] = 1      # one all-skip filter block
    rec = P3(*[p.ctypes.data for p in F.ref]); rs = I3(*[p.shape[1] for p in F.ref])
    src = P3(*[p.ctypes.data for p in F.cur]); ss = I3(*[p.shape[1] for p in F.cur])
    e_mse = np.zeros((2, F.n_sb, 64), np.uint64); g_mse = np.zeros_like(e_mse)
    orc.orc_cdef_search_frame(rec, rs, src, ss, 1, W, H, ptr(skip8), F.cdef_damping, 8, 0, ptr(e_mse), 0, F.n_sb)
    refb.refb_cdef_search_frame(rec, rs, src, ss, W, H, ptr(skip8), F.cdef_damping, ptr(g_mse), 0, F.n_sb)
    assert np.array_equal(e_mse, g_mse) and e_mse.any()
    ys = F.cdef_y.copy(); uvs = F.cdef_uv.copy(); ys[2] = 0; uvs[2] = 0; ys[3] = 0
    a = [p.copy() for p in F.ref]; b = [p.copy() for p in F.ref]
    orc.orc_cdef_apply_frame(rec, P3(*[p.ctypes.data for p in a]), rs, 1, W, H, ptr(skip8), ptr(ys), ptr(uvs), F.cdef_damping, 8)
    refb.refb_cdef_apply_frame(rec, P3(*[p.ctypes.data for p in b]), rs, W, H, ptr(skip8), ptr(ys), ptr(uvs), F.cdef_damping, 0, F.n_sb)
    for p in range(3):
        assert np.array_equal(a[p], b[p]) and (a[p] != F.ref[p]).any(), p
    # self-guided search: projection coefficients of every unit and set
    for p, ssub in ((0, 0), (1, 1)):
        pw, ph = W >> ssub, H >> ssub
        ext = np.ascontiguousarray(np.pad(F.ref[p], 3, mode="edge")); st = ext.shape[1]; off = 3 * st + 3
        for US in (64, 128):
            nu = max((pw + US // 2) // US, 1) * max((ph + US // 2) // US, 1)
            lim = np.zeros((nu, 4), np.int32)
            orc.orc_rest_unit_limits(pw, ph, ssub, US, ptr(lim))
            sums = np.zeros((nu, 16, 5), np.int64)
            orc.orc_sgr_search_plane(C.c_void_p(ext.ctypes.data + off), 1, st, ptr(F.cur[p]), F.cur[p].shape[1], pw, ph, ssub, ssub, US, 8, 0xFFFF, ptr(sums))
            e_xq = np.zeros((nu, 16, 2), np.int32)
            for u in range(nu):
                size = int((lim[u, 1] - lim[u, 0]) * (lim[u, 3] - lim[u, 2]))
                for ep in range(16):
                    orc.orc_sgr_solve(ptr(np.ascontiguousarray(sums[u, ep])), size, ep, C.c_void_p(e_xq.ctypes.data + (u * 16 + ep) * 8))
            g_xq = np.zeros((nu, 16, 2), np.int32)
            refb.refb_sgr_search_plane(C.c_void_p(ext.ctypes.data + off), st, ptr(F.cur[p]), F.cur[p].shape[1], ptr(lim), 0, nu, 64 >> ssub, 64 >> ssub, 0xFFFF, ptr(g_xq))
            assert np.array_equal(e_xq, g_xq), (p, US, np.argwhere(e_xq != g_xq)[:5])
            assert e_xq.any()


def test_hbd_drivers(orc, refb, pkg):
    """The 16-bit drivers of oracle/ref_bench.c (BASELINE configs[3]) vs the oracle: sad_16b + highbd_10 variance, the 64-point transform chain at
    bit depth 10 with svt_aom_highbd_quantize_b, the self-guided search with use_highbitdepth."""
    rng = np.random.default_rng(31)
    w, h, bd = 256, 192, 10
    cur = rng.integers(0, 1024, (h, w)).astype(np.uint16)
    prd = np.clip(cur.astype(np.int32) + rng.integers(-40, 41, (h, w)), 0, 1023).astype(np.uint16)
    # ---- block pairs
    pairs = [(x, y, min(x + 3, w - s), min(y + 2, h - s), s, s) for s in (64, 32, 16, 8) for y in range(0, h - s + 1, 64) for x in range(0, w - s + 1, 64)]
    n = len(pairs)
    P = (pkg.BlkPair * n)(*[pkg.BlkPair(*p) for p in pairs])
    sad, var, sse = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    refb.refb_hbd_sad_var_batch(ptr(cur), w, ptr(prd), w, P, 0, n, ptr(sad), ptr(var), ptr(sse))
    orc.orc_sad_16b.restype = C.c_uint32; orc.orc_variance_hbd10.restype = C.c_uint32
    for i, (ax, ay, bx, by, bw, bh) in enumerate(pairs):
        pa = C.c_void_p(cur.ctypes.data + (ay * w + ax) * 2); pb = C.c_void_p(prd.ctypes.data + (by * w + bx) * 2)
        s = C.c_uint32()
        assert sad[i] == orc.orc_sad_16b(pa, w, pb, w, bh, bw) and var[i] == orc.orc_variance_hbd10(pa, w, pb, w, bw, bh, C.byref(s)) and sse[i] == s.value, pairs[i]
    # ---- transform chain, every square size incl. 64x64
    g = np.load(os.path.join(ROOT, "tests", "golden", "txfm_tables.npz"))
    qp = np.ascontiguousarray(g["qp/10/60/0"])
    for ts, n_ in ((4, 64), (3, 32), (2, 16), (1, 8), (0, 4)):
        descs = np.array([pkg.tx_desc(x, y, (x // n_ + y // n_) % (1 if ts >= 3 else 4)) for y in range(0, h - n_ + 1, n_) for x in range(0, w - n_ + 1, n_)], np.uint32)
        nb = len(descs); nk = min(n_, 32) ** 2
        sc = [np.ascontiguousarray(g[f"scan/{ts}/{c}"]) if f"scan/{ts}/{c}" in g.files else None for c in range(3)]
        isc = [np.ascontiguousarray(g[f"iscan/{ts}/{c}"]) if f"iscan/{ts}/{c}" in g.files else None for c in range(3)]
        SC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in sc]); ISC = (C.c_void_p * 3)(*[s.ctypes.data if s is not None else None for s in isc])
        rec = np.zeros_like(cur); q = np.zeros((nb, nk), np.int32); dq = np.zeros((nb, nk), np.int32); eob = np.zeros(nb, np.uint16)
        refb.refb_txfm_chain_hbd(ptr(cur), w, ptr(prd), w, ptr(rec), w, ptr(descs), 0, nb, ts, bd, ptr(qp), SC, ISC, tc.TX_SCALE[ts], ptr(q), ptr(dq), ptr(eob))
        for i in range(0, nb, max(1, nb // 12)):
            x, y, tt = int(descs[i] & 0x3FFF), int((descs[i] >> 14) & 0x3FFF), int(descs[i] >> 28)
            res = (cur[y:y + n_, x:x + n_].astype(np.int32) - prd[y:y + n_, x:x + n_]).astype(np.int16)
            co = tc.orc_fwd(orc, np.ascontiguousarray(res), n_, tt, ts, bd)
            orc.orc_handle_transform.restype = C.c_uint64
            orc.orc_handle_transform(ptr(co), ts)
            cls = 0 if n_ > 16 else (0 if tt < 10 else (2 if tt & 1 else 1))
            eq, edq = np.zeros(nk, np.int32), np.zeros(nk, np.int32); e_eob = C.c_uint16()
            z = [np.array(r, np.int16) for r in qp[:5]]
            orc.orc_quantize(1, ptr(co), nk, ptr(z[0]), ptr(z[1]), ptr(z[2]), ptr(z[3]), ptr(eq), ptr(edq), ptr(z[4]), C.byref(e_eob), ptr(sc[cls]), tc.TX_SCALE[ts])
            assert np.array_equal(q[i], eq) and np.array_equal(dq[i], edq) and eob[i] == e_eob.value, (ts, i)
            er = np.zeros((n_, n_), np.uint16)
            orc.orc_inv_txfm2d_add(ptr(edq), ptr(np.ascontiguousarray(prd[y:y + n_, x:x + n_])), n_, ptr(er), n_, tt, ts, bd)
            assert np.array_equal(rec[y:y + n_, x:x + n_], er), (ts, i)
    # ---- self-guided search
    EXT, US = 3, 64
    ext = np.ascontiguousarray(np.pad(prd, EXT, mode="edge")); st = ext.shape[1]; off = (EXT * st + EXT) * 2
    nu = max((w + 32) // 64, 1) * max((h + 32) // 64, 1)
    lim = np.zeros((nu, 4), np.int32); orc.orc_rest_unit_limits(w, h, 0, US, ptr(lim))
    sums = np.zeros((nu, 16, 5), np.int64)
    orc.orc_sgr_search_plane(C.c_void_p(ext.ctypes.data + off), 2, st, ptr(cur), w, w, h, 0, 0, US, bd, 0xFFFF, ptr(sums))
    e_xq = np.zeros((nu, 16, 2), np.int32)
    for u in range(nu):
        size = int((lim[u, 1] - lim[u, 0]) * (lim[u, 3] - lim[u, 2]))
        for ep in range(16):
            orc.orc_sgr_solve(ptr(np.ascontiguousarray(sums[u, ep])), size, ep, C.c_void_p(e_xq.ctypes.data + (u * 16 + ep) * 8))
    g_xq = np.zeros((nu, 16, 2), np.int32)
    refb.refb_sgr_search_plane_hbd(C.c_void_p(ext.ctypes.data + off), st, ptr(cur), w, ptr(lim), 0, nu, 64, 64, 0xFFFF, bd, ptr(g_xq))
    assert np.array_equal(e_xq, g_xq) and e_xq.any()


def test_estimate_transform_coeff_shapes(orc, refb):
    """av1_estimate_transform through the RTCD pointers of either flavour (the AVX2 / AVX-512 N2 and N4 transform kernels on the SIMD
    build) for the four coefficient shapes vs the oracle; 8-bit residuals, all sizes and legal types."""
    import txfm_common as tc
    orc.orc_estimate_transform.restype = C.c_uint64
    rng = np.random.default_rng(34)

    def aligned(n, dtype):   # the SIMD kernels use aligned loads / stores on both buffers
        raw = np.zeros(n * np.dtype(dtype).itemsize + 64, np.uint8)
        o = (-raw.ctypes.data) % 64
        return raw[o:o + n * np.dtype(dtype).itemsize].view(dtype)
    for ts in range(19):
        w, h = tc.TXW[ts], tc.TXH[ts]
        kw, kh = min(w, 32), min(h, 32)
        for tt in tc.legal_types(ts):
            res = aligned(w * h, np.int16); res[:] = rng.integers(-255, 256, w * h)
            for shape in range(4):
                exp = aligned(w * h, np.int32); exp[:] = 0x5A5A5A
                e_en = C.c_uint64(0)
                assert refb.av1_estimate_transform(ptr(res), w, ptr(exp), w, ts, C.byref(e_en), 8, tt, 0, shape) == 0
                got = np.zeros(kw * kh, np.int32)
                g_en = orc.orc_estimate_transform(ptr(res), w, ptr(got), tt, ts, 8, shape)
                assert np.array_equal(got, exp[:kw * kh]), (tc.TX_NAMES[ts], tt, shape)
                assert g_en == e_en.value, (tc.TX_NAMES[ts], tt, shape)
