"""One small chained step of the hot path through the C ABI alone (svt_hip_malloc / memcpy + _dev entry points, no torch): integer ME -> deblock ->
CDEF search -> finish_cdef_search's strength decision -> CDEF apply with those strengths, on one synthetic frame per STREAM.  The same function drives the
product library on a GPU and the CPU test double (oracle/_ref/mock/libsvtav1_hip.so) — the multi-rank tests run it per rank and compare what comes back."""
import ctypes as C
import os
import zlib

import numpy as np

from conftest import ROOT

MOCK_LIB = os.path.join(ROOT, "oracle", "_ref", "mock", "libsvtav1_hip.so")
STATE_BYTES = 304 + 8192 + 4 * 128 * 4096 * 8   # SVT_HIP_CDEF_SELECT_STATE_BYTES
vp, i32 = C.c_void_p, C.c_int32
P3, I3 = C.c_void_p * 3, C.c_int * 3


def load(path):
    L = C.CDLL(path)
    L.svt_hip_init.argtypes = [i32, C.POINTER(vp)]
    L.svt_hip_malloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    L.svt_hip_free.argtypes = [vp, vp]
    L.svt_hip_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.svt_hip_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.svt_hip_memcpy_d2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.svt_hip_me_fullpel_frame.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp]
    L.svt_hip_deblock_frame_dev.argtypes = [vp, P3, i32, I3, i32, P3, P3, I3, I3, i32]
    L.svt_hip_cdef_search_frame_dev.argtypes = [vp, i32, P3, I3, P3, I3, i32, i32, vp, i32, i32, vp, vp, vp]
    L.svt_hip_cdef_strength_select_dev.argtypes = [vp, vp, vp, i32, i32, i32, vp, C.c_size_t]
    L.svt_hip_cdef_finish_dev.argtypes = [vp, vp, vp, i32, vp, C.c_uint64, vp, vp, vp, vp, vp]
    L.svt_hip_cdef_apply_frame_dev.argtypes = [vp, i32, P3, P3, I3, i32, i32, vp, vp, vp, i32, i32, vp, vp]
    L.svt_hip_last_error.restype = C.c_char_p; L.svt_hip_last_error.argtypes = [vp]
    L.svt_hip_destroy.argtypes = [vp]
    return L


class Dev:
    def __init__(self, L, device=0):
        self.L, self.h = L, vp()
        rc = L.svt_hip_init(device, C.byref(self.h))
        if rc:
            raise RuntimeError(f"svt_hip_init: status {rc}")

    def ck(self, rc, what):
        if rc:
            raise RuntimeError(f"{what}: status {rc}: {self.L.svt_hip_last_error(self.h)}")

    def up(self, a):
        a = np.ascontiguousarray(a)
        p = vp()
        self.ck(self.L.svt_hip_malloc(self.h, C.byref(p), max(a.nbytes, 8)), "malloc")
        self.ck(self.L.svt_hip_memcpy_h2d(self.h, p, a.ctypes.data_as(vp), a.nbytes), "h2d")
        return p

    def zeros(self, nbytes):
        return self.up(np.zeros(max(nbytes, 8), np.uint8))

    def down(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self.ck(self.L.svt_hip_memcpy_d2h(self.h, out.ctypes.data_as(vp), p, out.nbytes), "d2h")
        return out

    def close(self):
        self.L.svt_hip_destroy(self.h)


def stream_step(dev, stream_id, w=208, h=136):
    """-> (n_sb, {name: crc32}) of one stream's frame (seed = stream id)"""
    import workload
    F = workload.Frame(w, h, seed=100 + stream_id)
    L = dev.L
    # search windows (plain data): 32 x 32 candidates around every superblock, clipped to the padded plane
    n_sb = F.n_sb
    S = (workload.pkg.SbSearch * n_sb)()
    for i in range(n_sb):
        sx, sy = (i % F.sb_cols) * 64, (i // F.sb_cols) * 64
        x0, y0 = max(sx - 16, -F.pad + 1), max(sy - 16, -F.pad + 1)
        S[i] = workload.pkg.SbSearch(sx, sy, x0 - sx, y0 - sy, min(32, w + F.pad - 65 - x0 + 1), min(32, h + F.pad - 65 - y0 + 1))
    sad = np.zeros((n_sb, 85), np.uint32); mv = np.zeros((n_sb, 85), np.uint32)
    dev.ck(L.svt_hip_me_fullpel_frame(dev.h, F.cur_y_p.ctypes.data_as(vp), F.ref_y_p.ctypes.data_as(vp), F.cur_y_p.shape[1], F.cur_y_p.shape[0], F.pad, F.pad,
                                      C.cast(S, vp), n_sb, 0, sad.ctypes.data_as(vp), mv.ctypes.data_as(vp)), "me")
    strides = [p.shape[1] for p in F.ref]
    d_rec = [dev.up(p) for p in F.ref]; d_src = [dev.up(p) for p in F.cur]; d_out = [dev.up(p) for p in F.ref]
    d_e = [(dev.up(ev), dev.up(eh), ev.shape[1], ev.shape[0]) for ev, eh in F.edges]
    dev.ck(L.svt_hip_deblock_frame_dev(dev.h, P3(*[p.value for p in d_rec]), 1, I3(*strides), 8, P3(*[e[0].value for e in d_e]), P3(*[e[1].value for e in d_e]),
                                       I3(*[e[2] for e in d_e]), I3(*[e[3] for e in d_e]), 0), "deblock")
    d_skip = dev.up(F.skip8)
    d_mse = dev.zeros(2 * n_sb * 64 * 8); d_dir = dev.zeros(n_sb * 64); d_var = dev.zeros(n_sb * 64 * 4)
    dev.ck(L.svt_hip_cdef_search_frame_dev(dev.h, 1, P3(*[p.value for p in d_rec]), I3(*strides), P3(*[p.value for p in d_src]), I3(*strides), w, h, d_skip, F.cdef_damping, 8,
                                           d_mse, d_dir, d_var), "cdef search")
    d_state = dev.zeros(STATE_BYTES); d_fin = dev.zeros(80); d_sel = dev.zeros(4 * n_sb); d_cy = dev.zeros(n_sb); d_cuv = dev.zeros(n_sb)
    m1 = vp(d_mse.value + n_sb * 64 * 8)
    dev.ck(L.svt_hip_cdef_strength_select_dev(dev.h, d_mse, m1, n_sb, 0, 64, d_state, STATE_BYTES), "select")
    dev.ck(L.svt_hip_cdef_finish_dev(dev.h, d_mse, m1, n_sb, d_state, 55473, None, d_fin, d_sel, d_cy, d_cuv), "finish")
    for p in range(3):
        dev.ck(L.svt_hip_memcpy_d2d(dev.h, d_out[p], d_rec[p], F.ref[p].nbytes), "copy")
    dev.ck(L.svt_hip_cdef_apply_frame_dev(dev.h, 1, P3(*[p.value for p in d_rec]), P3(*[p.value for p in d_out]), I3(*strides), w, h, d_skip, d_cy, d_cuv, F.cdef_damping, 8,
                                          d_dir, d_var), "cdef apply")
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())
    res = {"me_sad": crc(sad), "me_mv": crc(mv), "mse": crc(dev.down(d_mse, (2 * n_sb * 64,), np.uint64)), "fin": crc(dev.down(d_fin, (72,), np.uint8)),
           "strengths": crc(np.concatenate([dev.down(d_cy, (n_sb,), np.uint8), dev.down(d_cuv, (n_sb,), np.uint8)]))}
    for p in range(3):
        res[f"deblocked{p}"] = crc(dev.down(d_rec[p], F.ref[p].shape, np.uint8)); res[f"cdef{p}"] = crc(dev.down(d_out[p], F.ref[p].shape, np.uint8))
    for d in d_rec + d_src + d_out + [d_skip, d_mse, d_dir, d_var, d_state, d_fin, d_sel, d_cy, d_cuv] + [x for e in d_e for x in e[:2]]:
        L.svt_hip_free(dev.h, d)
    return n_sb, res
