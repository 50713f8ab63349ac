"""Quick device-time probe of the ME kernel (dev pointers via the C ABI, HIP-event timer)."""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_package, ptr
import me_common as mc
pkg = load_package()
orc = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "oracle", "liboracle.so"))
w, h = (3840, 2160) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
ctx = pkg.Context(0); L = ctx.L
cur, refp = mc.synth.make_luma_pair(w, h, seed=11)
cur_p, ref_p = mc.synth.pad_plane(cur), mc.synth.pad_plane(refp)
stride = cur_p.shape[1]; sbs = mc.windows(orc, w, h, 64, 64); n = len(sbs)
def dmalloc(nb):
    p = C.c_void_p(); ctx.check(L.svt_hip_malloc(ctx.h, C.byref(p), nb)); return p
d_src, d_ref = dmalloc(cur_p.nbytes), dmalloc(ref_p.nbytes)
d_sbs = dmalloc(C.sizeof(sbs)); d_sad = dmalloc(n * 85 * 4); d_mv = dmalloc(n * 85 * 4)
L.svt_hip_memcpy_h2d(ctx.h, d_src, ptr(cur_p), cur_p.nbytes); L.svt_hip_memcpy_h2d(ctx.h, d_ref, ptr(ref_p), ref_p.nbytes)
L.svt_hip_memcpy_h2d(ctx.h, d_sbs, C.cast(sbs, C.c_void_p), C.sizeof(sbs))
for waves in (1, 2, 4):
    L.svt_hip_me_set_waves_per_sb(ctx.h, waves)
    for sub in (0, 1):
        for it in range(2):
            L.svt_hip_timer_start(ctx.h)
            for k in range(5):
                ctx.check(L.svt_hip_me_fullpel_frame_dev(ctx.h, d_src, d_ref, stride, 68, 68, d_sbs, n, sub, d_sad, d_mv))
            ms = C.c_float(); ctx.check(L.svt_hip_timer_stop_ms(ctx.h, C.byref(ms)))
        t = ms.value / 5
        print(f"{w}x{h} waves={waves} sub={sub}: {t:.3f} ms/frame  {n / t * 1e3:.0f} SB/s  {n*4096*4096/(1+sub)/t/1e9:.1f} Gpx-SAD/ms->{n*4096*4096/(1+sub)/t*1e3/1e12:.1f} Tpx-SAD/s", flush=True)
