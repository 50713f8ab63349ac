"""Quick device-time probe of the ME kernel (dev pointers via the C ABI, HIP-event timer).
SVT_HIP_LIB=<path> selects an alternative build of the library (tuning experiments)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_package, ptr
import me_common as mc
pkg = load_package()
if os.environ.get("SVT_HIP_LIB"):
    pkg.LIB_PATH = os.environ["SVT_HIP_LIB"]
orc = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "oracle", "liboracle.so"))
w, h = (3840, 2160)
ctx = pkg.Context(0); L = ctx.L
cur, refp = mc.synth.make_luma_pair(w, h, seed=11)
cur_p, ref_p = mc.synth.pad_plane(cur), mc.synth.pad_plane(refp)
stride = cur_p.shape[1]; sbs = mc.windows(orc, w, h, 64, 64); n = len(sbs)
d_src, d_ref = ctx.to_device(cur_p), ctx.to_device(ref_p)
d_sbs = ctx.to_device(np.frombuffer(bytes(sbs), np.uint8)); d_sad = ctx.empty(n * 85 * 4); d_mv = ctx.empty(n * 85 * 4)
for waves in [int(x) for x in os.environ.get("WAVES", "1,2,4").split(",")]:
    L.svt_hip_me_set_waves_per_sb(ctx.h, waves)
    for it in range(2):
        L.svt_hip_timer_start(ctx.h)
        for k in range(10):
            ctx.check(L.svt_hip_me_fullpel_frame_dev(ctx.h, d_src, d_ref, stride, 68, 68, d_sbs, n, 0, d_sad, d_mv))
        ms = C.c_float(); ctx.check(L.svt_hip_timer_stop_ms(ctx.h, C.byref(ms)))
    t = ms.value / 10
    print(f"{os.environ.get('SVT_HIP_LIB','default')} waves={waves}: {t:.3f} ms/frame  {n*4096*4096/t*1e3/1e12:.1f} Tpx-SAD/s", flush=True)
