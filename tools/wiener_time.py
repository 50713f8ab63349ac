"""Device-time probe of the Wiener statistics kernel (MFMA) and the loop-restoration apply pass with Wiener units on a 4K 8-bit frame,
next to the oracle's scalar C rate on a sample of units."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_package, ptr
pkg = load_package()
orc = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "oracle", "liboracle.so"))
ctx = pkg.Context(0); L = ctx.L
rng = np.random.default_rng(1)
for (name, w, h, win, ss) in (("luma 3840x2160 win 7", 3840, 2160, 7, 0), ("chroma 1920x1080 win 5", 1920, 1080, 5, 1)):
    dgd = rng.integers(0, 256, (h, w)).astype(np.uint8); src = rng.integers(0, 256, (h, w)).astype(np.uint8)
    ext = np.ascontiguousarray(np.pad(dgd, 3, mode="edge")); st = ext.shape[1]; off = 3 * st + 3
    for US in (64, 256):
        nu = max((w + US // 2) // US, 1) * max((h + US // 2) // US, 1); w2 = win * win
        d_ext, d_src, d_M, d_H = ctx.to_device(ext), ctx.to_device(src), ctx.empty(nu * w2 * 8), ctx.empty(nu * w2 * w2 * 8)
        for it in range(2):
            L.svt_hip_timer_start(ctx.h)
            for k in range(10):
                ctx.check(L.svt_hip_wiener_stats_plane_dev(ctx.h, 1, 8, win, d_ext.value + off, st, d_src, w, w, h, US, ss, d_M, d_H))
            ms = C.c_float(); ctx.check(L.svt_hip_timer_stop_ms(ctx.h, C.byref(ms)))
        t = ms.value / 10
        macs = w * h * (w2 * (w2 + 1) / 2 + w2)
        # CPU: oracle on 8 units
        n = 8 if US == 64 else 1
        Mo, Ho = np.zeros(w2, np.int64), np.zeros(w2 * w2, np.int64)
        t0 = time.perf_counter()
        for u in range(n):
            orc.orc_wiener_compute_stats(win, C.c_void_p(ext.ctypes.data + off), ptr(src), 1, 8, 64 * u, 64 * u + US, 56, 56 + US, st, w, ptr(Mo), ptr(Ho))
        tc = (time.perf_counter() - t0) / n * nu
        print(f"{name} unit {US}: {t*1e3:.1f} us/plane on the GPU ({macs/t*1e3/1e12:.1f} T useful MAC/s); scalar C oracle, 1 core: {tc*1e3:.0f} ms/plane", flush=True)
        ctx.free(d_ext, d_src, d_M, d_H)
# 16-bit planes (10-bit content): three int8 component planes through the same MFMA kernel + exact recombination
for (name, w, h, win, ss) in (("10-bit luma 3840x2160 win 7", 3840, 2160, 7, 0),):
    dgd = rng.integers(0, 1024, (h, w)).astype(np.uint16); src = rng.integers(0, 1024, (h, w)).astype(np.uint16)
    ext = np.ascontiguousarray(np.pad(dgd, 3, mode="edge")); st = ext.shape[1]; off = (3 * st + 3) * 2
    for US in (64, 256):
        nu = max((w + US // 2) // US, 1) * max((h + US // 2) // US, 1); w2 = win * win
        d_ext, d_src, d_M, d_H = ctx.to_device(ext), ctx.to_device(src), ctx.empty(nu * w2 * 8), ctx.empty(nu * w2 * w2 * 8)
        for it in range(2):
            L.svt_hip_timer_start(ctx.h)
            for k in range(10):
                ctx.check(L.svt_hip_wiener_stats_plane_dev(ctx.h, 2, 10, win, d_ext.value + off, st, d_src, w, w, h, US, ss, d_M, d_H))
            ms = C.c_float(); ctx.check(L.svt_hip_timer_stop_ms(ctx.h, C.byref(ms)))
        print(f"{name} unit {US}: {ms.value / 10 * 1e3:.1f} us/plane on the GPU", flush=True)
        ctx.free(d_ext, d_src, d_M, d_H)
