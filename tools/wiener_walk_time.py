"""Device time of the Wiener tap refinement of a whole 4K 4:2:0 picture (svt_hip_wiener_walk_units_picture_dev: finer_tile_search_wiener_seg of every unit of the three
planes in one launch), fed like search_wiener_seg feeds it: statistics -> initial filters -> walk, all on the device.  Content: a textured source, the "degraded" picture =
source blurred + noise (so that the solved filters sharpen and the walks move), the deblocked picture = degraded + a little more noise.
   python tools/wiener_walk_time.py [--size 3840x2160] [--bd 8] [--reps 3]
Prints one JSON line: ms per picture, units, probes per unit (mean / max), microseconds per probe of the longest walk."""
import argparse, ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_package
ap = argparse.ArgumentParser(); ap.add_argument("--size", default="3840x2160"); ap.add_argument("--bd", type=int, default=8); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--unit", type=int, default=256)
a = ap.parse_args()
W, H = map(int, a.size.split("x")); bd = a.bd; US = a.unit
pkg = load_package(); ctx = pkg.Context(0); L = ctx.L
rng = np.random.default_rng(7)
dt = np.uint8 if bd == 8 else np.uint16; top = (1 << bd) - 1; pb = 1 if bd == 8 else 2


def plane(w, h, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (0.5 + 0.22 * np.sin(xx / 13.0) * np.cos(yy / 9.0) + 0.12 * np.sin((xx + 2 * yy) / 5.0)) * top + r.normal(0, top / 30.0, (h, w))
    k = np.array([1, 3, 4, 3, 1], np.float64); k /= k.sum()
    deg = np.apply_along_axis(lambda v: np.convolve(v, k, "same"), 1, base); deg = np.apply_along_axis(lambda v: np.convolve(v, k, "same"), 0, deg)
    deg += r.normal(0, top / 60.0, (h, w))
    dbl = deg + r.normal(0, top / 120.0, (h, w))
    q = lambda p: np.clip(np.rint(p), 0, top).astype(dt)
    return q(base), np.ascontiguousarray(np.pad(q(deg), 3, mode="edge")), q(dbl)


planes, keep = (pkg.WienerWalkPlane * 3)(), []
tot_units = 0
for i, (w, h, ss, win) in enumerate(((W, H, 0, 7), (W // 2, H // 2, 1, 5), (W // 2, H // 2, 1, 5))):
    src, ext, dbl = plane(w, h, 11 + i)
    st = ext.shape[1]; off = (3 * st + 3) * pb
    nu = max((w + US // 2) // US, 1) * max((h + US // 2) // US, 1); w2 = win * win
    d_ext, d_src, d_dbl = ctx.to_device(ext), ctx.to_device(src), ctx.to_device(dbl)
    d_M, d_H = ctx.empty(nu * w2 * 8), ctx.empty(nu * w2 * w2 * 8)
    d_wn, d_act, d_stat, d_err, d_pr = ctx.empty(nu * 32), ctx.empty(nu), ctx.empty(nu), ctx.empty(nu * 8), ctx.empty(nu * 4)
    ctx.check(L.svt_hip_wiener_stats_plane_dev(ctx.h, pb, bd, win, d_ext.value + off, st, d_src, w, w, h, US, ss, d_M, d_H), "stats")
    ctx.check(L.svt_hip_wiener_init_units_dev(ctx.h, win, nu, d_M, d_H, d_wn, d_act, d_stat), "init")
    keep.append((d_ext, d_src, d_dbl, d_M, d_H, d_wn, d_act, d_stat, d_err, d_pr, nu, ctx.to_host(d_wn, (nu, 16), np.int16), win))
    planes[i] = pkg.WienerWalkPlane(d_ext.value + off, st, w, h, US, ss, d_dbl.value, w, d_src.value, w, d_wn.value, d_act.value, win, d_err.value, d_pr.value)
    tot_units += nu
best = 1e9
for rep in range(a.reps):
    for k in keep:   # every repetition starts from the initial filters
        ctx.check(L.svt_hip_memcpy_h2d(ctx.h, k[5], k[11].ctypes.data_as(C.c_void_p), k[11].nbytes), "reset")
    L.svt_hip_timer_start(ctx.h)
    ctx.check(L.svt_hip_wiener_walk_units_picture_dev(ctx.h, pb, bd, 3, planes), "walk")
    ms = C.c_float(); ctx.check(L.svt_hip_timer_stop_ms(ctx.h, C.byref(ms)))
    best = min(best, ms.value)
pr = np.concatenate([ctx.to_host(k[9], (k[10],), np.uint32) for k in keep]); act = np.concatenate([ctx.to_host(k[6], (k[10],), np.uint8) for k in keep])
err = np.concatenate([ctx.to_host(k[8], (k[10],), np.int64)[ctx.to_host(k[6], (k[10],), np.uint8) > 0] for k in keep])
wn = np.concatenate([ctx.to_host(k[5], (k[10], 16), np.int16) for k in keep])
on = act > 0
print(json.dumps({"size": a.size, "bd": bd, "unit": US, "walk_ms": round(best, 3), "units": int(tot_units), "active": int(on.sum()), "probes_mean": round(float(pr[on].mean()), 1) if on.any() else 0,
                  "probes_max": int(pr[on].max()) if on.any() else 0, "us_per_probe_longest": round(best * 1e3 / max(int(pr[on].max()), 1), 1) if on.any() else 0,
                  "checksum": int((wn.astype(np.int64) * (np.arange(1, 17) ** 2) * (1 + np.arange(wn.shape[0]))[:, None]).sum() % (1 << 31)), "err_sum": int(err.sum())}))
