"""Times svt_hip_tf_filter_frame_dev on a 2160p picture (3840 x 2176 walked extent) for a given window (HIP events), and the noise estimate.
    python tools/tf_time.py [--refs 6] [--bd 8]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package  # noqa: E402
import tf_common as tfc  # noqa: E402

pkg = load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--refs", type=int, default=6)
ap.add_argument("--bd", type=int, default=8)
args = ap.parse_args()
hip = pkg.Context(0)
w, h, bd = 3840, 2176, args.bd
rng = np.random.default_rng(0)
src, preds = tfc.make_pictures(rng, w, h, bd, 1, 1, 1, noise=1.5)
nb = (w // 64) * (h // 64)
blk = tfc.make_blocks(rng, nb, bd, err_max=12)
P3, I3 = C.c_void_p * 3, C.c_int * 3
d_src = [hip.to_device(p) for p in src]; d_dst = [hip.to_device(p) for p in src]
d_pred = [[hip.to_device(np.roll(p, f + 1, axis=1)) for p in preds[0]] for f in range(args.refs)]
d_blk = hip.to_device(blk)
refs = (pkg.TfRef * (args.refs + 1))()
for f in range(args.refs):
    for p in range(3):
        refs[f + 1].pred[p] = d_pred[f][p].value; refs[f + 1].pred_stride[p] = src[p].shape[1]
    refs[f + 1].blocks = d_blk.value
nl = (C.c_double * 3)(1.2, 0.8, 0.9)
strides = I3(*[p.shape[1] for p in src])
d_sse = hip.empty(16)


def once():
    hip.check(hip.L.svt_hip_tf_filter_frame_dev(hip.h, src[0].itemsize, bd, P3(*[p.value for p in d_src]), strides, P3(*[p.value for p in d_dst]), strides,
                                               w, h, 1, 1, 1, refs, args.refs + 1, nl, 4, 2160, d_sse), "tf")


for _ in range(3): once()
ms = C.c_float()
hip.L.svt_hip_timer_start(hip.h)
for _ in range(20): once()
hip.L.svt_hip_timer_stop_ms(hip.h, C.byref(ms))
t = ms.value / 20
samples = w * h * 3 // 2
bytes_ = samples * src[0].itemsize * (args.refs + 2)
print(f"tf_filter {w}x{h} bd{bd} window {args.refs}+1: {t:.3f} ms  {bytes_ / t / 1e6:.1f} GB/s algorithmic ({args.refs + 1} reads + 1 write per sample), "
      f"{samples * args.refs / t / 1e6:.2f} G weights/s")
d_out = hip.empty(16)
for _ in range(3): hip.L.svt_hip_tf_estimate_noise_dev(hip.h, d_src[0], src[0].itemsize, bd, w, 2160, w, d_out)
hip.L.svt_hip_timer_start(hip.h)
for _ in range(20): hip.L.svt_hip_tf_estimate_noise_dev(hip.h, d_src[0], src[0].itemsize, bd, w, 2160, w, d_out)
hip.L.svt_hip_timer_stop_ms(hip.h, C.byref(ms))
print(f"estimate_noise {w}x2160: {ms.value / 20:.3f} ms  {w * 2160 * src[0].itemsize / (ms.value / 20) / 1e6:.1f} GB/s")
