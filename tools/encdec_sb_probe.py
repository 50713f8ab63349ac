"""Run ON THE GPU BOX: the encdec_sb hook on the wall-clock clip under several knobs (which combination changes the bitstream?)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_common as E  # noqa: E402

wd = os.path.join(ROOT, "gpurun_out", "edsb")
os.makedirs(wd, exist_ok=True)
w, h, n, bd, preset, q = 1280, 720, 8, 8, 6, 36
clip = os.path.join(wd, "c.yuv")
E.make_clip(clip, w, h, n, seed=3, bd=bd)
SIMD, HIP_SIMD = os.path.join(E.REFDIR, "SvtAv1EncApp_simd"), os.path.join(E.REFDIR, "SvtAv1EncApp_hip_simd")
ref = E.encode(SIMD, clip, w, h, n, preset, q, bd, os.path.join(wd, "ref"))
for tag, app, env in (("c_build sb", E.APP_HIP, {"SVT_HIP_HOOKS": "encdec_sb"}), ("simd sb", HIP_SIMD, {"SVT_HIP_HOOKS": "encdec_sb"}), ("simd sb again", HIP_SIMD, {"SVT_HIP_HOOKS": "encdec_sb"}),
                      ("simd sb 1ctx", HIP_SIMD, {"SVT_HIP_HOOKS": "encdec_sb", "SVT_HIP_CONTEXTS": "1"}), ("simd sb lp1", HIP_SIMD, {"SVT_HIP_HOOKS": "encdec_sb"}),
                      ("simd tx", HIP_SIMD, {"SVT_HIP_HOOKS": "encdec_tx"}), ("simd all+sb", HIP_SIMD, {"SVT_HIP_HOOKS": "all,encdec_sb"}), ("simd all", HIP_SIMD, {"SVT_HIP_HOOKS": "all"})):
    got = E.encode(app, clip, w, h, n, preset, q, bd, os.path.join(wd, "hip"), env_extra=env, lp=1 if "lp1" in tag else 8)
    print(tag, got["ivf"] == ref["ivf"], got["recon"] == ref["recon"], got["ivf"][:8], flush=True)
for f in os.listdir(wd):
    os.remove(os.path.join(wd, f))
