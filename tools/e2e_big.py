"""Run ON THE GPU BOX: the hooked reference encoder against the unpatched one on large pictures (1080p 10-bit, 4K 8-bit), all hooks.
Too slow for the test suite (the reference side is the C-kernel build); prints one line per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_common as E  # noqa: E402

wd = os.path.join(ROOT, "gpurun_out", "e2e_big")
os.makedirs(wd, exist_ok=True)
CASES = {"1080p_10bit_m6": (1920, 1080, 3, 10, 6, 34), "2160p_8bit_m8": (3840, 2160, 3, 8, 8, 40), "2160p_8bit_m6": (3840, 2160, 2, 8, 6, 36), "2160p_10bit_m6": (3840, 2160, 2, 10, 6, 34), "1080p_8bit_m4": (1920, 1080, 3, 8, 4, 40),
         "2160p_8bit_m4": (3840, 2160, 2, 8, 4, 38), "1080p_10bit_m2": (1920, 1080, 2, 10, 2, 36), "2160p_8bit_m6_8frames": (3840, 2160, 8, 8, 6, 36)}
# the reference side: its SIMD build when it exists (codes the C build's bitstream — tests/test_encode_e2e.py::test_simd_build_... — several times faster)
APP_SIMD = os.path.join(E.REFDIR, "SvtAv1EncApp_simd")
REF_APP = APP_SIMD if os.path.exists(APP_SIMD) and not os.environ.get("E2E_BIG_C_REFERENCE") else E.APP_REF
HOOKS = os.environ.get("E2E_BIG_HOOKS", "all")
for name in sys.argv[1:] or list(CASES):
    w, h, n, bd, preset, q = CASES[name]
    clip = os.path.join(wd, name + ".yuv")
    E.make_clip(clip, w, h, n, seed=17, bd=bd)
    ref = E.encode(REF_APP, clip, w, h, n, preset, q, bd, os.path.join(wd, name + ".ref"), timeout=1200)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(wd, name + ".hip"), env_extra={"SVT_HIP_HOOKS": HOOKS}, timeout=1200)
    same = got["ivf"] == ref["ivf"] and got["recon"] == ref["recon"]
    res = [l for l in got["log"].splitlines() if l.startswith("svt_hip_resident")]   # SVT_HIP_RESIDENT=1 in the environment: the planes' report
    print(name, "identical" if same else "MISMATCH", "mock!" if "svt_hip MOCK" in got["log"] else "", {k: v for k, v in got["hooks"].items() if v != (0, 0)}, *res[-1:], flush=True)
    for f in os.listdir(wd):
        if f.startswith(name) and not f.endswith(".txt"):
            os.remove(os.path.join(wd, f))
