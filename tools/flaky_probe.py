"""Run ON THE GPU BOX: one option variant of tests/test_encode_e2e.py several times under different knobs; prints the bitstream hashes (is a mismatch a race or a defect?).
    python tools/flaky_probe.py film_grain 4"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_common as E  # noqa: E402
import test_encode_e2e as T  # noqa: E402

name, reps = sys.argv[1], int(sys.argv[2])
extra, lp = T.OPTION_VARIANTS[name]
wd = os.path.join(ROOT, "gpurun_out", "flaky")
os.makedirs(wd, exist_ok=True)
clip = os.path.join(wd, "c.yuv")
w, h, n, bd, preset, q = 352, 288, 6, 8, 6, 38
E.make_clip(clip, w, h, n, seed=5, bd=bd)
PERTURB = {"MALLOC_PERTURB_": os.environ["PROBE_PERTURB"]} if os.environ.get("PROBE_PERTURB") else {}
for tag, app, env, lp_ in (("ref", E.APP_REF, {}, lp), ("ref_lp1", E.APP_REF, {}, 1), ("hip", E.APP_HIP, {"SVT_HIP_HOOKS": "all"}, lp), ("hip_lp1", E.APP_HIP, {"SVT_HIP_HOOKS": "all"}, 1),
                           ("hip_nopin", E.APP_HIP, {"SVT_HIP_HOOKS": "all", "SVT_HIP_PIN": "0"}, lp), ("hip_nodefer", E.APP_HIP, {"SVT_HIP_HOOKS": "all", "SVT_HIP_DEFER": "0"}, lp),
                           ("hip_nopin_nodefer", E.APP_HIP, {"SVT_HIP_HOOKS": "all", "SVT_HIP_DEFER": "0", "SVT_HIP_PIN": "0"}, lp),
                           ("hip_src_only", E.APP_HIP, {"SVT_HIP_HOOKS": "pa,tf,tf_me,tf_subpel,hme,me"}, lp), ("hip_lf_only", E.APP_HIP, {"SVT_HIP_HOOKS": "dlf,dlf_search,cdef_search,cdef_apply,cdef_finish,sgr_search,wiener_search,rest_apply"}, lp)):
    hs = []
    for r in range(reps):
        got = E.encode(app, clip, w, h, n, preset, q, bd, os.path.join(wd, tag), env_extra=dict(env, **PERTURB), extra_args=extra, lp=lp_)
        hs.append(got["ivf"][:8] + "/" + got["recon"][:8])
    print(tag, hs, flush=True)
