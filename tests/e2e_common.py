"""End-to-end encodes: the reference's own encoder application, unpatched (SvtAv1EncApp_ref, the C path) and with the
hook calls of integration/patch_reference.py applied (SvtAv1EncApp_hip).  Both binaries are built by oracle/Makefile.enc into oracle/_ref/
(test infrastructure; they travel to the GPU box).  Inputs are synthetic clips generated here (numpy only, deterministic)."""
import hashlib
import os
import re
import subprocess

import numpy as np

from conftest import ROOT

REFDIR = os.path.join(ROOT, "oracle", "_ref")
APP_REF = os.path.join(REFDIR, "SvtAv1EncApp_ref")
APP_HIP = os.path.join(REFDIR, "SvtAv1EncApp_hip")
MOCK_DIR = os.path.join(REFDIR, "mock")
HOOKS = ["pa", "tf", "tf_me", "tf_subpel", "hme", "me", "cdef_finish", "dlf", "dlf_search", "cdef_search", "cdef_apply", "sgr_search", "wiener_stats", "rest_apply", "wiener_try", "wiener_search"]
OPT_IN_HOOKS = ["md_tx", "encdec_tx", "md_subpel", "encdec_sb"]   # not selected by SVT_HIP_HOOKS=all: named explicitly (one launch per transform block of every mode-decision candidate)
# wiener_search hooks the whole Wiener search of a picture and takes precedence over the two hooks of the per-unit path
PER_UNIT_WIENER = {"wiener_stats", "wiener_try"}
ALL_PICTURE_LEVEL = ",".join(h for h in HOOKS if h not in PER_UNIT_WIENER)            # = what SVT_HIP_HOOKS=all effectively runs
ALL_PER_UNIT = ",".join(h for h in HOOKS if h != "wiener_search")                     # the per-unit Wiener hooks instead


def have_apps():
    return os.path.exists(APP_REF) and os.path.exists(APP_HIP)


def make_clip(path, w, h, n, seed=1, bd=8):
    """Textured scene with global motion, an object moving against it, and sensor noise: inter prediction, deblocking, CDEF and both
    restoration filters all get work to do."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h + 64, 0:w + 64]
    base = (128 + 60 * np.sin(xx / 17.0) + 50 * np.cos(yy / 23.0) + 20 * np.sin((xx + yy) / 7.0)).astype(np.float32)
    tex = rng.normal(0, 18, (h + 64, w + 64)).astype(np.float32)
    k = np.array([1, 4, 6, 4, 1], np.float32)
    k /= k.sum()
    for ax in (0, 1):
        tex = np.apply_along_axis(lambda v: np.convolve(v, k, "same"), ax, tex)
    tex *= 3
    sc = 1 << (bd - 8)
    top = 255 * sc + sc - 1
    dt = np.uint8 if bd == 8 else "<u2"
    with open(path, "wb") as f:
        for i in range(n):
            dx, dy = (3 * i) % 48, (2 * i) % 40
            fr = (base + tex)[dy:dy + h, dx:dx + w].copy()
            ox, oy = (w // 2 - 5 * i) % (w - 40), h // 3
            fr[oy:oy + 40, ox:ox + 40] += 40
            fr += rng.normal(0, 2.0, (h, w))
            y = np.clip(fr * sc, 0, top)
            u = np.clip((128 + 0.3 * (fr[::2, ::2] - 128) + rng.normal(0, 1.5, (h // 2, w // 2))) * sc, 0, top)
            v = np.clip((128 - 0.2 * (fr[::2, ::2] - 128) + rng.normal(0, 1.5, (h // 2, w // 2))) * sc, 0, top)
            for p in (y, u, v):
                f.write(p.astype(dt).tobytes())


def _md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def encode(app, clip, w, h, n, preset, q, bd, out_prefix, env_extra=None, lp=8, timeout=600, extra_args=()):
    """-> dict(ivf=md5, recon=md5, hooks={name: (handled, fallback)}, log=str)"""
    env = dict(os.environ)
    for k in ("SVT_HIP_HOOKS", "SVT_HIP_RTCD", "SVT_HIP_MOCK_PERTURB", "SVT_HIP_VERBOSE"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [app, "-i", clip, "-w", str(w), "-h", str(h), "-n", str(n), "--preset", str(preset), "--fps", "30", "-q", str(q), "--lp", str(lp),
           "-b", out_prefix + ".ivf", "-o", out_prefix + ".yuv"] + (["--input-depth", "10"] if bd == 10 else []) + list(extra_args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    hooks = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"svt_hip_hook (\w+) handled=(\d+) fallback=(\d+)", log)}
    # per-call wrappers (SVT_HIP_RTCD): calls per wrapper function, delegations per dispatch-table entry (svt_hip_rtcd_report)
    calls = {m.group(1): int(m.group(2)) for m in re.finditer(r"svt_hip_rtcd_calls (\w+) calls=(\d+)", log)}
    deleg = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"svt_hip_rtcd_delegated (\w+) count=(\d+) device_failures=(\d+)", log)}
    return {"ivf": _md5(out_prefix + ".ivf"), "recon": _md5(out_prefix + ".yuv"), "hooks": hooks, "log": log, "rtcd_calls": calls, "rtcd_delegated": deleg}
