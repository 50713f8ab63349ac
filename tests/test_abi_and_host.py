"""CPU: the C-ABI library loads and exports every symbol include/svt_hip.h declares (no compute
calls without a GPU), host-side logic (search-window clamp) matches the oracle, and the product
refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    syms = set()
    for hdr in ("svt_hip.h", "svt_hip_rtcd.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"^(?:int|void|double|size_t|const char \*|SvtHipSbSearch)\s*\*?\s*(svt_hip_[a-z0-9_]+)\s*\(", txt, flags=re.M))
    return sorted(syms)


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/*.h but not exported"


def test_library_exports_nothing_else(pkg):
    """-fvisibility=hidden: the dynamic symbol table holds exactly the functions the two public headers declare (no launchers, no helpers)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TtWwBbDd" and ln.split()[1].isupper()}
    exported = {s for s in exported if not s.startswith(("_init", "_fini", "__"))}
    extra = sorted(exported - set(_declared_symbols()))
    assert not extra, f"exported but not declared in include/*.h: {extra[:20]}"


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        pkg.Context(0)


def test_product_does_not_reference_oracle():
    """The product path must never link/load anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "svt-av1_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle/" not in txt.replace("the oracle", ""), f


def test_search_window_matches_oracle(pkg, orc):
    import me_common as mc
    L = pkg.lib()
    orc.orc_me_search_window.restype = mc.OrcSearchWindow
    orc.orc_me_search_window.argtypes = [C.c_int] * 8
    rng = np.random.default_rng(5)
    for _ in range(3000):
        pw, ph = int(rng.choice([176, 200, 1920, 3840])), int(rng.choice([144, 136, 1080, 2160]))
        x = int(rng.integers(0, (pw + 63) // 64)) * 64; y = int(rng.integers(0, (ph + 63) // 64)) * 64
        cx, cy = int(rng.integers(-300, 300)), int(rng.integers(-300, 300))
        w = (int(rng.integers(1, 200)) + 7) & ~7; h = int(rng.integers(1, 200))
        a = L.svt_hip_me_search_window(x, y, cx, cy, w, h, pw, ph)
        b = orc.orc_me_search_window(x, y, cx, cy, w, h, pw, ph)
        assert (a.x_origin, a.y_origin, a.width, a.height) == (b.x_origin, b.y_origin, b.width, b.height)
        assert a.width >= 1 and a.height >= 1
        # the window never leaves the 63-px padded picture
        assert x + a.x_origin >= -63 and y + a.y_origin >= -63
        assert x + a.x_origin + a.width - 1 <= pw - 1 + 0 or a.width == 1 or x + a.x_origin + a.width <= pw


def test_every_entry_point_rejects_a_null_context(pkg):
    """Error behaviour without a GPU: every entry point that takes the context first returns an error code (never crashes, never touches HIP)
    when called with a NULL context and all-zero arguments.  The names come from include/svt_hip.h."""
    L = pkg.lib()
    hdr = open(os.path.join(ROOT, "include", "svt_hip.h")).read()
    names = sorted(set(re.findall(r"\b(svt_hip_[a-z0-9_]+)\s*\(SvtHipCtx \*ctx", hdr)))
    assert len(names) >= 45
    checked = 0
    for n in names:
        f = getattr(L, n)
        if f.restype is not C.c_int and f.restype is not None and f.restype is not int: continue   # svt_hip_last_error etc.
        assert f.argtypes, f"{n}: the ctypes binding declares no argtypes"
        args = []
        for a in f.argtypes:
            if isinstance(a, type) and issubclass(a, C.Array): args.append(a())
            elif a is C.c_void_p or a is C.c_char_p or hasattr(a, "contents"): args.append(None)
            elif a in (C.c_double, C.c_float): args.append(0.0)
            else: args.append(0)
        if n in ("svt_hip_sync", "svt_hip_destroy", "svt_hip_timer_start", "svt_hip_free", "svt_hip_me_set_waves_per_sb"): continue   # void / trivially valid on NULL
        r = f(*args)
        assert r != 0, f"{n} accepted a NULL context"
        checked += 1
    assert checked >= 40
