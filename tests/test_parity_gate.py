"""CPU: the gate bench.py puts in front of its number (tools/parity_gate.py) checked on its own — a snapshot that the reference's kernels alone produce passes every
stage, and one flipped byte in one stage's output turns exactly that stage false (each stage is recomputed from the snapshot's OWN input of that stage, so a defect
does not smear over the stages behind it).  Also: bench.py touches the checkers only after its timed region (AST), integration/ never does."""
import ast
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_package
import me_common as mc
import txfm_common as tc
import workload

sys.path.insert(0, os.path.join(ROOT, "tools"))
import parity_gate as G  # noqa: E402

SIMD = os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref_simd.so")


@pytest.fixture(scope="module")
def chain(orc):
    if not os.path.exists(SIMD):
        pytest.skip("oracle/_ref/libsvtav1_ref_simd.so not built")
    pkg = load_package()
    refb = G.setup_refb(C.CDLL(SIMD))
    F = workload.Frame(704, 400, seed=23)
    sbs = mc.windows(orc, F.w, F.h, 64, 64)
    frac = np.random.default_rng(3).integers(0, 16, ((F.w // 16) * (F.h // 16), 2)).astype(np.uint8)
    args = (refb, orc, pkg, tc, workload, 8, 55473)
    S = G.reference_chain(F, *args, sbs, frac, unit_size=64)
    return F, S, args


def test_reference_snapshot_passes_every_stage(chain):
    F, S, args = chain
    ok, bad = G.check_chain(F, S, *args, unit_size=64)
    assert list(ok) == G.STAGES and all(ok.values()), bad
    assert S["mse"].any() and S["eob_0"].any() and (S["dbl_0"] != S["recon_0"]).any() and (S["rest_0"] != S["cdef_0"]).any()


@pytest.mark.parametrize("stage,key", [("pyramids", "cur_4"), ("pyramids", "yvar"), ("hme_l0_l1_l2", "hme_xy_1"), ("me_fullpel_85pu", "sad"), ("subpel_convolve", "subpel"),
                                       ("fwd_quant_inv_recon", "q_3"), ("fwd_quant_inv_recon", "recon_1"), ("deblock", "dbl_2"), ("cdef_search", "mse"),
                                       ("cdef_strength_select", "cdef_uv"), ("cdef_apply", "cdef_0"), ("sgr_units_search", "unit_xqd_1"), ("sgr_apply", "rest_2")])
def test_one_wrong_output_fails_exactly_its_stage(chain, stage, key):
    F, S, args = chain
    T = dict(S)
    a = S[key].copy()
    flat = a.reshape(-1).view(np.uint8)
    flat[flat.size // 2] ^= 1
    T[key] = a
    ok, bad = G.check_chain(F, T, *args, stages=[stage], unit_size=64)
    assert ok == {stage: False} and any(d.startswith(key + ":") for d in bad[stage]), bad


def _names(node):
    return {n.id for n in ast.walk(node) if isinstance(n, ast.Name)}


def test_bench_uses_the_checkers_only_after_the_timed_region():
    """in bench.py's main(): no statement up to and including the headline `timed(...)` call refers to the oracle / reference libraries or the gate"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    checkers = {"orc", "refb", "parity_gate", "cpu_baseline_reference", "cpu_baseline", "run_parity_gate"}
    seen_timed = False
    for stmt in main.body:
        names = _names(stmt)
        is_headline = isinstance(stmt, ast.Assign) and any(isinstance(c, ast.Call) and getattr(c.func, "id", "") == "timed" for c in ast.walk(stmt)) and "elapsed" in {getattr(t, "id", "") for t in stmt.targets}
        if not seen_timed:
            used = names & checkers
            # the one allowed mention before the timed region: loading the oracle library handle (no call into it)
            if used:
                calls = [c for c in ast.walk(stmt) if isinstance(c, ast.Call) and (_names(c.func) & checkers)]
                assert not calls, f"bench.py line {stmt.lineno}: {sorted(used)} called before the timed region"
        if is_headline:
            seen_timed = True
    assert seen_timed, "bench.py: no `elapsed = ... timed(...)` statement found in main()"


def test_integration_and_product_do_not_reference_the_oracle():
    for d in ("svt-av1_amd", "integration", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".c", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    for word in ("liboracle", "libsvtav1_ref", "ref_bench", "orc_", "refb_"):
                        assert word not in txt, f"{d}/{f} mentions {word}"
                    assert "oracle/" not in txt.replace("the oracle", "").replace("oracle/Makefile", "").replace("oracle/_ref", "") or f in ("patch_reference.py",), f
