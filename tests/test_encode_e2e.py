"""End-to-end: the reference encoder with the process-loop hooks of integration/ must produce the SAME bitstream and the SAME reconstruction
as the unpatched reference encoder (BASELINE.json north_star: "drop-in replacement behind the existing C dispatch tables").

* CPU (`-m "not gpu"`): SvtAv1EncApp_hip runs against the CPU test double of the library (oracle/_ref/mock/libsvtav1_hip.so = the oracle
  kernels + the product's own host logic, svt-av1_amd/csrc/svt_hip_host.cpp).  This pins the glue — patched loops, bridges, the edge
  builder (set_lpf_parameters), the filter-level search control flow, the CDEF / restoration drivers — against the real encoder.
* GPU (`-m gpu`): the same binary loads svt-av1_amd/libsvtav1_hip.so; every hooked stage runs on the MI355X.
Every hook must report handled > 0 and fallback == 0 where the preset uses the stage: a silent fallback to the C loop would pass trivially.
"""
import os
import re

import pytest

import e2e_common as E

pytestmark = pytest.mark.skipif(not E.have_apps(), reason="oracle/_ref/SvtAv1EncApp_{ref,hip} not built (make -f oracle/Makefile.enc; needs /root/reference)")

# name: (w, h, frames, bit depth, preset, qp, hooks that must have run)
ALL = set(E.HOOKS) - E.PER_UNIT_WIENER   # SVT_HIP_HOOKS=all: the picture-level Wiener search takes the place of the per-unit hooks
ALL_UNIT = set(E.HOOKS) - {"wiener_search"}
NO_DLF_REST = {"pa", "tf", "tf_me", "tf_subpel", "hme", "me", "cdef_finish", "cdef_search", "cdef_apply"}        # presets > M6: deblocking inside EncDec (loop_filter_mode 1), restoration off
CASES = {
    "cif_8bit_m6": (352, 288, 8, 8, 6, 35, ALL),
    "cif_10bit_m6": (352, 288, 6, 10, 6, 30, ALL),
    "360p_8bit_m7": (640, 360, 5, 8, 7, 40, NO_DLF_REST),     # 40 SBs in many ME segments; width % 64 == 0, height % 64 == 40
    "cif_8bit_m4": (352, 288, 5, 8, 4, 45, ALL),              # full filter-level step search, chroma levels searched on their own
    "328x200_8bit_m6": (328, 200, 4, 8, 6, 33, ALL),
    "cif_10bit_m8": (352, 288, 5, 10, 8, 36, NO_DLF_REST),    # the fastest preset of this version, 10-bit          # width % 64 == height % 64 == 8: last filter blocks / restoration stripes narrower than the CDEF halo
}
GPU_ONLY_CASES = {
    "720p_8bit_m6": (1280, 720, 4, 8, 6, 38, ALL),
    "720p_10bit_m5": (1280, 720, 3, 10, 5, 32, ALL),
    "cif_8bit_q2": (352, 288, 4, 8, 6, 2, ALL - {"rest_apply"}),   # near-lossless: the largest coefficients; no unit picks a restoration filter
    "cif_10bit_q60": (352, 288, 4, 10, 6, 60, ALL),           # strongest filtering
    "cif_8bit_18_frames": (352, 288, 18, 8, 6, 42, ALL),      # more than one mini-GOP
    "cif_8bit_m2": (352, 288, 4, 8, 2, 40, ALL),              # slow presets: several reference pictures per list in ME, wider searches
    "qcif_8bit_m0": (176, 144, 3, 8, 0, 40, ALL - {"tf_me", "tf_subpel"}), # 3 frames: the alt-ref filter has a single neighbour pair, its ME batch stays empty
}


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("e2e"))


_ref_cache = {}


def _reference(case, spec, workdir):
    """clip + the unpatched reference encode (cached per module run)"""
    if case not in _ref_cache:
        w, h, n, bd, preset, q, _ = spec
        clip = os.path.join(workdir, case + ".src.yuv")
        E.make_clip(clip, w, h, n, seed=len(case) * 7 + w, bd=bd)
        _ref_cache[case] = (clip, E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, case + ".ref")))
    return _ref_cache[case]


def _check(case, spec, workdir, env, tag):
    w, h, n, bd, preset, q, must = spec
    clip, ref = _reference(case, spec, workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.{tag}"), env_extra=env)
    assert got["ivf"] == ref["ivf"], f"{case}: bitstream differs from the reference encoder\n" + got["log"][-2000:]
    assert got["recon"] == ref["recon"], f"{case}: reconstruction differs from the reference encoder"
    for hk in must:
        handled, fallback = got["hooks"].get(hk, (0, 0))
        assert handled > 0 and fallback == 0, f"{case}: hook {hk} handled={handled} fallback={fallback}\n" + got["log"][-2000:]
    return got


@pytest.mark.parametrize("case", list(CASES))
def test_patched_encoder_without_hooks_is_the_reference(case, workdir):
    """SVT_HIP_HOOKS unset: the patch itself changes nothing."""
    w, h, n, bd, preset, q, _ = CASES[case]
    if case != "cif_8bit_m6":
        pytest.skip("one case is enough for the no-op check")
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, case + ".nohook"), env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR})
    assert (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"]) and not got["hooks"]


@pytest.mark.parametrize("case", list(CASES))
def test_hooked_encode_on_cpu_test_double(case, workdir):
    got = _check(case, CASES[case], workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")
    assert "svt_hip MOCK" in got["log"]


def test_per_unit_wiener_hooks_on_cpu_test_double(workdir):
    """The per-unit path (statistics per unit from the picture-level pass, every refinement probe on the device) with the picture-level search off."""
    case = "cif_8bit_m6"
    spec = CASES[case][:6] + (ALL_UNIT,)
    _check(case, spec, workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": E.ALL_PER_UNIT}, "mock_unit")


def _check_md_tx(workdir, env, tag, case="cif_8bit_m4"):
    """hook "md_tx" (opt-in, not part of "all"): mode decision's transform-type search takes the forward transforms of a block for all its candidate types
    from ONE batched launch (svt_hip_md_bridge.c).  Preset 4 searches several types per block."""
    spec = CASES[case][:6] + ({"md_tx"},)
    return _check(case, spec, workdir, env, tag)


def test_md_tx_type_search_hook_on_cpu_test_double(workdir):
    got = _check_md_tx(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_tx"}, "mock_mdtx")
    assert got["hooks"]["md_tx"][0] > 100, got["hooks"]
    # together with the picture-level hooks
    both = _check("cif_8bit_m4", CASES["cif_8bit_m4"][:6] + (ALL | {"md_tx"},), workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all,md_tx"}, "mock_all_mdtx")
    assert both["hooks"]["md_tx"][0] > 100
    # "all" alone does not switch it on
    plain = E.encode(E.APP_HIP, os.path.join(workdir, "cif_8bit_m4.src.yuv"), 352, 288, 2, 4, 45, 8, os.path.join(workdir, "mdtx_off"),
                     env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"})
    assert "md_tx" not in plain["hooks"]


def test_md_tx_hook_matters(workdir):
    """a wrong coefficient out of the batched transforms changes the encode: the search really consumes them"""
    case = "cif_8bit_m4"
    w, h, n, bd, preset, q, _ = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.bad_mdtx"),
                   env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_tx", "SVT_HIP_MOCK_PERTURB": "md_tx"})
    assert (got["ivf"], got["recon"]) != (ref["ivf"], ref["recon"])


def _check_encdec_tx(workdir, env, tag, cases=("cif_8bit_m6", "cif_10bit_m6")):
    """hook "encdec_tx" (opt-in): the encode pass takes the forward transforms of every transform block of an inter-coded block from one batched launch; the
    bitstream must not change, blocks must really have been batched, and the number of av1_estimate_transform calls it replaced is reported"""
    out = {}
    for case in cases:
        got = _check(case, {**CASES, **GPU_ONLY_CASES}[case][:6] + ({"encdec_tx"},), workdir, env, tag + "_" + case)
        import re
        m = re.search(r"svt_hip_encdec_tx inter_blocks=(\d+) estimate_transform_calls_replaced=(\d+)", got["log"])
        assert m, got["log"][-800:]
        blocks, calls = int(m.group(1)), int(m.group(2))
        print(f"encdec_tx {case}: {blocks} inter blocks batched, {calls} av1_estimate_transform calls replaced")
        assert blocks > 50 and calls >= blocks, (blocks, calls)
        out[case] = got
    return out


def test_encdec_tx_hook_on_cpu_test_double(workdir):
    _check_encdec_tx(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "encdec_tx"}, "mock_edtx")
    # together with every picture-level hook and the mode-decision one
    case = "cif_8bit_m4"
    both = _check(case, CASES[case][:6] + (ALL | {"md_tx", "encdec_tx"},), workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all,md_tx,encdec_tx"}, "mock_all_edtx")
    assert both["hooks"]["encdec_tx"][0] > 50


def test_encdec_tx_hook_matters(workdir):
    """a wrong coefficient out of the batched transforms changes the encode: the encode pass really consumes them"""
    case = "cif_8bit_m6"
    w, h, n, bd, preset, q, _ = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.bad_edtx"),
                   env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "encdec_tx", "SVT_HIP_MOCK_PERTURB": "encdec_tx"})
    assert (got["ivf"], got["recon"]) != (ref["ivf"], ref["recon"])


def _encdec_sb_line(log):
    m = re.search(r"svt_hip_encdec_sb superblocks=(\d+) launches=(\d+) inter_blocks_predicted_ahead=(\d+) estimate_transform_calls_replaced=(\d+) kernel_launches=(\d+)", log)
    assert m, log[-800:]
    return tuple(int(v) for v in m.groups())


def _check_encdec_sb(workdir, env, tag, cases=("cif_8bit_m6", "cif_10bit_m6", "cif_8bit_m4")):
    """hook "encdec_sb" (opt-in): ONE launch per superblock -- the plain-translation inter blocks of a superblock's final partition are predicted ahead of the block loop
    of av1_encode_decode (the reference's own predictor, called early) and the forward transforms of all their transform blocks run in one launch.  The bitstream
    must not change; launches <= superblocks of the inter pictures; several blocks per launch on average."""
    out = {}
    for case in cases:
        spec = {**CASES, **GPU_ONLY_CASES}[case]
        got = _check(case, spec[:6] + ({"encdec_sb"},), workdir, env, tag + "_" + case)
        sbs, launches, blocks, calls, kernels = _encdec_sb_line(got["log"])
        assert launches <= kernels <= launches + launches // 8, (launches, kernels)   # one KERNEL launch per superblock (a second one only above 16 (plane, transform size) pairs)
        w, h, n = spec[:3]
        print(f"encdec_sb {case}: {launches} launches for {sbs} superblocks with inter blocks, {blocks} blocks predicted ahead, {calls} av1_estimate_transform calls replaced")
        assert 0 < launches <= sbs <= ((w + 63) // 64) * ((h + 63) // 64) * (n - 1), (sbs, launches)   # the first picture is intra-only
        assert blocks > 2 * launches and calls >= blocks, (blocks, calls, launches)
        out[case] = got
    return out


def test_encdec_sb_hook_on_cpu_test_double(workdir):
    _check_encdec_sb(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "encdec_sb"}, "mock_edsb")
    # with every picture-level hook and the other opt-in hooks of the encode pass / mode decision (encdec_tx then only sees the blocks the superblock launch left out)
    case = "cif_8bit_m4"
    both = _check(case, CASES[case][:6] + (ALL | {"md_tx", "encdec_sb"},), workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all,md_tx,encdec_tx,encdec_sb"}, "mock_all_edsb")
    assert both["hooks"]["encdec_sb"][0] > 20


def test_encdec_sb_geometries_on_cpu_test_double(workdir):
    """128 x 128 superblocks, a padded source size, the slowest and the fastest preset"""
    env = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "encdec_sb"}
    for name, w, h, n, bd, preset, q, seed in (("sb128_edsb", 640, 360, 2, 8, 4, 42, 13), ("padded_edsb", 130, 66, 5, 8, 6, 38, 11), ("m8_edsb", 352, 288, 5, 10, 8, 36, 6), ("m2_edsb", 176, 144, 3, 8, 2, 40, 5)):   # (sizes kept small: the C reference at presets 2 / 4 is what this test waits for)
        got = _check_geometry(name, w, h, n, bd, preset, q, seed, workdir, env, "mock", must={"encdec_sb"})
        assert _encdec_sb_line(got["log"])[2] > 0


def test_encdec_sb_hook_matters(workdir):
    """a wrong coefficient out of the superblock's launch changes the encode: the encode pass really consumes it"""
    case = "cif_8bit_m6"
    w, h, n, bd, preset, q, _ = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.bad_edsb"),
                   env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "encdec_sb", "SVT_HIP_MOCK_PERTURB": "encdec_tx"})
    assert (got["ivf"], got["recon"]) != (ref["ivf"], ref["recon"])


@pytest.mark.gpu
def test_encdec_sb_hook_on_gpu(workdir):
    got = _check_encdec_sb(workdir, {"SVT_HIP_HOOKS": "encdec_sb"}, "hip_edsb", cases=("cif_8bit_m6", "cif_10bit_m6", "cif_8bit_m2", "720p_8bit_m6"))
    assert all("svt_hip MOCK" not in g["log"] for g in got.values())


def _check_md_subpel(workdir, env, tag, cases=("cif_8bit_m4", "cif_8bit_m6")):
    """hook "md_subpel" (opt-in): every round of mode decision's sub-pel tree takes its candidates' (variance, sse) from one batched launch pair"""
    out = {}
    for case in cases:
        got = _check(case, {**CASES, **GPU_ONLY_CASES}[case][:6] + ({"md_subpel"},), workdir, env, tag + "_" + case)
        assert got["hooks"]["md_subpel"][0] > 100, got["hooks"]
        out[case] = got
    return out


def test_md_subpel_hook_on_cpu_test_double(workdir):
    _check_md_subpel(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_subpel"}, "mock_mdsp")
    # with every other hook as well
    case = "cif_8bit_m4"
    both = _check(case, CASES[case][:6] + (ALL | {"md_tx", "encdec_tx", "md_subpel"},), workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all,md_tx,encdec_tx,md_subpel"}, "mock_all_mdsp")
    assert both["hooks"]["md_subpel"][0] > 100


def test_md_subpel_hook_matters(workdir):
    """a wrong variance out of the batched round changes the encode: the sub-pel tree really consumes it"""
    case = "cif_8bit_m4"
    w, h, n, bd, preset, q, _ = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.bad_mdsp"),
                   env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_subpel", "SVT_HIP_MOCK_PERTURB": "md_subpel"})
    assert (got["ivf"], got["recon"]) != (ref["ivf"], ref["recon"])


@pytest.mark.gpu
def test_md_subpel_hook_on_gpu(workdir):
    got = _check_md_subpel(workdir, {"SVT_HIP_HOOKS": "md_subpel"}, "hip_mdsp", cases=("cif_8bit_m4", "cif_8bit_m6", "cif_8bit_m2"))
    assert all("svt_hip MOCK" not in g["log"] for g in got.values())
    print({k: g["hooks"]["md_subpel"] for k, g in got.items()})


@pytest.mark.gpu
def test_encdec_tx_hook_on_gpu(workdir):
    got = _check_encdec_tx(workdir, {"SVT_HIP_HOOKS": "encdec_tx"}, "hip_edtx", cases=("cif_8bit_m6", "cif_10bit_m6", "cif_8bit_m2"))
    assert all("svt_hip MOCK" not in g["log"] for g in got.values())


@pytest.mark.gpu
def test_md_tx_type_search_hook_on_gpu(workdir):
    got = _check_md_tx(workdir, {"SVT_HIP_HOOKS": "md_tx"}, "hip_mdtx")
    assert "svt_hip MOCK" not in got["log"] and got["hooks"]["md_tx"][0] > 100, got["hooks"]


def _md_pre_line(log):
    m = re.search(r"svt_hip_md_pre pictures=(\d+) launches=(\d+) blocks=(\d+) min_blocks_per_launch=(\d+) declined=(\d+) config_thread_ms=[0-9.]+ fast_loop_calls=(\d+) inter=(\d+) "
                  r"served_from_table=(\d+) predicted_late=(\d+)", log)
    assert m, log[-1500:]
    return dict(zip(("pictures", "launches", "blocks", "min_blocks", "declined", "calls", "inter", "served", "late"), map(int, m.groups())))


def _md_pre_subpel_line(log):
    m = re.search(r"svt_hip_md_pre_subpel grid_pictures=(\d+) probes=(\d+) served_from_grid=(\d+)", log)
    assert m, log[-1500:]
    return dict(zip(("pictures", "probes", "served"), map(int, m.groups())))


def _md_pre_compound_line(log):
    m = re.search(r"svt_hip_md_pre_compound pair_table_pictures=(\d+) served_from_pair_table=(\d+)", log)
    assert m, log[-1500:]
    return dict(zip(("pictures", "served"), map(int, m.groups())))


def _check_md_pre(workdir, env, tag, cases=("cif_8bit_m6", "cif_10bit_m6", "360p_8bit_m7", "328x200_8bit_m6", "cif_10bit_m8")):
    """hook "md_pre" (opt-in): ONE launch per picture, before the picture's mode decision starts, computes the stage-0 luma distortion of every (superblock, square PU,
    reference picture) at its open-loop ME vector; fast_loop_core (EbProductCodingLoop.c:907) reads the table instead of predicting + measuring, full_loop_core predicts
    the survivors.  Identical bitstream / reconstruction; every launch covers far more than 256 blocks; most of mode decision's inter fast-loop calls are served."""
    out = {}
    for case in cases:
        spec = {**CASES, **GPU_ONLY_CASES}[case]
        got = _check(case, spec[:6] + ({"md_pre"},), workdir, env, tag + "_" + case)
        st = _md_pre_line(got["log"])
        assert st["pictures"] > 0 and st["launches"] == st["pictures"] + _md_pre_subpel_line(got["log"])["pictures"] + _md_pre_compound_line(got["log"])["pictures"] and st["min_blocks"] >= 256, st   # one launch per table
        # (10-bit input: mode decision's fast loop works on 16-bit samples -- hbd_mode_decision 2 at this preset --, and the two distortion tables are made on the packed source and
        # reference_picture16bit: svt_hip_md_fullpel_sad_picture_hbd_dev)
        assert st["served"] * 2 > st["inter"], f"{case}: fewer than half of the inter fast-loop calls were served from the table: {st}"
        sp = _md_pre_subpel_line(got["log"])
        # the sub-pel grid: made for every picture with a table; md_subpel_search's own probes (the open-loop ME vectors' refinement) are served from it -- the other half of
        # the svt_upsampled_pref_error calls belongs to the predictive ME's sub-pel search, which starts where a neighbour-dependent full-pel search ended (8-bit and 10-bit input:
        # this search always works on the 8-bit planes)
        assert sp["pictures"] == st["pictures"] and (sp["probes"] == 0 or sp["served"] * 4 > sp["probes"]), sp
        print(f"md_pre {case}: {st} {sp}")
        out[case] = got
    return out


def test_md_pre_hook_on_cpu_test_double(workdir):
    _check_md_pre(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_pre"}, "mock_mdpre")
    # with every other hook as well, and with the slower presets, whose first pass refines the vectors (nothing to serve: the reference's path, same bitstream)
    for case, hooks in (("cif_8bit_m6", "all,md_pre"), ("cif_8bit_m4", "all,md_pre,md_subpel")):
        both = _check(case, CASES[case][:6] + (ALL | {"md_pre"},), workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": hooks}, "mock_all_mdpre")
        assert both["hooks"]["md_pre"][0] > 0


def test_md_pre_full_mini_gop_and_self_check_on_cpu_test_double(workdir):
    """A complete hierarchical mini-GOP (nine frames: B pictures with two reference lists, warped motion allowed -- the fast cost of stage 0 then reads the candidate's
    warped-motion sample count, which the reference's predictor computes on the side and a table hit has to supply), and the hook's self check: with
    SVT_HIP_MD_PRE_VERIFY=1 every table hit is computed by the reference's own code as well and must agree."""
    env = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_pre"}
    got = _check_geometry("gop9_mdpre", 352, 288, 9, 8, 6, 38, 29, workdir, env, "mock", must={"md_pre"})
    st = _md_pre_line(got["log"])
    assert st["served"] * 2 > st["inter"] and st["late"] > 0, st
    got = _check_geometry("gop9_mdpre", 352, 288, 9, 8, 6, 38, 29, workdir, {**env, "SVT_HIP_MD_PRE_VERIFY": "1"}, "mock_verify", must={"md_pre"})
    assert re.search(r"served_from_table=[1-9]\d* predicted_late=0 verify_mismatches=0\b", got["log"]), got["log"][-800:]


def _check_md_pre_compound(workdir, env, tag):
    """Seventeen frames (two hierarchical mini-GOPs): the B pictures' bi-directional ME candidates become NEW_NEWMV / COMPOUND_AVERAGE candidates of stage 0, served from the
    picture's pair table (svt_hip_md_fullpel_avg_sad_picture_dev); with the self check every hit equals the reference's own compound prediction + SAD."""
    got = _check_geometry("gop17_mdpre", 352, 288, 17, 8, 6, 36, 5, workdir, env, tag, must={"md_pre"})
    st, bi = _md_pre_line(got["log"]), _md_pre_compound_line(got["log"])
    assert bi["pictures"] > 0 and bi["served"] > 1000 and st["served"] * 2 > st["inter"], (st, bi)
    assert re.search(r"svt_hip_md_pre_misses compound=0 ", got["log"]), got["log"][-800:]   # this preset's compound candidates are all averages of two ME vectors
    got = _check_geometry("gop17_mdpre", 352, 288, 17, 8, 6, 36, 5, workdir, {**env, "SVT_HIP_MD_PRE_VERIFY": "1"}, tag + "_verify", must={"md_pre"})
    assert re.search(r"served_from_table=[1-9]\d* predicted_late=0 verify_mismatches=0\b", got["log"]), got["log"][-800:]
    return got


def _check_md_pre_compound_10bit(workdir, env, tag):
    """the same on a 10-bit clip: single-reference and compound-average candidates of the 16-bit fast loop from the tables made on 16-bit planes, the self check on"""
    got = _check_geometry("gop17_10bit_mdpre", 352, 288, 17, 10, 6, 36, 5, workdir, env, tag, must={"md_pre"})
    st, bi = _md_pre_line(got["log"]), _md_pre_compound_line(got["log"])
    assert bi["served"] > 1000 and st["served"] * 2 > st["inter"] and re.search(r"svt_hip_md_pre_misses compound=0 motion_mode=\d+ hbd=0 ", got["log"]), (st, bi)
    got = _check_geometry("gop17_10bit_mdpre", 352, 288, 17, 10, 6, 36, 5, workdir, {**env, "SVT_HIP_MD_PRE_VERIFY": "1"}, tag + "_verify", must={"md_pre"})
    assert re.search(r"served_from_table=[1-9]\d* predicted_late=0 verify_mismatches=0\b", got["log"]), got["log"][-800:]
    return got


def test_md_pre_compound_average_candidates_on_cpu_test_double(workdir):
    env = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_pre"}
    _check_md_pre_compound(workdir, env, "mock")
    _check_md_pre_compound_10bit(workdir, env, "mock")
    # a wrong distortion out of the pair table changes the encode; without the pair table (SVT_HIP_MD_PRE_COMPOUND=0) the compound candidates are the reference's again
    clip = os.path.join(workdir, "gop17_mdpre.src.yuv")
    ref = _ref_cache["gop17_mdpre"] if "gop17_mdpre" in _ref_cache else None
    got = E.encode(E.APP_HIP, clip, 352, 288, 17, 6, 36, 8, os.path.join(workdir, "gop17_mdpre.bad"), env_extra={**env, "SVT_HIP_MOCK_PERTURB": "md_pre_compound"})
    good = E.encode(E.APP_HIP, clip, 352, 288, 17, 6, 36, 8, os.path.join(workdir, "gop17_mdpre.nopairs"), env_extra={**env, "SVT_HIP_MD_PRE_COMPOUND": "0"})
    assert _md_pre_compound_line(good["log"])["served"] == 0 and re.search(r"svt_hip_md_pre_misses compound=[1-9]", good["log"])
    assert (got["ivf"], got["recon"]) != (good["ivf"], good["recon"])
    if ref is not None:
        assert (good["ivf"], good["recon"]) == (ref["ivf"], ref["recon"])


@pytest.mark.gpu
def test_md_pre_compound_average_candidates_on_gpu(workdir):
    got = _check_md_pre_compound(workdir, {"SVT_HIP_HOOKS": "md_pre"}, "hip")
    assert "svt_hip MOCK" not in got["log"]
    got = _check_md_pre_compound_10bit(workdir, {"SVT_HIP_HOOKS": "md_pre"}, "hip")
    assert "svt_hip MOCK" not in got["log"]


def test_md_pre_subpel_grid_matters(workdir):
    """a wrong variance out of the picture's sub-pel grid changes the encode: the sub-pel tree really consumes it"""
    case = "cif_8bit_m6"
    w, h, n, bd, preset, q, _ = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.bad_mdpre_grid"),
                   env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_pre", "SVT_HIP_MOCK_PERTURB": "md_pre_subpel"})
    assert (got["ivf"], got["recon"]) != (ref["ivf"], ref["recon"])
    # ... and without the grid (SVT_HIP_MD_PRE_SUBPEL=0) the hook still codes the reference's bitstream, serving stage 0 only
    got = _check(case, CASES[case][:6] + ({"md_pre"},), workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_pre", "SVT_HIP_MD_PRE_SUBPEL": "0"}, "mock_mdpre_nogrid")
    assert _md_pre_subpel_line(got["log"])["served"] == 0 and _md_pre_line(got["log"])["served"] > 0


def test_md_pre_hook_matters(workdir):
    """a wrong distortion out of the picture's table changes the encode: stage 0 really consumes it"""
    case = "cif_8bit_m6"
    w, h, n, bd, preset, q, _ = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.bad_mdpre"),
                   env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_pre", "SVT_HIP_MOCK_PERTURB": "md_pre"})
    assert (got["ivf"], got["recon"]) != (ref["ivf"], ref["recon"])


def test_md_pre_variants_on_cpu_test_double(workdir):
    """128 x 128 superblocks (the hook declines: the reference's path), a padded source size, several reference pictures per list"""
    env = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_pre"}
    got = _check_geometry("sb128_mdpre", 640, 360, 3, 8, 4, 42, 13, workdir, env, "mock", must=set())
    assert _md_pre_line(got["log"])["served"] == 0
    got = _check_geometry("padded_mdpre", 130, 66, 5, 8, 6, 38, 11, workdir, env, "mock", must={"md_pre"})
    assert _md_pre_line(got["log"])["served"] > 0
    got = _check_geometry("lowdelay_mdpre", 352, 288, 6, 8, 6, 40, 23, workdir, env, "mock", must={"md_pre"}, extra=["--pred-struct", "1"])
    assert _md_pre_line(got["log"])["served"] > 0


@pytest.mark.gpu
def test_md_pre_hook_on_gpu(workdir):
    got = _check_md_pre(workdir, {"SVT_HIP_HOOKS": "md_pre"}, "hip_mdpre", cases=("cif_8bit_m6", "cif_10bit_m6", "360p_8bit_m7", "720p_8bit_m6", "cif_8bit_18_frames", "cif_10bit_m8"))
    assert all("svt_hip MOCK" not in g["log"] for g in got.values())
    case = "720p_8bit_m6"
    both = _check(case, GPU_ONLY_CASES[case][:6] + (ALL | {"md_pre"},), workdir, {"SVT_HIP_HOOKS": "all,md_pre"}, "hip_all_mdpre")
    assert "svt_hip MOCK" not in both["log"] and _md_pre_line(both["log"])["served"] > 1000


@pytest.mark.parametrize("hooks", ["hme", "me"])
def test_motion_estimation_hooks_one_at_a_time_on_cpu_test_double(hooks, workdir):
    """The segment's SB loop runs a different pass sequence for every combination of the two ME hooks (svt_hip_me_bridge.c): "hme" alone = three
    level passes + the rest of motion_estimate_sb with the reference's integer search; "me" alone = the reference's HME inside the first pass.
    Both together are what every SVT_HIP_HOOKS=all case above runs.  40 SBs in many ME segments."""
    case = "360p_8bit_m7"
    _check(case, CASES[case][:6] + ({hooks},), workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": hooks}, "mock_" + hooks)


def _resident_line(log):
    m = re.findall(r"svt_hip_resident notes=(\d+) uploads=(\d+) uploaded_mb=([0-9.]+) hits=(\d+)", log)
    assert m, "no svt_hip_resident line in the report\n" + log[-1500:]
    notes, uploads, mb, hits = m[-1]   # one line per encoder instance of the process (two-pass runs), counters running on
    return int(notes), int(uploads), float(mb), int(hits)


@pytest.mark.parametrize("case", ["360p_8bit_m7", "cif_8bit_m4", "cif_10bit_m6"])
def test_resident_source_planes_on_cpu_test_double(case, workdir):
    """Resident planes (the default; set explicitly here): the padded luma plane of every picture and its 1/4 and 1/16 versions are uploaded once per (re)write -- picture analysis, the end of the
    temporal filter -- and every ME / HME / TF-ME segment reads that copy instead of uploading its own row band (integration/svt_hip_hooks.c).  Host logic only (which
    copy is current), so the CPU test double pins it: a stale plane changes the motion search and with it the bitstream.  The planes must really have been used."""
    # SVT_HIP_SEGMENTS=0: the reference's own ME / TF segmenting (60 segments per 360p picture), i.e. many readers per plane; the hooks' default makes the segments larger
    got = _check(case, CASES[case], workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_RESIDENT": "1", "SVT_HIP_SEGMENTS": "0"}, "mock_resident")
    notes, uploads, mb, hits = _resident_line(got["log"])
    w, h, n = CASES[case][:3]
    # every announced plane travels at most once per announcement, and the copies are read (CIF pictures are one ME segment each: few readers per plane;
    # the 360p case has many segments per picture: most reads find the plane on the device)
    assert notes >= 3 * n and n <= uploads <= notes and hits > 0, (notes, uploads, mb, hits)
    if case == "360p_8bit_m7":
        assert hits > 10 * uploads, (notes, uploads, mb, hits)


def _flushes(log, hook):
    m = re.findall(r"svt_hip_hook_time %s calls=(\d+)" % hook, log)
    return int(m[-1]) if m else 0


def test_larger_me_and_tf_segments_on_cpu_test_double(workdir):
    """With the ME / TF hooks on a segment is one batched launch per stage, so the patched load_default_buffer_configuration_settings makes the segments larger
    (svt_hip_hooks_segments: the reference cuts a 640 x 360 picture into 60 segments); the bitstream does not depend on the cut, the number of batched launches does."""
    case = "360p_8bit_m7"
    env = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}
    few = _check(case, CASES[case], workdir, env, "mock_segments_default")
    many = _check(case, CASES[case], workdir, dict(env, SVT_HIP_SEGMENTS="0"), "mock_segments_reference")
    assert 0 < _flushes(few["log"], "hme") * 8 < _flushes(many["log"], "hme"), (_flushes(few["log"], "hme"), _flushes(many["log"], "hme"))
    assert 0 < _flushes(few["log"], "me") * 8 < _flushes(many["log"], "me")


@pytest.mark.parametrize("case", ["360p_8bit_m7", "cif_10bit_m6"])
def test_without_resident_source_planes_on_cpu_test_double(case, workdir):
    """SVT_HIP_RESIDENT=0 (the default until round 4): every ME / HME / TF segment uploads its own row bands again; no table line in the report"""
    got = _check(case, CASES[case], workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_RESIDENT": "0"}, "mock_not_resident")
    assert not re.findall(r"svt_hip_resident notes=[1-9]", got["log"])


def test_resident_source_planes_are_the_default_on_cpu_test_double(workdir):
    case = "cif_8bit_m4"
    got = _check(case, CASES[case], workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock_resident_default")
    assert _resident_line(got["log"])[3] > 0


def test_resident_source_planes_with_a_small_budget_on_cpu_test_double(workdir):
    """SVT_HIP_RESIDENT_MB=1: hardly any plane fits, the oldest unused copies are dropped and uploaded again on demand (or the caller uploads its band as before)."""
    case = "360p_8bit_m7"
    got = _check(case, CASES[case], workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_RESIDENT": "1", "SVT_HIP_RESIDENT_MB": "1"}, "mock_resident_1mb")
    assert re.search(r"svt_hip_resident notes=\d+ uploads=\d+ uploaded_mb=[0-9.]+ hits=\d+ evictions=(\d+)", got["log"])


@pytest.mark.parametrize("stage", E.HOOKS)
def test_every_hook_matters(stage, workdir):
    """A deliberately wrong answer of ONE stage (SVT_HIP_MOCK_PERTURB) must change the bitstream or the reconstruction: the comparison above
    is sensitive to every hook's output, none of them is dead weight."""
    case = "cif_8bit_m6"
    w, h, n, bd, preset, q, _ = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.bad_{stage}"),
                   env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": E.ALL_PER_UNIT if stage in E.PER_UNIT_WIENER else "all", "SVT_HIP_MOCK_PERTURB": stage})
    assert (got["ivf"], got["recon"]) != (ref["ivf"], ref["recon"]), f"perturbing {stage} went unnoticed"


# Encoder options that change what reaches the hooks.  (extra arguments, --lp): one thread where the unpatched reference itself is not
# repeatable with eight (segment-based adaptive quantisation; VBR; a 0 x 0 HME level-2 area, whose centres are whatever the thread's previous
# search left behind -- the bridge reproduces that in the reference's block order).
OPTION_VARIANTS = {
    "tiles_2x2": (["-tile-columns", "1", "-tile-rows", "1"], 8),
    "hme_off": (["-hme", "0"], 8),
    "hme_level0_only": (["-hme-l1", "0", "-hme-l2", "0"], 8),
    "search_area_128x48_user_hme": (["-use-default-me-hme", "0", "-search-w", "128", "-search-h", "48"], 1),
    "screen_content": (["-scm", "1"], 8),             # ME search areas above 65 536 candidates
    "low_delay_p": (["-pred-struct", "0"], 8),
    "three_layers": (["-hierarchical-levels", "3"], 8),
    "altref_7_frames": (["-altref-nframes", "7", "-altref-strength", "6"], 8),
    "film_grain": (["-film-grain", "8"], 8),          # noise estimation + denoised source
    "segment_aq": (["-adaptive-quantization", "1"], 1),
    "sg_mode_1_wiener_mode_1_cdef_level_1": (["-sg-filter-mode", "1", "-wn-filter-mode", "1", "-cdef-level", "1"], 8),
    "sg_mode_4_wiener_mode_3_cdef_level_4": (["-sg-filter-mode", "4", "-wn-filter-mode", "3", "-cdef-level", "4"], 8),
    "two_pass_vbr": (["--passes", "2", "--stats", "{stats}", "--rc", "1", "--tbr", "500"], 1),   # the first pass drives the same hooks
    "vbr": (["-rc", "1", "-tbr", "400"], 1),         # with eight threads the rate-control feedback arrives when the pipeline's timing lets it
}


# The reference's film-grain path (v0.8.6) reads heap memory it has not written: the unpatched encoder's own bitstream changes with glibc's MALLOC_PERTURB_ (any fill
# value gives another one; without -film-grain it does not move), i.e. it depends on what the heap held before -- in a process that has loaded the HIP runtime that is
# no longer a fresh mapping's zeros (found on the MI355X in round 4: the hooked encoder's film-grain output differed from run to run with ANY subset of the hooks).
# A fixed fill makes both encoders read the same bytes; the comparison then pins the hooks as for every other variant.
VARIANT_ENV = {"film_grain": {"MALLOC_PERTURB_": "85"}}


def _check_variant(name, workdir, env, tag):
    extra, lp = OPTION_VARIANTS[name]
    env = dict(env, **VARIANT_ENV.get(name, {}))
    w, h, n, bd, preset, q = 352, 288, 6, 8, 6, 38
    clip = os.path.join(workdir, "variants.src.yuv")
    if not os.path.exists(clip):
        E.make_clip(clip, w, h, n, seed=5, bd=bd)
    key = "variant_" + name
    if key not in _ref_cache:
        _ref_cache[key] = E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, key + ".ref"), env_extra=VARIANT_ENV.get(name),
                                   extra_args=[a.replace("{stats}", os.path.join(workdir, key + ".ref.stat")) for a in extra], lp=lp)
    ref = _ref_cache[key]
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{key}.{tag}"), env_extra=env,
                   extra_args=[a.replace("{stats}", os.path.join(workdir, f"{key}.{tag}.stat")) for a in extra], lp=lp)
    assert got["ivf"] == ref["ivf"], f"{name}: bitstream differs from the reference encoder\n" + got["log"][-2000:]
    assert got["recon"] == ref["recon"], f"{name}: reconstruction differs from the reference encoder"
    assert sum(v[0] for v in got["hooks"].values()) > 20, got["hooks"]
    assert all(v[1] == 0 for v in got["hooks"].values()), f"{name}: a hook fell back to the C path: {got['hooks']}\n" + got["log"][-2000:]
    return got


def _check_geometry(name, w, h, n, bd, preset, q, seed, workdir, env, tag, must=None, extra=()):
    """hooked vs unpatched encoder on one more picture geometry: identical output, every hook the preset uses handled, nothing handed back"""
    clip = os.path.join(workdir, name + ".src.yuv")
    if not os.path.exists(clip):
        E.make_clip(clip, w, h, n, seed=seed, bd=bd)
    if name not in _ref_cache:
        _ref_cache[name] = E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, name + ".ref"), extra_args=extra)
    ref = _ref_cache[name]
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{name}.{tag}"), env_extra=env, extra_args=extra)
    assert got["ivf"] == ref["ivf"], f"{name}: bitstream differs from the reference encoder\n" + got["log"][-2000:]
    assert got["recon"] == ref["recon"], f"{name}: reconstruction differs from the reference encoder"
    for hk in (ALL if must is None else must):
        handled, fallback = got["hooks"].get(hk, (0, 0))
        assert handled > 0 and fallback == 0, f"{name}: hook {hk} handled={handled} fallback={fallback}\n" + got["log"][-2000:]
    assert all(v[1] == 0 for v in got["hooks"].values()), f"{name}: a hook handed a picture back to the C loops: {got['hooks']}"
    return got


def _check_sb128(workdir, env, tag):
    """Presets <= M4 code with 128 x 128 superblocks above the 240p range (EbEncHandle.c:2105-2111): the source-side hooks work on their own
    64 x 64 grid as in the reference; deblocking takes the last superblock row / column of the 128 grid, the CDEF search merges the filter blocks
    of unsplit 128-wide / 128-high blocks (cdef_seg_search), restoration units are independent of the superblock size.  640 x 360: the last
    superblock row is 104 rows high, the last column is complete."""
    return _check_geometry("sb128", 640, 360, 3, 8, 4, 42, 13, workdir, env, tag)


def _check_sb128_10bit(workdir, env, tag):
    """the same with the 16-bit pipeline, a width that ends inside a 128-wide superblock (480 = 3 x 128 + 96) and a padded height (270 -> 272)"""
    return _check_geometry("sb128_10bit", 480, 270, 3, 10, 4, 36, 17, workdir, env, tag)


def test_128_superblocks_on_cpu_test_double(workdir):
    _check_sb128(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")


@pytest.mark.gpu
def test_128_superblocks_on_gpu(workdir):
    _check_sb128(workdir, {"SVT_HIP_HOOKS": "all"}, "hip")


def _check_padded_size(workdir, env, tag):
    """Source sizes that are not multiples of 8 are coded padded (130 x 66 -> 136 x 72) while the reference deblocks the last superblock row / column
    only up to the unpadded extent (EbDeblockingFilter.c:343-367), searches CDEF and the filter level on the coded size, and runs the restoration
    search and filter on the cropped frame, whose 3-sample extension overwrites coded samples (link_eb_to_aom_buffer_desc, EbDlfProcess.c:247-251;
    svt_extend_frame, EbCdefProcess.c:552-572).  Every loop-filter hook takes such pictures."""
    return _check_geometry("padded", 130, 66, 5, 8, 6, 38, 11, workdir, env, tag, must=ALL - {"tf_me", "tf_subpel"})   # the alt-ref window of so small a picture is the central frame alone


def _check_padded_size_64(workdir, env, tag):
    """a padded size whose coded size IS a multiple of the superblock size (186 x 122 -> 192 x 128): the reference's crop test never fires and the
    padding is deblocked like picture content (the quirk svt_hip_dlf_filtered_units restates); restoration still works on the cropped 186 x 122"""
    return _check_geometry("padded64", 186, 122, 4, 10, 6, 34, 19, workdir, env, tag, must=ALL - {"tf_me", "tf_subpel"})   # the alt-ref window of so small a picture is the central frame alone


def test_padded_source_size_on_cpu_test_double(workdir):
    _check_padded_size(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")


@pytest.mark.gpu
def test_padded_source_size_on_gpu(workdir):
    _check_padded_size(workdir, {"SVT_HIP_HOOKS": "all"}, "hip")


def _check_480p_m4(workdir, env, tag):
    """854 x 480, preset 4: 128 x 128 superblocks AND a padded width (854 -> 856, the last superblock column is 88 samples of which 2 are padding)"""
    return _check_geometry("480p_m4", 854, 480, 3, 8, 4, 40, 23, workdir, env, tag)


def test_854x480_preset_4_on_cpu_test_double(workdir):
    _check_480p_m4(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")


@pytest.mark.gpu
def test_854x480_preset_4_on_gpu(workdir):
    _check_480p_m4(workdir, {"SVT_HIP_HOOKS": "all"}, "hip")


def _check_level_search_grid(workdir, env, tag):
    """The level search's mode-info grid stays on the device and serves the deblocking of the same picture (frame-uniform levels as arguments of the device
    edge builder); SVT_HIP_DLF_EDGES=host = the host builder and its upload for both stages.  Same bitstream either way."""
    got = _check_geometry("480p_m4_q63", 854, 480, 3, 8, 4, 63, 23, workdir, dict(env, SVT_HIP_VERBOSE="1"), tag + ".grid")   # q 63: the search settles on levels > 0
    filt = re.findall(r"dlf: levels [^\n]*", got["log"])
    assert filt and all("the level search's grid" in l for l in filt), got["log"][-1500:]
    host = _check_geometry("480p_m4_q63", 854, 480, 3, 8, 4, 63, 23, workdir, dict(env, SVT_HIP_VERBOSE="1", SVT_HIP_DLF_EDGES="host"), tag + ".hostedges")
    filt = re.findall(r"dlf: levels [^\n]*", host["log"])
    assert filt and all("mode info refilled" in l for l in filt), filt


def _check_wiener_initial_filters(workdir, env, tag):
    """The Wiener search's initial filters on the device (statistics, decomposition and walks queued back to back, one wait) and, SVT_HIP_WIENER_INIT=host, by the
    reference's own decomposition on downloaded statistics: same bitstream, and the log says which one ran."""
    dev = _check_geometry("wn_init", 352, 288, 4, 8, 6, 35, 41, workdir, dict(env, SVT_HIP_VERBOSE="1"), tag + ".dev")
    lines = re.findall(r"wiener_search: 1 launch[^\n]*", dev["log"])
    assert lines and all("initial filters (device)" in l for l in lines), dev["log"][-1500:]
    assert sum(int(re.search(r"(\d+) walks", l).group(1)) for l in lines) > 0
    host = _check_geometry("wn_init", 352, 288, 4, 8, 6, 35, 41, workdir, dict(env, SVT_HIP_VERBOSE="1", SVT_HIP_WIENER_INIT="host"), tag + ".host")
    hl = re.findall(r"wiener_search: 1 launch[^\n]*", host["log"])
    assert hl and all("initial filters (host)" in l for l in hl), hl
    counts = lambda ls: sorted(re.search(r"(\d+) walks, (\d+) probes", l).groups() for l in ls)   # pictures finish in any order
    assert counts(lines) == counts(hl)


def test_wiener_initial_filters_on_cpu_test_double(workdir):
    _check_wiener_initial_filters(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")


def test_wiener_initial_filters_matter(workdir):
    """a wrong initial tap from the device changes the encode (the perturbed test double is noticed)"""
    env = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_MOCK_PERTURB": "wiener_init"}
    with pytest.raises(AssertionError):
        _check_geometry("wn_init", 352, 288, 4, 8, 6, 35, 41, workdir, env, "mock.perturbed")


@pytest.mark.gpu
def test_wiener_initial_filters_on_gpu(workdir):
    _check_wiener_initial_filters(workdir, {"SVT_HIP_HOOKS": "all"}, "hip")


def test_level_search_grid_serves_the_filter_on_cpu_test_double(workdir):
    _check_level_search_grid(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")


@pytest.mark.gpu
def test_level_search_grid_serves_the_filter_on_gpu(workdir):
    _check_level_search_grid(workdir, {"SVT_HIP_HOOKS": "all"}, "hip")


@pytest.mark.gpu
def test_1080p_preset_4_on_gpu(workdir):
    """1920 x 1080 at a slow preset: 128 x 128 superblocks, the last superblock row 56 rows high -- the loop-filter hooks on the geometry the slow presets
    use at every practical resolution"""
    _check_geometry("1080p_m4", 1920, 1080, 2, 8, 4, 40, 29, workdir, {"SVT_HIP_HOOKS": "all"}, "hip")


def test_padded_source_size_multiple_of_64_on_cpu_test_double(workdir):
    _check_padded_size_64(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")


@pytest.mark.gpu
def test_padded_source_size_multiple_of_64_on_gpu(workdir):
    _check_padded_size_64(workdir, {"SVT_HIP_HOOKS": "all"}, "hip")


def test_128_superblocks_10bit_padded_on_cpu_test_double(workdir):
    _check_sb128_10bit(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")


@pytest.mark.gpu
def test_128_superblocks_10bit_padded_on_gpu(workdir):
    _check_sb128_10bit(workdir, {"SVT_HIP_HOOKS": "all"}, "hip")


@pytest.mark.parametrize("name", list(OPTION_VARIANTS))
def test_encoder_option_variants_on_cpu_test_double(name, workdir):
    _check_variant(name, workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}, "mock")


# SVT_HIP_TEST_RESIDENT_ALL=1: every option variant (the CPU suite keeps to the four that move the planes' writers)
RESIDENT_VARIANTS = list(OPTION_VARIANTS) if os.environ.get("SVT_HIP_TEST_RESIDENT_ALL") == "1" else ["altref_7_frames", "film_grain", "low_delay_p", "two_pass_vbr"]


@pytest.mark.parametrize("name", RESIDENT_VARIANTS)
def test_encoder_option_variants_with_resident_planes_on_cpu_test_double(name, workdir):
    """the writers of the resident planes under the options that move them: longer alt-ref windows, the denoised source of film grain, other prediction structures
    (which pictures are filtered, which are overlays), filtered / decimated pyramids, a second pass over the same pictures"""
    got = _check_variant(name, workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_RESIDENT": "1"}, "mock_resident")
    assert _resident_line(got["log"])[3] > 0


def test_resident_planes_announcements_matter(workdir):
    """SVT_HIP_RESIDENT_FAULT=1 ignores every announcement of a plane but its first: the copy of a picture that is filtered after it served as a window frame, or of
    a buffer that a later picture reuses, goes stale -- the encode must notice (the identity tests above are sensitive to exactly the bookkeeping they pin)."""
    case = "cif_8bit_18_frames"
    spec = GPU_ONLY_CASES[case]
    w, h, n, bd, preset, q, _ = spec
    clip, ref = _reference(case, spec, workdir)
    env = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_RESIDENT": "1"}
    good = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, case + ".res_ok"), env_extra=env)
    assert good["ivf"] == ref["ivf"] and good["recon"] == ref["recon"]
    bad = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, case + ".res_fault"), env_extra=dict(env, SVT_HIP_RESIDENT_FAULT="1"))
    assert bad["ivf"] != ref["ivf"] or bad["recon"] != ref["recon"], "stale resident planes went unnoticed"


# ---- deferred pictures (integration/svt_hip_lf_bridge.c): with every loop-filter hook on, the reconstructed picture stays on the device from deblocking to the restoration
# filter and comes back once; the patched process loops skip the host work only the C restoration path reads.  A hook that fails on the way must first bring the host up
# to date (lf_recover), so that the C code that takes over produces the reference's output: SVT_HIP_LF_FAULT makes one hook report a failure on every picture.
LF_FAULTS = ["cdef_search", "cdef_apply", "sgr_search", "wiener_search", "rest_apply"]


def _lf_pictures(log):
    m = re.findall(r"svt_hip_lf_pictures deferred=(\d+) recovered=(\d+) source_planes_resident=(\d+) planes_up=(\d+) up_mb=[0-9.]+ planes_down=(\d+)", log)
    assert m, "no svt_hip_lf_pictures line in the report\n" + log[-1500:]
    return dict(zip(("deferred", "recovered", "resident", "up", "down"), map(int, m[-1])))


def _encode_like(case, spec, workdir, env, tag):
    w, h, n, bd, preset, q, _ = spec
    clip, ref = _reference(case, spec, workdir)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, f"{case}.{tag}"), env_extra=env)
    assert got["ivf"] == ref["ivf"], f"{case} {tag}: bitstream differs from the reference encoder\n" + got["log"][-2000:]
    assert got["recon"] == ref["recon"], f"{case} {tag}: reconstruction differs from the reference encoder"
    return got


def _check_fault_recovery(case, spec, stage, workdir, env, tag):
    got = _encode_like(case, spec, workdir, dict(env, SVT_HIP_HOOKS="all", SVT_HIP_LF_FAULT=stage), f"{tag}_fault_{stage}")
    lf = _lf_pictures(got["log"])
    assert lf["deferred"] == spec[2] and lf["recovered"] == spec[2], lf   # every picture was deferred, every one was brought back for the C code
    assert got["hooks"][stage][1] > 0, got["hooks"]
    return got


@pytest.mark.parametrize("stage", LF_FAULTS)
def test_deferred_picture_recovers_from_a_failed_hook_on_cpu_test_double(stage, workdir):
    _check_fault_recovery("cif_8bit_m6", CASES["cif_8bit_m6"], stage, workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR}, "mock")


def _check_final_download_failure(case, spec, fault, workdir, env, tag):
    """The last step of a deferred picture -- its one download, in svt_hip_hook_picture_done -- fails: nothing has been written to the host yet
    (final_download: the reference's own deblocking / CDEF / restoration chain runs on the coded picture the host still has, with the decisions the hooks left in the
    reference's structures) or copies had been queued (final_download_late: a second complete download).  Either way the encoder must code what the reference codes."""
    got = _encode_like(case, spec, workdir, dict(env, SVT_HIP_HOOKS="all", SVT_HIP_LF_FAULT=fault), f"{tag}_{fault}")
    m = re.search(r"final_retries=(\d+) final_host_chains=(\d+) final_fatal=(\d+)", got["log"])
    assert m, got["log"][-1500:]
    retries, chains, fatal = map(int, m.groups())
    assert fatal == 0 and (chains, retries) == ((spec[2], 0) if fault == "final_download" else (0, spec[2])), (retries, chains, fatal)
    assert all(v[1] == 0 for v in got["hooks"].values()), got["hooks"]
    return got


@pytest.mark.parametrize("fault", ["final_download", "final_download_late"])
@pytest.mark.parametrize("case", ["cif_8bit_m6", "cif_10bit_m6", "328x200_8bit_m6"])
def test_deferred_picture_survives_a_failed_final_download_on_cpu_test_double(case, fault, workdir):
    _check_final_download_failure(case, CASES[case], fault, workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR}, "mock")


@pytest.mark.gpu
@pytest.mark.parametrize("fault", ["final_download", "final_download_late"])
def test_deferred_picture_survives_a_failed_final_download_on_gpu(fault, workdir):
    got = _check_final_download_failure("cif_8bit_m6", CASES["cif_8bit_m6"], fault, workdir, {}, "hip")
    assert "svt_hip MOCK" not in got["log"]


@pytest.mark.parametrize("stage", ["cdef_apply", "rest_apply"])
def test_deferred_10bit_picture_recovers_from_a_failed_hook_on_cpu_test_double(stage, workdir):
    _check_fault_recovery("cif_10bit_m6", CASES["cif_10bit_m6"], stage, workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR}, "mock")


def _padded_spec():
    return (130, 66, 5, 8, 6, 38, ALL - {"tf_me", "tf_subpel"})


def _padded_reference(workdir):
    """130 x 66 is coded 136 x 72: the border svt_extend_frame gives the CROPPED frame falls inside the coded picture -- the final download has to deliver it"""
    name = "padded_defer"
    if name not in _ref_cache:
        w, h, n, bd, preset, q, _ = _padded_spec()
        clip = os.path.join(workdir, name + ".src.yuv")
        E.make_clip(clip, w, h, n, seed=11, bd=bd)
        _ref_cache[name] = (clip, E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, name + ".ref")))
    return name


@pytest.mark.parametrize("stage", [None, "sgr_search", "rest_apply"])
def test_deferred_padded_picture_on_cpu_test_double(stage, workdir):
    name = _padded_reference(workdir)
    env = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}
    if stage:
        _check_fault_recovery(name, _padded_spec(), stage, workdir, env, "mock")
    else:
        assert _lf_pictures(_encode_like(name, _padded_spec(), workdir, env, "mock_defer")["log"])["deferred"] == 5


def test_deferred_pictures_cross_the_bus_once_on_cpu_test_double(workdir):
    """the copies of the loop-filter bridge, counted: deferred = reconstruction up once and down once (+ the source up once; with SVT_HIP_RESIDENT the source is
    the copy the temporal filter and the motion search already use); not deferred (SVT_HIP_DEFER=0) = every stage brings its result back"""
    case = "cif_8bit_m4"   # preset 4: the filter-level search runs too (it shares its upload with the deblocking hook)
    spec = CASES[case]
    n = spec[2]
    mock = {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"}
    a = _lf_pictures(_encode_like(case, spec, workdir, dict(mock, SVT_HIP_RESIDENT="0"), "mock_defer")["log"])
    assert (a["deferred"], a["recovered"], a["up"], a["down"]) == (n, 0, 6 * n, 3 * n), a
    b = _lf_pictures(_encode_like(case, spec, workdir, dict(mock, SVT_HIP_RESIDENT="1"), "mock_defer_res")["log"])
    assert (b["deferred"], b["resident"], b["up"], b["down"]) == (n, 3 * n, 3 * n, 3 * n), b
    c = _lf_pictures(_encode_like(case, spec, workdir, dict(mock, SVT_HIP_DEFER="0"), "mock_nodefer")["log"])
    assert c["deferred"] == 0 and c["down"] > 3 * n, c


@pytest.mark.gpu
@pytest.mark.parametrize("stage", ["cdef_search", "sgr_search", "rest_apply"])
def test_deferred_picture_recovers_from_a_failed_hook_on_gpu(stage, workdir):
    got = _check_fault_recovery("cif_8bit_m6", CASES["cif_8bit_m6"], stage, workdir, {}, "hip")
    assert "svt_hip MOCK" not in got["log"]


@pytest.mark.gpu
def test_deferred_padded_picture_on_gpu(workdir):
    name = _padded_reference(workdir)
    got = _encode_like(name, _padded_spec(), workdir, {"SVT_HIP_HOOKS": "all"}, "hip_defer")
    assert "svt_hip MOCK" not in got["log"] and _lf_pictures(got["log"])["deferred"] == 5
    got = _check_fault_recovery(name, _padded_spec(), "rest_apply", workdir, {}, "hip")
    assert "svt_hip MOCK" not in got["log"]


@pytest.mark.gpu
def test_not_deferred_pictures_on_gpu(workdir):
    """SVT_HIP_DEFER=0: every stage downloads its result (the round-3 behaviour, and what a subset of the hooks still does)"""
    got = _encode_like("cif_8bit_m6", CASES["cif_8bit_m6"], workdir, {"SVT_HIP_HOOKS": "all", "SVT_HIP_DEFER": "0"}, "hip_nodefer")
    assert "svt_hip MOCK" not in got["log"] and _lf_pictures(got["log"])["deferred"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["360p_8bit_m7", "cif_8bit_m4", "720p_8bit_m6", "cif_10bit_m6", "720p_10bit_m5"])
def test_resident_source_planes_on_gpu(case, workdir):
    """Resident planes on the device (the default; set explicitly here): planes uploaded by one context's stream and read by kernels of the other contexts of the pool, 8- and 10-bit"""
    spec = CASES.get(case) or GPU_ONLY_CASES[case]
    got = _check(case, spec, workdir, {"SVT_HIP_HOOKS": "all", "SVT_HIP_RESIDENT": "1"}, "hip_resident")
    assert "svt_hip MOCK" not in got["log"]
    assert _resident_line(got["log"])[3] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["360p_8bit_m7", "720p_10bit_m5"])
def test_without_resident_source_planes_on_gpu(case, workdir):
    """SVT_HIP_RESIDENT=0 on the device: the band uploads of the rounds before"""
    spec = CASES.get(case) or GPU_ONLY_CASES[case]
    got = _check(case, spec, workdir, {"SVT_HIP_HOOKS": "all", "SVT_HIP_RESIDENT": "0"}, "hip_not_resident")
    assert "svt_hip MOCK" not in got["log"] and not re.findall(r"svt_hip_resident notes=[1-9]", got["log"])


@pytest.mark.gpu
def test_resident_source_planes_2160p_on_gpu(workdir):
    """3840 x 2160, preset 6, planes resident: 40 ME segments per picture read one upload.  The reference side is the SIMD build (it codes the C build's
    bitstream: test_simd_build_codes_the_c_builds_bitstream), the C build needs minutes at this size."""
    name, w, h, n, bd, preset, q = "2160p_8bit_m6_res", 3840, 2160, 2, 8, 6, 36
    app_simd = os.path.join(E.REFDIR, "SvtAv1EncApp_simd")
    clip = os.path.join(workdir, name + ".src.yuv")
    E.make_clip(clip, w, h, n, seed=17, bd=bd)
    ref = E.encode(app_simd if os.path.exists(app_simd) else E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, name + ".ref"), timeout=1200)
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, name + ".hip"), env_extra={"SVT_HIP_HOOKS": "all", "SVT_HIP_RESIDENT": "1"}, timeout=1200)
    assert "svt_hip MOCK" not in got["log"]
    assert got["ivf"] == ref["ivf"] and got["recon"] == ref["recon"], got["log"][-2000:]
    assert all(v[1] == 0 for v in got["hooks"].values()), got["hooks"]
    notes, uploads, mb, hits = _resident_line(got["log"])
    assert hits > 10 * uploads, (notes, uploads, mb, hits)
    for f in os.listdir(workdir):
        if f.startswith(name):
            os.remove(os.path.join(workdir, f))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(OPTION_VARIANTS))
def test_encoder_option_variants_on_gpu(name, workdir):
    got = _check_variant(name, workdir, {"SVT_HIP_HOOKS": "all"}, "hip")
    assert "svt_hip MOCK" not in got["log"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES) + list(GPU_ONLY_CASES))
def test_hooked_encode_on_gpu(case, workdir):
    spec = CASES.get(case) or GPU_ONLY_CASES[case]
    got = _check(case, spec, workdir, {"SVT_HIP_HOOKS": "all"}, "hip")
    assert "svt_hip MOCK" not in got["log"], "the GPU test must load svt-av1_amd/libsvtav1_hip.so, not the CPU test double"


@pytest.mark.gpu
@pytest.mark.parametrize("hook", E.HOOKS)
def test_single_hook_on_gpu(hook, workdir):
    """Each hook alone (the others on the C path): a mismatch bisects to a stage."""
    case = "cif_8bit_m6"
    hooks = {hook, "tf"} if hook == "tf_subpel" else {hook}   # the sub-pel stage leaves its predictors on the device for hook "tf": it needs it
    spec = CASES[case][:6] + (hooks,)
    _check(case, spec, workdir, {"SVT_HIP_HOOKS": ",".join(sorted(hooks))}, "hip_" + hook)


@pytest.mark.gpu
def test_cdef_finish_hook_with_the_one_launch_selection_on_gpu(workdir):
    """SVT_HIP_CDEF_SELECT=resident: the strength-pair selection of finish_cdef_search as ONE launch (joint_resident_kernel) inside a real encode, with other
    threads' contexts at work beside it; identical bitstream, no fallback (an incomplete selection would be one: status[0])."""
    case = "cif_8bit_m6"
    hooks = {"cdef_finish", "cdef_search", "me", "hme"}
    spec = CASES[case][:6] + (hooks,)
    got = _check(case, spec, workdir, {"SVT_HIP_HOOKS": ",".join(sorted(hooks)), "SVT_HIP_CDEF_SELECT": "resident"}, "hip_select_resident")
    assert got["hooks"]["cdef_finish"][0] > 0 and got["hooks"]["cdef_finish"][1] == 0, got["hooks"]


# Dispatch-table entries that hand SOME of their calls to the saved C pointer in the encodes below, and why (everything else must stay on the device):
EXPECTED_DELEGATIONS = {
    "per_call": set(),
    "block_level": set(),
    "helpers": set(),
    "hbd": set(),
    "both": set(),
    "subpel": set(),
}


def _check_delegations(got, expected, tag):
    """The wrappers' own bookkeeping (svt_hip_rtcd_report at exit): exactly the table entries listed in `expected` may have handed calls to the saved C
    pointer (calls outside a kernel's domain -- each one explained next to the list), none because a device call failed, and wrappers did run."""
    dump = os.environ.get("SVT_E2E_DUMP")
    if dump:
        with open(dump, "a") as f:
            f.write(f"{tag}: delegated={got['rtcd_delegated']} calls={got['rtcd_calls']}\n")
    assert got["rtcd_calls"] and sum(got["rtcd_calls"].values()) > 0, "no per-call wrapper ran"
    assert all(v[1] == 0 for v in got["rtcd_delegated"].values()), f"{tag}: a wrapper delegated after a DEVICE failure: {got['rtcd_delegated']}\n" + got["log"][-1500:]
    assert set(got["rtcd_delegated"]) == set(expected), f"{tag}: delegated entries {sorted(got['rtcd_delegated'])}, expected {sorted(expected)}"


@pytest.mark.gpu
def test_per_call_wrappers_in_a_real_encode_on_gpu(workdir):
    """SVT_HIP_RTCD: the reference's dispatch-table entries themselves point at the svt_*_hip wrappers (include/svt_hip_rtcd.h) while the encoder
    runs its unchanged per-block loops: HME (svt_sad_loop_kernel), the self-guided filter of the restoration search and of the frame filter,
    the Wiener statistics.  Slow (one launch per call) but it is the literal "drop-in behind the dispatch table" statement on a real encode."""
    case = "cif_8bit_m6"
    w, h, n, bd, preset, q, _ = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    names = "svt_sad_loop_kernel,svt_av1_selfguided_restoration,svt_apply_selfguided_restoration,svt_av1_compute_stats"
    got = E.encode(E.APP_HIP, clip, w, h, 4, preset, q, bd, os.path.join(workdir, case + ".rtcd"), env_extra={"SVT_HIP_RTCD": names}, timeout=1200)
    ref4 = E.encode(E.APP_REF, clip, w, h, 4, preset, q, bd, os.path.join(workdir, case + ".ref4"))
    assert (got["ivf"], got["recon"]) == (ref4["ivf"], ref4["recon"])
    for nme in names.split(","):
        assert f"svt_hip_rtcd {nme} -> hip wrapper" in got["log"]
    assert "delegated to the installed C pointer" not in got["log"], got["log"][-1500:]
    _check_delegations(got, EXPECTED_DELEGATIONS["per_call"], "per_call")


@pytest.mark.gpu
def test_block_level_wrappers_in_a_real_encode_on_gpu(workdir):
    """The block-level pointers (residual, quantizers, inverse transform + add, the 16 loop filters, CDEF direction / block filter, the 8-candidate
    SAD ladders, variance intermediates) behind SVT_HIP_RTCD on a small clip: the mode-decision and encode loops call them hundreds of thousands of
    times, each call a launch of its own.  Bitstream and reconstruction must not change."""
    w, h, n, bd, preset, q = 176, 144, 2, 8, 6, 32
    clip = os.path.join(workdir, "qcif.yuv")
    E.make_clip(clip, w, h, n, seed=5, bd=bd)
    lpf = ",".join(f"svt_aom_lpf_{d}_{k}" for d in ("horizontal", "vertical") for k in (4, 6, 8, 14))
    names = ("svt_residual_kernel8bit,svt_aom_quantize_b,svt_av1_quantize_fp,svt_av1_quantize_fp_32x32,svt_av1_quantize_fp_64x64,svt_av1_inv_txfm_add,"
             "svt_cdef_find_dir,svt_cdef_filter_block,svt_ext_all_sad_calculation_8x8_16x16,svt_ext_eight_sad_calculation_32x32_64x64,"
             "svt_compute_interm_var_four8x8,svt_handle_transform64x64," + lpf)
    ref = E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif.ref"))
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif.rtcd"), env_extra={"SVT_HIP_RTCD": names}, timeout=1500)
    assert (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"])
    for nme in names.split(","):
        assert f"svt_hip_rtcd {nme} -> hip wrapper" in got["log"], nme
    _check_delegations(got, EXPECTED_DELEGATIONS["block_level"], "block_level")


@pytest.mark.gpu
def test_helper_pointers_in_a_real_encode_on_gpu(workdir):
    """The small helpers of the same dispatch-table rows behind SVT_HIP_RTCD in a real encode: picture-analysis block means, the single-candidate
    SAD ladders, subtract, the distortion sums of mode decision, CDEF's rectangle copy / filter-block distortion / strength-pair step, the
    self-guided projection on materialised filters, the Wiener convolution, svt_aom_convolve8_*.  Bitstream and reconstruction must not change."""
    w, h, n, bd, preset, q = 176, 144, 2, 8, 6, 32
    clip = os.path.join(workdir, "qcif2.yuv")
    E.make_clip(clip, w, h, n, seed=9, bd=bd)
    names = ("svt_aom_subtract_block,svt_nxm_sad_kernel_sub_sampled,svt_ext_sad_calculation_8x8_16x16,svt_ext_sad_calculation_32x32_64x64,"
             "svt_copy_rect8_8bit_to_16bit,svt_compute_cdef_dist_8bit,svt_search_one_dual,svt_full_distortion_kernel32_bits,"
             "svt_full_distortion_kernel_cbf_zero32_bits,svt_spatial_full_distortion_kernel,svt_aom_sse,svt_aom_satd,svt_av1_block_error,"
             "svt_get_proj_subspace,svt_av1_lowbd_pixel_proj_error,svt_compute_mean_square_values_8x8,svt_compute_sub_mean_8x8,"
             "svt_aom_convolve8_horiz,svt_aom_convolve8_vert,svt_av1_wiener_convolve_add_src,"
             "svt_av1_jnt_convolve_2d,svt_av1_jnt_convolve_x,svt_av1_jnt_convolve_y,svt_av1_jnt_convolve_2d_copy,"
             "svt_av1_build_compound_diffwtd_mask,svt_av1_build_compound_diffwtd_mask_d16,svt_aom_lowbd_blend_a64_d16_mask")
    ref = E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif2.ref"))
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif2.rtcd"), env_extra={"SVT_HIP_RTCD": names}, timeout=1500)
    assert (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"])
    for nme in names.split(","):
        assert f"svt_hip_rtcd {nme} -> hip wrapper" in got["log"], nme
    _check_delegations(got, EXPECTED_DELEGATIONS["helpers"], "helpers")


@pytest.mark.gpu
def test_high_bit_depth_pointers_in_a_real_encode_on_gpu(workdir):
    """The 16-bit flavours and the picture-format conversions behind SVT_HIP_RTCD in a 10-bit encode."""
    w, h, n, bd, preset, q = 176, 144, 2, 10, 6, 32
    clip = os.path.join(workdir, "qcif10.yuv")
    E.make_clip(clip, w, h, n, seed=11, bd=bd)
    names = ("svt_compressed_packmsb,svt_c_pack,svt_convert_8bit_to_16bit,svt_convert_16bit_to_8bit,svt_pack2d_16_bit_src_mul4,svt_un_pack2d_16_bit_src_mul4,"
             "svt_un_pack8_bit_data,svt_unpack_avg,svt_aom_highbd_subtract_block,svt_full_distortion_kernel16_bits,svt_aom_highbd_8_mse16x16,"
             "svt_av1_highbd_wiener_convolve_add_src,svt_av1_highbd_pixel_proj_error,variance_highbd,sad_16b_kernel,svt_compute_cdef_dist_16bit,"
             "svt_residual_kernel16bit,svt_aom_highbd_quantize_b,svt_av1_highbd_quantize_fp,svt_get_proj_subspace,svt_aom_highbd_sse,"
             "svt_av1_highbd_jnt_convolve_2d,svt_av1_highbd_jnt_convolve_x,svt_av1_highbd_jnt_convolve_y,svt_av1_highbd_jnt_convolve_2d_copy,"
             "svt_av1_build_compound_diffwtd_mask_highbd,svt_av1_build_compound_diffwtd_mask_d16,svt_aom_highbd_blend_a64_d16_mask")
    ref = E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif10.ref"))
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif10.rtcd"), env_extra={"SVT_HIP_RTCD": names}, timeout=1500)
    assert (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"])
    for nme in names.split(","):
        assert f"svt_hip_rtcd {nme} -> hip wrapper" in got["log"], nme
    _check_delegations(got, EXPECTED_DELEGATIONS["hbd"], "hbd")


APP_SIMD, APP_HIP_SIMD = os.path.join(E.REFDIR, "SvtAv1EncApp_simd"), os.path.join(E.REFDIR, "SvtAv1EncApp_hip_simd")
need_simd = pytest.mark.skipif(not (os.path.exists(APP_SIMD) and os.path.exists(APP_HIP_SIMD)), reason="SIMD-build applications not built (make -f oracle/Makefile.enc simd)")


@need_simd
@pytest.mark.parametrize("case", ["cif_8bit_m6", "cif_10bit_m6", "cif_8bit_m4"])
def test_simd_build_of_the_reference_and_hooks_on_top_of_it(case, workdir):
    """The reference as its x86 build dispatches it (SSE2 .. AVX-512 intrinsics, the NASM entry points as C stand-ins of oracle/ref_asm_stubs.c) codes the same
    bitstream as its C build -- which pins the stand-ins --, and so does the hooked encoder built on top of the SIMD objects (CPU test double): the applications of
    the wall-clock comparison (tools/encoder_walltime.sh) are the same encoder."""
    w, h, n, bd, preset, q, must = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    simd = E.encode(APP_SIMD, clip, w, h, n, preset, q, bd, os.path.join(workdir, case + ".simd"))
    assert (simd["ivf"], simd["recon"]) == (ref["ivf"], ref["recon"])
    got = E.encode(APP_HIP_SIMD, clip, w, h, n, preset, q, bd, os.path.join(workdir, case + ".hipsimd"), env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all"})
    assert (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"])
    assert all(got["hooks"].get(hk, (0, 0))[0] > 0 and got["hooks"][hk][1] == 0 for hk in must), got["hooks"]


def _check_md_pre_720p_gop(workdir, env, tag):
    """1280 x 720, nine frames (a complete hierarchical mini-GOP), preset 6, on the SIMD build: the case in which a table hit that did not supply the candidate's warped-motion
    sample count (the reference's predictor computes it on the side, the fast cost of the later passes reads it) changed the bitstream -- found on the MI355X in round 5"""
    clip = os.path.join(workdir, "720p_gop9.src.yuv")
    if not os.path.exists(clip):
        E.make_clip(clip, 1280, 720, 9, seed=3, bd=8)
    if "720p_gop9" not in _ref_cache:
        _ref_cache["720p_gop9"] = E.encode(APP_SIMD, clip, 1280, 720, 9, 6, 36, 8, os.path.join(workdir, "720p_gop9.simd"))
    ref = _ref_cache["720p_gop9"]
    got = E.encode(APP_HIP_SIMD, clip, 1280, 720, 9, 6, 36, 8, os.path.join(workdir, "720p_gop9." + tag), env_extra=env)
    assert (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"]), got["log"][-1500:]
    st = _md_pre_line(got["log"])
    assert got["hooks"]["md_pre"] == (8, 0) and st["served"] * 2 > st["inter"], (got["hooks"], st)
    sp = _md_pre_subpel_line(got["log"])
    assert (st["served"] + sp["served"]) * 2 > st["inter"] + sp["probes"], f"fewer than half of mode decision's inter prediction + distortion leaf calls came from the picture's launches: {st} {sp}"
    return got


@need_simd
def test_md_pre_720p_mini_gop_on_cpu_test_double(workdir):
    _check_md_pre_720p_gop(workdir, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "md_pre"}, "mock_mdpre")


@need_simd
@pytest.mark.gpu
def test_md_pre_720p_mini_gop_on_gpu(workdir):
    got = _check_md_pre_720p_gop(workdir, {"SVT_HIP_HOOKS": "all,md_pre"}, "hip_mdpre")
    assert "svt_hip MOCK" not in got["log"]
    got = _check_md_pre_720p_gop(workdir, {"SVT_HIP_HOOKS": "md_pre", "SVT_HIP_MD_PRE_VERIFY": "1"}, "hip_mdpre_verify")
    assert re.search(r"verify_mismatches=0\b", got["log"])


@need_simd
@pytest.mark.gpu
def test_hooks_on_the_simd_build_on_gpu(workdir):
    case = "cif_8bit_m6"
    w, h, n, bd, preset, q, must = CASES[case]
    clip, ref = _reference(case, CASES[case], workdir)
    got = E.encode(APP_HIP_SIMD, clip, w, h, n, preset, q, bd, os.path.join(workdir, case + ".hipsimd_gpu"), env_extra={"SVT_HIP_HOOKS": "all"})
    assert "svt_hip MOCK" not in got["log"] and (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"])
    assert all(got["hooks"].get(hk, (0, 0))[0] > 0 and got["hooks"][hk][1] == 0 for hk in must), got["hooks"]


BLOCK_SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64), (128, 128),
               (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


@pytest.mark.gpu
def test_subpel_search_pointers_in_a_real_encode_on_gpu(workdir):
    """the sub-pel search trees (md_subpel_search -> svt_av1_find_best_sub_pixel_tree, Encoder/Codec/mcomp.c:350; the temporal filter's rounds) are
    host control flow of the reference; every kernel they evaluate a candidate with -- svt_aom_upsampled_pred, svt_aom_variance{W}x{H}, svt_aom_sad{W}x{H} and
    its x4d form (through mefn_ptr[], which is built from these pointers), the four *_sr convolves behind av1_inter_prediction -- runs on the device here,
    inside a real encode, one launch per call.  Bitstream and reconstruction must not change, and the wrappers must really have been called."""
    w, h, n, bd, preset, q = 176, 144, 3, 8, 4, 36
    clip = os.path.join(workdir, "qcif_sp.yuv")
    E.make_clip(clip, w, h, n, seed=31, bd=bd)
    fam = [f"svt_aom_variance{a}x{b}" for a, b in BLOCK_SIZES] + [f"svt_aom_sad{a}x{b}" for a, b in BLOCK_SIZES] + [f"svt_aom_sad{a}x{b}x4d" for a, b in BLOCK_SIZES]
    names = ",".join(["svt_aom_upsampled_pred", "svt_av1_convolve_2d_sr", "svt_av1_convolve_x_sr", "svt_av1_convolve_y_sr", "svt_av1_convolve_2d_copy_sr"] + fam)
    ref = E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif_sp.ref"))
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif_sp.rtcd"), env_extra={"SVT_HIP_RTCD": names}, timeout=2400)
    assert (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"])
    for nme in names.split(","):
        assert f"svt_hip_rtcd {nme} -> hip wrapper" in got["log"], nme
    calls = got["rtcd_calls"]
    print("sub-pel pointers, calls per wrapper:", {k: v for k, v in sorted(calls.items())})
    assert any("upsampled_pred" in k and v > 0 for k, v in calls.items()), calls
    assert any("var" in k and v > 0 for k, v in calls.items()) and any("conv" in k and v > 0 for k, v in calls.items()), calls
    _check_delegations(got, EXPECTED_DELEGATIONS["subpel"], "subpel")


@pytest.mark.gpu
def test_hooks_and_wrappers_together_10bit_on_gpu(workdir):
    """SVT_HIP_HOOKS and SVT_HIP_RTCD in one 10-bit encode: the picture-level hooks (one context behind the hooks' lock) and per-call wrappers that use
    the library-owned scratch at 16 bits (the Wiener statistics, the self-guided filter) run from different process threads at the same time -- the
    wrappers own a second context, so neither side can free or overwrite the other's scratch."""
    w, h, n, bd, preset, q = 176, 144, 3, 10, 6, 34
    clip = os.path.join(workdir, "qcif10b.yuv")
    E.make_clip(clip, w, h, n, seed=21, bd=bd)
    names = "svt_av1_compute_stats_highbd,svt_av1_selfguided_restoration,svt_apply_selfguided_restoration,svt_aom_highbd_quantize_b,svt_residual_kernel16bit"
    ref = E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif10b.ref"))
    hooks = "pa,tf,tf_me,tf_subpel,hme,me,cdef_finish,dlf,dlf_search,cdef_search,cdef_apply,sgr_search,rest_apply"   # the Wiener search stays the reference's loop: it calls svt_av1_compute_stats_highbd
    got = E.encode(E.APP_HIP, clip, w, h, n, preset, q, bd, os.path.join(workdir, "qcif10b.both"), env_extra={"SVT_HIP_HOOKS": hooks, "SVT_HIP_RTCD": names}, timeout=1500)
    assert (got["ivf"], got["recon"]) == (ref["ivf"], ref["recon"])
    assert all(v[1] == 0 for v in got["hooks"].values()) and sum(v[0] for v in got["hooks"].values()) > 10, got["hooks"]
    _check_delegations(got, EXPECTED_DELEGATIONS["both"], "both")
