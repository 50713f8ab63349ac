#!/bin/bash
# CPU only (needs /root/reference): the hooked reference encoder (integration/*.c + the patched reference files) and the CPU test double of the library, built
# with AddressSanitizer or ThreadSanitizer, run through end-to-end encodes with every hook on (incl. the opt-in ones).  The unpatched reference objects stay
# uninstrumented (they come from Makefile.ref's object directory).  Reports are written to $OUT/log_*; a run is clean when no report names integration/, svt_hip or
# the mock.      usage: tools/sanitize_e2e.sh address|thread|undefined [out_dir]
set -eu
SAN=${1:-address}; OUT=${2:-/tmp/sanitize_$SAN}
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="-fsanitize=$SAN -g -fno-omit-frame-pointer"; [ "$SAN" = address ] && FLAGS="$FLAGS -fsanitize-recover=address"; [ "$SAN" = undefined ] && FLAGS="$FLAGS -fno-sanitize=alignment"
mkdir -p "$OUT/mock" "$OUT/keep/mock"
# Everything the sanitizer build shares with the ordinary one is brought up to date FIRST, with the ordinary flags: the reference library and its shims
# (oracle/_ref/libsvtav1_ref.so, objects under /tmp/svtav1_ref_obj) are prerequisites of the application, and a stale one would otherwise be rebuilt WITH the
# sanitizer and left behind for every other test (round 6: the reference library came back linked against libtsan).  The library is kept / restored as well.
make -C "$R/oracle" -f Makefile.ref -j16 -s
make -C "$R/oracle" -f Makefile.enc -j16 -s
cp "$R/oracle/_ref/SvtAv1EncApp_hip" "$OUT/keep/"; cp "$R/oracle/_ref/mock/libsvtav1_hip.so" "$OUT/keep/mock/"; cp "$R/oracle/_ref/libsvtav1_ref.so" "$OUT/keep/"
restore() { cp "$OUT/keep/SvtAv1EncApp_hip" "$R/oracle/_ref/"; cp "$OUT/keep/mock/libsvtav1_hip.so" "$R/oracle/_ref/mock/"; cp "$OUT/keep/libsvtav1_ref.so" "$R/oracle/_ref/"; }
trap restore EXIT
rm -f "$R/oracle/_ref/SvtAv1EncApp_hip" "$R/oracle/_ref/mock/libsvtav1_hip.so"; rm -rf "$OUT/obj"
make -C "$R" -f oracle/Makefile.enc -j16 -s EOBJDIR="$OUT/obj" CC="gcc $FLAGS" CXX="g++ $FLAGS" "$R/oracle/_ref/SvtAv1EncApp_hip" "$R/oracle/_ref/mock/libsvtav1_hip.so"
mv "$R/oracle/_ref/SvtAv1EncApp_hip" "$OUT/SvtAv1EncApp_hip_$SAN"; mv "$R/oracle/_ref/mock/libsvtav1_hip.so" "$OUT/mock/"
restore; trap - EXIT
SAN=$SAN OUT=$OUT REPO=$R python3 - <<'PY'
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(os.environ.get("_", ""))))
sys.path.insert(0, os.path.join(os.environ.get("REPO", os.getcwd()), "tests"))
import e2e_common as E
san, out = os.environ["SAN"], os.environ["OUT"]
app, mock, wd = f"{out}/SvtAv1EncApp_hip_{san}", f"{out}/mock", f"{out}/work"
os.makedirs(wd, exist_ok=True)
opt = "all,md_tx,encdec_tx,md_subpel,encdec_sb,md_pre"
cases = {"cif_8bit_m6": (352, 288, 6, 8, 6, 35, opt, []), "cif_10bit_m6": (352, 288, 4, 10, 6, 30, opt, []), "cif_8bit_m4": (352, 288, 4, 8, 4, 45, "all", []),
         "328x200_8bit_m6": (328, 200, 4, 8, 6, 33, "all", []), "cif_8bit_m2": (352, 288, 3, 8, 2, 40, "all", []), "unit_wiener": (352, 288, 4, 8, 6, 35, E.ALL_PER_UNIT, []),
         "854x480_m4": (854, 480, 3, 8, 4, 40, "all", []), "640x360_10bit_m5": (640, 360, 4, 10, 5, 32, "all", []), "cif_10bit_m8": (352, 288, 5, 10, 8, 36, "all", []),
         "qcif_m0": (176, 144, 3, 8, 0, 40, "all", []), "tiles_2x2": (352, 288, 6, 8, 6, 38, "all", ["-tile-columns", "1", "-tile-rows", "1"]),
         "screen_content": (352, 288, 6, 8, 6, 38, "all", ["-scm", "1"]), "altref_7_frames": (352, 288, 6, 8, 6, 38, "all", ["-altref-nframes", "7", "-altref-strength", "6"]),
         "film_grain": (352, 288, 6, 8, 6, 38, "all", ["-film-grain", "8"]), "low_delay_p": (352, 288, 6, 8, 6, 38, "all", ["-pred-struct", "0"]),
         "cif_gop17_md_pre": (352, 288, 17, 8, 6, 36, "all,md_pre", [])}   # two mini-GOPs: bi-directional ME candidates, the compound pair table of md_pre
if san in ("thread", "undefined"):
    cases = {k: cases[k] for k in ("cif_8bit_m6", "cif_10bit_m6", "tiles_2x2", "cif_gop17_md_pre")}
bad = 0
for name, (w, h, n, bd, preset, q, hooks, extra) in cases.items():
    clip = os.path.join(wd, name + ".yuv"); E.make_clip(clip, w, h, n, seed=7 + w, bd=bd)
    ref = E.encode(E.APP_REF, clip, w, h, n, preset, q, bd, os.path.join(wd, name + ".ref"), extra_args=extra)
    env = {"LD_LIBRARY_PATH": mock, "SVT_HIP_HOOKS": hooks, {"address": "ASAN_OPTIONS", "thread": "TSAN_OPTIONS", "undefined": "UBSAN_OPTIONS"}[san]: f"print_stacktrace=1:detect_leaks=0:halt_on_error=0:report_signal_unsafe=0:exitcode=0:log_path={out}/log_{name}"}
    got = E.encode(app, clip, w, h, n, preset, q, bd, os.path.join(wd, name + "." + san), env_extra=env, extra_args=extra, timeout=2400)
    same = got["ivf"] == ref["ivf"] and got["recon"] == ref["recon"]
    logs = [f for f in os.listdir(out) if f.startswith("log_" + name + ".")]
    def ours_in(txt):   # UBSan names the offending source line itself; for the other two any frame of ours in a report counts
        if san == "undefined":
            return sum(1 for l in txt.splitlines() if "runtime error:" in l and any(k in l.split(":")[0] for k in ("/integration/", "/oracle/", "/svt-av1_amd/")))
        return int(any(k in txt for k in ("integration/", "svt_hip", "hip_mock", "_oracle.c")))
    ours = sum(ours_in(open(os.path.join(out, f), errors="replace").read()) for f in logs)
    total = sum(sum(open(os.path.join(out, f), errors="replace").read().count(k) for k in ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "runtime error:")) for f in logs)
    bad += (not same) + ours
    print(f"{name}: {'identical' if same else 'DIFFERENT'}; {san} reports {total}, naming the hooks / the library / the test double: {ours}", flush=True)
sys.exit(1 if bad else 0)
PY
