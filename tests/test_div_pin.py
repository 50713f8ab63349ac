"""CPU: the three-operation quotient of the temporal filter's weight (svt-av1_amd/csrc/tfilter.hip quot3: q = x r, e = fma(-y, q, x), q' = fma(e, r, q)) against the
IEEE division the reference performs — every uint32 dividend for the window's sample counts, constructed hard cases for 6 and for arbitrary launch constants
(tools/div_pin.c; the CPU's fused multiply-add is the device's arithmetic)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quot3_is_the_ieee_quotient(tmp_path):
    exe = str(tmp_path / "div_pin")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(ROOT, "tools", "div_pin.c"), "-lm", "-lpthread", "-lquadmath"])
    out = subprocess.check_output([exe, "1"], timeout=1800).decode()
    m = re.search(r"every uint32 dividend: (\d+) quotients, (\d+) differ", out)
    assert m and int(m.group(1)) == 4 << 32 and int(m.group(2)) == 0, out
    for label in ("divisor 6, hard cases", "arbitrary divisors"):
        m = re.search(re.escape(label) + r"[^:]*: (\d+) quotients, (\d+) differ", out)
        assert m and int(m.group(1)) > 10 ** 8 and int(m.group(2)) == 0, out


def test_kernel_uses_the_pinned_sequence_and_sample_counts():
    txt = open(os.path.join(ROOT, "svt-av1_amd", "csrc", "tfilter.hip")).read()
    assert "const double q = x * r;" in txt and "__builtin_fma(-y, q, x)" in txt and "__builtin_fma(e, r, q)" in txt
    assert "NUM == 25 || NUM == 26 || NUM == 27 || NUM == 29" in txt     # the divisors tools/div_pin.c covers exhaustively
    pin = open(os.path.join(ROOT, "tools", "div_pin.c")).read()
    assert "{25.0, 26.0, 27.0, 29.0}" in pin
