"""CPU: the device expf of the temporal filter (svt-av1_amd/csrc/tfilter.hip, glibc's table + cubic in double) against the host libm,
exhaustively over [-7, -0] (tools/expf_pin.c), and the kernel's table against 2^(i/32) computed independently."""
import os
import re
import subprocess
import struct
from decimal import Decimal, getcontext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_expf_matches_libm_on_the_filter_domain(tmp_path):
    exe = str(tmp_path / "expf_pin")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(ROOT, "tools", "expf_pin.c"), "-lm", "-lpthread"])
    out = subprocess.check_output([exe], timeout=900).decode()
    m = re.search(r"n=(\d+) mismatches: plain=(\d+) fma_poly=(\d+) fma_k=(\d+) fma_both=(\d+)", out)
    assert m, out
    assert int(m.group(1)) == 0xC0E00001 - 0x80000000
    assert int(m.group(2)) == 0, out      # the evaluation order the kernel uses (no contraction)
    assert [int(m.group(i)) for i in (3, 4, 5)] == [0, 0, 0], out


def test_kernel_table_is_exp2_i_over_32():
    getcontext().prec = 80
    want = []
    for i in range(32):
        u = struct.unpack("<Q", struct.pack("<d", float(Decimal(2) ** (Decimal(i) / Decimal(32)))))[0]
        want.append((u - (i << 47)) & 0xFFFFFFFFFFFFFFFF)
    for path in ("svt-av1_amd/csrc/tfilter.hip", "tools/expf_pin.c"):
        txt = open(os.path.join(ROOT, path)).read()
        body = txt[txt.index("[32]"):]
        got = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", body)[:32]]
        assert got == want, path
