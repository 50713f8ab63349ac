"""CPU: the reference-side glue in integration/  is real C against the reference's own headers: syntax-check it
with the include paths oracle/Makefile.ref uses.  Skipped where /root/reference does not exist (GPU box)."""
import glob
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Source", "Lib")), reason="needs the reference headers")
def test_integration_sources_compile_against_reference_headers():
    L = os.path.join(REF, "Source", "Lib")
    gen = os.path.join(ROOT, "oracle", "_ref", "gen")
    if not os.path.exists(os.path.join(gen, "EbVersion.h")):   # the 5-line version header the reference's own build generates
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-s", os.path.join(gen, "EbVersion.h")])
    incs = [gen, os.path.join(REF, "Source", "API"), os.path.join(L, "Common", "Codec"), os.path.join(L, "Common", "C_DEFAULT"),
            os.path.join(L, "Encoder", "Codec"), os.path.join(L, "Encoder", "C_DEFAULT"), os.path.join(L, "Encoder", "Globals"),
            os.path.join(ROOT, "include"), os.path.join(ROOT, "integration")]
    srcs = sorted(glob.glob(os.path.join(ROOT, "integration", "*.c")))
    assert srcs
    for s in srcs:
        cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-D_GNU_SOURCE"] + [f"-I{i}" for i in incs] + [s]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
