"""CPU: the table of resident planes behind SVT_HIP_RESIDENT (integration/svt_hip_resident.c) on its own — announcements, uploads, hits, the reader / writer rule,
the budget's evictions, failure paths, eight threads — with the library's three calls stubbed on host memory (tests/resident_table_test.c), plain and under
AddressSanitizer + UndefinedBehaviorSanitizer and ThreadSanitizer.  The encoder-level behaviour is tests/test_encode_e2e.py::test_resident_*."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.parametrize("flags", [[], ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], ["-fsanitize=thread"]], ids=["plain", "asan_ubsan", "tsan"])
def test_resident_table(flags, tmp_path):
    exe = str(tmp_path / "resident_table_test")
    cmd = ["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-Werror", "-pthread", *flags, f"-I{ROOT}/include", f"-I{ROOT}/integration",
           os.path.join(ROOT, "tests", "resident_table_test.c"), os.path.join(ROOT, "integration", "svt_hip_resident.c"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), (r.stdout + r.stderr)[-3000:]
