import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import the `svt-av1_amd/` directory as module `svt_av1_amd`."""
    if "svt_av1_amd" in sys.modules:
        return sys.modules["svt_av1_amd"]
    spec = importlib.util.spec_from_file_location(
        "svt_av1_amd", os.path.join(ROOT, "svt-av1_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(ROOT, "svt-av1_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["svt_av1_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    mod = load_package()
    if not os.path.exists(mod.LIB_PATH):   # fresh checkout: same recipe as __graft_entry__.build() (hipcc cross-compiles without a GPU)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "svt-av1_amd", "csrc"), "-j", str(min(16, os.cpu_count() or 4)), "-s"])
    return mod


@pytest.fixture(scope="session")
def orc():
    """The CPU restatement (oracle/liboracle.so) — the checker, never the product."""
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return C.CDLL(path)


@pytest.fixture(scope="session")
def ref():
    """The real reference C path (oracle/_ref/libsvtav1_ref.so) when it has been built."""
    path = os.path.join(ROOT, "oracle", "_ref", "libsvtav1_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libsvtav1_ref.so not built (needs /root/reference; make -f oracle/Makefile.ref)")
    L = C.CDLL(path)
    L.setup_common_rtcd_internal(0)
    L.setup_rtcd_internal(0)
    return L


@pytest.fixture(scope="session")
def hip(pkg):
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()


def ptr(a, t=C.c_void_p):
    return a.ctypes.data_as(t)
