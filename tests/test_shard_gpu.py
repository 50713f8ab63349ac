"""GPU: the chained mini-step of tests/shard_common.py on the product library, two contexts (two "ranks" of one process on the one visible GPU, each
on its own thread and stream) — every stream's outputs must equal what the CPU test double (the oracle kernels behind the same ABI) produces."""
import os
import threading

import pytest

from conftest import load_package
import shard_common as sc

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(sc.MOCK_LIB), reason="oracle/_ref/mock/libsvtav1_hip.so not built")
def test_streams_on_two_contexts_match_the_test_double():
    pkg = load_package()
    L = sc.load(pkg.LIB_PATH)
    got = {}

    def rank(r):
        dev = sc.Dev(L, 0)
        for s in range(r, 4, 2):
            got[s] = sc.stream_step(dev, s)
        dev.close()
    ts = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    ref = sc.Dev(sc.load(sc.MOCK_LIB))
    for s in range(4):
        assert got[s] == sc.stream_step(ref, s), s
    ref.close()
