"""GPU: several contexts in one process, each driven by its own host thread ("host worker i owns GPU i").  Every entry point
makes its context's device current (hipSetDevice is per thread), contexts share nothing, so concurrent calls must give the serial results.
The gpurun box exposes one GPU: the contexts all name device 0 (on an 8-GPU node the same code runs with device i)."""
import threading

import numpy as np
import pytest

import me_common as mc

pytestmark = pytest.mark.gpu


def test_two_contexts_two_threads(pkg, orc):
    w, h = 256, 192
    cases = []
    for seed in (31, 32):
        cur, refp = mc.synth.make_luma_pair(w, h, seed=seed)
        cur_p, ref_p = mc.synth.pad_plane(cur), mc.synth.pad_plane(refp)
        sbs = mc.windows(orc, w, h, 72, 40)
        exp = mc.oracle_frame(orc, cur_p, ref_p, cur_p.shape[1], mc.synth.PAD, sbs, 0)
        cases.append((cur_p, ref_p, sbs, exp))
    ctxs = [pkg.Context(0), pkg.Context(0)]
    errors = []

    def worker(i):
        try:
            cur_p, ref_p, sbs, exp = cases[i]
            for _ in range(20):
                got = mc.hip_frame(ctxs[i], cur_p, ref_p, cur_p.shape[1], mc.synth.PAD, sbs, 0)
                if not (np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])):
                    errors.append(f"context {i}: mismatch")
                    return
        except Exception as e:   # noqa: BLE001
            errors.append(f"context {i}: {e!r}")

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for c in ctxs:
        c.close()
    assert not errors, errors
