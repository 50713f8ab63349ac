"""Per-phase wall-clock trace of the one-launch CDEF strength selection (joint_resident_kernel): needs percall2.hip built with -DSVT_RES_TRACE, which makes
workgroups 0 and 133 write wall_clock64() at eight points of every step into the state's unused partial[3] rows.  argv[1] = byte offset of those rows in the state
(offsetof(JointState, partial) + 3 * 64 * 4096 * 8 = 6327624 for the current layout).  Development tool; not part of bench.py or the tests."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package(); hip = pkg.Context(0); L = hip.L
rng = np.random.default_rng(3); n = 2040
m0 = rng.integers(1000, 1 << 22, (n, 64)).astype(np.uint64); m1 = rng.integers(1000, 1 << 21, (n, 64)).astype(np.uint64)
d_m0, d_m1 = hip.to_device(m0), hip.to_device(m1)
SB = 304 + 8192 + 4 * 128 * 4096 * 8
d_state = hip.empty(SB)
for _ in range(3): hip.check(L.svt_hip_set_cdef_select_form(hip.h, 1), "form"); hip.check(L.svt_hip_cdef_strength_select_dev(hip.h, d_m0, d_m1, n, 0, 64, d_state, SB), "select")
torch.cuda.synchronize()
off = int(sys.argv[1])
raw = hip.to_host(C.c_void_p(d_state.value + off), (2, 4096), np.uint64)
for wg in range(2):
    t = raw[wg, :320].reshape(40, 8).astype(np.int64)
    d = np.diff(t, axis=1) * 10   # ns at 100 MHz
    print("wg", wg, "per-step phases ns [compute, sync, slot, gather, sync, phaseC, sync] ; step total")
    for s in (0, 1, 4, 5, 10, 19, 20, 30, 39):
        nxt = (t[s + 1, 0] - t[s, 0]) * 10 if s < 39 else -1
        print(s, d[s].tolist(), nxt)
    print("total us", (t[39, 7] - t[0, 0]) / 100.0)
