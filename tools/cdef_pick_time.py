"""HIP-event time of the CDEF strength-pair selection (svt_hip_cdef_joint_strength_search_dev, nb = 1, 2, 4, 8) for one 4K frame (2040 filter blocks)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
pkg = load_package(); hip = pkg.Context(0); L = hip.L
rng = np.random.default_rng(3)
n = 2040
MAG = int(os.environ.get("PICK_MAG", 22))
m0 = rng.integers(1000, 1 << MAG, (n, 64)).astype(np.uint64); m1 = rng.integers(1000, 1 << (MAG - 1), (n, 64)).astype(np.uint64)
d_m0, d_m1 = hip.to_device(m0), hip.to_device(m1)
d_lev = hip.to_device(np.zeros(16, np.int32)); d_work = hip.empty(8 * (4097 + n))
st = torch.cuda.Stream(); L.svt_hip_set_stream(hip.h, C.c_void_p(st.cuda_stream))
def run():
    for nb in (1, 2, 4, 8):
        hip.check(L.svt_hip_cdef_joint_strength_search_dev(hip.h, d_m0, d_m1, n, d_lev, C.c_void_p(d_lev.value + 32), nb, 0, 64, d_work), "joint")
with torch.cuda.stream(st):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5): run()
    e1.record(st); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream(); cap.wait_stream(st)
L.svt_hip_set_stream(hip.h, C.c_void_p(cap.cuda_stream))
with torch.cuda.graph(g, stream=cap):
    run()
L.svt_hip_set_stream(hip.h, C.c_void_p(st.cuda_stream))
with torch.cuda.stream(st):
    g.replay(); torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record(st)
    for _ in range(5): g.replay()
    g1.record(st); torch.cuda.synchronize()
print(f"the same as one captured HIP graph: {g0.elapsed_time(g1) / 5:.3f} ms per frame")
d_state = hip.empty(304 + 8192 + 4 * 128 * 4096 * 8)
def run2(): hip.check(L.svt_hip_cdef_strength_select_dev(hip.h, d_m0, d_m1, n, 0, 64, d_state, 304 + 8192 + 4 * 128 * 4096 * 8), "select")
with torch.cuda.stream(st):
    run2(); torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(st)
    for _ in range(5): run2()
    s1.record(st); torch.cuda.synchronize()
FORM = "one resident launch" if os.environ.get("SVT_HIP_CDEF_SELECT") == "resident" else "two launches per step index, 80 launches"
print(f"svt_hip_cdef_strength_select_dev (the four chains side by side, {FORM}; tables below 2^{MAG}): {s0.elapsed_time(s1) / 5:.3f} ms per frame")
print(f"cdef strength-pair selection, 2040 filter blocks, nb = 1 + 2 + 4 + 8 (75 steps): {e0.elapsed_time(e1) / 5:.3f} ms per frame (eager launches)")
g2 = torch.cuda.CUDAGraph()
cap2 = torch.cuda.Stream(); cap2.wait_stream(st)
L.svt_hip_set_stream(hip.h, C.c_void_p(cap2.cuda_stream))
with torch.cuda.graph(g2, stream=cap2):
    run2()
L.svt_hip_set_stream(hip.h, C.c_void_p(st.cuda_stream))
with torch.cuda.stream(st):
    g2.replay(); torch.cuda.synchronize()
    h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0.record(st)
    for _ in range(10): g2.replay()
    h1.record(st); torch.cuda.synchronize()
print(f"svt_hip_cdef_strength_select_dev ({FORM}; tables below 2^{MAG}) as one captured HIP graph: {h0.elapsed_time(h1) / 10:.3f} ms per frame")
