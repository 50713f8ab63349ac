"""Run ON THE GPU BOX: host <-> device copy rates of a 3840 x 2160 plane through the library ABI -- 2-D (rect) against 1-D copies, pageable against page-locked host memory."""
import sys, time, ctypes as C, numpy as np, importlib.util, os
ROOT="/root/repo" if os.path.isdir("/root/repo/svt-av1_amd") else os.getcwd()
spec=importlib.util.spec_from_file_location("pkg", os.path.join(ROOT,"svt-av1_amd","__init__.py")); pkg=importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
ctx=pkg.Context(0); L=ctx.L; h=ctx.h
W,H,ST=3840,2160,3840+2*160+64
L.svt_hip_malloc.argtypes=[C.c_void_p,C.POINTER(C.c_void_p),C.c_size_t]
d=C.c_void_p(); assert L.svt_hip_malloc(h,C.byref(d),ST*H)==0
def t(fn,n=10):
    fn(); L.svt_hip_sync(h)
    t0=time.perf_counter()
    for _ in range(n): fn(); L.svt_hip_sync(h)
    return (time.perf_counter()-t0)/n*1e3
pageable=np.random.randint(0,255,(H+320,ST),dtype=np.uint8)
reg=np.random.randint(0,255,(H+320,ST),dtype=np.uint8)
print("register rc", L.svt_hip_host_register(h, C.c_void_p(reg.ctypes.data), C.c_size_t(reg.nbytes)))
for name,buf in (("pageable",pageable),("registered",reg)):
    src=C.c_void_p(buf.ctypes.data+160*ST+160)
    ms2=t(lambda: L.svt_hip_memcpy2d_h2d_async(h, d, C.c_size_t(ST), src, C.c_size_t(ST), C.c_size_t(W), C.c_size_t(H)))
    ms1=t(lambda: L.svt_hip_memcpy_h2d_async(h, d, C.c_void_p(buf.ctypes.data+160*ST), C.c_size_t(ST*H)))
    print(f"{name:10s}  2-D {W}x{H} of stride {ST}: {ms2:.2f} ms ({W*H/ms2/1e6:.1f} GB/s)   1-D {ST*H/1e6:.1f} MB: {ms1:.2f} ms ({ST*H/ms1/1e6:.1f} GB/s)")
    ms2d=t(lambda: L.svt_hip_memcpy2d_d2h_async(h, src, C.c_size_t(ST), d, C.c_size_t(ST), C.c_size_t(W), C.c_size_t(H)))
    ms1d=t(lambda: L.svt_hip_memcpy_d2h_async(h, C.c_void_p(buf.ctypes.data+160*ST), d, C.c_size_t(ST*H)))
    print(f"{name:10s}  d2h 2-D {ms2d:.2f} ms ({W*H/ms2d/1e6:.1f} GB/s)   1-D {ms1d:.2f} ms ({ST*H/ms1d/1e6:.1f} GB/s)")
