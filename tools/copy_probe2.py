"""Run ON THE GPU BOX: why a reconstruction plane goes up at 5 GB/s inside the hooked encoder when tools/copy_probe.py moves the same plane at 52 GB/s.
One 3840 x 2160 plane of stride 4224 through svt_hip_memcpy2d_h2d_async (the hooks' entry point), host side page-locked in place (svt_hip_host_register), timed with
HIP events: the first copy after registering, a repeat, a copy right after the CPU rewrote the plane (lines dirty in the caches), after another thread on another core
rewrote it, from posix_memalign memory instead of numpy's (huge pages or not), and d2h the same way."""
import ctypes as C, importlib.util, os, threading, time
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/svt-av1_amd") else os.getcwd()
spec = importlib.util.spec_from_file_location("pkg", os.path.join(ROOT, "svt-av1_amd", "__init__.py")); pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
ctx = pkg.Context(0); L = ctx.L; h = ctx.h
W, H, ST = 3840, 2160, 3840 + 2 * 160 + 64
ROWS = H + 320
L.svt_hip_malloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]
d = C.c_void_p(); assert L.svt_hip_malloc(h, C.byref(d), ST * H) == 0
libc = C.CDLL("libc.so.6"); libc.posix_memalign.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t]; libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]


def up(addr):
    t0 = time.perf_counter()
    L.svt_hip_memcpy2d_h2d_async(h, d, C.c_size_t(ST), C.c_void_p(addr + 160 * ST + 160), C.c_size_t(ST), C.c_size_t(W), C.c_size_t(H)); L.svt_hip_sync(h)
    return (time.perf_counter() - t0) * 1e3


def down(addr):
    t0 = time.perf_counter()
    L.svt_hip_memcpy2d_d2h_async(h, C.c_void_p(addr + 160 * ST + 160), C.c_size_t(ST), d, C.c_size_t(ST), C.c_size_t(W), C.c_size_t(H)); L.svt_hip_sync(h)
    return (time.perf_counter() - t0) * 1e3


def gbs(ms): return W * H / ms / 1e6


def case(name, arr_addr, view, madv=None):
    if madv is not None:
        print(f"{name}: madvise({madv}) rc", libc.madvise(C.c_void_p(arr_addr & ~0x1fffff), C.c_size_t(ROWS * ST), madv))
    view[:] = 7   # first touch by the CPU
    t0 = time.perf_counter(); rc = L.svt_hip_host_register(h, C.c_void_p(arr_addr), C.c_size_t(ROWS * ST)); treg = (time.perf_counter() - t0) * 1e3
    first = up(arr_addr); again = min(up(arr_addr) for _ in range(5))
    view[:] = 9; dirty = up(arr_addr)
    def other(): view[:] = 11
    th = threading.Thread(target=other); th.start(); th.join(); dirty2 = up(arr_addr)
    dfirst = down(arr_addr); dagain = min(down(arr_addr) for _ in range(5))
    view[:] = 3; ddirty = down(arr_addr)
    print(f"{name:28s} register rc {rc} {treg:6.2f} ms | h2d first {first:5.2f} ms ({gbs(first):5.1f} GB/s) repeat {again:5.2f} ({gbs(again):5.1f}) after CPU rewrite {dirty:5.2f} ({gbs(dirty):5.1f}) after another thread's rewrite {dirty2:5.2f} ({gbs(dirty2):5.1f})"
          f" | d2h first {dfirst:5.2f} ({gbs(dfirst):5.1f}) repeat {dagain:5.2f} ({gbs(dagain):5.1f}) after CPU rewrite {ddirty:5.2f} ({gbs(ddirty):5.1f})")
    L.svt_hip_host_unregister(h, C.c_void_p(arr_addr))


a = np.empty((ROWS, ST), np.uint8); case("numpy", a.ctypes.data, a)
for name, madv in (("posix_memalign 64", None), ("posix_memalign 64 + MADV_HUGEPAGE", 14), ("posix_memalign 64 + MADV_NOHUGEPAGE", 15)):
    p = C.c_void_p(); assert libc.posix_memalign(C.byref(p), 64 if madv is None else 1 << 21, ROWS * ST + (1 << 21)) == 0
    v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (ROWS, ST))
    case(name, p.value, v, madv)
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| nodes:", os.listdir("/sys/devices/system/node") if os.path.isdir("/sys/devices/system/node") else "?")
