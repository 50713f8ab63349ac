"""Where does sgr_search8_kernel<uint8_t, 8, 1> spend its time?  Builds timing variants of the kernel from a PATCHED COPY of sgr.hip (the product file is not
touched) -- each variant drops one part of the per-set loop (results are then wrong: these are clocks, not checks) -- plus a main() that runs the luma plane of
a 3840x2160 picture 20 times under HIP events.

    python tools/ubench/sgr_filter_probe.py build     # here: hipcc, one executable per variant under tools/ubench/probe_bin/
    python tools/ubench/sgr_filter_probe.py run       # on the GPU box: runs them all, prints one line each

PROBE bits: 1 no difference-word stores, 2 no projection sums (DPP reductions + LDS atomics), 4 no per-set barrier, 8 no A'/B' build, 16 no filter."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
SRC = os.path.join(ROOT, "svt-av1_amd", "csrc", "sgr.hip")
BIN = os.path.join(HERE, "probe_bin")
VARIANTS = [0, 1, 3, 11, 19, 27, 32, 64, 64 + 27]

MAIN = r'''
#include <cstdio>
#include <vector>
int main() {
    const int pw = 3840, ph = 2160, EXT = 3, stride = pw + 2 * EXT + 58, unit = 256, ux = 15, uy = 8;
    std::vector<uint8_t> h((size_t)stride * (ph + 2 * EXT)), s((size_t)pw * ph);
    uint32_t x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint8_t)(100 + ((x >> 24) & 31)); }
    for (auto& v : s) { x = x * 1664525u + 1013904223u; v = (uint8_t)(100 + ((x >> 24) & 31)); }
    uint8_t *d_dgd, *d_src; unsigned long long *d_sums, *d_d2; uint32_t* d_pairs; int16_t* d_sd;
    const int dstride = (pw + 63) & ~63; const size_t dplane = (size_t)dstride * ph;
    hipMalloc(&d_dgd, h.size()); hipMalloc(&d_src, s.size()); hipMalloc(&d_sums, 8 * ux * uy * 16 * 5); hipMalloc(&d_d2, 8 * ux * uy);
    hipMalloc(&d_pairs, 16 * dplane * 4); hipMalloc(&d_sd, dplane * 2);
    hipMemcpy(d_dgd, h.data(), h.size(), hipMemcpyHostToDevice); hipMemcpy(d_src, s.data(), s.size(), hipMemcpyHostToDevice);
    hipMemset(d_sums, 0, 8 * ux * uy * 16 * 5); hipMemset(d_d2, 0, 8 * ux * uy);
    SgrSearchPic a = {};
    a.p[0] = sgr_search_plane_args(d_dgd + EXT * stride + EXT, stride, d_src, pw, pw, ph, unit, ux, uy, 0, 0xFFFFu, (int64_t*)d_sums, d_pairs, d_sd, dstride, dplane, (int64_t*)d_d2, nullptr, nullptr, 1024);
    for (int i = 1; i <= kSgrMaxPlanes; i++) a.first_tile[i] = a.p[0].n_tiles;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { hipLaunchKernelGGL((sgr_search8_kernel<uint8_t, 8, 1>), dim3(a.first_tile[kSgrMaxPlanes]), dim3(256), 0, 0, a); };
    for (int i = 0; i < 3; i++) go();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("PROBE=%2d  %8.1f us per luma launch (%s)\n", PROBE, ms * 1000.f / 20, hipGetErrorString(hipGetLastError()));
    return 0;
}
'''


def patched():
    t = open(SRC).read()

    def rep(old, new, count=1):
        nonlocal t
        assert t.count(old) >= count, old
        t = t.replace(old, new, count)

    # the patches go into sgr8_set_interior, the straight-line body every tile inside the picture runs
    rep("    if (STORE == 1) {\n#pragma unroll\n        for (int r = 0; r < 8; r++) SGR_ST(&pairs_px", "    if (STORE == 1 && !(PROBE & 1)) {\n#pragma unroll\n        for (int r = 0; r < 8; r++) SGR_ST(&pairs_px")
    rep("    int32_t h[5] = {0, 0, 0, 0, 0};   // H00, H01, H11, C0, C1\n",
        "    if (PROBE & 2) { int32_t sk = 0;\n#pragma unroll\n for (int r = 0; r < 8; r++) sk ^= (H0 ? D0[r] : 0) ^ (H1 ? D1[r] : 0);\n if (sk == 0x7ABCDEF1) part_ep[0] = 1; return; }\n"
        "    int32_t h[5] = {0, 0, 0, 0, 0};   // H00, H01, H11, C0, C1\n")
    rep("    __syncthreads();   // also orders this set's build after every lane's reads of the set before last (double buffer)\n    int32_t D0[8], D1[8];",
        "    if (!(PROBE & 4)) __syncthreads();\n    int32_t D0[8], D1[8];")
    rep("        if (live) {\n            const uint32_t z = (__umul24(P[k], is1 ? s1 : s0)", "        if (live && !(PROBE & 8)) {\n            const uint32_t z = (__umul24(P[k], is1 ? s1 : s0)")
    rep("    if (BD == 8) sgr8_filter(abw, i0, j, X, CX, H0, H1, D0, D1); else sgr10_filter(abw, i0, j, X, CX, H0, H1, D0, D1);\n    if (STORE == 1",
        "    if (!(PROBE & 16)) sgr8_filter(abw, i0, j, X, CX, H0, H1, D0, D1);\n    else {\n#pragma unroll\n for (int r = 0; r < 8; r++) { D0[r] = (int32_t)X[r] + (int32_t)s0; D1[r] = CX[r] - (int32_t)s1; } }\n    if (STORE == 1")
    rep("    int buf = 0;\n    for (int ep = 0; ep < 16; ep++) {\n        if (!((cmask >> ep) & 1)) continue;", "    int buf = 0;\n    if (PROBE & 32) cmask = 0;\n    for (int ep = 0; ep < 16; ep++) {\n        if (!((cmask >> ep) & 1)) continue;", 1)
    rep("    const bool interior = x0 + S_TW <= pw && y0 >= 0 && y0 + S_TH <= ph;", "    const bool interior = x0 + S_TW <= pw && y0 >= 0 && y0 + S_TH <= ph && !(PROBE & 64);")
    # the general path (edge tiles; every tile with bit 64)
    rep("            if (STORE == 1 && colvalid) {   // (flt0 - u) | (flt1 - u) << 16", "            if (STORE == 1 && colvalid && !(PROBE & 1)) {   // (flt0 - u) | (flt1 - u) << 16")
    rep("            int32_t h00 = 0, h01 = 0, h11 = 0, c0 = 0, c1 = 0;\n    #pragma unroll",
        "            if (PROBE & 2) { int32_t sk = 0;\n#pragma unroll\n for (int r = 0; r < 8; r++) sk ^= (has0 ? D0[r] : 0) ^ (has1 ? D1[r] : 0);\n if (sk == 0x7ABCDEF1) atomicAdd(&acc[ep][0], 1ull); }\n"
        "            int32_t h00 = 0, h01 = 0, h11 = 0, c0 = 0, c1 = 0;\n            if (!(PROBE & 2)) {\n    #pragma unroll")
    rep("                if (has0 && has1) atomicAdd(&acc[ep][1], (unsigned long long)(long long)h01);\n            }\n",
        "                if (has0 && has1) atomicAdd(&acc[ep][1], (unsigned long long)(long long)h01);\n            }\n            }\n")
    rep("        sgr8_build(abw, xt, P, M, has0, has1, s0, s1, tid);", "        if (!(PROBE & 8)) sgr8_build(abw, xt, P, M, has0, has1, s0, s1, tid);")
    rep("            sgr8_filter(abw, i0, j, X, CX, has0, has1, D0, D1);",
        "            if (!(PROBE & 16)) sgr8_filter(abw, i0, j, X, CX, has0, has1, D0, D1);\n            else {\n#pragma unroll\n for (int r = 0; r < 8; r++) { D0[r] = (int32_t)X[r] + ep; D1[r] = CX[r] - ep; } }")
    return t + MAIN


def build():
    os.makedirs(BIN, exist_ok=True)
    src = os.path.join(BIN, "sgr_probe.hip")
    open(src, "w").write(patched())
    procs = []
    for v in VARIANTS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-DPROBE={v}", "-I", os.path.join(ROOT, "svt-av1_amd", "csrc"),
               "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(BIN, f"probe_{v}")]
        procs.append((v, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) % 4 == 0:
            for _, p in procs[-4:]: p.wait()
    for v, p in procs:
        out = p.communicate()[0]
        print(v, "ok" if p.returncode == 0 else out[-2000:])


def run():
    for v in VARIANTS:
        r = subprocess.run([os.path.join(BIN, f"probe_{v}")], capture_output=True, text=True, timeout=120)
        print(r.stdout.strip() or r.stderr[-500:], flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
