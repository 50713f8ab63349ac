// valu_peak.hip — chip-wide issue rate of vector instructions by WALL CLOCK (HIP events), in T lane-operations per second: the integer-VALU roofline of the search kernels.
// (tools/valu_rate.cpp of round 2 divided s_memtime deltas by instruction counts; s_memtime does not tick once per shader cycle on this part, which made every rate look
// 1.6x better than it is.)  512 workgroups of 1024 threads (two rounds of one workgroup per compute unit... the runtime packs two per CU: 8 waves per SIMD), each wave runs
// ITER x 16 independent instances of one instruction.  Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_peak.hip -o tools/ubench/probe_bin/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
#define KERNEL(NAME, BODY)                                                                                         \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long* out, int seed) {                         \
        int r[16];                                                                                                 \
        for (int i = 0; i < 16; i++) r[i] = seed + i * 7 + threadIdx.x;                                            \
        int a = seed * 3 + threadIdx.x, b = seed + 11; long long q[8];                                             \
        for (int i = 0; i < 8; i++) q[i] = seed + i + threadIdx.x;                                                 \
        long long a64 = a; (void)a64; (void)q;                                                                     \
        for (int it = 0; it < ITER; it++) { BODY }                                                                 \
        int s = 0;                                                                                                 \
        for (int i = 0; i < 16; i++) s += r[i];                                                                    \
        for (int i = 0; i < 8; i++) s += (int)q[i];                                                                \
        if (s == 0x12345678) out[1] = s;                                                                           \
    }
#define X16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define X8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define OP3(OP) asm volatile(OP " %0, %1, %2, %0" : "+v"(r[0]) : "v"(a), "v"(b));
#define D3(OP, i) asm volatile(OP " %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define D3S(OP, i) asm volatile(OP " %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "s"(seed));
#define D2(OP, i) asm volatile(OP " %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define DQ(OP, i) asm volatile(OP " %0, %1, %2, %0" : "+v"(q[i]) : "v"(a64), "v"(b));
#define M_mad(i) D3("v_mad_u32_u24", i)
#define M_add(i) D2("v_add_u32", i)
#define M_fma(i) D3("v_fma_f32", i)
#define M_dot2(i) D3("v_dot2_i32_i16", i)
#define M_sad16(i) D3("v_sad_u16", i)
#define M_sad16s(i) D3S("v_sad_u16", i)
#define M_sad8(i) D3("v_sad_u8", i)
#define M_perm(i) D3("v_perm_b32", i)
#define M_pkadd(i) D2("v_pk_add_u16", i)
#define M_andor(i) D3("v_and_or_b32", i)
#define M_qsad(i) DQ("v_qsad_pk_u16_u8", i)
KERNEL(mad_u32_u24, X16(M_mad))
KERNEL(add_u32, X16(M_add))
KERNEL(fma_f32, X16(M_fma))
KERNEL(dot2_i32_i16, X16(M_dot2))
KERNEL(sad_u16, X16(M_sad16))
KERNEL(sad_u16_sgpr, X16(M_sad16s))
KERNEL(sad_u8, X16(M_sad8))
KERNEL(perm_b32, X16(M_perm))
KERNEL(pk_add_u16, X16(M_pkadd))
KERNEL(and_or_b32, X16(M_andor))
KERNEL(qsad_pk_u16_u8, X8(M_qsad) X8(M_qsad))
template <typename K> void run(const char* name, K kern, unsigned long long* d, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<wgs, 1024>>>(d, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<wgs, 1024>>>(d, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)wgs * 1024.0 * ITER * 16.0;
    std::printf("%-16s %4d workgroups: %7.3f ms  %6.2f T lane-op/s  (%5.1f lanes per clock and compute unit at 2.4 GHz)\n", name, wgs, ms, lane_ops / (ms * 1e-3) / 1e12,
                lane_ops / (ms * 1e-3) / 256.0 / 2.4e9);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
#define R(N) run(#N, k_##N, d, 256); run(#N, k_##N, d, 512);
    R(add_u32) R(fma_f32) R(mad_u32_u24) R(and_or_b32) R(perm_b32) R(pk_add_u16) R(dot2_i32_i16) R(sad_u8) R(sad_u16) R(sad_u16_sgpr) R(qsad_pk_u16_u8)
    return 0;
}
