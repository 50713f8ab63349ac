// valu_rate.cpp — issue rate of the integer / packed VALU instructions the search kernels are built from, on gfx950.
// SUPERSEDED (round 6) by tools/ubench/valu_peak.hip: this tool divides s_memtime deltas by instruction counts, and s_memtime does not advance once per shader cycle on this
// part -- its "2.78 cycles" for v_mad_u32_u24 are 4.4 by the wall clock (profiles/r06/valu_peak.txt).  The RATIOS between instructions it reports hold.
// Every workgroup is 1024 threads (4 waves per SIMD); each wave runs ITER iterations of 16 independent instances of one instruction.
// Reports shader cycles per wave-instruction per SIMD (s_memtime around the loop, averaged over waves) and the implied lane rate.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.cpp -o gpurun_out/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 2048
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, ASM3)                                                                                         \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long* out, int seed) {                         \
        int r[16];                                                                                                 \
        for (int i = 0; i < 16; i++) r[i] = seed + i * 7 + threadIdx.x;                                            \
        int a = seed * 3 + threadIdx.x, b = seed + 11;                                                             \
        long long a64 = a, b64 = b; (void)a64; (void)b64;                                                          \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                \
        for (int it = 0; it < ITER; it++) {                                                                        \
            ASM3                                                                                                   \
        }                                                                                                          \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                \
        int s = 0;                                                                                                 \
        for (int i = 0; i < 16; i++) s += r[i];                                                                    \
        if (s == 0x12345678) out[1] = s;                                                                           \
        if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);                                                      \
    }
#define V3(OP) asm volatile(OP " %0, %1, %2, %0" : "+v"(r[0]) : "v"(a), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(r[1]) : "v"(a), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(r[2]) : "v"(a), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(r[3]) : "v"(a), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(r[4]) : "v"(a), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(r[5]) : "v"(a), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(r[6]) : "v"(a), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(r[7]) : "v"(a), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(r[8]) : "v"(a), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(r[9]) : "v"(a), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(r[10]) : "v"(a), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(r[11]) : "v"(a), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(r[12]) : "v"(a), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(r[13]) : "v"(a), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(r[14]) : "v"(a), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(r[15]) : "v"(a), "v"(b));
#define V2(OP) asm volatile(OP " %0, %1, %0" : "+v"(r[0]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[1]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[2]) : "v"(a)); \
    asm volatile(OP " %0, %1, %0" : "+v"(r[3]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[4]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[5]) : "v"(a)); \
    asm volatile(OP " %0, %1, %0" : "+v"(r[6]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[7]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[8]) : "v"(a)); \
    asm volatile(OP " %0, %1, %0" : "+v"(r[9]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[10]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[11]) : "v"(a)); \
    asm volatile(OP " %0, %1, %0" : "+v"(r[12]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[13]) : "v"(a)); asm volatile(OP " %0, %1, %0" : "+v"(r[14]) : "v"(a)); \
    asm volatile(OP " %0, %1, %0" : "+v"(r[15]) : "v"(a));

KERNEL(dot2_i32_i16, V3("v_dot2_i32_i16"))
KERNEL(dot2_u32_u16, V3("v_dot2_u32_u16"))
KERNEL(dot4_i32_i8, V3("v_dot4_i32_i8"))
KERNEL(dot4_u32_u8, V3("v_dot4_u32_u8"))
KERNEL(dot2c_i32_i16, V2("v_dot2c_i32_i16"))
KERNEL(mad_i32_i24, V3("v_mad_i32_i24"))
KERNEL(mad_u32_u24, V3("v_mad_u32_u24"))
KERNEL(perm_b32, V3("v_perm_b32"))
KERNEL(pk_add_u16, V2("v_pk_add_u16"))
KERNEL(pk_sub_i16, V2("v_pk_sub_i16"))
KERNEL(pk_max_i16, V2("v_pk_max_i16"))
KERNEL(pk_min_u16, V2("v_pk_min_u16"))
KERNEL(pk_mul_lo_u16, V2("v_pk_mul_lo_u16"))
KERNEL(pk_mad_i16, V3("v_pk_mad_i16"))
KERNEL(pk_lshrrev_b16, V2("v_pk_lshrrev_b16"))
KERNEL(add_u32, V2("v_add_u32"))
KERNEL(add3_u32, V3("v_add3_u32"))
KERNEL(lshl_add_u32, V3("v_lshl_add_u32"))
KERNEL(and_or_b32, V3("v_and_or_b32"))
KERNEL(bfe_i32, V3("v_bfe_i32"))
KERNEL(alignbit_b32, V3("v_alignbit_b32"))
KERNEL(min3_u32, V3("v_min3_u32"))
KERNEL(sad_u8, V3("v_sad_u8"))
KERNEL(sad_u16, V3("v_sad_u16"))
KERNEL(msad_u8, V3("v_msad_u8"))
KERNEL(mul_lo_u32, V2("v_mul_lo_u32"))
KERNEL(mul_hi_u32, V2("v_mul_hi_u32"))
KERNEL(fma_f32, V3("v_fma_f32"))
KERNEL(cndmask, V2("v_cndmask_b32"))
KERNEL(max_u32, V2("v_max_u32"))
KERNEL(ashrrev_i32, V2("v_ashrrev_i32"))

// 64-bit destination forms: 8 independent register pairs
#define V3Q(OP) asm volatile(OP " %0, %1, %2, %0" : "+v"(q[0]) : "v"(a64), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(q[1]) : "v"(a64), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(q[2]) : "v"(a64), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(q[3]) : "v"(a64), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(q[4]) : "v"(a64), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(q[5]) : "v"(a64), "v"(b)); \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(q[6]) : "v"(a64), "v"(b)); asm volatile(OP " %0, %1, %2, %0" : "+v"(q[7]) : "v"(a64), "v"(b));
#define KERNELQ(NAME, OP)                                                                                          \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long* out, int seed) {                         \
        long long q[8];                                                                                            \
        for (int i = 0; i < 8; i++) q[i] = seed + i * 7 + threadIdx.x;                                             \
        long long a64 = seed * 3 + threadIdx.x; int b = seed + 11;                                                 \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                \
        for (int it = 0; it < ITER; it++) { V3Q(OP) V3Q(OP) }                                                      \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                \
        long long s = 0;                                                                                           \
        for (int i = 0; i < 8; i++) s += q[i];                                                                     \
        if (s == 0x12345678) out[1] = s;                                                                           \
        if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);                                                      \
    }
KERNELQ(qsad_pk_u16_u8, "v_qsad_pk_u16_u8")
KERNELQ(mqsad_pk_u16_u8, "v_mqsad_pk_u16_u8")

template <typename K> void run(const char* name, K kern, unsigned long long* d) {
    hipMemset(d, 0, 16);
    const int wgs = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<wgs, 1024>>>(d, 3);   // warm
    hipMemset(d, 0, 16);
    hipEventRecord(e0);
    kern<<<wgs, 1024>>>(d, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    const double per_wave = (double)cyc / (wgs * 16.0);        // cycles one wave spent in its loop
    const double per_instr = per_wave / (ITER * 16.0) / 4.0;   // 4 waves share a SIMD: cycles of SIMD time per wave-instruction
    std::printf("%-18s %6.2f cycles per wave-instruction per SIMD  (%5.1f lanes/clk/SIMD)   kernel %.3f ms\n", name, per_instr, 64.0 / per_instr, ms);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
#define R(N) run(#N, k_##N, d);
    R(fma_f32) R(add_u32) R(add3_u32) R(lshl_add_u32) R(and_or_b32) R(bfe_i32) R(alignbit_b32) R(min3_u32) R(max_u32) R(cndmask) R(ashrrev_i32) R(perm_b32)
    R(mad_i32_i24) R(mad_u32_u24) R(mul_lo_u32) R(mul_hi_u32)
    R(pk_add_u16) R(pk_sub_i16) R(pk_max_i16) R(pk_min_u16) R(pk_mul_lo_u16) R(pk_mad_i16) R(pk_lshrrev_b16)
    R(dot2_i32_i16) R(dot2_u32_u16) R(dot2c_i32_i16) R(dot4_i32_i8) R(dot4_u32_u8)
    R(sad_u8) R(sad_u16) R(msad_u8) R(qsad_pk_u16_u8) R(mqsad_pk_u16_u8)
    return 0;
}
