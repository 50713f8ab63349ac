// wiener.hip — Wiener restoration statistics (the autocorrelation / cross-correlation sums of the filter search) on the matrix
// cores; gfx950.  (The Wiener *filter* pass lives in sgr.hip's loop-restoration apply kernel.)
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Encoder/Codec/EbRestorationPick.c:704  svt_av1_compute_stats_c  (+ find_average, EbRestorationPick.h:24) as called per restoration
//   unit by search_wiener_seg (:1347): M[k] = sum (y_k - avg)(x - avg), H[k][l] = sum (y_k - avg)(y_l - avg), k, l over the win x win
//   window offsets -- a Gram matrix of the (pixels x features) matrix Z = [window samples | source sample | 1].  This is the one
//   genuinely GEMM-shaped reduction on the hot path, so it runs on MFMA:
//     * samples are biased to int8 (d' = d - 128) and fed to v_mfma_i32_32x32x32_i8: exact int32 accumulation (|products| <= 2^14,
//       a wave sees < 2^17 pixels), A and B operands are THE SAME registers (G = Z^T Z), so any k-ordering inside the instruction
//       cancels out; the constant-1 feature yields sum d'_k and the pixel count, from which the kernel recovers the unit average
//       (find_average) and removes it algebraically:  H = G_kl - a (S_k + S_l) + N a^2,  M = G_kx - a (S_k + S_x) + N a^2,  a = avg - 128;
//     * a lane's operand = 16 consecutive pixels of one window offset (dx, dy): one aligned ds_read_b128 out of one of 7 LDS copies of
//       the row ring, copy dx being the rows shifted by dx bytes;
//     * one workgroup (4 waves) per restoration unit, wave w takes rows v_start + w, + 4, ...; 7x7: features 0..50 -> 2 blocks of 32 ->
//       3 MFMA tiles per 32 pixels; 5x5 / 3x3: one block, one MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kRing = 16;        // rows kept in LDS (10 needed per 4-row step)
constexpr int kPitch = 416;      // bytes per row copy: unit width <= 384 (1.5 x 256) + 2 x 3 window + 16-B over-read, multiple of 16

template <int WIN>
__global__ void __launch_bounds__(256)
wiener_stats8_kernel(const uint8_t* __restrict__ dgd, int dgd_stride, const uint8_t* __restrict__ src, int src_stride, int pw, int ph,
                     int unit_size, int units_x, int units_y, int voff, long long* __restrict__ M_out, long long* __restrict__ H_out) {
    constexpr int HALF = WIN / 2, NF = WIN * WIN, XF = NF, ONE = NF + 1, NBLK = (NF + 2 + 31) / 32, NT = NBLK == 1 ? 1 : 3;
    __shared__ __attribute__((aligned(16))) int8_t ring[7][kRing][kPitch];   // copy c holds d'[col + (c - 3)] at byte col + 16
    __shared__ __attribute__((aligned(16))) int8_t sring[4][kPitch];         // source rows of the current step (x' = s - 128)
    const int unit = blockIdx.x, ui = unit / units_x, uj = unit - ui * units_x;
    const int h0 = uj * unit_size, h1 = uj == units_x - 1 ? pw : h0 + unit_size;
    const int v0 = max(0, ui * unit_size - voff), v1 = ui == units_y - 1 ? ph : (ui + 1) * unit_size - voff;
    const int uw = h1 - h0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // lane -> features (block 0: lane & 31, block 1: 32 + (lane & 31)), k-group g = lane >> 5 (16 pixels each)
    const int g = lane >> 5;
    int copy[2], dy[2], kind[2];   // kind 0: window sample, 1: source sample, 2: ones, 3: unused
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int f = 32 * b + (lane & 31);
        kind[b] = f < NF ? 0 : (f == XF ? 1 : (f == ONE ? 2 : 3));
        copy[b] = f < NF ? f / WIN - HALF + 3 : 3;
        dy[b] = f < NF ? f % WIN - HALF : 0;
    }
    v16i C[3];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) C[t][r] = 0;

    auto stage_row = [&](int y) {   // all 256 threads: picture row y (clamped rows are never used by a valid pixel's window... they are:
        // rows outside the picture come from the caller's 3-px extension, like the reference reads them)
        const uint8_t* rowp = dgd + (ptrdiff_t)y * dgd_stride + h0;
        const int slot = y & (kRing - 1);
        for (int i = tid; i < (uw + 6 + 16 + 3) / 4 * 7; i += 256) {
            const int c = i % 7, q = i / 7;            // copy c, dword q of the row: bytes col = 4q .. 4q+3  (col 0 <-> picture column h0 - 16)
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = 4 * q + b - 16 + (c - 3);     // picture column offset from h0 of the sample stored at byte 4q + b
                const int px = (x >= -3 && x < uw + 3) ? (int)rowp[x] - 128 : 0;
                v |= (uint32_t)(px & 0xFF) << (8 * b);
            }
            *(uint32_t*)&ring[c][slot][4 * q] = v;
        }
    };

    int staged_to = v0 - 4;   // last staged picture row
    for (int y = v0 - 3; y <= min(v0 + 2, v1 + 2); y++) stage_row(y);
    staged_to = min(v0 + 2, v1 + 2);
    for (int r0 = v0; r0 < v1; r0 += 4) {
        __syncthreads();   // previous step's reads are done before the ring / source rows are overwritten
        for (int y = staged_to + 1; y <= min(r0 + 6, v1 + 2); y++) stage_row(y);
        staged_to = max(staged_to, min(r0 + 6, v1 + 2));
        for (int i = tid; i < 4 * ((uw + 15) / 16 * 4); i += 256) {   // source rows r0 .. r0+3, dwords
            const int q = i % ((uw + 15) / 16 * 4), rr = i / ((uw + 15) / 16 * 4);
            uint32_t v = 0;
            if (r0 + rr < v1) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int x = 4 * q + b;
                    const int px = x < uw ? (int)src[(size_t)(r0 + rr) * src_stride + h0 + x] - 128 : 0;
                    v |= (uint32_t)(px & 0xFF) << (8 * b);
                }
            }
            *(uint32_t*)&sring[rr][4 * q] = v;
        }
        __syncthreads();
        const int row = r0 + wave;
        if (row < v1) {
            for (int x0 = 0; x0 < uw; x0 += 32) {
                const int p0 = x0 + 16 * g;                       // first pixel of this lane's 16
                const int nvalid = min(max(uw - p0, 0), 16);
                v4i A[2];
#pragma unroll
                for (int b = 0; b < NBLK; b++) {
                    v4i a = {0, 0, 0, 0};
                    if (kind[b] == 0) a = *(const v4i*)&ring[copy[b]][(row + dy[b]) & (kRing - 1)][16 + p0];
                    else if (kind[b] == 1) a = *(const v4i*)&sring[wave][p0];
                    else if (kind[b] == 2) a = v4i{0x01010101, 0x01010101, 0x01010101, 0x01010101};
                    if (nvalid < 16) {   // pixels past the unit's right edge contribute nothing (only the last 32-pixel step)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int nb = min(max(nvalid - 4 * q, 0), 4);
                            a[q] &= nb == 4 ? -1 : (int)((1u << (8 * nb)) - 1u);
                        }
                    }
                    A[b] = a;
                }
                C[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], A[0], C[0], 0, 0, 0);
                if (NBLK == 2) {
                    C[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], A[1], C[1], 0, 0, 0);
                    C[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], A[1], C[2], 0, 0, 0);
                }
            }
        }
    }
    // ---- add the four waves' tiles (int32 -> int64) in LDS; element (row, col) of a tile: lane = col + 32 * ((row >> 2) & 1),
    //      register = (row & 3) + 4 * (row >> 3)   (C/D layout of the 32x32 MFMAs)
    __syncthreads();
    unsigned long long* acc = (unsigned long long*)&ring[0][0][0];   // [NT][32 * 32]
    for (int i = tid; i < NT * 1024; i += 256) acc[i] = 0ull;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), tcol = lane & 31;
            atomicAdd(&acc[t * 1024 + trow * 32 + tcol], (unsigned long long)(long long)C[t][r]);
        }
    __syncthreads();
    auto G = [&](int f1, int f2) -> long long {
        if (f1 > f2) { const int t = f1; f1 = f2; f2 = t; }
        const int t = f2 < 32 ? 0 : (f1 < 32 ? 1 : 2);
        return (long long)acc[t * 1024 + (f1 & 31) * 32 + (f2 & 31)];
    };
    const long long N = G(ONE, ONE);
    const long long sum_d = G(HALF * WIN + HALF, ONE) + 128 * N;
    const long long a = (long long)((unsigned long long)sum_d / (unsigned long long)N) - 128;   // find_average() - 128
    const long long Sx = G(XF, ONE);
    long long* Mo = M_out + (size_t)unit * NF;
    long long* Ho = H_out + (size_t)unit * NF * NF;
    for (int i = tid; i < NF * NF + NF; i += 256) {
        if (i < NF * NF) {
            const int k = i / NF, l = i - k * NF;
            Ho[i] = G(k, l) - a * (G(k, ONE) + G(l, ONE)) + N * a * a;
        } else {
            const int k = i - NF * NF;
            Mo[k] = G(k, XF) - a * (G(k, ONE) + Sx) + N * a * a;
        }
    }
}

}  // namespace

extern "C" int svt_hip_launch_wiener_stats8(hipStream_t st, int win, const uint8_t* dgd, int dgd_stride, const uint8_t* src, int src_stride, int pw,
                                            int ph, int unit_size, int units_x, int units_y, int ss_y, int64_t* M, int64_t* H) {
    const int voff = 8 >> ss_y, n = units_x * units_y;
    if (n <= 0) return 0;
    if (win == 7) hipLaunchKernelGGL((wiener_stats8_kernel<7>), dim3(n), dim3(256), 0, st, dgd, dgd_stride, src, src_stride, pw, ph, unit_size, units_x, units_y, voff, (long long*)M, (long long*)H);
    else if (win == 5) hipLaunchKernelGGL((wiener_stats8_kernel<5>), dim3(n), dim3(256), 0, st, dgd, dgd_stride, src, src_stride, pw, ph, unit_size, units_x, units_y, voff, (long long*)M, (long long*)H);
    else hipLaunchKernelGGL((wiener_stats8_kernel<3>), dim3(n), dim3(256), 0, st, dgd, dgd_stride, src, src_stride, pw, ph, unit_size, units_x, units_y, voff, (long long*)M, (long long*)H);
    return (int)hipGetLastError();
}
