// distortion.hip — the RD-side distortion reductions of SURVEY 8(a) D9 for lists of blocks; gfx950.
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Common/Codec/EbPictureOperators.c:156  svt_full_distortion_kernel32_bits_c       (sum (c - r)^2, sum c^2)
//   Common/Codec/EbPictureOperators.c:212  svt_full_distortion_kernel_cbf_zero32_bits_c (both = sum c^2)
//   Common/Codec/common_dsp_rtcd.c:56      svt_av1_block_error_c                      (same two sums on a flat block)
//   Common/Codec/common_dsp_rtcd.c:47      svt_aom_satd_c                             (sum |c|)
//   Encoder/Codec/EbEncInterPrediction.c:803 svt_aom_sse_c / svt_aom_highbd_sse_c, and svt_spatial_full_distortion_kernel_c
//                                            (sum (a - b)^2 over a pixel block)
// One wave per block; lanes stride over the block, 64-bit partial sums, one DPP-free shuffle tree at the end (the blocks are
// small: the kernel is a pure HBM stream — 8 B of coefficients per sample).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"

namespace {

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += ((unsigned long long)(uint32_t)__shfl_xor((int)(v >> 32), m, 64) << 32) | (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}

// out[blk][0] = sum (c - r)^2 (= sum c^2 when recon == nullptr), [1] = sum c^2, [2] = sum |c|
__global__ void __launch_bounds__(256)
coeff_distortion_kernel(const int32_t* __restrict__ coeff, const int32_t* __restrict__ recon, int n, int nblk, unsigned long long* __restrict__ out) {
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= nblk) return;
    const int32_t* c = coeff + (size_t)blk * n;
    const int32_t* r = recon ? recon + (size_t)blk * n : nullptr;
    unsigned long long res = 0, pred = 0, satd = 0;
    for (int i = lane; i < n; i += 64) {
        const long long cv = c[i], d = r ? cv - r[i] : cv;
        res += (unsigned long long)(d * d); pred += (unsigned long long)(cv * cv); satd += (unsigned long long)(cv < 0 ? -cv : cv);
    }
    res = wave_sum_u64(res); pred = wave_sum_u64(pred); satd = wave_sum_u64(satd);
    if (lane == 0) { out[3 * (size_t)blk] = res; out[3 * (size_t)blk + 1] = pred; out[3 * (size_t)blk + 2] = satd; }
}

template <typename PIX>
__global__ void __launch_bounds__(256)
block_sse_kernel(const PIX* __restrict__ a, int a_stride, const PIX* __restrict__ b, int b_stride, const SvtHipBlkPair* __restrict__ pairs, int n,
                 unsigned long long* __restrict__ out) {
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= n) return;
    const SvtHipBlkPair p = pairs[blk];
    unsigned long long s = 0;
    for (int i = lane; i < p.w * p.h; i += 64) {
        const int y = i / p.w, x = i - y * p.w;
        const long long d = (long long)a[(size_t)(p.a_y + y) * a_stride + p.a_x + x] - (long long)b[(size_t)(p.b_y + y) * b_stride + p.b_x + x];
        s += (unsigned long long)(d * d);
    }
    s = wave_sum_u64(s);
    if (lane == 0) out[blk] = s;
}

// The same sums for lists of LARGE rectangles (restoration units, whole stripes): blockIdx.x = pair, blockIdx.y = a slice of 32 rows repeating every
// 32 * gridDim.y rows; a wave takes 8 rows of the slice, lanes run along x; partial sums meet in out[pair] by 64-bit atomics (out cleared before).
template <typename PIX>
__global__ void __launch_bounds__(256)
block_sse_rows_kernel(const PIX* __restrict__ a, int a_stride, const PIX* __restrict__ b, int b_stride, const SvtHipBlkPair* __restrict__ pairs,
                      unsigned long long* __restrict__ out) {
    const SvtHipBlkPair p = pairs[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long s = 0;
    for (int row0 = blockIdx.y * 32 + wave * 8; row0 < p.h; row0 += gridDim.y * 32)
        for (int y = row0; y < min(row0 + 8, (int)p.h); y++) {
            const PIX* ra = a + (size_t)(p.a_y + y) * a_stride + p.a_x;
            const PIX* rb = b + (size_t)(p.b_y + y) * b_stride + p.b_x;
            for (int x = lane; x < p.w; x += 64) { const long long d = (long long)ra[x] - (long long)rb[x]; s += (unsigned long long)(d * d); }
        }
    s = wave_sum_u64(s);
    if (lane == 0 && s) atomicAdd(&out[blockIdx.x], s);
}

}  // namespace

extern "C" int svt_hip_launch_coeff_distortion(hipStream_t st, const int32_t* coeff, const int32_t* recon, int n, int nblk, uint64_t* out) {
    if (nblk <= 0) return 0;
    hipLaunchKernelGGL(coeff_distortion_kernel, dim3((nblk + 3) / 4), dim3(256), 0, st, coeff, recon, n, nblk, (unsigned long long*)out);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_block_sse(hipStream_t st, int pix_bytes, const void* a, int a_stride, const void* b, int b_stride,
                                        const SvtHipBlkPair* pairs, int n, uint64_t* out) {
    if (n <= 0) return 0;
    if (n <= 2048) {   // few pairs: probably large rectangles (restoration units) — row-sliced form; long lists are small blocks, one wave each
        if (hipMemsetAsync(out, 0, sizeof(uint64_t) * n, st) != hipSuccess) return (int)hipGetLastError();
        if (pix_bytes == 1) hipLaunchKernelGGL((block_sse_rows_kernel<uint8_t>), dim3(n, 8), dim3(256), 0, st, (const uint8_t*)a, a_stride, (const uint8_t*)b, b_stride, pairs, (unsigned long long*)out);
        else hipLaunchKernelGGL((block_sse_rows_kernel<uint16_t>), dim3(n, 8), dim3(256), 0, st, (const uint16_t*)a, a_stride, (const uint16_t*)b, b_stride, pairs, (unsigned long long*)out);
        return (int)hipGetLastError();
    }
    if (pix_bytes == 1) hipLaunchKernelGGL((block_sse_kernel<uint8_t>), dim3((n + 3) / 4), dim3(256), 0, st, (const uint8_t*)a, a_stride, (const uint8_t*)b, b_stride, pairs, n, (unsigned long long*)out);
    else hipLaunchKernelGGL((block_sse_kernel<uint16_t>), dim3((n + 3) / 4), dim3(256), 0, st, (const uint16_t*)a, a_stride, (const uint16_t*)b, b_stride, pairs, n, (unsigned long long*)out);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(distortion)
