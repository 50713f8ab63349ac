// conv.hip — batched sub-pel prediction (AV1 *_sr convolve and the sub-pel-search predictor), block
// SAD and block variance; gfx950.
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Common/Codec/EbInterPrediction.c:349-469   svt_av1_convolve_{2d,x,y,2d_copy}_sr_c
//   Common/Codec/EbInterPrediction.c:744-866   svt_av1_highbd_convolve_{2d_copy,x,y,2d}_sr_c
//   Encoder/C_DEFAULT/variance.c:212-269       svt_aom_upsampled_pred_c  (+ Common/Codec/convolve.c:249-307 convolve8_*)
//   Encoder/C_DEFAULT/EbComputeVariance_C.c:14-77, Encoder/Codec/EbPsnr.c:170-233   svt_aom_[highbd_10_]variance{W}x{H}_c
//   Encoder/C_DEFAULT/EbComputeSAD_C.c:20,39    svt_nxm_sad_kernel / sad_16b_kernel (and the svt_aom_sad{W}x{H} family)
// One workgroup = one block; the block is walked in 16x16 output tiles: (16+7)^2 source samples are staged
// in LDS, the horizontal pass writes a 23x16 intermediate to LDS, the vertical pass writes the output —
// the separable structure and every rounding step of the reference are kept.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"
#include "interp_kernels.h"

namespace {

__device__ __forceinline__ int rp2(int v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); }

template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
subpel_predict_kernel(const PIX* __restrict__ ref, int ref_stride, PIX* __restrict__ dst, int dst_stride,
                      const SvtHipConvBlk* __restrict__ blks) {
    __shared__ int s_src[23 * 24];
    __shared__ int s_im[23 * 16];
    const SvtHipConvBlk b = blks[svt_xcd_order(blockIdx.x, gridDim.x)];   // the job list is in raster order of the blocks: an XCD takes a band of them
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int sx = b.subpel_x & 15, sy = b.subpel_y & 15;
    constexpr int pix_max = (1 << BD) - 1;
    int xf[8], yf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { xf[k] = kInterp[b.bank_x][sx][k]; yf[k] = kInterp[b.bank_y][sy][k]; }
    for (int oy = 0; oy < b.h; oy += 16)
        for (int ox = 0; ox < b.w; ox += 16) {
            __syncthreads();
            {   // 529 samples / 256 threads: issue all three loads before the first LDS store
                PIX v[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int i = tid + 256 * u, r = i / 23, c = i - r * 23;
                    if (i < 23 * 23) v[u] = ref[(ptrdiff_t)(b.src_y + oy + r - 3) * ref_stride + (b.src_x + ox + c - 3)];
                }
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int i = tid + 256 * u, r = i / 23, c = i - r * 23;
                    if (i < 23 * 23) s_src[r * 24 + c] = v[u];
                }
            }
            __syncthreads();
            const bool live = (ox + tx < b.w) && (oy + ty < b.h);
            int out = 0;
            if (!sx && !sy) {
                out = s_src[(ty + 3) * 24 + tx + 3];
            } else if (!sy) {
                int res = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) res += xf[k] * s_src[(ty + 3) * 24 + tx + k];
                out = b.mode ? rp2(res, 7) : rp2(rp2(res, 3), 4);            // x_sr: round_0 then FILTER_BITS - round_0 (EbInterPrediction.c:425-453)
                out = min(max(out, 0), pix_max);
            } else if (!sx) {
                int res = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) res += yf[k] * s_src[(ty + k) * 24 + tx + 3];
                out = min(max(rp2(res, 7), 0), pix_max);                     // y_sr (:395-423) / convolve8_vert
            } else {
                for (int i = tid; i < 23 * 16; i += 256) {                   // horizontal pass over 16 + 7 rows
                    const int r = i >> 4, c = i & 15;
                    int sum = b.mode ? 0 : (1 << (BD + 6));
#pragma unroll
                    for (int k = 0; k < 8; k++) sum += xf[k] * s_src[r * 24 + c + k];
                    s_im[i] = b.mode ? min(max(rp2(sum, 7), 0), 255)         // 8-bit intermediate (variance.c:245-254)
                                     : (int)(int16_t)rp2(sum, 3);           // im_block, :366-374
                }
                __syncthreads();
                if (b.mode) {
                    int res = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) res += yf[k] * s_im[(ty + k) * 16 + tx];
                    out = min(max(rp2(res, 7), 0), 255);
                } else {
                    constexpr int offset_bits = BD + 14 - 3;
                    int sum = 1 << offset_bits;
#pragma unroll
                    for (int k = 0; k < 8; k++) sum += yf[k] * s_im[(ty + k) * 16 + tx];
                    int res = rp2(sum, 11) - ((1 << (offset_bits - 11)) + (1 << (offset_bits - 12)));
                    if (sizeof(PIX) == 1) res = (int16_t)res;
                    out = min(max(res, 0), pix_max);                         // bits = 0 (:376-392)
                }
            }
            if (live) dst[(ptrdiff_t)(b.dst_y + oy + ty) * dst_stride + (b.dst_x + ox + tx)] = (PIX)out;
        }
}

// ---- block SAD / variance: one wave per block pair
template <typename PIX>
__global__ void __launch_bounds__(64)
block_sad_kernel(const PIX* __restrict__ a, int a_stride, const PIX* __restrict__ b, int b_stride, const SvtHipBlkPair* __restrict__ d,
                 uint32_t* __restrict__ out) {
    const SvtHipBlkPair p = d[blockIdx.x];
    uint32_t s = 0;
    for (int i = threadIdx.x; i < p.w * p.h; i += 64) {
        const int y = i / p.w, x = i - y * p.w;
        const int va = a[(ptrdiff_t)(p.a_y + y) * a_stride + p.a_x + x], vb = b[(ptrdiff_t)(p.b_y + y) * b_stride + p.b_x + x];
        s += (uint32_t)abs(va - vb);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) s += (uint32_t)__shfl_xor((int)s, m, 64);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
template <typename PIX, int BD>
__global__ void __launch_bounds__(64)
block_variance_kernel(const PIX* __restrict__ a, int a_stride, const PIX* __restrict__ b, int b_stride, const SvtHipBlkPair* __restrict__ d,
                      uint32_t* __restrict__ var_out, uint32_t* __restrict__ sse_out) {
    const SvtHipBlkPair p = d[blockIdx.x];
    long long sum = 0;
    unsigned long long sse = 0;
    for (int i = threadIdx.x; i < p.w * p.h; i += 64) {
        const int y = i / p.w, x = i - y * p.w;
        const int df = (int)a[(ptrdiff_t)(p.a_y + y) * a_stride + p.a_x + x] - (int)b[(ptrdiff_t)(p.b_y + y) * b_stride + p.b_x + x];
        sum += df;
        sse += (unsigned)(df * df);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        sum += ((long long)__shfl_xor((int)(sum >> 32), m, 64) << 32) | (unsigned)__shfl_xor((int)sum, m, 64);
        sse += ((unsigned long long)(unsigned)__shfl_xor((int)(sse >> 32), m, 64) << 32) | (unsigned)__shfl_xor((int)sse, m, 64);
    }
    if (threadIdx.x == 0) {
        const int n = p.w * p.h;
        uint32_t s2, v;
        if (BD == 8) {   // svt_aom_variance{W}x{H}_c
            s2 = (uint32_t)sse;
            v = s2 - (uint32_t)((sum * sum) / n);
        } else if (BD == 16) {   // variance_highbd_c (EbComputeVariance_C.c:34-52): 32-bit sums, int arithmetic for sum * sum / n
            s2 = (uint32_t)sse;
            const int sm = (int)sum;
            v = s2 - (uint32_t)((int)((unsigned)sm * (unsigned)sm) / n);
        } else {         // svt_aom_highbd_10_variance{W}x{H}_c: sse >> 4, sum >> 2 with rounding, clamp at 0
            s2 = (uint32_t)((sse + 8) >> 4);
            const int sm = (int)((sum + 2) >> 2);
            const long long vv = (long long)s2 - (((long long)sm * sm) / n);
            v = vv >= 0 ? (uint32_t)vv : 0u;
        }
        var_out[blockIdx.x] = v;
        if (sse_out) sse_out[blockIdx.x] = s2;
    }
}

}  // namespace

extern "C" int svt_hip_launch_subpel_predict(hipStream_t st, int pix_bytes, int bd, const void* ref, int ref_stride, void* dst,
                                             int dst_stride, const SvtHipConvBlk* blks, int n) {
    if (n <= 0) return 0;
    if (pix_bytes == 1) hipLaunchKernelGGL((subpel_predict_kernel<uint8_t, 8>), dim3(n), dim3(256), 0, st, (const uint8_t*)ref, ref_stride, (uint8_t*)dst, dst_stride, blks);
    else if (bd == 8) hipLaunchKernelGGL((subpel_predict_kernel<uint16_t, 8>), dim3(n), dim3(256), 0, st, (const uint16_t*)ref, ref_stride, (uint16_t*)dst, dst_stride, blks);
    else hipLaunchKernelGGL((subpel_predict_kernel<uint16_t, 10>), dim3(n), dim3(256), 0, st, (const uint16_t*)ref, ref_stride, (uint16_t*)dst, dst_stride, blks);
    return (int)hipGetLastError();
}
// The SvtHipConvBlk list of every whole 16x16 luma block from the open-loop ME table: block (bx, by) of the picture reads its integer vector from
// its superblock's 16x16 PU (EbMeTierZeroPu order: 5 + z-order index of the 16x16 inside the 64x64; MV word = y << 16 | x in quarter-pel, full-pel
// vectors) and gets the eighth-pel phase pair frac[2 * k] / frac[2 * k + 1] (q4, 0..15; NULL: full-pel).  k = raster index over whole blocks.
__global__ void __launch_bounds__(256)
subpel_jobs_from_me_kernel(const uint32_t* __restrict__ mv, int sb_cols, int w, int h, const uint8_t* __restrict__ frac, SvtHipConvBlk* __restrict__ out) {
    const int bw = w >> 4, bh = h >> 4, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= bw * bh) return;
    const int bx = k % bw, by = k / bw, sb = (by >> 2) * sb_cols + (bx >> 2), qx = bx & 3, qy = by & 3;
    const int z = ((qy >> 1) * 2 + (qx >> 1)) * 4 + (qy & 1) * 2 + (qx & 1);
    const uint32_t word = mv[(size_t)sb * 85 + 5 + z];
    const int mx = (int)(int16_t)(word & 0xffff) >> 2, my = (int)(int16_t)(word >> 16) >> 2;
    SvtHipConvBlk b;
    b.src_x = bx * 16 + mx; b.src_y = by * 16 + my; b.dst_x = bx * 16; b.dst_y = by * 16;
    b.w = 16; b.h = 16; b.bank_x = 0; b.bank_y = 0;
    b.subpel_x = frac ? frac[2 * k] & 15 : 0; b.subpel_y = frac ? frac[2 * k + 1] & 15 : 0;
    b.mode = 0; b.reserved = 0;
    out[k] = b;
}
extern "C" int svt_hip_launch_subpel_jobs_from_me(hipStream_t st, const uint32_t* mv, int sb_cols, int w, int h, const uint8_t* frac, SvtHipConvBlk* out) {
    const int n = (w >> 4) * (h >> 4);
    if (n > 0) hipLaunchKernelGGL(subpel_jobs_from_me_kernel, dim3((n + 255) / 256), dim3(256), 0, st, mv, sb_cols, w, h, frac, out);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_block_sad(hipStream_t st, int pix_bytes, const void* a, int a_stride, const void* b, int b_stride,
                                        const SvtHipBlkPair* d, int n, uint32_t* out) {
    if (n <= 0) return 0;
    if (pix_bytes == 1) hipLaunchKernelGGL((block_sad_kernel<uint8_t>), dim3(n), dim3(64), 0, st, (const uint8_t*)a, a_stride, (const uint8_t*)b, b_stride, d, out);
    else hipLaunchKernelGGL((block_sad_kernel<uint16_t>), dim3(n), dim3(64), 0, st, (const uint16_t*)a, a_stride, (const uint16_t*)b, b_stride, d, out);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_block_variance(hipStream_t st, int pix_bytes, int bd, const void* a, int a_stride, const void* b, int b_stride,
                                             const SvtHipBlkPair* d, int n, uint32_t* var_out, uint32_t* sse_out) {
    if (n <= 0) return 0;
    if (pix_bytes == 1) hipLaunchKernelGGL((block_variance_kernel<uint8_t, 8>), dim3(n), dim3(64), 0, st, (const uint8_t*)a, a_stride, (const uint8_t*)b, b_stride, d, var_out, sse_out);
    else if (bd == 16) hipLaunchKernelGGL((block_variance_kernel<uint16_t, 16>), dim3(n), dim3(64), 0, st, (const uint16_t*)a, a_stride, (const uint16_t*)b, b_stride, d, var_out, sse_out);
    else hipLaunchKernelGGL((block_variance_kernel<uint16_t, 10>), dim3(n), dim3(64), 0, st, (const uint16_t*)a, a_stride, (const uint16_t*)b, b_stride, d, var_out, sse_out);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(conv)
