// compound.hip — compound (two-reference) inter prediction for a list of blocks in one launch; gfx950.  SURVEY 8(f) rank 4.
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Common/Codec/EbInterPrediction.c:552-741   svt_av1_jnt_convolve_{2d,y,x,2d_copy}_c          (common_dsp_rtcd.h:221-243), and :944-1143 highbd
//   Common/C_DEFAULT/EbInterPrediction_c.c:15  svt_av1_build_compound_diffwtd_mask_d16_c        (common_dsp_rtcd.h:115)
//   Common/Codec/EbBlend_a64_mask.c:34 / :110  svt_aom_{lowbd,highbd}_blend_a64_d16_mask_c      (as build_masked_compound_no_round uses them)
// The reference predicts reference 0 into a 16-bit "ConvBufType" buffer, then runs the second convolve with do_average (or blends the two
// 16-bit buffers under a mask).  Here one workgroup owns one block: per 16x16 tile both references are staged through LDS (23 x 23
// samples each), both 16-bit intermediates stay in a register, and only the final sample (and, for COMPOUND_DIFFWTD, the segmentation
// mask the chroma planes reuse) is written: algorithmic bytes = 2 reads + 1 write per sample, no intermediate buffer in HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"
#include "interp_kernels.h"

namespace {

__device__ __forceinline__ int rp2(int v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); }

// the 16-bit intermediate of sample (tx, ty) of the staged 23 x 23 tile (s_src, pitch 24, origin (-3, -3)): the do_average == 0 value
// of the four jnt_convolve flavours; s_im is scratch for the 2-D case (uniform control flow: sx / sy are per-block)
template <int BD>
__device__ __forceinline__ int d16_sample(const int* s_src, int* s_im, const int* xf, const int* yf, int sx, int sy, int tid, int tx, int ty) {
    constexpr int r0 = BD == 12 ? 5 : 3, r1 = 7, offset_bits = BD + 14 - r0, round_offset = (1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1));
    if (!sx && !sy) return (s_src[(ty + 3) * 24 + tx + 3] << (14 - r1 - r0)) + round_offset;
    if (!sy) {
        int res = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) res += xf[k] * s_src[(ty + 3) * 24 + tx + k];
        return (1 << (7 - r1)) * rp2(res, r0) + round_offset;
    }
    if (!sx) {
        int res = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) res += yf[k] * s_src[(ty + k) * 24 + tx + 3];
        return rp2(res * (1 << (7 - r0)), r1) + round_offset;
    }
    for (int i = tid; i < 23 * 16; i += 256) {
        const int r = i >> 4, c = i & 15;
        int sum = 1 << (BD + 6);
#pragma unroll
        for (int k = 0; k < 8; k++) sum += xf[k] * s_src[r * 24 + c + k];
        s_im[i] = (int)(int16_t)rp2(sum, r0);
    }
    __syncthreads();
    int sum = 1 << offset_bits;
#pragma unroll
    for (int k = 0; k < 8; k++) sum += yf[k] * s_im[(ty + k) * 16 + tx];
    return (int)(uint16_t)rp2(sum, r1);
}

template <typename PIX>
__device__ __forceinline__ void stage_tile(int* s_src, const PIX* __restrict__ ref, int ref_stride, int x0, int y0, int tid) {
    PIX v[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const int i = tid + 256 * u, r = i / 23, c = i - r * 23;
        if (i < 23 * 23) v[u] = ref[(ptrdiff_t)(y0 + r - 3) * ref_stride + (x0 + c - 3)];
    }
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const int i = tid + 256 * u, r = i / 23, c = i - r * 23;
        if (i < 23 * 23) s_src[r * 24 + c] = v[u];
    }
}

template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
compound_predict_kernel(const PIX* __restrict__ ref0, int ref0_stride, const PIX* __restrict__ ref1, int ref1_stride, PIX* __restrict__ dst, int dst_stride,
                        uint8_t* __restrict__ masks, const SvtHipCompBlk* __restrict__ blks) {
    __shared__ int s_src[23 * 24];
    __shared__ int s_im[23 * 16];
    const SvtHipCompBlk b = blks[blockIdx.x];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    constexpr int pix_max = (1 << BD) - 1;
    constexpr int r0 = BD == 12 ? 5 : 3, r1 = 7, offset_bits = BD + 14 - r0, round_offset = (1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1));
    constexpr int round_bits = 14 - r0 - r1;
    int xf[8], yf[8];
    for (int oy = 0; oy < b.h; oy += 16)
        for (int ox = 0; ox < b.w; ox += 16) {
            int d[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int sx = (r ? b.subpel1_x : b.subpel0_x) & 15, sy = (r ? b.subpel1_y : b.subpel0_y) & 15;
#pragma unroll
                for (int k = 0; k < 8; k++) { xf[k] = kInterp[b.bank_x][sx][k]; yf[k] = kInterp[b.bank_y][sy][k]; }
                __syncthreads();
                if (r == 0) stage_tile(s_src, ref0, ref0_stride, b.src0_x + ox, b.src0_y + oy, tid);
                else stage_tile(s_src, ref1, ref1_stride, b.src1_x + ox, b.src1_y + oy, tid);
                __syncthreads();
                d[r] = d16_sample<BD>(s_src, s_im, xf, yf, sx, sy, tid, tx, ty);
            }
            const bool live = (ox + tx < b.w) && (oy + ty < b.h);
            if (!live) continue;
            const int x = ox + tx, y = oy + ty;
            int tmp;
            if (b.type <= 1) {
                tmp = b.type ? (d[0] * b.fwd_offset + d[1] * b.bck_offset) >> 4 : (d[0] + d[1]) >> 1;
            } else {
                int m;
                if (b.type == 2) {
                    const int diff = rp2(abs(d[0] - d[1]), round_bits + (BD - 8));
                    m = min(38 + (diff >> 4), 64);
                    if (b.mask_type) m = 64 - m;
                    if (b.mask_off >= 0) masks[(size_t)b.mask_off + (size_t)y * b.w + x] = (uint8_t)m;
                } else {
                    const uint8_t* mp = masks + (size_t)b.mask_off;
                    const int ms = b.mask_stride;
                    if (b.mask_sub) m = rp2(mp[(2 * y) * ms + 2 * x] + mp[(2 * y + 1) * ms + 2 * x] + mp[(2 * y) * ms + 2 * x + 1] + mp[(2 * y + 1) * ms + 2 * x + 1], 2);
                    else m = mp[y * ms + x];
                }
                tmp = (m * d[0] + (64 - m) * d[1]) >> 6;
            }
            tmp -= round_offset;
            dst[(ptrdiff_t)(b.dst_y + y) * dst_stride + (b.dst_x + x)] = (PIX)min(max(rp2(tmp, round_bits), 0), pix_max);
        }
}


// ---- OBMC motion-search costs: svt_aom_obmc_sad{W}x{H} (Encoder/C_DEFAULT/sad_av1.c:18-38), svt_aom_obmc_variance{W}x{H} and
// svt_aom_obmc_sub_pixel_variance{W}x{H} (Encoder/C_DEFAULT/variance.c:270-318): one wave per block; out[blk] = {sad, sse, variance}.
// The bilinear sample is recomputed per pixel from its 2 x 2 neighbourhood (first pass rounded to 16 bits, second to 8, as the two passes do).
__global__ void __launch_bounds__(256)
obmc_cost_kernel(const uint8_t* __restrict__ pre, int pre_stride, const int32_t* __restrict__ wsrc, const int32_t* __restrict__ mask,
                 const SvtHipObmcBlk* __restrict__ blks, int n, uint32_t* __restrict__ out) {
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= n) return;
    const SvtHipObmcBlk b = blks[blk];
    const uint8_t* p = pre + (ptrdiff_t)b.pre_y * pre_stride + b.pre_x;
    const int32_t *ws = wsrc + b.wm_off, *mk = mask + b.wm_off;
    const int fx0 = 128 - 16 * b.xoffset, fx1 = 16 * b.xoffset, fy0 = 128 - 16 * b.yoffset, fy1 = 16 * b.yoffset;
    uint32_t sad = 0, sse = 0;
    int sum = 0;
    for (int i = lane; i < b.w * b.h; i += 64) {
        const int y = i / b.w, x = i - y * b.w;
        const uint8_t* q = p + (ptrdiff_t)y * pre_stride + x;
        const int a00 = q[0], a01 = q[1], a10 = q[pre_stride], a11 = q[pre_stride + 1];
        const int m = mk[i], w0 = ws[i];
        sad += (uint32_t)rp2(abs(w0 - a00 * m), 12);
        const int r0 = rp2(a00 * fx0 + a01 * fx1, 7), r1 = rp2(a10 * fx0 + a11 * fx1, 7);
        const int pf = rp2(r0 * fy0 + r1 * fy1, 7) & 0xff;
        const int v = w0 - pf * m;
        const int d = v < 0 ? -rp2(-v, 12) : rp2(v, 12);
        sum += d; sse += (uint32_t)(d * d);
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { sad += (uint32_t)__shfl_xor((int)sad, s, 64); sse += (uint32_t)__shfl_xor((int)sse, s, 64); sum += __shfl_xor(sum, s, 64); }
    if (lane == 0) {
        out[3 * (size_t)blk] = sad; out[3 * (size_t)blk + 1] = sse;
        out[3 * (size_t)blk + 2] = sse - (uint32_t)(((long long)sum * sum) / (b.w * b.h));
    }
}


// ---- pixel-domain mask blends: svt_aom_[highbd_]blend_a64_{mask,hmask,vmask} (Common/Codec/EbBlend_a64_mask.c:214-434): OBMC, inter-intra,
// pixel-domain masked compound.  One workgroup per block, 256 threads stride over its samples.
template <typename PIX>
__global__ void __launch_bounds__(256)
blend_a64_kernel(const PIX* __restrict__ src0, int src0_stride, const PIX* __restrict__ src1, int src1_stride, PIX* __restrict__ dst, int dst_stride,
                 const uint8_t* __restrict__ masks, const SvtHipBlendBlk* __restrict__ blks) {
    const SvtHipBlendBlk b = blks[blockIdx.x];
    const uint8_t* mk = masks + b.mask_off;
    const int ms = b.mask_stride;
    for (int i = threadIdx.x; i < b.w * b.h; i += 256) {
        const int y = i / b.w, x = i - y * b.w;
        int m;
        if (b.mode == 1) m = mk[x];
        else if (b.mode == 2) m = mk[y];
        else if (!b.subw && !b.subh) m = mk[y * ms + x];
        else if (b.subw && b.subh) m = rp2(mk[(2 * y) * ms + 2 * x] + mk[(2 * y + 1) * ms + 2 * x] + mk[(2 * y) * ms + 2 * x + 1] + mk[(2 * y + 1) * ms + 2 * x + 1], 2);
        else if (b.subw) m = rp2(mk[y * ms + 2 * x] + mk[y * ms + 2 * x + 1], 1);
        else m = rp2(mk[(2 * y) * ms + x] + mk[(2 * y + 1) * ms + x], 1);
        const int v0 = src0[(ptrdiff_t)(b.src0_y + y) * src0_stride + b.src0_x + x], v1 = src1[(ptrdiff_t)(b.src1_y + y) * src1_stride + b.src1_x + x];
        dst[(ptrdiff_t)(b.dst_y + y) * dst_stride + b.dst_x + x] = (PIX)rp2(m * v0 + (64 - m) * v1, 6);
    }
}

}  // namespace

extern "C" int svt_hip_launch_compound_predict(hipStream_t st, int pix_bytes, int bd, const void* ref0, int ref0_stride, const void* ref1, int ref1_stride,
                                               void* dst, int dst_stride, uint8_t* masks, const SvtHipCompBlk* blks, int n) {
    if (n <= 0) return 0;
#define LAUNCH(P, B) hipLaunchKernelGGL((compound_predict_kernel<P, B>), dim3(n), dim3(256), 0, st, (const P*)ref0, ref0_stride, (const P*)ref1, ref1_stride, \
                                        (P*)dst, dst_stride, masks, blks)
    if (pix_bytes == 1) LAUNCH(uint8_t, 8);
    else if (bd == 8) LAUNCH(uint16_t, 8);
    else if (bd == 10) LAUNCH(uint16_t, 10);
    else LAUNCH(uint16_t, 12);
#undef LAUNCH
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_obmc_cost(hipStream_t st, const uint8_t* pre, int pre_stride, const int32_t* wsrc, const int32_t* mask, const SvtHipObmcBlk* blks, int n,
                                        uint32_t* out) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(obmc_cost_kernel, dim3((n + 3) / 4), dim3(256), 0, st, pre, pre_stride, wsrc, mask, blks, n, out);
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_blend_a64(hipStream_t st, int pix_bytes, const void* src0, int src0_stride, const void* src1, int src1_stride, void* dst, int dst_stride,
                                        const uint8_t* masks, const SvtHipBlendBlk* blks, int n) {
    if (n <= 0) return 0;
    if (pix_bytes == 1) hipLaunchKernelGGL((blend_a64_kernel<uint8_t>), dim3(n), dim3(256), 0, st, (const uint8_t*)src0, src0_stride, (const uint8_t*)src1, src1_stride,
                                           (uint8_t*)dst, dst_stride, masks, blks);
    else hipLaunchKernelGGL((blend_a64_kernel<uint16_t>), dim3(n), dim3(256), 0, st, (const uint16_t*)src0, src0_stride, (const uint16_t*)src1, src1_stride, (uint16_t*)dst,
                            dst_stride, masks, blks);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(compound)
