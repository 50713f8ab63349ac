// percall.hip — stand-alone forms of operations that otherwise only exist fused inside the frame kernels:
// quantizers, residual, the 8-candidate SAD ladders of the open-loop search, the variance intermediates of picture
// analysis, the 64-point coefficient re-pack and the sub-pel prediction of the OBMC search.  Each kernel takes a LIST of
// units so a caller with many of them pays one launch; the per-call table of include/svt_hip_rtcd.h launches them with
// a list of one.  Bit-exact restatements of (paths under /root/reference/Source/Lib):
//   quantize_blocks        Encoder/Codec/EbFullLoop.c:37-93, :171-225, :314-377, :467-532
//   residual               Common/Codec/EbPictureOperators.c (svt_residual_kernel8bit_c / 16bit_c)
//   ext_all_sad            Encoder/Codec/EbMotionEstimation.c:230-388
//   ext_eight_sad_32_64    Encoder/Codec/EbMotionEstimation.c:394-458
//   interm_var_four8x8     Encoder/Codec/EbPictureAnalysisProcess.c:309-381
//   handle_transform64     Encoder/Codec/EbTransforms.c:2750-2932
//   upsampled_pred         Encoder/C_DEFAULT/variance.c:212-266 + Common/Codec/convolve.c:244-316
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"
#include "interp_kernels.h"
#include "quant_dev.h"

namespace {

__constant__ int16_t kTaps[6][16][8] = SVT_HIP_INTERP_TABLE;

// ------------------------------------------------------------------------------------------------ quantizers
// one workgroup per block of n coefficients; eob = 1 + the largest scan position holding a non-zero level
__global__ void __launch_bounds__(256)
quantize_blocks_kernel(const int32_t* __restrict__ coeff, int n, SvtHipQuantParams qp, const int16_t* __restrict__ iscan, int32_t* __restrict__ qcoeff,
                       int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob) {
    __shared__ unsigned s_eob;
    const size_t base = (size_t)blockIdx.x * n;
    if (threadIdx.x == 0) s_eob = 0;
    __syncthreads();
    unsigned e = 0;
    for (int rc = threadIdx.x; rc < n; rc += 256) {
        const int32_t c = coeff[base + rc];
        int32_t dq;
        const int32_t lvl = quant_one(qp, c, rc != 0, dq);
        qcoeff[base + rc] = c < 0 ? -lvl : lvl;
        dqcoeff[base + rc] = c < 0 ? -dq : dq;
        if (lvl) e = max(e, (unsigned)iscan[rc] + 1u);
    }
    if (e) atomicMax(&s_eob, e);
    __syncthreads();
    if (threadIdx.x == 0) eob[blockIdx.x] = (uint16_t)s_eob;
}

// ------------------------------------------------------------------------------------------------ residual
template <typename PIX>
__global__ void __launch_bounds__(256)
residual_kernel(const PIX* __restrict__ src, int ss, const PIX* __restrict__ pred, int ps, int16_t* __restrict__ res, int rs, int w, int h) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < w && y < h) res[(size_t)y * rs + x] = (int16_t)((int)src[(size_t)y * ss + x] - (int)pred[(size_t)y * ps + x]);
}

// ------------------------------------------------------------------------------------------------ 8-candidate SAD ladders
// state layout (uint32): best_sad8x8[64] best_sad16x16[16] best_mv8x8[64] best_mv16x16[16] eight_sad16x16[16][8] eight_sad8x8[64][8]
// (indices in the reference's z-order: 16x16 block (y, x) -> kZ16[4 * y + x], its 8x8 quadrant k -> 4 * that + k)
__constant__ uint8_t kZ16[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
__device__ __forceinline__ uint32_t step_mv(uint32_t mv, int j) {   // x advances 4 quarter-pel units per candidate
    const int16_t x = (int16_t)((int16_t)(mv & 0xffff) + (int16_t)(j * 4)), y = (int16_t)(mv >> 16);
    return ((uint32_t)(uint16_t)y << 16) | (uint16_t)x;
}
__global__ void __launch_bounds__(512)
ext_all_sad_kernel(const uint8_t* __restrict__ src, int ss, const uint8_t* __restrict__ ref, int rs, const SvtHipExtSadJob* __restrict__ jobs, uint32_t* __restrict__ state) {
    __shared__ uint32_t s8[64][8], s16[16][8];
    const SvtHipExtSadJob j = jobs[blockIdx.x];
    uint32_t* st = state + (size_t)blockIdx.x * 800;
    const int tid = threadIdx.x, b = tid >> 3, c = tid & 7, by = b >> 3, bx = b & 7;
    const uint8_t* s = src + j.src_off + (by * 8) * ss + bx * 8;
    const uint8_t* r = ref + j.ref_off + (by * 8) * rs + bx * 8 + c;
    uint32_t sad = 0;
    const int rstep = j.sub_sad ? 2 : 1;
    for (int y = 0; y < 8; y += rstep)
#pragma unroll
        for (int x = 0; x < 8; x++) sad += (uint32_t)abs((int)s[y * ss + x] - (int)r[y * rs + x]);
    if (j.sub_sad) sad <<= 1;
    const int z16 = kZ16[4 * (by >> 1) + (bx >> 1)], z8 = 4 * z16 + 2 * (by & 1) + (bx & 1);
    s8[z8][c] = sad;
    st[160 + 128 + z8 * 8 + c] = sad;
    __syncthreads();
    if (tid < 128) {
        const int k = tid >> 3;
        const uint32_t v = s8[4 * k][c] + s8[4 * k + 1][c] + s8[4 * k + 2][c] + s8[4 * k + 3][c];
        s16[k][c] = v;
        st[160 + k * 8 + c] = v;
    }
    __syncthreads();
    if (tid < 80) {   // candidates in order, strictly-smaller wins: the first smallest keeps the vector
        const bool is8 = tid < 64;
        const int k = is8 ? tid : tid - 64;
        uint32_t* bs = st + (is8 ? 0 : 64) + k;
        uint32_t* bm = st + (is8 ? 80 : 144) + k;
        uint32_t best = *bs, mv = *bm;
        for (int q = 0; q < 8; q++) {
            const uint32_t v = is8 ? s8[k][q] : s16[k][q];
            if (v < best) { best = v; mv = step_mv(j.mv, q); }
        }
        *bs = best; *bm = mv;
    }
}
// state layout (uint32): sad16x16[16][8] (in) best_sad32x32[4] best_sad64x64 best_mv32x32[4] best_mv64x64 (in/out) sad32x32[4][8] (out) = 170
__global__ void __launch_bounds__(64)
ext_eight_sad_32_64_kernel(const uint32_t* __restrict__ mvs, uint32_t* __restrict__ state) {
    __shared__ uint32_t s32[5][8];
    uint32_t* st = state + (size_t)blockIdx.x * 170;
    const int tid = threadIdx.x;
    if (tid < 32) {
        const int k = tid >> 3, c = tid & 7;
        const uint32_t v = st[(4 * k) * 8 + c] + st[(4 * k + 1) * 8 + c] + st[(4 * k + 2) * 8 + c] + st[(4 * k + 3) * 8 + c];
        s32[k][c] = v;
        st[138 + k * 8 + c] = v;
    }
    __syncthreads();
    if (tid < 8) s32[4][tid] = s32[0][tid] + s32[1][tid] + s32[2][tid] + s32[3][tid];
    __syncthreads();
    if (tid < 5) {
        uint32_t best = st[128 + tid], mv = st[133 + tid];
        for (int q = 0; q < 8; q++)
            if (s32[tid][q] < best) { best = s32[tid][q]; mv = step_mv(mvs[blockIdx.x], q); }
        st[128 + tid] = best; st[133 + tid] = mv;
    }
}

// ------------------------------------------------------------------------------------------------ variance intermediates
// four horizontally adjacent 8x8 blocks, rows 0 2 4 6 only: mean << 3 and mean of squares << 11 with the reference's scaling
__global__ void __launch_bounds__(64)
interm_var_kernel(const uint8_t* __restrict__ plane, int stride, const int32_t* __restrict__ offs, uint64_t* __restrict__ mean, uint64_t* __restrict__ mean_sq) {
    const int lane = threadIdx.x, blk = lane >> 4, col = (lane >> 1) & 7, half = lane & 1;
    const uint8_t* p = plane + offs[blockIdx.x] + blk * 8 + col;
    uint32_t s = 0, q = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t v = p[(size_t)(2 * (2 * half + k)) * stride];
        s += v; q += v * v;
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { s += __shfl_xor(s, m); q += __shfl_xor(q, m); }
    if ((lane & 15) == 0) { mean[blockIdx.x * 4 + blk] = (uint64_t)s << 3; mean_sq[blockIdx.x * 4 + blk] = (uint64_t)q << 11; }
}

// ------------------------------------------------------------------------------------------------ 64-point coefficient re-pack
// in place on a W x H block of coefficients: energy of everything outside the top-left 32 x 32, zero it, pack the kept rows to stride min(W, 32).
// Every thread reads all it needs before anybody writes, so the final buffer equals the reference's (including the stale middle rows it leaves).
template <int W, int H>
__global__ void __launch_bounds__(1024)
handle_transform64_kernel(int32_t* __restrict__ coeff, uint64_t* __restrict__ energy) {
    constexpr int KW = W > 32 ? 32 : W, KH = H > 32 ? 32 : H, N = W * H, NK = KW * KH, PER = (N + 1023) / 1024;
    __shared__ unsigned long long s_e;
    int32_t* c = coeff + (size_t)blockIdx.x * N;
    const int tid = threadIdx.x;
    if (tid == 0) s_e = 0;
    __syncthreads();
    int32_t nv[PER];
    unsigned long long e = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = tid + k * 1024;
        if (i >= N) { nv[k] = 0; continue; }
        const int r = i / W, col = i % W;
        const int32_t old = c[i];
        const bool dropped = r >= 32 || col >= 32;
        if (dropped) e += (unsigned long long)((int64_t)old * (int64_t)old);
        nv[k] = i < NK ? c[(i / KW) * W + (i % KW)] : (dropped ? 0 : old);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) e += __shfl_xor(e, m);
    if ((tid & 63) == 0 && e) atomicAdd(&s_e, e);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = tid + k * 1024;
        if (i < N) c[i] = nv[k];
    }
    if (tid == 0) energy[blockIdx.x] = s_e;
}

// ------------------------------------------------------------------------------------------------ up-sampled prediction
// two 8-tap passes with a round-to-8-bit clip after each (the libvpx-style convolve8, not the AV1 normative one)
__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__global__ void __launch_bounds__(256)
upsampled_pred_kernel(const uint8_t* __restrict__ ref, int rs, uint8_t* __restrict__ dst, const SvtHipUpsampledBlk* __restrict__ blks) {
    __shared__ uint8_t tmp[(128 + 7) * 128];
    const SvtHipUpsampledBlk b = blks[blockIdx.x];
    const int w = b.w, h = b.h, tid = threadIdx.x;
    const uint8_t* r0 = ref + b.ref_off;
    uint8_t* d = dst + b.dst_off;
    const int16_t* fx = kTaps[b.bank][(b.subpel_x_q3 << 1) & 15];
    const int16_t* fy = kTaps[b.bank][(b.subpel_y_q3 << 1) & 15];
    const int row0 = b.subpel_y_q3 ? -3 : 0, rows = b.subpel_y_q3 ? h + 7 : h;
    for (int i = tid; i < rows * w; i += 256) {
        const int y = i / w, x = i % w;
        const uint8_t* p = r0 + (ptrdiff_t)(y + row0) * rs + x;
        int v;
        if (b.subpel_x_q3) {
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) sum += (int)p[k - 3] * (int)fx[k];
            v = clip8((sum + 64) >> 7);
        } else {
            v = p[0];
        }
        if (b.subpel_y_q3) tmp[y * 128 + x] = (uint8_t)v; else d[y * w + x] = (uint8_t)v;
    }
    if (!b.subpel_y_q3) return;
    __syncthreads();
    for (int i = tid; i < h * w; i += 256) {
        const int y = i / w, x = i % w;
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) sum += (int)tmp[(y + k) * 128 + x] * (int)fy[k];
        d[y * w + x] = (uint8_t)clip8((sum + 64) >> 7);
    }
}


// ---- the lossless 4x4 inverse: svt_av1_highbd_iwht4x4_16_add_c / svt_av1_highbd_iwht4x4_1_add_c (Common/Codec/EbInvTransforms.c:2771-2857), selected by eob like
// highbd_iwht4x4_add (:2858-2864).  One thread per block: 16 coefficients (four 16-byte loads), the reversible Walsh-Hadamard butterfly along rows then columns
// (3.5 adds and half a shift per sample, no multiplications), added to the prediction and clipped to the bit depth.  UNIT_QUANT_SHIFT = 2 (EbInvTransforms.h:23).
__device__ __forceinline__ void iwht4(int& a1, int& b1, int& c1, int& d1) {   // in: (a, c, d, b) as the reference names its inputs; out: the four outputs in order
    a1 += c1; d1 -= b1;
    const int e1 = (a1 - d1) >> 1;
    b1 = e1 - b1; c1 = e1 - c1;
    a1 -= b1; d1 += c1;
}
template <typename PIX>
__global__ void __launch_bounds__(64)
iwht4x4_add_kernel(const int32_t* __restrict__ dq, const uint16_t* __restrict__ eob, const PIX* pred, int ps, PIX* recon, int rs, const uint32_t* __restrict__ descs, int n, int bd) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n) return;
    const uint32_t d = descs[b];
    const int x = d & 0x3FFF, y = (d >> 14) & 0x3FFF, mx = (1 << bd) - 1;
    const PIX* pr = pred + (size_t)y * ps + x; PIX* rc = recon + (size_t)y * rs + x;
    int o[4][4];   // [row][column] of the residual
    if (!eob || eob[b] > 1) {
        const int4* q = (const int4*)(dq + (size_t)b * 16);
#pragma unroll
        for (int i = 0; i < 4; i++) {   // rows: ip[0], ip[1], ip[2], ip[3] play a, c, d, b
            const int4 v = q[i];
            int a1 = v.x >> 2, c1 = v.y >> 2, d1 = v.z >> 2, b1 = v.w >> 2;
            iwht4(a1, b1, c1, d1);
            o[i][0] = a1; o[i][1] = b1; o[i][2] = c1; o[i][3] = d1;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {   // columns
            int a1 = o[0][i], c1 = o[1][i], d1 = o[2][i], b1 = o[3][i];
            iwht4(a1, b1, c1, d1);
            o[0][i] = a1; o[1][i] = b1; o[2][i] = c1; o[3][i] = d1;
        }
    } else {   // DC only
        int a1 = dq[(size_t)b * 16] >> 2;
        int e1 = a1 >> 1;
        a1 -= e1;
        const int t[4] = {a1, e1, e1, e1};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = t[i] >> 1, a = t[i] - e;
            o[0][i] = a; o[1][i] = e; o[2][i] = e; o[3][i] = e;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) rc[(size_t)r * rs + c] = (PIX)min(max((int)pr[(size_t)r * ps + c] + o[r][c], 0), mx);
}

}  // namespace

extern "C" {
int svt_hip_launch_quantize_blocks(hipStream_t st, const int32_t* coeff, int n, int nblk, const SvtHipQuantParams* qp, const int16_t* iscan, int32_t* qcoeff,
                                   int32_t* dqcoeff, uint16_t* eob) {
    if (nblk <= 0) return 0;
    hipLaunchKernelGGL(quantize_blocks_kernel, dim3(nblk), dim3(256), 0, st, coeff, n, *qp, iscan, qcoeff, dqcoeff, eob);
    return (int)hipGetLastError();
}
int svt_hip_launch_residual(hipStream_t st, int pix_bytes, const void* src, int ss, const void* pred, int ps, int16_t* res, int rs, int w, int h) {
    if (w <= 0 || h <= 0) return 0;
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    if (pix_bytes == 1) hipLaunchKernelGGL((residual_kernel<uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)src, ss, (const uint8_t*)pred, ps, res, rs, w, h);
    else hipLaunchKernelGGL((residual_kernel<uint16_t>), grid, dim3(256), 0, st, (const uint16_t*)src, ss, (const uint16_t*)pred, ps, res, rs, w, h);
    return (int)hipGetLastError();
}
int svt_hip_launch_ext_all_sad(hipStream_t st, const uint8_t* src, int ss, const uint8_t* ref, int rs, const void* jobs, int n, uint32_t* state) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(ext_all_sad_kernel, dim3(n), dim3(512), 0, st, src, ss, ref, rs, (const SvtHipExtSadJob*)jobs, state);
    return (int)hipGetLastError();
}
int svt_hip_launch_ext_eight_sad_32_64(hipStream_t st, const uint32_t* mvs, int n, uint32_t* state) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(ext_eight_sad_32_64_kernel, dim3(n), dim3(64), 0, st, mvs, state);
    return (int)hipGetLastError();
}
int svt_hip_launch_interm_var(hipStream_t st, const uint8_t* plane, int stride, const int32_t* offs, int n, uint64_t* mean, uint64_t* mean_sq) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(interm_var_kernel, dim3(n), dim3(64), 0, st, plane, stride, offs, mean, mean_sq);
    return (int)hipGetLastError();
}
int svt_hip_launch_handle_transform64(hipStream_t st, int tx_size, int32_t* coeff, int nblk, uint64_t* energy) {
    if (nblk <= 0) return 0;
    switch (tx_size) {   // TxSize: 4 TX_64X64, 11 TX_32X64, 12 TX_64X32, 17 TX_16X64, 18 TX_64X16
    case 4:  hipLaunchKernelGGL((handle_transform64_kernel<64, 64>), dim3(nblk), dim3(1024), 0, st, coeff, energy); break;
    case 11: hipLaunchKernelGGL((handle_transform64_kernel<32, 64>), dim3(nblk), dim3(1024), 0, st, coeff, energy); break;
    case 12: hipLaunchKernelGGL((handle_transform64_kernel<64, 32>), dim3(nblk), dim3(1024), 0, st, coeff, energy); break;
    case 17: hipLaunchKernelGGL((handle_transform64_kernel<16, 64>), dim3(nblk), dim3(1024), 0, st, coeff, energy); break;
    case 18: hipLaunchKernelGGL((handle_transform64_kernel<64, 16>), dim3(nblk), dim3(1024), 0, st, coeff, energy); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}
int svt_hip_launch_upsampled_pred(hipStream_t st, const uint8_t* ref, int rs, uint8_t* dst, const void* blks, int n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(upsampled_pred_kernel, dim3(n), dim3(256), 0, st, ref, rs, dst, (const SvtHipUpsampledBlk*)blks);
    return (int)hipGetLastError();
}
int svt_hip_launch_iwht4x4_add(hipStream_t st, int pix_bytes, int bd, const int32_t* dq, const uint16_t* eob, const void* pred, int ps, void* recon, int rs, const uint32_t* descs, int n) {
    if (n <= 0) return 0;
    if (pix_bytes == 1) hipLaunchKernelGGL((iwht4x4_add_kernel<uint8_t>), dim3((n + 63) / 64), dim3(64), 0, st, dq, eob, (const uint8_t*)pred, ps, (uint8_t*)recon, rs, descs, n, bd);
    else hipLaunchKernelGGL((iwht4x4_add_kernel<uint16_t>), dim3((n + 63) / 64), dim3(64), 0, st, dq, eob, (const uint16_t*)pred, ps, (uint16_t*)recon, rs, descs, n, bd);
    return (int)hipGetLastError();
}
}

SVT_HIP_TU_PROBE(percall)
