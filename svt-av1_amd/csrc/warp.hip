// warp.hip — AV1 warped (affine) prediction for a list of blocks in one launch: single reference, and both halves of a compound prediction
// (first reference into the 16-bit compound buffer, second reference averaged / distance-weighted with it); gfx950.  SURVEY 8(f) rank 4.
//
// Replaces (file:line under /root/reference/Source/Lib): Common/Codec/EbWarpedMotion.c:577-694 svt_av1_warp_affine_c and :733-842
// svt_av1_highbd_warp_affine_c (common_dsp_rtcd.h), the non-compound path svt_warp_plane / svt_highbd_warp_plane take for local-warp and
// global-motion blocks, and their is_compound branches (:660-683, :812-835).  One workgroup per block; each of its 4 waves takes 8x8 sub-blocks in turn: 15 x 8 horizontally filtered
// samples go through a per-wave LDS tile, then the 64 lanes produce the 8 x 8 outputs.  Both passes pick their 8-tap kernel per sample
// from Warped_Filters (LDS copy: the index diverges across lanes), with the reference's rounding at every step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"
#include "warp_filter_table.h"

namespace {

__device__ const int16_t kWarpedFilter[193][8] = SVT_WARPED_FILTER_TABLE;

__device__ __forceinline__ int rp2(int v, int n) { return (v + ((1 << n) >> 1)) >> n; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// COMP: blks is a SvtHipWarpCompBlk list; round_1 = COMPOUND_ROUND1_BITS = 7 (av1 get_conv_params_no_round), DIST_PRECISION_BITS = 4
template <typename PIX, int BD, bool COMP>
__global__ void __launch_bounds__(256)
warp_predict_kernel(const PIX* __restrict__ ref, int width, int height, int stride, PIX* __restrict__ dst, int dst_stride, int ss_x, int ss_y,
                    const void* __restrict__ blks_v, uint16_t* __restrict__ convbuf) {
    constexpr int round_0 = BD == 12 ? 5 : 3, round_1 = 7;
    constexpr int extra = BD + 7 - round_0 - 14;
    constexpr int rbh = sizeof(PIX) == 1 ? round_0 : round_0 + (extra > 0 ? extra : 0);
    constexpr int rbv = COMP ? round_1 : 14 - rbh, obh = BD + 6, obv = BD + 14 - rbh;
    constexpr int round_bits = 14 - round_0 - round_1, offset_bits = BD + 14 - round_0;
    __shared__ int16_t filt[193][8];
    __shared__ int tmp[4][15 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 193 * 8; i += 256) (&filt[0][0])[i] = (&kWarpedFilter[0][0])[i];
    __syncthreads();
    const SvtHipWarpBlk b = COMP ? ((const SvtHipWarpCompBlk*)blks_v)[blockIdx.x].blk : ((const SvtHipWarpBlk*)blks_v)[blockIdx.x];
    SvtHipWarpCompBlk cb = {};
    if (COMP) cb = ((const SvtHipWarpCompBlk*)blks_v)[blockIdx.x];
    const int nbx = (b.p_width + 7) >> 3, nby = (b.p_height + 7) >> 3;
    for (int sb = wave; sb < nbx * nby; sb += 4) {
        const int j = b.p_col + 8 * (sb % nbx), i = b.p_row + 8 * (sb / nbx);
        const int src_x = (j + 4) << ss_x, src_y = (i + 4) << ss_y;
        const int dst_x = b.mat[2] * src_x + b.mat[3] * src_y + b.mat[0], dst_y = b.mat[4] * src_x + b.mat[5] * src_y + b.mat[1];
        const int x4 = dst_x >> ss_x, y4 = dst_y >> ss_y;
        const int ix4 = x4 >> 16, iy4 = y4 >> 16;
        int sx4 = x4 & 0xffff, sy4 = y4 & 0xffff;
        sx4 += b.alpha * (-4) + b.beta * (-4); sy4 += b.gamma * (-4) + b.delta * (-4);
        sx4 &= ~63; sy4 &= ~63;
        // horizontal: 15 rows x 8 columns = 120 samples, two rounds of 64 lanes
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int idx = lane + 64 * r;
            if (idx < 120) {
                const int k = (idx >> 3) - 7, l = (idx & 7) - 4;
                const int iy = clampi(iy4 + k, 0, height - 1);
                const int sx = sx4 + b.beta * (k + 4) + b.alpha * (l + 4);
                const int16_t* c = filt[rp2(sx, 10) + 64];
                const int ix = ix4 + l - 3;
                const PIX* row = ref + (ptrdiff_t)iy * stride;
                int sum = 1 << obh;
#pragma unroll
                for (int m = 0; m < 8; m++) sum += (int)row[clampi(ix + m, 0, width - 1)] * c[m];
                tmp[wave][idx] = rp2(sum, rbh);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed
        {
            const int k = (lane >> 3) - 4, l = (lane & 7) - 4;
            if (k < b.p_row + b.p_height - i - 4 && l < b.p_col + b.p_width - j - 4) {
                const int sy = sy4 + b.delta * (k + 4) + b.gamma * (l + 4);
                const int16_t* c = filt[rp2(sy, 10) + 64];
                int sum = 1 << obv;
#pragma unroll
                for (int m = 0; m < 8; m++) sum += tmp[wave][(k + m + 4) * 8 + (l + 4)] * c[m];
                if (COMP) {
                    sum = rp2(sum, rbv);
                    uint16_t* p = convbuf + cb.cb_off + (ptrdiff_t)(i - b.p_row + k + 4) * cb.cb_stride + (j - b.p_col + l + 4);
                    if (cb.do_average) {
                        int t = *p;
                        t = cb.use_jnt_comp_avg ? (t * cb.fwd_offset + sum * cb.bck_offset) >> 4 : (t + sum) >> 1;
                        t -= (1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1));
                        dst[(ptrdiff_t)(i + k + 4) * dst_stride + (j + l + 4)] = (PIX)clampi(rp2(t, round_bits), 0, (1 << BD) - 1);
                    } else {
                        *p = (uint16_t)sum;
                    }
                } else {
                    sum = rp2(sum, rbv) - (1 << (BD - 1)) - (1 << BD);
                    dst[(ptrdiff_t)(i + k + 4) * dst_stride + (j + l + 4)] = (PIX)clampi(sum, 0, (1 << BD) - 1);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();      // the tile is rewritten by the next sub-block
    }
}

}  // namespace

extern "C" int svt_hip_launch_warp_predict(hipStream_t st, int pix_bytes, int bd, const void* ref, int width, int height, int stride, void* dst, int dst_stride,
                                           int ss_x, int ss_y, const SvtHipWarpBlk* blks, int n) {
    if (n <= 0) return 0;
#define LAUNCH(P, B) hipLaunchKernelGGL((warp_predict_kernel<P, B, false>), dim3(n), dim3(256), 0, st, (const P*)ref, width, height, stride, (P*)dst, dst_stride, ss_x, ss_y, (const void*)blks, (uint16_t*)nullptr)
    if (pix_bytes == 1) LAUNCH(uint8_t, 8);
    else if (bd == 8) LAUNCH(uint16_t, 8);
    else if (bd == 10) LAUNCH(uint16_t, 10);
    else LAUNCH(uint16_t, 12);
#undef LAUNCH
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_warp_compound(hipStream_t st, int pix_bytes, int bd, const void* ref, int width, int height, int stride, void* dst, int dst_stride,
                                            int ss_x, int ss_y, uint16_t* convbuf, const SvtHipWarpCompBlk* blks, int n) {
    if (n <= 0) return 0;
#define LAUNCH(P, B) hipLaunchKernelGGL((warp_predict_kernel<P, B, true>), dim3(n), dim3(256), 0, st, (const P*)ref, width, height, stride, (P*)dst, dst_stride, ss_x, ss_y, (const void*)blks, convbuf)
    if (pix_bytes == 1) LAUNCH(uint8_t, 8);
    else if (bd == 8) LAUNCH(uint16_t, 8);
    else if (bd == 10) LAUNCH(uint16_t, 10);
    else LAUNCH(uint16_t, 12);
#undef LAUNCH
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(warp)
