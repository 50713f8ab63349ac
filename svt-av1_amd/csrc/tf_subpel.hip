// tf_subpel.hip — the sub-pel stage of the alt-ref temporal filter on the device; gfx950.
//
// Replaces, for every (64x64 block, window frame) pair of a TF segment (file:line under /root/reference/Source/Lib/Encoder/Codec):
//   EbTemporalFiltering.c:1469-1766  tf_32x32_sub_pel_search   three rounds (1/2, 1/4, 1/8 pel when tf_hp) of nine candidates per 32x32 block
//   EbTemporalFiltering.c:1133-1467  tf_16x16_sub_pel_search   the same per 16x16 block of the 32x32 blocks whose error reaches tf_block_32x32_16x16_th
//   EbTemporalFiltering.c:284-324    derive_tf_32x32_block_split_flag
//   EbTemporalFiltering.c:1768-1941  tf_inter_prediction       MULTITAP_SHARP prediction of luma (+ 4:2:0 chroma) with the chosen vectors
// Each candidate is av1_inter_prediction's single-reference, unscaled path (EbEncInterPrediction.c:4040 -> enc_make_inter_predictor :3663 ->
// compute_subpel_params :3593 -> clamp_mv_to_umv_border_sb :24 -> convolve[sx != 0][sy != 0][0], round_0 = 3, round_1 = 11) followed by
// svt_aom_variance{W}x{H} (C_DEFAULT/EbComputeVariance_C.c:54) or variance_highbd (:34; 32-bit sums, wrapping like the C) against the central picture.
// One workgroup (nine waves) owns one 32x32 quadrant of one pair from the first half-pel candidate to the final predictor: wave c evaluates candidate c of a
// round in its own LDS region (window -> horizontal pass -> 16-bit intermediate -> vertical pass, every rounding step of the reference kept; the source samples
// of a lane stay in registers), the rounds are sequential inside the workgroup — 3 + 12 of them instead of 27 + 108 candidate evaluations one after the other —,
// no host round trip between rounds, and the predictors are written straight into the planes svt_hip_tf_filter_frame_dev reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"
#include "lds_stage.h"
#include "interp_kernels.h"

namespace {

__device__ __forceinline__ int rp2(int v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct TfSubpelArgs {
    const void* src[3]; const void* ref[3]; void* pred[3];
    int src_stride[3], ref_stride[3], pred_stride[3];
    int mi_cols, mi_rows, tf_hp, tf_chroma;
    unsigned long long th16;
    const SvtHipTfSubpelBlk* jobs; SvtHipTfBlk64* blocks;
};

constexpr int kWaves = 9;   // one wave per candidate of a round
struct WaveLds {
    short src[(32 + 7) * 40];   // the candidate's window, rows -3 .. bs + 3, row stride bs + 8
    short im[(32 + 7) * 32];    // horizontal pass
};
struct Lds {
    WaveLds w[kWaves];
    unsigned dist[kWaves];
    short win[(32 + 9) * 44];   // the window the nine candidates of a search round share: rows / columns -4 .. bs + 4 around the round's centre position, row stride bs + 12
};

// compute_subpel_params (EbEncInterPrediction.c:3593-3660), unscaled branch: the vector (1/8 pel, luma units) of a bw x bh block of the plane with
// subsampling ss whose top-left luma sample is (px, py) -> integer position in the plane and q4 phase.  mb_to_*_edge as the TF callers set them
// (EbTemporalFiltering.c:1214-1223: the bottom edge from the block WIDTH in mi, the right edge from its height — the blocks are square).
__device__ __forceinline__ void subpel_params(int mvx, int mvy, int px, int py, int bs_luma, int bw, int bh, int ss, int mi_cols, int mi_rows, int pre_x, int pre_y,
                                              int& pos_x, int& pos_y, int& sx, int& sy) {
    const int micol = px >> 2, mirow = py >> 2, mi = bs_luma >> 2;
    const int to_left = -(micol * 4 * 8), to_right = (mi_cols - mi - micol) * 4 * 8, to_top = -(mirow * 4 * 8), to_bottom = (mi_rows - mi - mirow) * 4 * 8;
    const int spel_left = (4 + bw) << 4, spel_right = spel_left - 16, spel_top = (4 + bh) << 4, spel_bottom = spel_top - 16;
    const int sh = 1 - ss;
    int col = (short)(mvx * (1 << sh)), row = (short)(mvy * (1 << sh));   // MV fields are int16
    col = (short)clampi(col, to_left * (1 << sh) - spel_left, to_right * (1 << sh) + spel_right);
    row = (short)clampi(row, to_top * (1 << sh) - spel_top, to_bottom * (1 << sh) + spel_bottom);
    sx = col & 15; sy = row & 15;
    pos_x = ((pre_x << 4) + col) >> 4; pos_y = ((pre_y << 4) + row) >> 4;
}

// One bw x bh prediction (bw, bh <= 32) by ONE wave in its own LDS region: lane l owns outputs l, l + 64, ... (raster), out[u] receives them.  `on` = this wave
// has a block to predict; every wave of the workgroup passes the two barriers whatever it does in between (the candidates of a round differ in their phases).
// the two passes on a staged window: W[r * ws + c] = reference sample (pos_x - 3 + c, pos_y - 3 + r); the horizontal pass goes through the wave's own L.im (one wave writes and
// reads it: LDS operations of a wave execute in order, no workgroup barrier between the passes)
template <typename PIX, int BD>
__device__ void predict_core(WaveLds& L, const short* __restrict__ W, int ws, bool on, int sx, int sy, int bank, int bw, int bh, int out[16]);
template <typename PIX, int BD>
__device__ void predict(WaveLds& L, bool on, const PIX* __restrict__ ref, int ref_stride, int pos_x, int pos_y, int sx, int sy, int bank, int bw, int bh, int out[16]) {
    const int lane = threadIdx.x & 63, ws = bw + 8;
    __syncthreads();   // the previous users of the LDS windows are done
    if (on)   // (eight loads in flight per lane through batched_stage measured SLOWER here: 165 -> 183 us per launch in the hooked 4K encode — the windows are small and nine waves stage at once)
        for (int i = lane; i < (bh + 7) * (bw + 7); i += 64) {
            const int r = i / (bw + 7), c = i - r * (bw + 7);
            L.src[r * ws + c] = (short)ref[(ptrdiff_t)(pos_y + r - 3) * ref_stride + (pos_x + c - 3)];
        }
    __syncthreads();
    predict_core<PIX, BD>(L, L.src, ws, on, sx, sy, bank, bw, bh, out);
}
template <typename PIX, int BD>
__device__ void predict_core(WaveLds& L, const short* __restrict__ W, int ws, bool on, int sx, int sy, int bank, int bw, int bh, int out[16]) {
    const int lane = threadIdx.x & 63;
    constexpr int pix_max = (1 << BD) - 1;
    int xf[8], yf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { xf[k] = kInterp[bank][sx][k]; yf[k] = kInterp[bank][sy][k]; }
    const int n = bw * bh;
    if (on && sx && sy)
        for (int i = lane; i < (bh + 7) * bw; i += 64) {   // horizontal pass over bh + 7 rows (im_block, EbInterPrediction.c:366-374)
            const int r = i / bw, c = i - r * bw;
            int sum = 1 << (BD + 6);
#pragma unroll
            for (int k = 0; k < 8; k++) sum += xf[k] * W[r * ws + c + k];
            L.im[i] = (short)rp2(sum, 3);
        }
    // the wave's own intermediate: written and read by the same wave, whose LDS operations execute in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int i = lane + 64 * u;
        int o = 0;
        if (on && i < n) {
            const int y = i / bw, x = i - y * bw;
            if (!sx && !sy) o = W[(y + 3) * ws + x + 3];
            else if (!sy) {
                int res = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) res += xf[k] * W[(y + 3) * ws + x + k];
                o = clampi(rp2(rp2(res, 3), 4), 0, pix_max);   // x_sr: round_0, then FILTER_BITS - round_0 (:425-453)
            } else if (!sx) {
                int res = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) res += yf[k] * W[(y + k) * ws + x + 3];
                o = clampi(rp2(res, 7), 0, pix_max);            // y_sr (:395-423)
            } else {
                constexpr int offset_bits = BD + 14 - 3;
                int sum = 1 << offset_bits;
#pragma unroll
                for (int k = 0; k < 8; k++) sum += yf[k] * L.im[(y + k) * bw + x];
                int res = rp2(sum, 11) - ((1 << (offset_bits - 11)) + (1 << (offset_bits - 12)));
                if (sizeof(PIX) == 1) res = (short)res;
                o = clampi(res, 0, pix_max);                    // bits = 0 (:376-392)
            }
        }
        out[u] = o;
    }
}

// the three rounds of one bs x bs block at luma position (px, py) (picture) / (lx, ly) (central and predictor planes); word = the open-loop ME vector.
// Wave c of the workgroup evaluates candidate c of the round (i outer, j inner: the reference's order); the first strictly smaller distortion wins.
template <typename PIX, int BD>
__device__ void search(Lds& L, const TfSubpelArgs& a, int bs, int px, int py, int lx, int ly, uint32_t word, int& best_x, int& best_y, unsigned long long& best_err) {
    const PIX* __restrict__ src = (const PIX*)a.src[0];
    const PIX* __restrict__ ref = (const PIX*)a.ref[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = bs * bs;
    int s[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int i = min(lane + 64 * u, n - 1), y = i / bs, x = i - y * bs;   // unconditional loads (a load under a condition is a branch that waits for it)
        s[u] = (int)src[(ptrdiff_t)(ly + y) * a.src_stride[0] + lx + x];
    }
    short mv_x = (short)((short)(word & 0xffff) << 1), mv_y = (short)((short)(word >> 16) << 1);   // AV1 vectors are 1/8 pel (:1225-1232)
    short bx = mv_x, by = mv_y;
    unsigned long long berr = 0x7fffffffull;   // INT_MAX
    for (int round = 0; round < (a.tf_hp ? 3 : 2); round++) {
        const int step = 4 >> round;
        const short cx = (short)(mv_x + (wave / 3 - 1) * step), cy = (short)(mv_y + (wave % 3 - 1) * step);
        int pos_x, pos_y, sx, sy, o[16];
        subpel_params(cx, cy, px, py, bs, bs, bs, 0, a.mi_cols, a.mi_rows, px, py, pos_x, pos_y, sx, sy);
        // The nine candidates of a round lie within one sample of the round's centre: the workgroup stages ONE window (rows / columns -4 .. bs + 4 around the centre's integer
        // position) instead of nine overlapping ones, and every wave filters from it at its own offset.  (The vector clamp of compute_subpel_params can move a candidate
        // further at the picture's edge: such a wave stages a window of its own afterwards.)
        int bpx, bpy, bsx, bsy;
        subpel_params(mv_x, mv_y, px, py, bs, bs, bs, 0, a.mi_cols, a.mi_rows, px, py, bpx, bpy, bsx, bsy);
        constexpr int wws = 44;
        __syncthreads();   // the previous round's readers are done
        for (int i = threadIdx.x; i < (bs + 9) * (bs + 9); i += 64 * kWaves) {
            const int r = i / (bs + 9), c = i - r * (bs + 9);
            L.win[r * wws + c] = (short)ref[(ptrdiff_t)(bpy + r - 4) * a.ref_stride[0] + (bpx + c - 4)];
        }
        const int ddx = pos_x - bpx, ddy = pos_y - bpy;
        const bool near = ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1;   // wave-uniform
        if (!near)
            for (int i = lane; i < (bs + 7) * (bs + 7); i += 64) {
                const int r = i / (bs + 7), c = i - r * (bs + 7);
                L.w[wave].src[r * (bs + 8) + c] = (short)ref[(ptrdiff_t)(pos_y + r - 3) * a.ref_stride[0] + (pos_x + c - 3)];
            }
        __syncthreads();
        // (One horizontal pass per COLUMN of the 3 x 3 round — its three candidates share position and phase — made by the column's three waves together was measured as well:
        // the barrier it needs between the passes costs what the saved multiplications gain, 145 -> ~150 us per launch; not kept.)
        if (near) predict_core<PIX, BD>(L.w[wave], L.win + (ddy + 1) * wws + (ddx + 1), wws, true, sx, sy, 0, bs, bs, o);
        else predict_core<PIX, BD>(L.w[wave], L.w[wave].src, bs + 8, true, sx, sy, 0, bs, bs, o);
        int sum = 0, sse = 0;   // sse < 2^31: 1024 x 1023^2
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (lane + 64 * u < n) { const int d = o[u] - s[u]; sum += d; sse += d * d; }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { sum += __shfl_xor(sum, m, 64); sse += __shfl_xor(sse, m, 64); }
        uint32_t dist;
        if (sizeof(PIX) == 1) dist = (uint32_t)sse - (uint32_t)(((long long)sum * sum) / n);
        else dist = (uint32_t)sse - (uint32_t)((int)((unsigned)sum * (unsigned)sum) / n);   // variance_highbd_c: int arithmetic
        if (lane == 0) L.dist[wave] = dist;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kWaves; c++) {
            const unsigned long long d = L.dist[c];
            if (d < berr) { berr = d; bx = (short)(mv_x + (c / 3 - 1) * step); by = (short)(mv_y + (c % 3 - 1) * step); }
        }
        mv_x = bx; mv_y = by;
    }
    best_x = bx; best_y = by; best_err = berr;
}

// tf_inter_prediction: the workgroup's waves share the quadrant's final predictions — job j of `njobs` luma blocks (bs x bs at (ox[j], oy[j]) inside the 64x64
// block) goes to wave j, its two chroma blocks (tf_chroma, 4:2:0) to waves njobs + 2 j and njobs + 2 j + 1; jobs beyond the ninth wave take a second turn.
template <typename PIX, int BD>
__device__ void final_predict(Lds& L, const TfSubpelArgs& a, const SvtHipTfSubpelBlk& J, int bs, int njobs, const int* ox, const int* oy, const int* mvx, const int* mvy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = njobs * (a.tf_chroma ? 3 : 1);
    for (int base = 0; base < total; base += kWaves) {
        const int  t = base + wave;
        const bool on = t < total;
        const int  j = !on ? 0 : (t < njobs ? t : (t - njobs) >> 1), plane = !on || t < njobs ? 0 : 1 + ((t - njobs) & 1);
        const int  px = J.x + ox[j], py = J.y + oy[j], lx = J.dst_x + ox[j], ly = J.dst_y + oy[j];
        const int  bw = plane ? bs >> 1 : bs;
        const int  pre_x = plane ? ((px >> 3) << 3) / 2 : px, pre_y = plane ? ((py >> 3) << 3) / 2 : py;
        const int  dx = plane ? ((lx >> 3) << 3) / 2 : lx, dy = plane ? ((ly >> 3) << 3) / 2 : ly;
        int pos_x, pos_y, sx, sy, o[16];
        subpel_params(mvx[j], mvy[j], px, py, bs, bw, bw, plane ? 1 : 0, a.mi_cols, a.mi_rows, pre_x, pre_y, pos_x, pos_y, sx, sy);
        predict<PIX, BD>(L.w[wave], on, (const PIX*)a.ref[plane], a.ref_stride[plane], pos_x, pos_y, sx, sy, 2, bw, bw, o);   // MULTITAP_SHARP
        if (on) {
            PIX* __restrict__ d = (PIX*)a.pred[plane];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int i = lane + 64 * u, y = i / bw, x = i - y * bw;
                if (i < bw * bw) d[(ptrdiff_t)(dy + y) * a.pred_stride[plane] + dx + x] = (PIX)o[u];
            }
        }
    }
}

// grid: 4 x n_jobs (quadrant = blockIdx.x & 3); block 9 waves
template <typename PIX, int BD>
__global__ void __launch_bounds__(64 * kWaves)
tf_subpel_kernel(const TfSubpelArgs a) {
    __shared__ Lds L;
    const SvtHipTfSubpelBlk J = a.jobs[blockIdx.x >> 2];
    const int q = blockIdx.x & 3, qx = q & 1, qy = q >> 1;
    SvtHipTfBlk64* __restrict__ B = a.blocks + J.blk_index;
    int mv32x, mv32y; unsigned long long err32;
    search<PIX, BD>(L, a, 32, J.x + 32 * qx, J.y + 32 * qy, J.dst_x + 32 * qx, J.dst_y + 32 * qy, J.mv32[q], mv32x, mv32y, err32);
    const bool do16 = !(err32 < a.th16);   // tf_16x16_search_do (:1190-1193)
    int mv16x[4] = {0, 0, 0, 0}, mv16y[4] = {0, 0, 0, 0}; unsigned long long err16[4] = {0, 0, 0, 0};
    int ox[4], oy[4];
    for (int k = 0; k < 4; k++) { ox[k] = 32 * qx + 16 * (k & 1); oy[k] = 32 * qy + 16 * (k >> 1); }   // z-order inside the quadrant = index_16x16_from_subindexes / tab16x16 (:54, EbMotionEstimation.h:108)
    int split = 0;
    if (do16) {
        for (int k = 0; k < 4; k++) search<PIX, BD>(L, a, 16, J.x + ox[k], J.y + oy[k], J.dst_x + ox[k], J.dst_y + oy[k], J.mv16[4 * q + k], mv16x[k], mv16y[k], err16[k]);
        // derive_tf_32x32_block_split_flag (:284-324), int arithmetic
        const int block_error = (int)err32;
        int mn = 0x7fffffff, mx = (int)0x80000000, sum = 0;
        for (int k = 0; k < 4; k++) { const int e = (int)err16[k]; sum += e; mn = e < mn ? e : mn; mx = e > mx ? e : mx; }
        const bool no_split = ((block_error * 15 < sum * 16) && mx - mn < 12000) || ((block_error * 14 < sum * 16) && mx - mn < 6000);
        split = no_split ? 0 : 1;
    }
    if (threadIdx.x == 0) {
        B->mv32_x[q] = (int16_t)mv32x; B->mv32_y[q] = (int16_t)mv32y; B->err32[q] = err32; B->split[q] = split;
        for (int k = 0; k < 4; k++) { B->mv16_x[4 * q + k] = (int16_t)mv16x[k]; B->mv16_y[4 * q + k] = (int16_t)mv16y[k]; B->err16[4 * q + k] = err16[k]; }
    }
    if (split) final_predict<PIX, BD>(L, a, J, 16, 4, ox, oy, mv16x, mv16y);
    else {
        const int o32x = 32 * qx, o32y = 32 * qy;
        final_predict<PIX, BD>(L, a, J, 32, 1, &o32x, &o32y, &mv32x, &mv32y);
    }
}

}  // namespace

extern "C" int svt_hip_launch_tf_subpel(hipStream_t st, int pix_bytes, int bd, const void* const src[3], const int src_stride[3], const void* const ref[3],
                                        const int ref_stride[3], void* const pred[3], const int pred_stride[3], int mi_cols, int mi_rows, uint64_t th16,
                                        int tf_hp, int tf_chroma, const SvtHipTfSubpelBlk* jobs, int n_jobs, SvtHipTfBlk64* blocks) {
    if (n_jobs <= 0) return 0;
    TfSubpelArgs a;
    for (int p = 0; p < 3; p++) {
        a.src[p] = src[p]; a.ref[p] = ref[p]; a.pred[p] = pred[p];
        a.src_stride[p] = src_stride[p]; a.ref_stride[p] = ref_stride[p]; a.pred_stride[p] = pred_stride[p];
    }
    a.mi_cols = mi_cols; a.mi_rows = mi_rows; a.tf_hp = tf_hp; a.tf_chroma = tf_chroma; a.th16 = th16; a.jobs = jobs; a.blocks = blocks;
    const dim3 grid(4 * n_jobs), block(64 * kWaves);
    if (pix_bytes == 1) hipLaunchKernelGGL((tf_subpel_kernel<uint8_t, 8>), grid, block, 0, st, a);
    else if (bd == 8) hipLaunchKernelGGL((tf_subpel_kernel<uint16_t, 8>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((tf_subpel_kernel<uint16_t, 10>), grid, block, 0, st, a);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(tf_subpel)
