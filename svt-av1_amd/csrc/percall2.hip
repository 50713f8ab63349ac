// percall2.hip — the remaining stand-alone forms behind the per-call table of include/svt_hip_rtcd.h: the small helpers of the reference's
// dispatch table on this path whose work is otherwise fused into the frame kernels (picture-analysis block means, the single-candidate
// SAD ladders, CDEF's distortion of one filter block and its strength-pair selection step, the self-guided projection on materialised
// flt0 / flt1 planes, the 8-tap scaled convolution of svt_aom_upsampled_pred and the Wiener convolution of one stripe).
// Bit-exact restatements of (paths under /root/reference/Source/Lib):
//   block_mean            Encoder/Codec/EbPictureAnalysisProcess.c:287-326   svt_compute_mean_squared_values_c, svt_compute_sub_mean_8x8_c
//   ext_sad_16 / _32_64   Encoder/Codec/EbMotionEstimation.c:122-228         svt_ext_sad_calculation_8x8_16x16_c, _32x32_64x64_c
//   cdef_dist             Encoder/Codec/EbEncCdef.c:25-220                   compute_cdef_dist_c / _8bit_c
//   search_one_dual       Encoder/Codec/EbEncCdef.c:1070-1118                svt_search_one_dual_c
//   sgr_flt_proj          Encoder/Codec/EbRestorationPick.c:174-316, :448-538  svt_av1_{lowbd,highbd}_pixel_proj_error_c, svt_get_proj_subspace_c
//   convolve8             Common/Codec/convolve.c:249-307                    svt_aom_convolve8_horiz_c / _vert_c
//   wiener_convolve       Common/Codec/convolve.c:57-242                     svt_av1_[highbd_]wiener_convolve_add_src_c
//   repack64              Encoder/Codec/EbTransforms.c:2933-2969             handle_transform*_N2_N4_c
//   jnt_convolve          Common/Codec/EbInterPrediction.c:552-741, :868-1143  svt_av1_[highbd_]jnt_convolve_{2d,x,y,2d_copy}_c
//   diffwtd_mask          Common/Codec/EbInterPrediction.c:78-175; Common/C_DEFAULT/EbInterPrediction_c.c:15-45  svt_av1_build_compound_diffwtd_mask[_highbd|_d16]_c
//   blend_d16             Common/Codec/EbBlend_a64_mask.c:34-215            svt_aom_{lowbd,highbd}_blend_a64_d16_mask_c
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstddef>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "svt_hip_internal.h"

namespace {

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
        v += ((unsigned long long)(unsigned)__shfl_xor((int)(v >> 32), m, 64) << 32) | (unsigned)__shfl_xor((int)v, m, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) { return (long long)wave_sum_u64((unsigned long long)v); }
__device__ __forceinline__ int clip_px(int v, int bd) { const int mx = (1 << bd) - 1; return v < 0 ? 0 : (v > mx ? mx : v); }

// ------------------------------------------------------------------------------------------------ picture-analysis block means
// mode 0: (sum of squares << 16) / (w * h) over a w x h area; mode 1: sum over rows 0, 2, 4, 6 of an 8 x 8 block, << 3.  One wave per block.
__global__ void __launch_bounds__(64)
block_mean_kernel(const uint8_t* __restrict__ plane, int stride, const int32_t* __restrict__ offs, int mode, int w, int h, uint64_t* __restrict__ out) {
    const uint8_t* p = plane + offs[blockIdx.x];
    unsigned long long acc = 0;
    if (mode == 0) {
        for (int i = threadIdx.x; i < w * h; i += 64) { const int y = i / w, x = i - y * w; const unsigned v = p[(size_t)y * stride + x]; acc += v * v; }
    } else if (threadIdx.x < 32) {
        const int y = 2 * (threadIdx.x >> 3), x = threadIdx.x & 7;
        acc = p[(size_t)y * stride + x];
    }
    acc = wave_sum_u64(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = mode == 0 ? (acc << 16) / (unsigned long long)(w * h) : acc << 3;
}

// ------------------------------------------------------------------------------------------------ single-candidate SAD ladders
// state of one 16x16 job (uint32): best_sad8x8[4] best_sad16x16 best_mv8x8[4] best_mv16x16 | sad16x16 sad8x8[4]   (15 words, the last five are outputs)
__global__ void __launch_bounds__(64)
ext_sad_16_kernel(const uint8_t* __restrict__ src, int ss, const uint8_t* __restrict__ ref, int rs, const SvtHipExtSadJob* __restrict__ jobs, uint32_t* __restrict__ state) {
    const SvtHipExtSadJob j = jobs[blockIdx.x];
    const uint8_t *s = src + j.src_off, *r = ref + j.ref_off;
    const int      lane = threadIdx.x, q = lane >> 4, l16 = lane & 15;   // quadrant q: 8x8 block (q >> 1, q & 1); 16 lanes x 4 pixels
    const int      qy = 8 * (q >> 1), qx = 8 * (q & 1);
    unsigned       acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = l16 * 4 + k, y = p >> 3, x = p & 7;
        if (!j.sub_sad || !(y & 1)) acc += (unsigned)abs((int)s[(size_t)(qy + y) * ss + qx + x] - (int)r[(size_t)(qy + y) * rs + qx + x]);
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) acc += (unsigned)__shfl_xor((int)acc, m, 64);
    if (j.sub_sad) acc <<= 1;   // compute8x4_sad_kernel on every other row, doubled
    const unsigned s0 = (unsigned)__shfl((int)acc, 0, 64), s1 = (unsigned)__shfl((int)acc, 16, 64), s2 = (unsigned)__shfl((int)acc, 32, 64), s3 = (unsigned)__shfl((int)acc, 48, 64);
    if (lane == 0) {
        uint32_t*      st = state + (size_t)blockIdx.x * 15;
        const unsigned sad8[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            st[11 + k] = sad8[k];
            if (sad8[k] < st[k]) { st[k] = sad8[k]; st[5 + k] = j.mv; }
        }
        const unsigned sad16 = s0 + s1 + s2 + s3;
        if (sad16 < st[4]) { st[4] = sad16; st[9] = j.mv; }
        st[10] = sad16;
    }
}
// state of one 64x64 job (uint32): sad16x16[16] (input) best_sad32x32[4] best_sad64x64 best_mv32x32[4] best_mv64x64 | sad32x32[4]   (30 words)
__global__ void __launch_bounds__(64)
ext_sad_32_64_kernel(uint32_t* __restrict__ state, const uint32_t* __restrict__ mv, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t* st = state + (size_t)i * 30;
    uint32_t  s64 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t s32 = st[4 * k] + st[4 * k + 1] + st[4 * k + 2] + st[4 * k + 3];
        st[26 + k] = s32;
        if (s32 < st[16 + k]) { st[16 + k] = s32; st[21 + k] = mv[i]; }
        s64 += s32;
    }
    if (s64 < st[20]) { st[20] = s64; st[25] = mv[i]; }
}

// ------------------------------------------------------------------------------------------------ CDEF: distortion of one filter block
// dst: the picture plane (source pixels), src: the filtered blocks packed one after the other (bw x bh each) — the reference's argument names.
template <typename PIX>
__global__ void __launch_bounds__(256)
cdef_dist_kernel(const PIX* __restrict__ dst, int dstride, const PIX* __restrict__ src, const uint8_t* __restrict__ list, int n, int bw_log2, int bh_log2, int cs,
                 int pli, uint64_t* __restrict__ out) {
    __shared__ unsigned long long part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bw = 1 << bw_log2, npx = 1 << (bw_log2 + bh_log2);
    unsigned long long acc = 0;
    for (int bi = wave; bi < n; bi += 4) {
        const int by = list[3 * bi], bx = list[3 * bi + 1];   // CdefList {by, bx, skip}
        const bool on = lane < npx;
        const int  i = lane >> bw_log2, jx = lane & (bw - 1);
        const unsigned s = on ? src[((size_t)bi << (bw_log2 + bh_log2)) + lane] : 0u;
        const unsigned d = on ? dst[(size_t)((by << bh_log2) + i) * dstride + (bx << bw_log2) + jx] : 0u;
        if (pli == 0 && bw_log2 == 3 && bh_log2 == 3) {   // dist_8x8_*: the perceptual luma metric
            const unsigned long long sum_s = wave_sum_u64(s), sum_d = wave_sum_u64(d), sum_s2 = wave_sum_u64((unsigned long long)s * s),
                                     sum_d2 = wave_sum_u64((unsigned long long)d * d), sum_sd = wave_sum_u64((unsigned long long)s * d);
            const uint64_t svar = sum_s2 - ((sum_s * sum_s + 32) >> 6);
            const uint64_t dvar = sum_d2 - ((sum_d * sum_d + 32) >> 6);
            const double   num = (double)(sum_d2 + sum_s2 - 2 * sum_sd) * .5 * (double)(svar + dvar + (uint64_t)(400 << 2 * cs));
            const double   den = sqrt((double)(20000 << 4 * cs) + (double)svar * (double)dvar);
            acc += (unsigned long long)floor(.5 + num / den);
        } else {
            const int e = (int)d - (int)s;
            acc += wave_sum_u64((unsigned long long)(unsigned)(e * e));
        }
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (part[0] + part[1] + part[2] + part[3]) >> (2 * cs);
}

// ------------------------------------------------------------------------------------------------ CDEF: one step of the strength-pair selection
__global__ void __launch_bounds__(256)
one_dual_best_kernel(const uint64_t* __restrict__ mse0, const uint64_t* __restrict__ mse1, int sb_count, const int* __restrict__ lev0, const int* __restrict__ lev1,
                     int nb, uint64_t* __restrict__ best, uint64_t* __restrict__ tot, int n_tot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    for (int t = i; t < n_tot; t += gridDim.x * 256) tot[t] = 0;   // the totals this step accumulates into
    if (i >= sb_count) return;
    uint64_t b = (uint64_t)1 << 63;
    for (int g = 0; g < nb; g++) {
        const uint64_t c = mse0[(size_t)i * 64 + lev0[g]] + mse1[(size_t)i * 64 + lev1[g]];
        if (c < b) b = c;
    }
    best[i] = b;
}
// tot[j][k] += sum over a chunk of filter blocks of min(best, mse0[i][j] + mse1[i][k]): one workgroup per (j, chunk), lanes along k so that the
// mse1 rows are read as whole 512-byte lines and mse0[i][j] is a broadcast; the totals are accumulated with 64-bit atomics (integers: any order)
constexpr int kDualChunk = 128;
__global__ void __launch_bounds__(256)
one_dual_total_kernel(const uint64_t* __restrict__ mse0, const uint64_t* __restrict__ mse1, int sb_count, const uint64_t* __restrict__ best, int start_gi, int ng,
                      unsigned long long* __restrict__ tot) {
    __shared__ unsigned long long part[4][64];
    const int j = start_gi + blockIdx.x, kk = threadIdx.x & 63, r = threadIdx.x >> 6, k = start_gi + kk;
    const int i0 = blockIdx.y * kDualChunk, i1 = min(i0 + kDualChunk, sb_count);
    unsigned long long acc = 0;
    if (kk < ng)
        for (int i = i0 + r; i < i1; i += 4) {
            const uint64_t c = mse0[(size_t)i * 64 + j] + mse1[(size_t)i * 64 + k], b = best[i];
            acc += c < b ? c : b;
        }
    part[r][kk] = acc;
    __syncthreads();
    if (r == 0 && kk < ng) atomicAdd(&tot[blockIdx.x * ng + kk], part[0][kk] + part[1][kk] + part[2][kk] + part[3][kk]);
}
// the first minimum in (j, k) raster order; out[0] = its total, lev0[nb] / lev1[nb] = the pair
// first minimum of tot[0..n) in index order, by the 256 threads of a workgroup (result valid in thread 0); loads issued back to back
__device__ __forceinline__ void block_argmin_first(const unsigned long long* tot, int n, unsigned long long& bv, int& bi) {
    __shared__ unsigned long long s_v[256];
    __shared__ int                s_i[256];
    unsigned long long v[16];
#pragma unroll
    for (int t = 0; t < 16; t++) { const int i = threadIdx.x + 256 * t; v[t] = i < n ? ((const volatile unsigned long long*)tot)[i] : ~0ull; }
    bv = (unsigned long long)1 << 63; bi = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < 16; t++) { const int i = threadIdx.x + 256 * t; if (i < n && v[t] < bv) { bv = v[t]; bi = i; } }
    s_v[threadIdx.x] = bv; s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) {
            const unsigned long long ov = s_v[threadIdx.x + m];
            const int                oi = s_i[threadIdx.x + m];
            if (ov < s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x])) { s_v[threadIdx.x] = ov; s_i[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    bv = s_v[0]; bi = s_i[0];
}
// shift_after: joint_strength_search_dual's refinement drops the oldest pair before its next search (lev[j] = lev[j + 1], j < n_shift - 1)
// the first minimum in (j, k) raster order; out[0] = its total, lev0[nb] / lev1[nb] = the pair
__global__ void __launch_bounds__(256)
one_dual_pick_kernel(const uint64_t* __restrict__ tot, int start_gi, int ng, int nb, int* __restrict__ lev0, int* __restrict__ lev1, uint64_t* __restrict__ out, int n_shift) {
    unsigned long long bv;
    int                bi;
    block_argmin_first((const unsigned long long*)tot, ng * ng, bv, bi);
    if (threadIdx.x == 0) {
        const bool any = bi != 0x7fffffff;   // nothing below 1 << 63: the reference keeps (0, 0)
        lev0[nb] = any ? start_gi + bi / ng : 0;
        lev1[nb] = any ? start_gi + bi % ng : 0;
        out[0] = bv;
        for (int j = 0; j < n_shift - 1; j++) { lev0[j] = lev0[j + 1]; lev1[j] = lev1[j + 1]; }
    }
}

// ------------------------------------------------------------------------------------------------ CDEF: the whole strength-pair selection of a picture
// finish_cdef_search runs joint_strength_search_dual for 1, 2, 4 and 8 pairs (EbEncCdef.c:1258): four independent chains of 5, 10, 20 and 40
// svt_search_one_dual steps.  One launch per step index advances every chain that is still running (blockIdx.z = chain); a step is ONE kernel
// (joint_step_kernel below).  The workgroup that finishes last picks the chain's pair (first minimum in (j, k) raster order), shifts the list when the next
// step is a refinement step and resets the counter; the totals are triple-buffered and every workgroup clears its share of the buffer of step + 2 on the way.
constexpr int kJointMaxSlices = 64;
struct JointState {
    int lev0[4][8], lev1[4][8];
    unsigned int counter[4];
    unsigned long long result[4];          // total of the chain's last step
    unsigned long long max0, max1;         // largest entry of each table (joint_init_kernel)
    unsigned long long cand_v[4][kJointMaxSlices];   // per reduce workgroup: its first minimum ...
    int                cand_i[4][kJointMaxSlices];   // ... and where
    unsigned int       done, err;          // one-launch form: "the four chains are complete", "a gather timed out"
    int                stable[4], drop0[4], drop1[4];   // step form: refinement steps in a row that changed nothing, the pair the last shift dropped (see joint_resident: EARLY END OF A CHAIN)
    unsigned int       ended[4];           // step form: the chain reached its fixed point -- the remaining launches of the chain return at once
    unsigned long long slot[2][4][256];    // one-launch form: per step parity, chain and workgroup {step tag | total | raster index} (wide tables: {tag | index})
    unsigned long long slot_v[2][4][256];  // wide tables: the totals
    unsigned long long partial[4][kJointMaxSlices][4096];   // step form: totals of one slice of the filter blocks, [j][k] with row stride 64
};
// Several pictures' selections share the launches (blockIdx.y = picture): the stage is bound by its 80 dependent launches, not by their arithmetic, so four pictures
// in one set of launches cost what one costs.
constexpr int kJointMaxPics = 8;
struct JointPics { const uint64_t* mse0[kJointMaxPics]; const uint64_t* mse1[kJointMaxPics]; JointState* S[kJointMaxPics]; };
__global__ void __launch_bounds__(256)
joint_init_kernel(const JointPics P, int n) {
    const uint64_t* __restrict__ mse0 = P.mse0[blockIdx.y]; const uint64_t* __restrict__ mse1 = P.mse1[blockIdx.y]; JointState* __restrict__ S = P.S[blockIdx.y];
    unsigned long long m0 = 0, m1 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { m0 = max(m0, (unsigned long long)mse0[i]); m1 = max(m1, (unsigned long long)mse1[i]); }
    for (int o = 32; o > 0; o >>= 1) { m0 = max(m0, (unsigned long long)__shfl_xor((long long)m0, o)); m1 = max(m1, (unsigned long long)__shfl_xor((long long)m1, o)); }
    __shared__ unsigned long long w0[4], w1[4];
    if ((threadIdx.x & 63) == 0) { w0[threadIdx.x >> 6] = m0; w1[threadIdx.x >> 6] = m1; }
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic pair per workgroup: same-address 64-bit atomics serialise at ~25 ns each
        atomicMax(&S->max0, max(max(w0[0], w0[1]), max(w0[2], w0[3])));
        atomicMax(&S->max1, max(max(w1[0], w1[1]), max(w1[2], w1[3])));
    }
}
// A step of all running chains is two launches.  The totals are a (min, +) product, tot[j][k] = sum_i min(best_i, a[i][j] + b[i][k]), laid out like a small GEMM:
// joint_partial_kernel -- a workgroup (1024 threads) takes a slice of the filter blocks, stages their two distortion rows through LDS in bulk (every load of a
// stage is in flight at once: the tables live in L2, and a per-block dependent load chain is what bounded the earlier forms), forms the running best of each
// staged block once, and every thread accumulates a 2 x 2 tile of pairs over the slice (T = uint32_t when every distortion is below 2^27: add, min, and a
// 64-bit accumulate); the slice's 4096 totals go to its own row of `partial` -- no atomics (64-bit device-scope atomics ran at ~30 G/s here, 18 us per step).
// joint_reduce_kernel -- 64 workgroups per chain add the slices' rows (coalesced), each finds the first minimum of its 64 pairs, and the one that finishes
// last picks the chain's pair (first minimum in (j, k) raster order), shifts the list when the next step is a refinement step and resets the counter.
constexpr int kJointStage = 64;   // filter blocks per LDS stage
template <typename T>
__device__ __forceinline__ void joint_tile(const uint64_t* __restrict__ mse0, const uint64_t* __restrict__ mse1, int p0, int p1, int start_gi, int ng, int idx, const int* s_l0,
                                           const int* s_l1, unsigned char* lds, unsigned long long* __restrict__ out) {
    T* A = (T*)lds;                          // [kJointStage][64]
    T* B = A + kJointStage * 64;             // [kJointStage][64]
    T* best = B + kJointStage * 64;          // [kJointStage]
    const int tid = threadIdx.x, tj = (tid >> 5) * 2, tk = (tid & 31) * 2;
    const int ja = start_gi + min(tj, ng - 1), jb = start_gi + min(tj + 1, ng - 1), ka = start_gi + min(tk, ng - 1), kb = start_gi + min(tk + 1, ng - 1);
    unsigned long long acc[4] = {0, 0, 0, 0};
    for (int s0 = p0; s0 < p1; s0 += kJointStage) {
        const int ns = min(kJointStage, p1 - s0);
        __syncthreads();
        for (int e = tid; e < ns * 64; e += 1024) {   // whole rows: entries outside [start_gi, start_gi + ng) are staged but never used
            A[e] = (T)mse0[(size_t)s0 * 64 + e];
            B[e] = (T)mse1[(size_t)s0 * 64 + e];
        }
        __syncthreads();
        if (tid < ns) {
            T bm = sizeof(T) == 4 ? (T)0xffffffffu : (T)((unsigned long long)1 << 63);   // narrow: every candidate is below 2^28, so this start value never wins
            for (int g = 0; g < idx; g++) { const T v = A[tid * 64 + s_l0[g]] + B[tid * 64 + s_l1[g]]; bm = v < bm ? v : bm; }
            best[tid] = bm;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < ns; i++) {
            const T a0 = A[i * 64 + ja], a1 = A[i * 64 + jb], b0 = B[i * 64 + ka], b1 = B[i * 64 + kb], bb = best[i];
            T v;
            v = a0 + b0; acc[0] += v < bb ? v : bb;
            v = a0 + b1; acc[1] += v < bb ? v : bb;
            v = a1 + b0; acc[2] += v < bb ? v : bb;
            v = a1 + b1; acc[3] += v < bb ? v : bb;
        }
    }
    // [j][k], row stride 64: a thread's two k are adjacent, a wave writes two whole rows
    *(ulonglong2*)&out[tj * 64 + tk] = make_ulonglong2(acc[0], acc[1]);
    *(ulonglong2*)&out[(tj + 1) * 64 + tk] = make_ulonglong2(acc[2], acc[3]);
}
__global__ void __launch_bounds__(1024)
joint_partial_kernel(const JointPics P, int sb_count, int start_gi, int ng, int step) {
    const uint64_t* __restrict__ mse0 = P.mse0[blockIdx.y]; const uint64_t* __restrict__ mse1 = P.mse1[blockIdx.y]; JointState* __restrict__ S = P.S[blockIdx.y];
    const int c = blockIdx.z, nb = 1 << c;
    if (step >= 5 * nb || S->ended[c]) return;
    const int idx = step < nb ? step : nb - 1;   // pairs already selected = the slot this step fills
    __shared__ int s_l0[8], s_l1[8];
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[(2 * kJointStage * 64 + kJointStage) * 8];
    // the selected pairs are only needed after the first stage is in LDS (joint_tile's barrier orders this store before their first use): their load
    // latency overlaps the staging loads instead of preceding them
    if (threadIdx.x < 8) { s_l0[threadIdx.x] = S->lev0[c][threadIdx.x]; s_l1[threadIdx.x] = S->lev1[c][threadIdx.x]; }
    const int p0 = (int)((long long)sb_count * blockIdx.x / gridDim.x), p1 = (int)((long long)sb_count * (blockIdx.x + 1) / gridDim.x);
    const bool narrow = S->max0 < (1ull << 27) && S->max1 < (1ull << 27);
    if (narrow) joint_tile<uint32_t>(mse0, mse1, p0, p1, start_gi, ng, idx, s_l0, s_l1, s_stage, S->partial[c][blockIdx.x]);
    else joint_tile<unsigned long long>(mse0, mse1, p0, p1, start_gi, ng, idx, s_l0, s_l1, s_stage, S->partial[c][blockIdx.x]);
}
__global__ void __launch_bounds__(256)
joint_reduce_kernel(const JointPics P, int slices, int start_gi, int ng, int step, int early) {
    JointState* __restrict__ S = P.S[blockIdx.y];
    const int c = blockIdx.z, nb = 1 << c, total_steps = 5 * nb;
    if (step >= total_steps || S->ended[c]) return;
    const int idx = step < nb ? step : nb - 1;
    __shared__ unsigned long long r_v[256];
    __shared__ int                r_i[64];
    __shared__ bool               s_last;
    // 64 pairs per workgroup, four threads per pair: thread (pair, q) adds the slices s = q, q + 4, ... -- at most 16 loads, all in flight at once (the rows
    // were written by other XCDs: every load is a memory-side round trip, so batches of dependent loads were what this kernel's time consisted of)
    const int q = threadIdx.x >> 6, p = blockIdx.x * 64 + (threadIdx.x & 63), j = p >> 6, k = p & 63;
    unsigned long long v = 0;
    const unsigned long long* src = &S->partial[c][0][p];
#pragma unroll
    for (int t = 0; t < kJointMaxSlices / 4; t++) {
        const int sl = q + 4 * t;
        v += sl < slices ? src[(size_t)sl * 4096] : 0ull;
    }
    r_v[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        const bool on = j < ng && k < ng;
        unsigned long long bv = on ? r_v[threadIdx.x] + r_v[threadIdx.x + 64] + r_v[threadIdx.x + 128] + r_v[threadIdx.x + 192] : ~0ull;
        int                bi = on ? j * ng + k : 0x7fffffff;          // the reference's raster index over the ng x ng table
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ov = (unsigned long long)__shfl_xor((long long)bv, o);
            const int                oi = __shfl_xor(bi, o);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (threadIdx.x == 0) {
            // write-through (agent-scope) stores, drained, then the arrival: no L2 write-back / invalidate as __threadfence() would do (~3.5 us per workgroup here)
            __hip_atomic_store(&S->cand_v[c][blockIdx.x], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&S->cand_i[c][blockIdx.x], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = atomicAdd(&S->counter[c], 1u) == gridDim.x - 1;
        }
    }
    (void)r_i;
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    // the last workgroup of the chain: its first wave reads the candidates side by side (agent-scope loads: they were stored write-through and drained before each arrival)
    unsigned long long bv = (unsigned long long)1 << 63;   // "tot < best" with best = 1 << 63 (EbEncCdef.c:1104): nothing below it keeps (0, 0)
    int                bi = 0x7fffffff;
    if (threadIdx.x < gridDim.x) {
        const unsigned long long ov = __hip_atomic_load(&S->cand_v[c][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int                oi = __hip_atomic_load(&S->cand_i[c][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (oi != 0x7fffffff && ov < bv) { bv = ov; bi = oi; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ov = (unsigned long long)__shfl_xor((long long)bv, o);
        const int                oi = __shfl_xor(bi, o);
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (threadIdx.x != 0) return;
    const bool any = bi != 0x7fffffff;
    const int n0 = any ? start_gi + bi / ng : 0, n1 = any ? start_gi + bi % ng : 0;
    S->lev0[c][idx] = n0;
    S->lev1[c][idx] = n1;
    S->result[c] = bv;
    // the exact early end of a chain (joint_resident, "EARLY END OF A CHAIN"): nb refinement steps in a row that put back the pair they had dropped = a fixed point
    bool fixed = false;
    if (step >= nb) {
        const int st = (n0 == S->drop0[c] && n1 == S->drop1[c]) ? S->stable[c] + 1 : 0;
        S->stable[c] = st;
        fixed = early && st >= nb && step + 1 < total_steps;
    }
    if (fixed) {   // the remaining steps would rotate the list by one place each: do that, and let the chain's remaining launches return at once
        const int rot = (total_steps - (step + 1)) % nb;
        int t0[8], t1[8];
        for (int g = 0; g < nb; g++) { t0[g] = S->lev0[c][g]; t1[g] = S->lev1[c][g]; }
        for (int g = 0; g < nb; g++) { S->lev0[c][g] = t0[(g + rot) % nb]; S->lev1[c][g] = t1[(g + rot) % nb]; }
        S->ended[c] = 1u;
    } else if (step + 1 >= nb && step + 1 < total_steps) {   // the next step is a refinement step: drop the oldest pair
        S->drop0[c] = S->lev0[c][0]; S->drop1[c] = S->lev1[c][0];
        for (int g = 0; g < nb - 1; g++) { S->lev0[c][g] = S->lev0[c][g + 1]; S->lev1[c][g] = S->lev1[c][g + 1]; }
    }
    S->counter[c] = 0;
}

// Two variations that were built, verified and measured, and dropped (MI355X, 2040 filter blocks):
//  - a pair-divided launch-per-step form (one launch per step, 256 workgroups of 512 threads per chain streaming their 4 x 4 tile's columns and the filter blocks'
//    running best through LDS, the last workgroup advancing the list): 1.26 ms per picture, 31 us per step -- re-reading 150 MB of columns from L2 every step;
//  - sums in doubles (exact below 2^53; add / min / add are three instructions where 64-bit integers take seven) for tables between 2^27 and 2^40: 0.81 against
//    0.83 ms in steps, 0.57 against 0.58 ms resident -- neither form is bound by its arithmetic.
// ---- the same selection in ONE launch (svt_hip_set_cdef_select_form; for pictures of up to kResMaxSb filter blocks).  The step-by-step form above spends its time moving 16 MB
// of slice totals per step across the XCDs and on ~8 dependent memory-side round trips per step; none of that is arithmetic (a whole selection is ~0.6 G
// add / min / add).  Here the PAIRS are divided, not the filter blocks: 256 workgroups (one per compute unit, 1024 threads) each own a 4 x 4 tile of strength
// pairs for ALL four chains and keep that tile's eight table columns of every filter block in LDS for the whole selection (64 KB, read once), so a workgroup's
// totals are complete and the only thing exchanged per step is its first minimum: ONE 8-byte word per chain {step tag | total | raster index}, written
// write-through into a slot array and gathered by one wave of every workgroup (no counters, no fences: the tag makes the word self-validating, slots alternate by
// step parity so that a fast workgroup cannot overwrite a word a slow one is still waiting for).  After the gather every workgroup knows every chain's pick,
// reads the two new columns of its filter blocks (2 per thread; the list members' sums live in registers) and forms the running best for the next step.
// Tables above 2^27 (T = 64-bit) take the same path with the columns read from L2 instead of LDS and a two-word exchange (total, then tag | index).
// The spin is bounded and raises `err` instead of hanging (the 256 workgroups must be resident together: launches of this kernel are serialised by the caller).
constexpr int kResWgs = 256;
constexpr int kResMaxSb = 2048;
constexpr unsigned long long kResEmpty = (1ull << 56) - 1;
// The selection reads table COLUMNS (one strength, every filter block): joint_transpose_kernel lays both tables out column-major once ([strength][kResMaxSb],
// in the state's `partial` area, which this form does not otherwise use) through an LDS tile, and takes the tables' maxima on the way (joint_init_kernel's job).
__global__ void __launch_bounds__(256)
joint_transpose_kernel(const uint64_t* __restrict__ mse0, const uint64_t* __restrict__ mse1, int sb_count, JointState* __restrict__ S) {
    __shared__ unsigned long long tile[64][65];
    __shared__ unsigned long long wmax[4];
    const uint64_t* __restrict__ src = blockIdx.y ? mse1 : mse0;
    unsigned long long* __restrict__ dst = &S->partial[0][0][0] + (size_t)blockIdx.y * 64 * kResMaxSb;
    const int sb0 = blockIdx.x * 64, ns = min(64, sb_count - sb0);
    unsigned long long m = 0;
    for (int e = threadIdx.x; e < ns * 64; e += 256) { const unsigned long long v = src[(size_t)sb0 * 64 + e]; tile[e >> 6][e & 63] = v; m = max(m, v); }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned long long)__shfl_xor((long long)m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) { const int gi = e >> 6, i = e & 63; if (i < ns) dst[(size_t)gi * kResMaxSb + sb0 + i] = tile[i][gi]; }
    if (threadIdx.x == 0) atomicMax(blockIdx.y ? &S->max1 : &S->max0, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}
// sum over the lanes l, l ^ 4, l ^ 8, ... ^ 32 (the 16 lanes of a wave that share l & 3)
__device__ __forceinline__ unsigned long long sum_stride4(uint32_t v) {   // every partial sum of one 16-lane row fits 32 bits (narrow tables)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);   // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);   // row_ror:8
    unsigned long long r = v;
    r += (unsigned long long)__shfl_xor((long long)r, 16);
    r += (unsigned long long)__shfl_xor((long long)r, 32);
    return r;
}
__device__ __forceinline__ unsigned long long sum_stride4(unsigned long long r) {
    for (int o = 4; o < 64; o <<= 1) r += (unsigned long long)__shfl_xor((long long)r, o);
    return r;
}
// MODE 0 (tables below 2^26): 32-bit columns in LDS, 32-bit arithmetic, packed one-word exchange, the list members' sums of a thread's two filter blocks in
// registers.  MODE 1 (below 2^32): 32-bit columns in LDS, 64-bit arithmetic.  MODE 2: columns read from L2, 64-bit arithmetic.  Modes 1 and 2 exchange two
// words (total, then tag | index) and rebuild the running best from the list's columns each step (the 64-bit register copy of fifteen members would spill).
template <int MODE>
__device__ void joint_resident(int sb_count, int start_gi, int ng, JointState* __restrict__ S, unsigned char* lds, int early) {
    constexpr bool kPacked = MODE <= 1, kLdsCols = MODE <= 1, kRing = MODE == 0;   // packed: totals below 2^44 (2048 filter blocks x 2^33)
    typedef typename std::conditional<MODE == 0, uint32_t, unsigned long long>::type T;
    const unsigned long long* __restrict__ col0 = &S->partial[0][0][0];            // [64][kResMaxSb], joint_transpose_kernel
    const unsigned long long* __restrict__ col1 = col0 + 64 * kResMaxSb;
    const T kMax = MODE == 0 ? (T)0xffffffffu : (T)((unsigned long long)1 << 63);   // narrow: every candidate is below 2^28, so this start value never wins
    T* best = (T*)lds;                                                            // [4][kResMaxSb]
    unsigned long long* red = (unsigned long long*)(best + 4 * kResMaxSb);        // [4][16 waves][16 pairs]
    uint32_t* AB = (uint32_t*)(red + 4 * 16 * 16);                                // narrow: [kResMaxSb][8] = four columns of each table
    __shared__ int                s_pick_i[4], s_err;
    __shared__ unsigned long long s_pick_v[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int tj = (w >> 4) * 4, tk = (w & 15) * 4, pg = tid & 3, slice = tid >> 2;
    const int ca0 = start_gi + min(tj + 2 * (pg >> 1), ng - 1), ca1 = start_gi + min(tj + 2 * (pg >> 1) + 1, ng - 1);
    const int cb0 = start_gi + min(tk + 2 * (pg & 1), ng - 1), cb1 = start_gi + min(tk + 2 * (pg & 1) + 1, ng - 1);
    if (kLdsCols)
        for (int x = 0; x < 8; x++) {
            const unsigned long long* __restrict__ cp = x < 4 ? col0 + (size_t)(start_gi + min(tj + x, ng - 1)) * kResMaxSb : col1 + (size_t)(start_gi + min(tk + x - 4, ng - 1)) * kResMaxSb;
            for (int sb = tid; sb < sb_count; sb += 1024) AB[sb * 8 + x] = (uint32_t)cp[sb];
        }
    for (int e = tid; e < 4 * kResMaxSb; e += 1024) best[e] = kMax;
    if (tid == 0) s_err = 0;
    T cr[kRing ? 15 : 1][2];   // the sums a[i][lev0[g]] + b[i][lev1[g]] of this thread's two filter blocks (tid, tid + 1024) for the members of each chain's list: chain c at [nb - 1 + g]
#pragma unroll
    for (int g = 0; g < (kRing ? 15 : 1); g++) cr[g][0] = cr[g][1] = kMax;
    int ul0[kRing ? 1 : 15], ul1[kRing ? 1 : 15];   // modes 1, 2: every thread's own (wave-uniform, scalar-register) copy of the four lists, same layout;
    __shared__ int s_l0[4][8], s_l1[4][8];          // mode 0: thread 0's copy
    // EARLY END OF A CHAIN.  A refinement step drops the oldest pair of the list and searches a replacement given the others (EbEncCdef.c:1140-1165).  When nb
    // refinement steps in a row put back exactly the pair that was dropped, every member of the list has been confirmed against the other nb - 1: the list is a fixed
    // point, the remaining steps would each rotate it by one place and change nothing else -- so the chain ends here, its list rotated by (remaining steps mod nb)
    // and its total as it is: exactly what the remaining steps would have left.  On coded pictures that happens after 8 - 13 of the 32 refinement steps of the
    // eight-pair chain (1 of 4, 2 of 8, 4 of 16 for the shorter ones), which halves the number of dependent steps.  Every workgroup sees the same picks, so every
    // workgroup ends a chain at the same step (s_end: the first step the chain no longer takes part in).
    __shared__ int s_end[4];
#pragma unroll
    for (int g = 0; g < (kRing ? 1 : 15); g++) ul0[g] = ul1[g] = 0;
    if (tid < 32) { s_l0[tid >> 3][tid & 7] = 0; s_l1[tid >> 3][tid & 7] = 0; }
    if (tid < 4) s_end[tid] = 5 << tid;
    int stable[4] = {0, 0, 0, 0}, drop0[4] = {0, 0, 0, 0}, drop1[4] = {0, 0, 0, 0}, n_fin = 0;   // thread 0: refinement steps in a row that changed nothing, the pair the last shift dropped
    __syncthreads();
#ifdef SVT_RES_TRACE
#define RES_T(k) do { if (tid == 0 && (w == 0 || w == 133)) S->partial[3][w == 0 ? 0 : 1][step * 8 + (k)] = wall_clock64(); } while (0)
#else
#define RES_T(k) do {} while (0)
#endif
    for (int step = 0; step < 40; step++) {
        const int par = step & 1;
        const unsigned long long tag = (unsigned long long)(step + 1) << 56;
        const int endc[4] = {__builtin_amdgcn_readfirstlane(s_end[0]), __builtin_amdgcn_readfirstlane(s_end[1]), __builtin_amdgcn_readfirstlane(s_end[2]), __builtin_amdgcn_readfirstlane(s_end[3])};
        if (step >= endc[0] && step >= endc[1] && step >= endc[2] && step >= endc[3]) break;   // every chain has ended
        RES_T(0);
        // ---- this workgroup's 16 pairs, every running chain: thread = (2 x 2 pairs, every 256th filter block)
        T acc[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[c][q] = 0;
#pragma unroll 2
        for (int sb = slice; sb < sb_count; sb += 256) {
            T a0, a1, b0, b1;
            if (kLdsCols) {
                const uint2 av = *(const uint2*)&AB[sb * 8 + (pg >> 1) * 2], bw = *(const uint2*)&AB[sb * 8 + 4 + (pg & 1) * 2];
                a0 = av.x; a1 = av.y; b0 = bw.x; b1 = bw.y;
            } else {
                a0 = (T)col0[(size_t)ca0 * kResMaxSb + sb]; a1 = (T)col0[(size_t)ca1 * kResMaxSb + sb]; b0 = (T)col1[(size_t)cb0 * kResMaxSb + sb]; b1 = (T)col1[(size_t)cb1 * kResMaxSb + sb];
            }
            const T v00 = a0 + b0, v01 = a0 + b1, v10 = a1 + b0, v11 = a1 + b1;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (step >= endc[c]) continue;
                const T bb = best[c * kResMaxSb + sb];
                acc[c][0] += v00 < bb ? v00 : bb; acc[c][1] += v01 < bb ? v01 : bb; acc[c][2] += v10 < bb ? v10 : bb; acc[c][3] += v11 < bb ? v11 : bb;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (step >= endc[c]) continue;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long r = sum_stride4(acc[c][q]);
                if (lane < 4) red[(c * 16 + wave) * 16 + lane * 4 + q] = r;
            }
        }
        RES_T(1);
        __syncthreads();
        RES_T(2);
        // ---- first wave: the workgroup's first minimum per chain -> its slot; then the gather of everybody's
        if (wave == 0) {
            const int c = lane >> 4, p = lane & 15;
            const bool running = step < s_end[c];
            unsigned long long tot = 0;
            if (running)
#pragma unroll
                for (int wv = 0; wv < 16; wv++) tot += red[(c * 16 + wv) * 16 + p];
            const int jl = tj + 2 * (p >> 3) + ((p >> 1) & 1), kl = tk + 2 * ((p >> 2) & 1) + (p & 1);
            const bool on = running && jl < ng && kl < ng;
            unsigned long long bv = on ? tot : ~0ull;
            int                bi = on ? jl * ng + kl : 0x7fffffff;   // the reference's raster index over the ng x ng table
            for (int o = 1; o < 16; o <<= 1) {
                const unsigned long long ov = (unsigned long long)__shfl_xor((long long)bv, o);
                const int                oi = __shfl_xor(bi, o);
                if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (p == 0 && running) {
                if (kPacked) {
                    __hip_atomic_store(&S->slot[par][c][w], tag | (bi == 0x7fffffff ? kResEmpty : (bv << 12 | (unsigned)bi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    __hip_atomic_store(&S->slot_v[par][c][w], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the total is in memory before the word that validates it
                    __hip_atomic_store(&S->slot[par][c][w], tag | (bi == 0x7fffffff ? kResEmpty : (unsigned long long)bi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            RES_T(3);
            // one poll covers all running chains (sixteen loads in flight at most)
            // a slot that is not there yet carries an older (smaller) tag, so the minimum of a lane's four words says whether all four are this step's
            unsigned long long k[4];
            bool     failed = false;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int cc = 0; cc < 4; cc++)
                    if (step < endc[cc]) {
                        unsigned long long t4[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) t4[t] = __hip_atomic_load(&S->slot[par][cc][lane + 64 * t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        k[cc] = min(min(t4[0], t4[1]), min(t4[2], t4[3]));
                    }
#pragma unroll
                for (int cc = 0; cc < 4; cc++)
                    if (step < endc[cc]) ok = ok && (k[cc] >> 56) == (unsigned)(step + 1);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 16) || ((spins & 63) == 0 && __hip_atomic_load(&S->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { failed = true; break; }
            }
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                if (step >= endc[cc] || failed) continue;
                unsigned long long gv = (unsigned long long)1 << 63;   // "tot < best" with best = 1 << 63 (EbEncCdef.c:1104): nothing below it keeps (0, 0)
                int                gi = 0x7fffffff;
                if (kPacked) {
                    unsigned long long m = k[cc] & kResEmpty;
                    for (int o = 1; o < 64; o <<= 1) m = min(m, (unsigned long long)__shfl_xor((long long)m, o));
                    if (m != kResEmpty) { gv = m >> 12; gi = (int)(m & 4095); }
                } else {   // every word is this step's: now the totals and indices of the lane's four workgroups
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const unsigned long long ix = __hip_atomic_load(&S->slot[par][cc][lane + 64 * t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kResEmpty;
                        if (ix == kResEmpty) continue;
                        const unsigned long long ov = __hip_atomic_load(&S->slot_v[par][cc][lane + 64 * t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ov < gv || (ov == gv && (int)ix < gi)) { gv = ov; gi = (int)ix; }
                    }
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned long long ov = (unsigned long long)__shfl_xor((long long)gv, o);
                        const int                oi = __shfl_xor(gi, o);
                        if (ov < gv || (ov == gv && oi < gi)) { gv = ov; gi = oi; }
                    }
                }
                if (lane == 0) { s_pick_v[cc] = gv; s_pick_i[cc] = gi; }
            }
            if (failed && lane == 0) { s_err = 1; atomicExch(&S->err, 1u); atomicExch(&S->counter[0], 0xdeadu); }   // counter[0] = SvtHipCdefSelectResult.status[0]
            RES_T(4);
        }
        __syncthreads();
        RES_T(5);
        if (s_err) return;
        // ---- every workgroup advances its copy of each chain's list; a thread's two filter blocks get the new member's sum and the next step's running best
        T   cn[4][2];
        int pl0[4], pl1[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {   // the column loads of every running chain are in flight together
            if (step >= endc[c]) continue;
            const int  bi = s_pick_i[c];
            const bool any = bi != 0x7fffffff;
            pl0[c] = __builtin_amdgcn_readfirstlane(any ? start_gi + bi / ng : 0); pl1[c] = __builtin_amdgcn_readfirstlane(any ? start_gi + bi % ng : 0);   // scalar registers
            if constexpr (kRing)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int sb = tid + 1024 * u;
                    cn[c][u] = sb < sb_count ? (T)col0[(size_t)pl0[c] * kResMaxSb + sb] + (T)col1[(size_t)pl1[c] * kResMaxSb + sb] : kMax;
                }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            constexpr int kOff[4] = {0, 1, 3, 7};
            const int nb = 1 << c, total_steps = 5 * nb;
            if (step >= endc[c]) continue;
            const int idx = step < nb ? step : nb - 1;
#pragma unroll
            for (int g = 0; g < nb; g++)
                if (g == idx) {
                    if constexpr (kRing) { cr[kOff[c] + g][0] = cn[c][0]; cr[kOff[c] + g][1] = cn[c][1]; }
                    else { ul0[kOff[c] + g] = pl0[c]; ul1[kOff[c] + g] = pl1[c]; }
                }
            if (tid == 0) {
                if constexpr (kRing) { s_l0[c][idx] = pl0[c]; s_l1[c][idx] = pl1[c]; }
                bool fin = step + 1 == total_steps;
                int  rot = 0;
                if (step >= nb) {   // a refinement step: did it put back the pair the shift before it dropped?
                    stable[c] = (pl0[c] == drop0[c] && pl1[c] == drop1[c]) ? stable[c] + 1 : 0;
                    if (early && stable[c] >= nb && !fin) { fin = true; rot = (total_steps - (step + 1)) % nb; }   // a fixed point: the remaining steps only rotate the list
                }
                if (fin) {
                    s_end[c] = step + 1;   // read by every thread at the top of the next step (the barrier at the end of this one is in between)
                    n_fin++;
                    if (w == 0) {   // the chain is complete: its pairs and total
#pragma unroll
                        for (int g = 0; g < 8; g++) {
                            const int gs = g < nb ? (g + rot) % nb : g;
                            if constexpr (kRing) { S->lev0[c][g] = g < nb ? s_l0[c][gs] : 0; S->lev1[c][g] = g < nb ? s_l1[c][gs] : 0; }
                            else {
                                int v0 = 0, v1 = 0;
#pragma unroll
                                for (int q = 0; q < nb; q++) if (q == gs) { v0 = ul0[kOff[c] + q]; v1 = ul1[kOff[c] + q]; }   // compile-time indices: the lists live in registers
                                S->lev0[c][g] = g < nb ? v0 : 0; S->lev1[c][g] = g < nb ? v1 : 0;
                            }
                        }
                        S->result[c] = s_pick_v[c];
                        if (n_fin == 4) S->done = 1;
                    }
                }
                if constexpr (kRing) { drop0[c] = s_l0[c][0]; drop1[c] = s_l1[c][0]; }   // what the shift below is about to drop
                else { drop0[c] = ul0[kOff[c]]; drop1[c] = ul1[kOff[c]]; }
            }
            if (step + 1 >= total_steps) continue;
            if (step + 1 >= nb) {   // the next step is a refinement step: drop the oldest pair
#pragma unroll
                for (int g = 0; g < nb - 1; g++) {
                    if constexpr (kRing) { cr[kOff[c] + g][0] = cr[kOff[c] + g + 1][0]; cr[kOff[c] + g][1] = cr[kOff[c] + g + 1][1]; }
                    else { ul0[kOff[c] + g] = ul0[kOff[c] + g + 1]; ul1[kOff[c] + g] = ul1[kOff[c] + g + 1]; }
                }
                if constexpr (kRing)
                    if (tid == 0)
                        for (int g = 0; g < nb - 1; g++) { s_l0[c][g] = s_l0[c][g + 1]; s_l1[c][g] = s_l1[c][g + 1]; }
            }
            const int nidx = step + 1 < nb ? step + 1 : nb - 1;
            T bm0 = kMax, bm1 = kMax;
#pragma unroll
            for (int g = 0; g < nb; g++)
                if (g < nidx) {
                    T m0, m1;
                    if constexpr (kRing) { m0 = cr[kOff[c] + g][0]; m1 = cr[kOff[c] + g][1]; }
                    if constexpr (!kRing) {
                        const unsigned long long* __restrict__ c0 = col0 + (size_t)ul0[kOff[c] + g] * kResMaxSb; const unsigned long long* __restrict__ c1 = col1 + (size_t)ul1[kOff[c] + g] * kResMaxSb;
                        m0 = tid < sb_count ? c0[tid] + c1[tid] : kMax; m1 = tid + 1024 < sb_count ? c0[tid + 1024] + c1[tid + 1024] : kMax;
                    }
                    bm0 = m0 < bm0 ? m0 : bm0; bm1 = m1 < bm1 ? m1 : bm1;
                }
            best[c * kResMaxSb + tid] = bm0; best[c * kResMaxSb + tid + 1024] = bm1;
        }
        RES_T(6);
        __syncthreads();
        RES_T(7);
    }
}
// Three kernels, launched back to back: the two whose table width does not apply return at once (~1.5 us each).  One kernel with all bodies would carry the
// 64-bit bodies' register pressure into the narrow one.
template <int MODE>
__global__ void __launch_bounds__(1024)
joint_resident_kernel(int sb_count, int start_gi, int ng, JointState* __restrict__ S, int early) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_res[];
    const unsigned long long m = max(S->max0, S->max1);
    const int mode = m < (1ull << 26) ? 0 : m < (1ull << 32) ? 1 : 2;   // below 2^26: the sums of a 16-lane row (32 filter blocks) fit 32 bits
    if (mode != MODE) return;
    joint_resident<MODE>(sb_count, start_gi, ng, S, s_res, early);
}
constexpr size_t kResLdsBytes = 4 * kResMaxSb * 8 + 4 * 16 * 16 * 8 + (size_t)kResMaxSb * 8 * 4;   // sized for the wider `best`

// finish_cdef_search after the four searches (EbEncCdef.c:1258-1298): the count of strength pairs by rate-distortion cost, then every filter block's
// pair.  One thread per filter block; every workgroup redoes the four-way cost comparison (a handful of scalar operations).
struct CdefFinishOut { int cdef_bits, nb_strengths, y_strength[8], uv_strength[8]; unsigned long long best_cost; };
__global__ void __launch_bounds__(256)
cdef_finish_kernel(const uint64_t* __restrict__ mse0, const uint64_t* __restrict__ mse1, int sb_count, const JointState* __restrict__ S, unsigned long long lambda,
                   const int* __restrict__ sb_fb, CdefFinishOut* __restrict__ out, int* __restrict__ sel_gi, uint8_t* __restrict__ fb_y, uint8_t* __restrict__ fb_uv) {
    if (S->counter[0]) {   // the selection did not complete (svt_hip.h: status[0])
        if (blockIdx.x == 0 && threadIdx.x == 0) { out->cdef_bits = -1; out->nb_strengths = 0; }
        return;
    }
    unsigned long long best = (unsigned long long)1 << 63;
    int bits = 0;
    for (int i = 0; i <= 3; i++) {
        const int nb = 1 << i;
        const long long total_bits = (long long)sb_count * i + nb * 6 * 2;                       // CDEF_STRENGTH_BITS = 6
        const unsigned long long rate = (unsigned long long)total_bits << 9, dist = S->result[i] * 16;   // av1_cost_literal
        const unsigned long long cost = ((rate * lambda + 256) >> 9) + dist * 128;              // RDCOST: AV1_PROB_COST_SHIFT 9, RDDIV_BITS 7
        if (cost < best) { best = cost; bits = i; }
    }
    const int nb = 1 << bits;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out->cdef_bits = bits; out->nb_strengths = nb; out->best_cost = best;
        for (int g = 0; g < 8; g++) { out->y_strength[g] = g < nb ? S->lev0[bits][g] : 0; out->uv_strength[g] = g < nb ? S->lev1[bits][g] : 0; }
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= sb_count) return;
    unsigned long long bm = (unsigned long long)1 << 63;
    int bg = 0;
    for (int g = 0; g < nb; g++) {
        const unsigned long long c = mse0[(size_t)i * 64 + S->lev0[bits][g]] + mse1[(size_t)i * 64 + S->lev1[bits][g]];
        if (c < bm) { bm = c; bg = g; }
    }
    if (sel_gi) sel_gi[i] = bg;
    const int fb = sb_fb ? sb_fb[i] : i;
    if (fb_y) fb_y[fb] = (uint8_t)S->lev0[bits][bg];
    if (fb_uv) fb_uv[fb] = (uint8_t)S->lev1[bits][bg];
}
extern "C" int svt_hip_launch_cdef_finish(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, const void* state, unsigned long long lambda, const int* sb_fb,
                                          void* out, int* sel_gi, uint8_t* fb_y, uint8_t* fb_uv) {
    hipLaunchKernelGGL(cdef_finish_kernel, dim3(sb_count > 0 ? (sb_count + 255) / 256 : 1), dim3(256), 0, st, mse0, mse1, sb_count, (const JointState*)state, lambda, sb_fb,
                       (CdefFinishOut*)out, sel_gi, fb_y, fb_uv);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ self-guided projection on materialised filters
// mode 0: acc[0..4] += {H00, H01, H11, C0, C1} (exact integers; the reference accumulates the same integers in doubles, exact below 2^53)
// mode 1: acc[0] += squared error of the projection with xq
template <typename PIX>
__global__ void __launch_bounds__(256)
sgr_flt_proj_kernel(const PIX* __restrict__ src, int ss, const PIX* __restrict__ dat, int ds, const int32_t* __restrict__ f0, int f0s, const int32_t* __restrict__ f1,
                    int f1s, int w, int h, int r0, int r1, int mode, int xq0, int xq1, long long* __restrict__ acc) {
    long long a[5] = {0, 0, 0, 0, 0};
    for (int y = blockIdx.x; y < h; y += gridDim.x)
        for (int x = threadIdx.x; x < w; x += 256) {
            const int d = dat[(size_t)y * ds + x], s = src[(size_t)y * ss + x];
            const int u = d << 4;
            if (mode == 0) {
                const long long sv = (long long)((s << 4) - u);
                const long long v1 = r0 > 0 ? (long long)f0[(size_t)y * f0s + x] - u : 0, v2 = r1 > 0 ? (long long)f1[(size_t)y * f1s + x] - u : 0;
                a[0] += v1 * v1; a[1] += v1 * v2; a[2] += v2 * v2; a[3] += v1 * sv; a[4] += v2 * sv;
            } else {
                int e;
                if (r0 > 0 || r1 > 0) {
                    int v = u << 7;
                    if (r0 > 0) v += xq0 * (f0[(size_t)y * f0s + x] - u);
                    if (r1 > 0) v += xq1 * (f1[(size_t)y * f1s + x] - u);
                    e = ((v + (1 << 10)) >> 11) - s;
                } else
                    e = d - s;
                a[0] += (long long)e * e;
            }
        }
    __shared__ long long part[4][5];
    const int nacc = mode == 0 ? 5 : 1;
    for (int k = 0; k < nacc; k++) {
        const long long t = wave_sum_i64(a[k]);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < nacc)
        atomicAdd((unsigned long long*)&acc[threadIdx.x],
                  (unsigned long long)(part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]));
}
// svt_get_proj_subspace_c's solve: every operation an IEEE double operation in the reference's order (-ffp-contract=off)
__global__ void sgr_flt_solve_kernel(const long long* __restrict__ sums, int size, int r0, int r1, int32_t* __restrict__ xq) {
    double H00 = (double)sums[0], H01 = (double)sums[1], H11 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    const double dsize = (double)size;
    H00 /= dsize; H01 /= dsize; H11 /= dsize; C0 /= dsize; C1 /= dsize;
    const double H10 = H01;
    int q0 = 0, q1 = 0;
    if (r0 == 0) { if (!(H11 < 1e-8)) q1 = (int)rint((C1 / H11) * 128.0); }
    else if (r1 == 0) { if (!(H00 < 1e-8)) q0 = (int)rint((C0 / H00) * 128.0); }
    else {
        const double det = H00 * H11 - H01 * H10;
        if (!(det < 1e-8)) {
            const double x0 = (H11 * C0 - H01 * C1) / det, x1 = (H00 * C1 - H10 * C0) / det;
            q0 = (int)rint(x0 * 128.0); q1 = (int)rint(x1 * 128.0);
        }
    }
    xq[0] = q0; xq[1] = q1;
}

// ------------------------------------------------------------------------------------------------ 8-tap convolution with a phase table (scaling allowed)
// filters[16][8]: the 256-byte-aligned kernel table the reference derives from its filter pointer; q0 = first phase, step = phase advance per sample
template <bool VERT>
__global__ void __launch_bounds__(256)
convolve8_kernel(const uint8_t* __restrict__ src, int ss, uint8_t* __restrict__ dst, int ds, const int16_t* __restrict__ filters, int q0, int step, int w, int h) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int      q = q0 + (VERT ? y : x) * step, base = (q >> 4) - 3;
    const int16_t* f = filters + 8 * (q & 15);
    int            sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) sum += (int)(VERT ? src[(ptrdiff_t)(base + k) * ss + x] : src[(ptrdiff_t)y * ss + base + k]) * f[k];
    dst[(size_t)y * ds + x] = (uint8_t)clip_px((sum + 64) >> 7, 8);
}

// ------------------------------------------------------------------------------------------------ Wiener convolution of one processing unit
// horizontal pass with the source added and the intermediate clamp, vertical pass on the clamped rows; step 16 (no scaling), taps fx / fy.
template <typename PIX>
__global__ void __launch_bounds__(256)
wiener_convolve_kernel(const PIX* __restrict__ src, int ss, PIX* __restrict__ dst, int ds, const int16_t* __restrict__ taps, int w, int h, int round0, int round1, int bd) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int16_t *fx = taps, *fy = taps + 8;
    const int      lim = (1 << (bd + 1 + 7 - round0)) - 1;
    int            mid[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const PIX* row = src + (ptrdiff_t)(y + k - 3) * ss + x - 3;
        int        sum = ((int)row[3] << 7) + (1 << (bd + 7 - 1));
#pragma unroll
        for (int t = 0; t < 8; t++) sum += (int)row[t] * fx[t];
        const int v = (sum + ((1 << round0) >> 1)) >> round0;
        mid[k] = v < 0 ? 0 : (v > lim ? lim : v);
    }
    int sum = (mid[3] << 7) - (1 << (bd + round1 - 1));
#pragma unroll
    for (int k = 0; k < 8; k++) sum += mid[k] * fy[k];
    dst[(size_t)y * ds + x] = (PIX)clip_px((sum + ((1 << round1) >> 1)) >> round1, bd);
}

// ------------------------------------------------------------------------------------------------ 64-point re-pack of the N2 / N4 coefficient shapes
// handle_transform64x{16,32,64}_N2_N4_c (EbTransforms.c:2947-2969): rows 1 .. rows-1 move from stride 64 to stride 32, nothing is zeroed, no energy
__global__ void __launch_bounds__(1024)
repack64_kernel(int32_t* __restrict__ coeff, int rows, int per_block) {
    int32_t* c = coeff + (size_t)blockIdx.x * per_block;
    const int r = threadIdx.x >> 5, x = threadIdx.x & 31;
    const int32_t v = r < rows ? c[r * 64 + x] : 0;
    __syncthreads();
    if (r < rows) c[r * 32 + x] = v;
}

// ------------------------------------------------------------------------------------------------ one reference of a compound prediction
// variant 0 = 2d, 1 = x, 2 = y, 3 = 2d_copy.  do_average = 0: the 16-bit result goes to the compound buffer; 1: it is combined with the buffer
// (plain or distance-weighted average) and written to dst as pixels.  taps[0..7] horizontal, taps[8..15] vertical kernel.
struct JntArgs { int variant, w, h, round0, round1, do_average, use_jnt, fwd, bck, bd; };
template <typename PIX>
__global__ void __launch_bounds__(256)
jnt_convolve_kernel(const PIX* __restrict__ src, int ss, PIX* __restrict__ dst, int ds, uint16_t* __restrict__ cb, int cbs, const int16_t* __restrict__ taps, JntArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.w || y >= a.h) return;
    const int16_t *fx = taps, *fy = taps + 8;
    const int      offset_bits = a.bd + 14 - a.round0, round_offset = (1 << (offset_bits - a.round1)) + (1 << (offset_bits - a.round1 - 1));
    const int      round_bits = 14 - a.round0 - a.round1;
    int            res;
    if (a.variant == 0) {
        int sum = 1 << offset_bits;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const PIX* row = src + (ptrdiff_t)(y + k - 3) * ss + x - 3;
            int        hs = 1 << (a.bd + 6);
#pragma unroll
            for (int t = 0; t < 8; t++) hs += fx[t] * (int)row[t];
            sum += fy[k] * (int)(int16_t)((hs + ((1 << a.round0) >> 1)) >> a.round0);
        }
        res = (int)(uint16_t)((sum + ((1 << a.round1) >> 1)) >> a.round1);
    } else if (a.variant == 1) {
        int hs = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) hs += fx[t] * (int)src[(ptrdiff_t)y * ss + x - 3 + t];
        res = (1 << (7 - a.round1)) * ((hs + ((1 << a.round0) >> 1)) >> a.round0) + round_offset;
    } else if (a.variant == 2) {
        int vs = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) vs += fy[k] * (int)src[(ptrdiff_t)(y + k - 3) * ss + x];
        vs *= 1 << (7 - a.round0);
        res = ((vs + ((1 << a.round1) >> 1)) >> a.round1) + round_offset;
    } else
        res = (int)(uint16_t)(((int)src[(ptrdiff_t)y * ss + x] << round_bits) + round_offset);
    if (!a.do_average) { cb[(size_t)y * cbs + x] = (uint16_t)res; return; }
    int tmp = cb[(size_t)y * cbs + x];
    tmp = a.use_jnt ? (tmp * a.fwd + res * a.bck) >> 4 : (tmp + res) >> 1;
    tmp -= round_offset;
    dst[(size_t)y * ds + x] = (PIX)clip_px((tmp + ((1 << round_bits) >> 1)) >> round_bits, a.bd);
}

// ------------------------------------------------------------------------------------------------ difference-weighted compound mask, d16 blend
// mask[i * w + j] = 38 + (|a - b| rounded by `round`, shifted by `shift`) / 16, clamped to [0, 64], inverted for DIFFWTD_38_INV
template <typename T>
__global__ void __launch_bounds__(256)
diffwtd_mask_kernel(uint8_t* __restrict__ mask, const T* __restrict__ a, int as, const T* __restrict__ b, int bs, int w, int h, int inverse, int round, int shift) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    int d = abs((int)a[(size_t)y * as + x] - (int)b[(size_t)y * bs + x]);
    if (round > 0) d = (d + (1 << (round - 1))) >> round;
    d >>= shift;
    const int m = min(max(38 + d / 16, 0), 64);
    mask[i] = (uint8_t)(inverse ? 64 - m : m);
}
template <typename PIX>
__global__ void __launch_bounds__(256)
blend_d16_kernel(PIX* __restrict__ dst, int ds, const uint16_t* __restrict__ s0, int s0s, const uint16_t* __restrict__ s1, int s1s, const uint8_t* __restrict__ mask, int ms, int w,
                 int h, int subw, int subh, int round_offset, int round_bits, int bd) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    int m;
    if (!subw && !subh) m = mask[(size_t)y * ms + x];
    else if (subw && subh) m = (mask[(size_t)(2 * y) * ms + 2 * x] + mask[(size_t)(2 * y + 1) * ms + 2 * x] + mask[(size_t)(2 * y) * ms + 2 * x + 1] + mask[(size_t)(2 * y + 1) * ms + 2 * x + 1] + 2) >> 2;
    else if (subw) m = (mask[(size_t)y * ms + 2 * x] + mask[(size_t)y * ms + 2 * x + 1] + 1) >> 1;
    else m = (mask[(size_t)(2 * y) * ms + x] + mask[(size_t)(2 * y + 1) * ms + x] + 1) >> 1;
    int res = (m * (int)s0[(size_t)y * s0s + x] + (64 - m) * (int)s1[(size_t)y * s1s + x]) >> 6;
    res -= round_offset;
    dst[(size_t)y * ds + x] = (PIX)clip_px((res + ((1 << round_bits) >> 1)) >> round_bits, bd);
}

}  // namespace

extern "C" int svt_hip_launch_diffwtd_mask(hipStream_t st, int elem_bytes, uint8_t* mask, const void* a, int as, const void* b, int bs, int w, int h, int inverse, int round,
                                           int shift) {
    const dim3 grid((w * h + 255) / 256);
    if (elem_bytes == 1) hipLaunchKernelGGL(diffwtd_mask_kernel<uint8_t>, grid, dim3(256), 0, st, mask, (const uint8_t*)a, as, (const uint8_t*)b, bs, w, h, inverse, round, shift);
    else hipLaunchKernelGGL(diffwtd_mask_kernel<uint16_t>, grid, dim3(256), 0, st, mask, (const uint16_t*)a, as, (const uint16_t*)b, bs, w, h, inverse, round, shift);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_blend_d16(hipStream_t st, int pix_bytes, int bd, void* dst, int ds, const uint16_t* s0, int s0s, const uint16_t* s1, int s1s, const uint8_t* mask,
                                        int ms, int w, int h, int subw, int subh, int round0, int round1) {
    const dim3 grid((w + 63) / 64, (h + 3) / 4);
    const int  b = pix_bytes == 1 ? 8 : bd, offset_bits = b + 14 - round0, round_offset = (1 << (offset_bits - round1)) + (1 << (offset_bits - round1 - 1));
    if (pix_bytes == 1) hipLaunchKernelGGL(blend_d16_kernel<uint8_t>, grid, dim3(256), 0, st, (uint8_t*)dst, ds, s0, s0s, s1, s1s, mask, ms, w, h, subw, subh, round_offset, 14 - round0 - round1, b);
    else hipLaunchKernelGGL(blend_d16_kernel<uint16_t>, grid, dim3(256), 0, st, (uint16_t*)dst, ds, s0, s0s, s1, s1s, mask, ms, w, h, subw, subh, round_offset, 14 - round0 - round1, b);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_jnt_convolve(hipStream_t st, int pix_bytes, int bd, int variant, const void* src, int ss, void* dst, int ds, uint16_t* cb, int cbs,
                                           const int16_t* taps, int w, int h, int round0, int round1, int do_average, int use_jnt, int fwd, int bck) {
    const dim3    grid((w + 63) / 64, (h + 3) / 4);
    const JntArgs a = {variant, w, h, round0, round1, do_average, use_jnt, fwd, bck, pix_bytes == 1 ? 8 : bd};
    if (pix_bytes == 1) hipLaunchKernelGGL(jnt_convolve_kernel<uint8_t>, grid, dim3(256), 0, st, (const uint8_t*)src, ss, (uint8_t*)dst, ds, cb, cbs, taps, a);
    else hipLaunchKernelGGL(jnt_convolve_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)src, ss, (uint16_t*)dst, ds, cb, cbs, taps, a);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_repack64(hipStream_t st, int32_t* coeff, int rows, int per_block, int nblk) {
    if (nblk <= 0) return 0;
    hipLaunchKernelGGL(repack64_kernel, dim3(nblk), dim3(1024), 0, st, coeff, rows, per_block);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_block_mean(hipStream_t st, const uint8_t* plane, int stride, const int32_t* offs, int n, int mode, int w, int h, uint64_t* out) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(block_mean_kernel, dim3(n), dim3(64), 0, st, plane, stride, offs, mode, w, h, out);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_ext_sad_16(hipStream_t st, const uint8_t* src, int ss, const uint8_t* ref, int rs, const SvtHipExtSadJob* jobs, int n, uint32_t* state) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(ext_sad_16_kernel, dim3(n), dim3(64), 0, st, src, ss, ref, rs, jobs, state);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_ext_sad_32_64(hipStream_t st, uint32_t* state, const uint32_t* mv, int n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(ext_sad_32_64_kernel, dim3((n + 63) / 64), dim3(64), 0, st, state, mv, n);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_cdef_dist(hipStream_t st, int pix_bytes, const void* dst, int dstride, const void* src, const uint8_t* list, int n, int bw_log2, int bh_log2,
                                        int cs, int pli, uint64_t* out) {
    if (pix_bytes == 1) hipLaunchKernelGGL(cdef_dist_kernel<uint8_t>, dim3(1), dim3(256), 0, st, (const uint8_t*)dst, dstride, (const uint8_t*)src, list, n, bw_log2, bh_log2, cs, pli, out);
    else hipLaunchKernelGGL(cdef_dist_kernel<uint16_t>, dim3(1), dim3(256), 0, st, (const uint16_t*)dst, dstride, (const uint16_t*)src, list, n, bw_log2, bh_log2, cs, pli, out);
    return (int)hipGetLastError();
}
static int one_dual_step(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, int* lev0, int* lev1, int nb, int start_gi, int end_gi, uint64_t* best,
                         uint64_t* tot, uint64_t* out, int n_shift) {
    const int ng = end_gi - start_gi > 0 ? end_gi - start_gi : 0;
    hipLaunchKernelGGL(one_dual_best_kernel, dim3(sb_count > 0 ? (sb_count + 255) / 256 : 1), dim3(256), 0, st, mse0, mse1, sb_count, lev0, lev1, nb, best, tot, ng * ng);
    if (ng > 0 && sb_count > 0)
        hipLaunchKernelGGL(one_dual_total_kernel, dim3(ng, (sb_count + kDualChunk - 1) / kDualChunk), dim3(256), 0, st, mse0, mse1, sb_count, best, start_gi, ng,
                           (unsigned long long*)tot);
    hipLaunchKernelGGL(one_dual_pick_kernel, dim3(1), dim3(256), 0, st, tot, start_gi, ng, nb, lev0, lev1, out, n_shift);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_search_one_dual(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, int* lev0, int* lev1, int nb, int start_gi, int end_gi,
                                              uint64_t* best, uint64_t* tot, uint64_t* out) {
    return one_dual_step(st, mse0, mse1, sb_count, lev0, lev1, nb, start_gi, end_gi, best, tot, out, 0);
}
// joint_strength_search_dual (EbEncCdef.c:1140-1164): nb greedy steps, then 4 * nb refinement steps (each preceded by the shift, which the previous
// step's last kernel performs), all queued back to back: three launches per step
extern "C" int svt_hip_launch_joint_strength_search(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, int* lev0, int* lev1, int nb, int start_gi,
                                                    int end_gi, uint64_t* best, uint64_t* tot, uint64_t* out) {
    for (int i = 0; i < nb; i++) {
        const int rc = one_dual_step(st, mse0, mse1, sb_count, lev0, lev1, i, start_gi, end_gi, best, tot, out, i == nb - 1 ? nb : 0);
        if (rc) return rc;
    }
    for (int i = 0; i < 4 * nb; i++) {
        const int rc = one_dual_step(st, mse0, mse1, sb_count, lev0, lev1, nb - 1, start_gi, end_gi, best, tot, out, i == 4 * nb - 1 ? 0 : nb);
        if (rc) return rc;
    }
    return (int)hipGetLastError();
}
extern "C" size_t svt_hip_joint_state_bytes(void) { return sizeof(JointState); }
// out[c] = {total, lev0[8], lev1[8]} as 64-bit words (17 per chain) is assembled by the caller from the state; here: clear, 40 steps
// form: SVT_HIP_CDEF_SELECT_* of svt_hip.h (-1: SVT_HIP_CDEF_SELECT=steps|resident from the environment, else steps)
extern "C" int svt_hip_strength_select_is_resident(int form, int sb_count) {
    static int env_form = -1;
    if (env_form < 0) { const char* e = getenv("SVT_HIP_CDEF_SELECT"); env_form = e && !strcmp(e, "resident") ? 1 : 0; }
    return (form < 0 ? env_form : form) == 1 && sb_count > 0 && sb_count <= kResMaxSb;
}
// n_pics pictures of the same size (same sb_count and strength range) in one set of launches; states[i]: SVT_HIP_CDEF_SELECT_STATE_BYTES each
extern "C" int svt_hip_launch_strength_select_multi(hipStream_t st, int n_pics, const uint64_t* const* mse0, const uint64_t* const* mse1, int sb_count, int start_gi, int end_gi,
                                                    void* const* states, int resident_form) {
    const int ng = end_gi - start_gi;
    for (int i = 0; i < n_pics; i++)
        if (hipMemsetAsync(states[i], 0, offsetof(JointState, partial), st) != hipSuccess) return (int)hipGetLastError();
    if (ng <= 0 || n_pics <= 0) return 0;
    // resident_form: the one-launch form (joint_resident_kernel), for pictures of up to kResMaxSb filter blocks.  It needs its 256 workgroups on the chip together:
    // two of them launched on different streams could each hold a part of the compute units and wait for the rest, so the C ABI layer issues them on one stream
    // per device (svt_hip_api.cpp); here they are simply in stream order.
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)joint_resident_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResLdsBytes);
        (void)hipFuncSetAttribute((const void*)joint_resident_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResLdsBytes);
        (void)hipFuncSetAttribute((const void*)joint_resident_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResLdsBytes);
        attr_set = true;
    }
    const char* early_env = getenv("SVT_HIP_CDEF_SELECT_EARLY");   // read per call: 0 = every chain runs its 5 nb steps (the reference's loop as written); tests compare both
    const int early = !(early_env && !atoi(early_env));
    static int forced = -1;   // debug: SVT_HIP_CDEF_SELECT_SLICES
    if (forced < 0) { const char* e = getenv("SVT_HIP_CDEF_SELECT_SLICES"); forced = e ? atoi(e) : 0; }
    int slices = forced > 0 ? forced : (sb_count + 31) / 32;
    slices = slices < 1 ? 1 : (slices > kJointMaxSlices ? kJointMaxSlices : slices);
    for (int p0 = 0; p0 < n_pics; p0 += kJointMaxPics) {
        const int np = min(kJointMaxPics, n_pics - p0);
        JointPics P = {};
        for (int i = 0; i < np; i++) { P.mse0[i] = mse0[p0 + i]; P.mse1[i] = mse1[p0 + i]; P.S[i] = (JointState*)states[p0 + i]; }
        const bool resident = resident_form && sb_count > 0 && sb_count <= kResMaxSb;
        if (sb_count > 0 && !resident) hipLaunchKernelGGL(joint_init_kernel, dim3(min((sb_count * 64 + 255) / 256, 64), np), dim3(256), 0, st, P, sb_count * 64);
        if (resident) {
            for (int i = 0; i < np; i++) {
                hipLaunchKernelGGL(joint_transpose_kernel, dim3((sb_count + 63) / 64, 2), dim3(256), 0, st, P.mse0[i], P.mse1[i], sb_count, P.S[i]);
                hipLaunchKernelGGL(joint_resident_kernel<0>, dim3(kResWgs), dim3(1024), kResLdsBytes, st, sb_count, start_gi, ng, P.S[i], early);
                hipLaunchKernelGGL(joint_resident_kernel<1>, dim3(kResWgs), dim3(1024), kResLdsBytes, st, sb_count, start_gi, ng, P.S[i], early);
                hipLaunchKernelGGL(joint_resident_kernel<2>, dim3(kResWgs), dim3(1024), kResLdsBytes, st, sb_count, start_gi, ng, P.S[i], early);
            }
            continue;
        }
        for (int step = 0; step < 40; step++) {
            hipLaunchKernelGGL(joint_partial_kernel, dim3(slices, np, 4), dim3(1024), 0, st, P, sb_count, start_gi, ng, step);
            hipLaunchKernelGGL(joint_reduce_kernel, dim3(64, np, 4), dim3(256), 0, st, P, slices, start_gi, ng, step, early);
        }
    }
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_strength_select(hipStream_t st, const uint64_t* mse0, const uint64_t* mse1, int sb_count, int start_gi, int end_gi, void* state) {
    return svt_hip_launch_strength_select_multi(st, 1, &mse0, &mse1, sb_count, start_gi, end_gi, &state, 0);
}
extern "C" int svt_hip_launch_sgr_flt_proj(hipStream_t st, int pix_bytes, const void* src, int ss, const void* dat, int ds, const int32_t* f0, int f0s, const int32_t* f1, int f1s,
                                           int w, int h, int r0, int r1, int mode, int xq0, int xq1, long long* acc, int32_t* xq_out) {
    const int blocks = h < 256 ? h : 256;
    if (pix_bytes == 1)
        hipLaunchKernelGGL(sgr_flt_proj_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (const uint8_t*)src, ss, (const uint8_t*)dat, ds, f0, f0s, f1, f1s, w, h, r0, r1, mode, xq0, xq1, acc);
    else
        hipLaunchKernelGGL(sgr_flt_proj_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, (const uint16_t*)src, ss, (const uint16_t*)dat, ds, f0, f0s, f1, f1s, w, h, r0, r1, mode, xq0, xq1, acc);
    if (mode == 0) hipLaunchKernelGGL(sgr_flt_solve_kernel, dim3(1), dim3(1), 0, st, acc, w * h, r0, r1, xq_out);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_convolve8(hipStream_t st, int vert, const uint8_t* src, int ss, uint8_t* dst, int ds, const int16_t* filters, int q0, int step, int w, int h) {
    const dim3 grid((w + 63) / 64, (h + 3) / 4);
    if (vert) hipLaunchKernelGGL(convolve8_kernel<true>, grid, dim3(256), 0, st, src, ss, dst, ds, filters, q0, step, w, h);
    else hipLaunchKernelGGL(convolve8_kernel<false>, grid, dim3(256), 0, st, src, ss, dst, ds, filters, q0, step, w, h);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_wiener_convolve(hipStream_t st, int pix_bytes, int bd, const void* src, int ss, void* dst, int ds, const int16_t* taps, int w, int h, int round0,
                                              int round1) {
    const dim3 grid((w + 63) / 64, (h + 3) / 4);
    if (pix_bytes == 1) hipLaunchKernelGGL(wiener_convolve_kernel<uint8_t>, grid, dim3(256), 0, st, (const uint8_t*)src, ss, (uint8_t*)dst, ds, taps, w, h, round0, round1, 8);
    else hipLaunchKernelGGL(wiener_convolve_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)src, ss, (uint16_t*)dst, ds, taps, w, h, round0, round1, bd);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(percall2)
