// cdef.hip — CDEF strength search (distortion table per 64x64 filter block) and frame application, gfx950.
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Encoder/Codec/EbCdefProcess.c:80-475      cdef_seg_search / cdef_seg_search16bit (per-fb loops)
//   Common/Codec/EbCdef.c:132,202,294         svt_cdef_find_dir_c, svt_cdef_filter_block_c, svt_cdef_filter_fb
//   Encoder/Codec/EbEncCdef.c:25-220          dist_8x8_* / mse_* / compute_cdef_dist_{8bit,}_c
//   Encoder/Codec/EbEncCdef.c:292-1031        svt_av1_cdef_frame / av1_cdef_frame16bit
//
// Search design.  The reference filters every block once per strength pair (64 x 3 planes).  The
// filter sum is separable: sum = primary(pri, dir) + secondary(sec, dir or 0), and min/max do not
// depend on the strengths — so per pixel we compute the 15 primary sums and the 2 x 3 secondary
// sums once and then only combine/round/clamp 64 times.  One wave owns one 8x8 luma block (lane =
// pixel) or four 4x4 chroma blocks; the 64 filtered values of a lane go to a [strength][lane] byte
// table in LDS, after which lane g becomes the reducer of strength g: sum(y), sum(y^2), sum(y*s)
// come from v_dot4_u32_u8 over 16 dwords (v_dot2 for 16-bit), and the luma perceptual distortion
// (FP64, sqrt, floor — Encoder/Codec/EbEncCdef.c:100-104) is evaluated per (block, strength) in the
// reference's operation order with contraction off.
// The staging tile marks everything outside the picture CDEF_VERY_LARGE exactly like the
// reference's inbuf (EbCdefProcess.c:210-226).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "svt_hip_internal.h"
#include "lds_stage.h"

namespace {

constexpr int kVeryLarge = 16384;  // CDEF_VERY_LARGE, Common/Codec/EbCdef.h:37
constexpr int kVB = 3, kHB = 8;    // CDEF_VBORDER / CDEF_HBORDER

__device__ __constant__ int8_t kDirDy[8][2] = {{-1, -2}, {0, -1}, {0, 0}, {0, 1}, {1, 2}, {1, 2}, {1, 2}, {1, 2}};
__device__ __constant__ int8_t kDirDx[8][2] = {{1, 2}, {1, 2}, {1, 2}, {1, 2}, {1, 2}, {0, 1}, {0, 0}, {0, -1}};

__device__ __forceinline__ int msb(int n) { return 31 - __clz(n); }
// Common/Codec/EbCdef.c:87-93; shift precomputed by the caller (uniform per strength)
__device__ __forceinline__ int constrain(int diff, int threshold, int shift) {
    const int a = abs(diff);
    const int v = min(a, max(0, threshold - (a >> shift)));
    return diff < 0 ? -v : v;
}
__device__ __forceinline__ int adjust_strength(int strength, int var) {  // EbCdef.c:112-116
    const int i = (var >> 6) ? min(msb(var >> 6), 12) : 0;
    return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

template <typename PIX>
__device__ __forceinline__ void stage_tile(uint16_t* tile, int tstride, const PIX* plane, int stride, int pw, int ph, int x0, int y0,
                                           int tw, int th, int tid, int nt) {
    // tile covers [x0 - 8, x0 + tw + 8) x [y0 - 3, y0 + th + 3)
    const int cols = tw + 2 * kHB, rows = th + 2 * kVB;
    batched_stage<8, uint16_t>(rows * cols, tid, nt,
        [&](int i) {
            const int r = i / cols, c = i - r * cols;
            const int x = x0 - kHB + c, y = y0 - kVB + r;
            const uint16_t v = (uint16_t)plane[(size_t)min(max(y, 0), ph - 1) * stride + min(max(x, 0), pw - 1)];   // (unconditional: see batched_stage)
            return (x >= 0 && y >= 0 && x < pw && y < ph) ? v : (uint16_t)kVeryLarge;
        },
        [&](int i, uint16_t v) { const int r = i / cols, c = i - r * cols; tile[r * tstride + c] = v; });
}

// svt_cdef_find_dir_c (EbCdef.c:132-196) for one 8x8 block by one wave; lane = pixel on entry.
// The 8 x 15 directional line sums are independent dot products: after the 64 samples are parked in LDS,
// lane (d & 3, bin) sums the <= 8 pixels of its line (pixel lists precomputed below), squares and weights it
// (div_table), and a 16-lane DPP row reduction yields the cost of direction d; two passes cover d = 0..3, 4..7.
// xs: per-wave LDS scratch of 64 ints.  Returns dir, writes var (both wave-uniform).
__device__ __constant__ uint8_t kDirBinPix[8][16][8] = {
    {{0,255,255,255,255,255,255,255},{1,8,255,255,255,255,255,255},{2,9,16,255,255,255,255,255},{3,10,17,24,255,255,255,255},{4,11,18,25,32,255,255,255},{5,12,19,26,33,40,255,255},{6,13,20,27,34,41,48,255},{7,14,21,28,35,42,49,56},{15,22,29,36,43,50,57,255},{23,30,37,44,51,58,255,255},{31,38,45,52,59,255,255,255},{39,46,53,60,255,255,255,255},{47,54,61,255,255,255,255,255},{55,62,255,255,255,255,255,255},{63,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255}},
    {{0,1,255,255,255,255,255,255},{2,3,8,9,255,255,255,255},{4,5,10,11,16,17,255,255},{6,7,12,13,18,19,24,25},{14,15,20,21,26,27,32,33},{22,23,28,29,34,35,40,41},{30,31,36,37,42,43,48,49},{38,39,44,45,50,51,56,57},{46,47,52,53,58,59,255,255},{54,55,60,61,255,255,255,255},{62,63,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255}},
    {{0,1,2,3,4,5,6,7},{8,9,10,11,12,13,14,15},{16,17,18,19,20,21,22,23},{24,25,26,27,28,29,30,31},{32,33,34,35,36,37,38,39},{40,41,42,43,44,45,46,47},{48,49,50,51,52,53,54,55},{56,57,58,59,60,61,62,63},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255}},
    {{6,7,255,255,255,255,255,255},{4,5,14,15,255,255,255,255},{2,3,12,13,22,23,255,255},{0,1,10,11,20,21,30,31},{8,9,18,19,28,29,38,39},{16,17,26,27,36,37,46,47},{24,25,34,35,44,45,54,55},{32,33,42,43,52,53,62,63},{40,41,50,51,60,61,255,255},{48,49,58,59,255,255,255,255},{56,57,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255}},
    {{7,255,255,255,255,255,255,255},{6,15,255,255,255,255,255,255},{5,14,23,255,255,255,255,255},{4,13,22,31,255,255,255,255},{3,12,21,30,39,255,255,255},{2,11,20,29,38,47,255,255},{1,10,19,28,37,46,55,255},{0,9,18,27,36,45,54,63},{8,17,26,35,44,53,62,255},{16,25,34,43,52,61,255,255},{24,33,42,51,60,255,255,255},{32,41,50,59,255,255,255,255},{40,49,58,255,255,255,255,255},{48,57,255,255,255,255,255,255},{56,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255}},
    {{48,56,255,255,255,255,255,255},{32,40,49,57,255,255,255,255},{16,24,33,41,50,58,255,255},{0,8,17,25,34,42,51,59},{1,9,18,26,35,43,52,60},{2,10,19,27,36,44,53,61},{3,11,20,28,37,45,54,62},{4,12,21,29,38,46,55,63},{5,13,22,30,39,47,255,255},{6,14,23,31,255,255,255,255},{7,15,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255}},
    {{0,8,16,24,32,40,48,56},{1,9,17,25,33,41,49,57},{2,10,18,26,34,42,50,58},{3,11,19,27,35,43,51,59},{4,12,20,28,36,44,52,60},{5,13,21,29,37,45,53,61},{6,14,22,30,38,46,54,62},{7,15,23,31,39,47,55,63},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255}},
    {{0,8,255,255,255,255,255,255},{1,9,16,24,255,255,255,255},{2,10,17,25,32,40,255,255},{3,11,18,26,33,41,48,56},{4,12,19,27,34,42,49,57},{5,13,20,28,35,43,50,58},{6,14,21,29,36,44,51,59},{7,15,22,30,37,45,52,60},{23,31,38,46,53,61,255,255},{39,47,54,62,255,255,255,255},{55,63,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255},{255,255,255,255,255,255,255,255}}};
__device__ __constant__ int kDirBinWeight[8][16] = {{840,420,280,210,168,140,120,105,120,140,168,210,280,420,840,0},{420,210,140,105,105,105,105,105,140,210,420,0,0,0,0,0},{105,105,105,105,105,105,105,105,0,0,0,0,0,0,0,0},{420,210,140,105,105,105,105,105,140,210,420,0,0,0,0,0},{840,420,280,210,168,140,120,105,120,140,168,210,280,420,840,0},{420,210,140,105,105,105,105,105,140,210,420,0,0,0,0,0},{105,105,105,105,105,105,105,105,0,0,0,0,0,0,0,0},{420,210,140,105,105,105,105,105,140,210,420,0,0,0,0,0}};
__device__ __forceinline__ int row_sum16(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror
    return v;
}
__device__ __forceinline__ int find_dir_wave(int x_px, int lane, int* xs, int& var_out) {
    xs[lane] = x_px - 128;
    __builtin_amdgcn_wave_barrier();
    int costs[8];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int d = 4 * pass + (lane >> 4), b = lane & 15;
        const uint2 pix = *(const uint2*)kDirBinPix[d][b];
        int s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t idx = ((k < 4 ? pix.x : pix.y) >> (8 * (k & 3))) & 0xFFu;
            s += idx < 64 ? xs[idx] : 0;
        }
        const int c = row_sum16(s * s * kDirBinWeight[d][b]);
#pragma unroll
        for (int q = 0; q < 4; q++) costs[4 * pass + q] = __builtin_amdgcn_readlane(c, 16 * q);
    }
    __builtin_amdgcn_wave_barrier();
    int best = 0, best_cost = 0;
#pragma unroll
    for (int d = 0; d < 8; d++)
        if (costs[d] > best_cost) { best_cost = costs[d]; best = d; }
    int orth = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) if (d == ((best + 4) & 7)) orth = costs[d];
    var_out = (best_cost - orth) >> 10;
    return best;
}

// ---- packed 16-bit search arithmetic ------------------------------------------------------------
// Everything the 64 strength pairs need for one pixel: primary sums for pri = 1..15, secondary sums
// for sec in {1,2,4} with the block's direction (A) and with direction 0 (B, used when pri == 0), and
// the two min/max pairs.  All quantities fit int16 (taps <= 16384, sums <= 12 * 60), so the taps are
// kept as PAIRS in one VGPR (the two mirrored taps of a (direction, distance), which share a weight)
// and constrain() runs on both halves at once:
//   |d|            v_pk_sub_i16 / v_pk_max_i16, once per pair
//   t - (|d|>>s)   v_pk_lshrrev_b16 + v_pk_sub_u16 with unsigned saturation (= max(0, .))
//   min(|d|, .)    v_pk_min_u16
//   weight * sign  folded into a per-pair constant (+-w, +-w), applied together with the horizontal
//                  add by one v_dot2_i32_i16 per pair and strength.
// A CDEF_VERY_LARGE tap yields |d| >> s > t for every strength (EbCdef.c:87-93 relies on the same),
// so it contributes 0 without a test.
typedef short  s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 dup2(int v) { return s16x2{(short)v, (short)v}; }
__device__ __forceinline__ s16x2 pack2(int lo, int hi) { return s16x2{(short)lo, (short)hi}; }
__device__ __forceinline__ u16x2 as_u(s16x2 v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ s16x2 as_s(u16x2 v) { return __builtin_bit_cast(s16x2, v); }

struct TapPair {
    u16x2 a;    // |tap - x| of both taps
    s16x2 sg;   // 0 / -1 per half: sign of (tap - x)
};
__device__ __forceinline__ TapPair make_pair(int t0, int t1, s16x2 x2) {
    const s16x2 d = pack2(t0, t1) - x2;
    TapPair p;
    p.a = as_u(__builtin_elementwise_max(d, -d));
    p.sg = d >> 15;
    return p;
}
// (w * sign0, w * sign1)
__device__ __forceinline__ s16x2 signed_weight(const TapPair& p, int w) { return (dup2(w) ^ p.sg) - p.sg; }
// sum over the two taps of weight * constrain(tap - x, t, shift), accumulated into acc
__device__ __forceinline__ int constrain_pair(const TapPair& p, s16x2 sw, int t, int shift, int acc) {
    const u16x2 hi = __builtin_elementwise_sub_sat(as_u(dup2(t)), p.a >> (unsigned short)shift);
    const u16x2 v = __builtin_elementwise_min(p.a, hi);
    return __builtin_amdgcn_sdot2(as_s(v), sw, acc, false);
}
__device__ __forceinline__ void minmax_pair(int t0, int t1, s16x2& mn, s16x2& mx) {
    const s16x2 v = pack2(t0, t1);
    mn = __builtin_elementwise_min(mn, v);
    mx = __builtin_elementwise_max(mx, v & dup2(0x3FFF));   // CDEF_VERY_LARGE (0x4000) never wins the max (EbCdef.c:229-247)
}

struct PixelTerms {
    s16x2 x2;          // (x, x)
    s16x2 pri2[16];    // (pri sum, pri sum); [0] = 0
    s16x2 secA[2], secB[2];   // [0] = (0, sec 1), [1] = (sec 2, sec 4)
    s16x2 mnA, mxA, mnB, mxB; // duplicated in both halves
};

// tile points at the pixel; tstride in elements.  t_of[idx] = effective primary strength of index idx
// (luma: adjusted by the block variance), cs = coeff_shift, damping already includes "+ cs - (pli != 0)".
__device__ __forceinline__ void pixel_terms(const uint16_t* px, int tstride, int dir, const int (&t_of)[16], int cs, int damping,
                                            PixelTerms& T) {
    const int x = (int)(int16_t)px[0];
    const s16x2 x2 = dup2(x);
    T.x2 = x2;
    TapPair pp[2], sa[2][2], sb[2][2];
    s16x2 mnA = x2, mxA = x2, mnB = x2, mxB = x2;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int o = kDirDy[dir][k] * tstride + kDirDx[dir][k];
        const int d2 = (dir + 2) & 7, d6 = (dir + 6) & 7;
        const int o2 = kDirDy[d2][k] * tstride + kDirDx[d2][k], o6 = kDirDy[d6][k] * tstride + kDirDx[d6][k];
        const int p0 = px[o], p1 = px[-o], a0 = px[o2], a1 = px[-o2], a2 = px[o6], a3 = px[-o6];
        pp[k] = make_pair(p0, p1, x2); sa[k][0] = make_pair(a0, a1, x2); sa[k][1] = make_pair(a2, a3, x2);
        minmax_pair(p0, p1, mnA, mxA); minmax_pair(a0, a1, mnA, mxA); minmax_pair(a2, a3, mnA, mxA);
        // direction 0 variant (pri == 0 -> filter_block is called with dir 0, EbCdef.c:371): its primary taps only feed min/max
        const int ob = kDirDy[0][k] * tstride + kDirDx[0][k];
        const int ob2 = kDirDy[2][k] * tstride + kDirDx[2][k], ob6 = kDirDy[6][k] * tstride + kDirDx[6][k];
        const int q0 = px[ob], q1 = px[-ob], b0 = px[ob2], b1 = px[-ob2], b2 = px[ob6], b3 = px[-ob6];
        sb[k][0] = make_pair(b0, b1, x2); sb[k][1] = make_pair(b2, b3, x2);
        minmax_pair(q0, q1, mnB, mxB); minmax_pair(b0, b1, mnB, mxB); minmax_pair(b2, b3, mnB, mxB);
    }
    // fold the two halves of the running min / max and duplicate
    T.mnA = __builtin_elementwise_min(mnA, mnA.yx); T.mxA = __builtin_elementwise_max(mxA, mxA.yx);
    T.mnB = __builtin_elementwise_min(mnB, mnB.yx); T.mxB = __builtin_elementwise_max(mxB, mxB.yx);
    // primary: eb_cdef_pri_taps (EbCdef.c:198) = {4, 2} or {3, 3} by the parity of (t >> cs)
    const s16x2 w40 = signed_weight(pp[0], 4), w30 = signed_weight(pp[0], 3), w21 = signed_weight(pp[1], 2), w31 = signed_weight(pp[1], 3);
    T.pri2[0] = dup2(0);
#pragma unroll
    for (int idx = 1; idx < 16; idx++) {
        const int t = t_of[idx];   // wave-uniform
        // luma: adjust_strength() maps the 16 frame-header strengths onto fewer effective ones in a low-variance block (it is monotone, so equal values are
        // neighbours): the sum of a repeated value is the previous index's
        if (t == t_of[idx - 1]) { T.pri2[idx] = T.pri2[idx - 1]; continue; }
        int s = 0;
        if (t) {
            const int shift = max(0, damping - msb(t));
            const bool odd = (t >> cs) & 1;
            s = constrain_pair(pp[0], odd ? w30 : w40, t, shift, 0);
            s = constrain_pair(pp[1], odd ? w31 : w21, t, shift, s);
        }
        T.pri2[idx] = dup2(s);
    }
    // secondary: eb_cdef_sec_taps {2, 1}; strengths 1, 2, 4 (index 3 means 4: "sec += sec == 3")
    s16x2 wa[2][2], wb[2][2];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int h = 0; h < 2; h++) { wa[k][h] = signed_weight(sa[k][h], 2 - k); wb[k][h] = signed_weight(sb[k][h], 2 - k); }
    int a[3], b[3];
#pragma unroll
    for (int si = 0; si < 3; si++) {
        const int st = (1 << si) << cs;
        const int shift = max(0, damping - msb(st));
        int va = 0, vb = 0;
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int h = 0; h < 2; h++) { va = constrain_pair(sa[k][h], wa[k][h], st, shift, va); vb = constrain_pair(sb[k][h], wb[k][h], st, shift, vb); }
        a[si] = va; b[si] = vb;
    }
    T.secA[0] = pack2(0, a[0]); T.secA[1] = pack2(a[1], a[2]);
    T.secB[0] = pack2(0, b[0]); T.secB[1] = pack2(b[1], b[2]);
}

// filtered values of strengths (pri_idx, 2 * pair) and (pri_idx, 2 * pair + 1): y = x + ((8 + sum - (sum < 0)) >> 4), clamped
__device__ __forceinline__ s16x2 combine2(const PixelTerms& T, int pri_idx, int pair) {
    const s16x2 sum = T.pri2[pri_idx] + (pri_idx ? T.secA[pair] : T.secB[pair]);
    const s16x2 y = T.x2 + ((sum + (sum >> 15) + dup2(8)) >> 4);
    return pri_idx ? __builtin_elementwise_min(__builtin_elementwise_max(y, T.mnA), T.mxA)
                   : __builtin_elementwise_min(__builtin_elementwise_max(y, T.mnB), T.mxB);
}

__device__ __forceinline__ void tap_minmax(int v, int& mn, int& mx) {
    if (v != kVeryLarge) mx = max(mx, v);
    mn = min(mn, v);
}

// One pixel, one (pri, sec, dir): svt_cdef_filter_block_c (EbCdef.c:202-257).  The two mirrored taps of every (direction, distance) share a weight,
// so they are handled as a packed pair with the search's helpers (|d|, constrain and the weighted sum of both taps in ~9 instructions).
__device__ __forceinline__ int filter_px_single(const uint16_t* px, int tstride, int pri, int sec, int dir, int cs, int damping, int sec_damping = -1) {
    const int x = (int)(int16_t)px[0];
    const s16x2 x2 = dup2(x);
    s16x2 mn = x2, mx = x2;
    int sum = 0;
    const int pshift = pri ? max(0, damping - msb(pri)) : 0, sshift = sec ? max(0, (sec_damping < 0 ? damping : sec_damping) - msb(sec)) : 0;
    const int w0 = ((pri >> cs) & 1) ? 3 : 4, w1 = ((pri >> cs) & 1) ? 3 : 2;
    const int d2 = (dir + 2) & 7, d6 = (dir + 6) & 7;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int o = kDirDy[dir][k] * tstride + kDirDx[dir][k];
        const int p0 = px[o], p1 = px[-o];
        minmax_pair(p0, p1, mn, mx);
        if (pri) {   // wave-uniform
            const TapPair pp = make_pair(p0, p1, x2);
            sum = constrain_pair(pp, signed_weight(pp, k ? w1 : w0), pri, pshift, sum);
        }
        const int o2 = kDirDy[d2][k] * tstride + kDirDx[d2][k], o6 = kDirDy[d6][k] * tstride + kDirDx[d6][k];
        const int s0 = px[o2], s1 = px[-o2], s2 = px[o6], s3 = px[-o6];
        minmax_pair(s0, s1, mn, mx); minmax_pair(s2, s3, mn, mx);
        if (sec) {
            const TapPair a = make_pair(s0, s1, x2), b = make_pair(s2, s3, x2);
            sum = constrain_pair(a, signed_weight(a, 2 - k), sec, sshift, sum);
            sum = constrain_pair(b, signed_weight(b, 2 - k), sec, sshift, sum);
        }
    }
    const int lo = min((int)mn.x, (int)mn.y), hi = max((int)mx.x, (int)mx.y);
    const int y = x + ((8 + sum - (sum < 0)) >> 4);
    return min(max(y, lo), hi);
}

// sum(a), sum(a*a), sum(a*b) over 64 samples held as packed rows in LDS
template <typename PIX> struct Dots;
template <> struct Dots<uint8_t> {
    static __device__ __forceinline__ void run(const uint8_t* a, const uint8_t* b, uint32_t& sa, uint32_t& saa, uint32_t& sab) {
        const uint4* pa = (const uint4*)a; const uint4* pb = (const uint4*)b;   // rows are 16-byte aligned (ytab / stab declarations): four 128-bit LDS reads each
        sa = saa = sab = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 qa = pa[k], qb = pb[k];
            const uint32_t va[4] = {qa.x, qa.y, qa.z, qa.w}, vb[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
            for (int m = 0; m < 4; m++) {
                sa  = __builtin_amdgcn_udot4(va[m], 0x01010101u, sa, false);
                saa = __builtin_amdgcn_udot4(va[m], va[m], saa, false);
                sab = __builtin_amdgcn_udot4(va[m], vb[m], sab, false);
            }
        }
    }
};
template <> struct Dots<uint16_t> {
    // packed pairs through v_dot2_u32_u16; 64 samples of <= 12 bits: every sum stays below 2^32
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void run(const uint16_t* a, const uint16_t* b, uint32_t& sa, uint32_t& saa, uint32_t& sab) {
        const uint4* pa = (const uint4*)a; const uint4* pb = (const uint4*)b;
        sa = saa = sab = 0;
        const u16x2 ones = {1, 1};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 qa = pa[k], qb = pb[k];
            const uint32_t wa[4] = {qa.x, qa.y, qa.z, qa.w}, wb[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const u16x2 va = __builtin_bit_cast(u16x2, wa[m]), vb = __builtin_bit_cast(u16x2, wb[m]);
                sa  = __builtin_amdgcn_udot2(va, ones, sa, false);
                saa = __builtin_amdgcn_udot2(va, va, saa, false);
                sab = __builtin_amdgcn_udot2(va, vb, sab, false);
            }
        }
    }
};

// total of v over the 64 lanes of a wave (wave-uniform result): DPP inside the 16-lane rows, then the four row totals
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);   // row_mirror
    return (uint32_t)(__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

// ------------------------------------------------------------------------------- search, luma ---
template <typename PIX, bool DEDUPE_OFF = false>
__global__ void __launch_bounds__(256)
cdef_search_luma_kernel(const PIX* __restrict__ rec, int rec_stride, const PIX* __restrict__ src, int src_stride, int w, int h,
                        const uint8_t* __restrict__ skip8, int pri_damping, int cs, uint64_t* __restrict__ mse,
                        uint8_t* __restrict__ dir_out, int32_t* __restrict__ var_out) {
    constexpr int TS = 64 + 2 * kHB;  // tile stride (elements)
    __shared__ uint16_t tile[(64 + 2 * kVB) * TS];
    __shared__ __attribute__((aligned(16))) PIX ytab[4][64][64 + 16 / sizeof(PIX)];  // [wave][strength][pixel] (+pad)
    __shared__ __attribute__((aligned(16))) PIX stab[4][64];
    __shared__ int part[4][128];
    __shared__ unsigned long long accum[4][64];
    const int nhfb = (w + 63) >> 6, fb = svt_xcd_order(blockIdx.x, gridDim.x), fbr = fb / nhfb, fbc = fb - fbr * nhfb, c8 = w >> 3;
    const int nbx = min(8, c8 - 8 * fbc), nby = min(8, (h >> 3) - 8 * fbr);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // svt_sb_all_skip (EbEncCdef.c:222): nothing to do, table entry stays untouched
    int any = 0;
    for (int b = lane; b < 64; b += 64) { const int by = b >> 3, bx = b & 7; if (by < nby && bx < nbx && !skip8[(8 * fbr + by) * c8 + 8 * fbc + bx]) any = 1; }
    if (!__any(any)) return;
    stage_tile(tile, TS, rec, rec_stride, w, h, 64 * fbc, 64 * fbr, 64, 64, tid, 256);
    __syncthreads();
    const int damping = pri_damping + cs;  // pli == 0 (EbCdef.c:306-307)
    unsigned long long acc = 0;            // lane g: sum over this wave's blocks of dist(block, strength g)
    const int i = lane >> 3, j = lane & 7;
    for (int b = wave; b < 64; b += 4) {
        const int by = b >> 3, bx = b & 7;
        if (by >= nby || bx >= nbx || skip8[(8 * fbr + by) * c8 + 8 * fbc + bx]) {
            if (lane == 0 && dir_out) { dir_out[fb * 64 + b] = 0; var_out[fb * 64 + b] = 0; }
            continue;
        }
        const uint16_t* px = tile + (8 * by + i + kVB) * TS + 8 * bx + j + kHB;
        int var;
        const int dir = find_dir_wave(((int)px[0] >> cs), lane, part[wave], var);
        if (lane == 0 && dir_out) { dir_out[fb * 64 + b] = (uint8_t)dir; var_out[fb * 64 + b] = var; }
        int t_of[16];
#pragma unroll
        for (int idx = 0; idx < 16; idx++) t_of[idx] = adjust_strength(idx << cs, var);
        PixelTerms T;
        pixel_terms(px, TS, dir, t_of, cs, damping, T);
        // Primary indices whose effective strength equals the previous index's filter identically (same sums, same direction, same clamps): only the first index of
        // such a run is combined and stored; rep4 = 16 x 4 bits, the first index of every index's run (wave-uniform).  A block of low directional variance has 5 .. 9
        // distinct effective strengths out of 16 (adjust_strength, EbCdef.c:112-116).
        unsigned long long rep4 = 0;
        int first = 0;
#pragma unroll
        for (int pi = 0; pi < 16; pi++) {
            const bool fresh = DEDUPE_OFF || pi <= 1 || t_of[pi] != t_of[pi - 1];   // index 0 filters along direction 0 (EbCdef.c:371 `t ? dir : 0` tests the frame-header strength): it never shares a row, even when index 1 is adjusted to 0
            if (fresh) {
                first = pi;
#pragma unroll
                for (int pair = 0; pair < 2; pair++) {
                    const s16x2 y = combine2(T, pi, pair);
                    ytab[wave][4 * pi + 2 * pair][lane] = (PIX)y.x; ytab[wave][4 * pi + 2 * pair + 1][lane] = (PIX)y.y;
                }
            }
            rep4 |= (unsigned long long)first << (4 * pi);
        }
        const uint32_t sv = src[(size_t)(64 * fbr + 8 * by + i) * src_stride + 64 * fbc + 8 * bx + j];
        stab[wave][lane] = (PIX)sv;
        __builtin_amdgcn_wave_barrier();
        // lane g reduces strength g: dist_8x8 (EbEncCdef.c:79-105), names as in the reference:
        //   s = filtered ("src" there), d = source picture ("dst" there)
        uint32_t sum_s, sum_s2, sum_sd;
        const int row = (int)((rep4 >> (4 * (lane >> 2))) & 15) * 4 + (lane & 3);
        Dots<PIX>::run(&ytab[wave][row][0], &stab[wave][0], sum_s, sum_s2, sum_sd);
        const uint32_t sum_d = wave_sum_u32(sv), sum_d2 = wave_sum_u32(sv * sv);   // the same for every strength
        const uint64_t svar = (uint64_t)sum_s2 - (((uint64_t)sum_s * sum_s + 32) >> 6);
        const uint64_t dvar = (uint64_t)sum_d2 - (((uint64_t)sum_d * sum_d + 32) >> 6);
        const double num = (double)((uint64_t)sum_d2 + sum_s2 - 2 * (uint64_t)sum_sd) * .5 * (double)(svar + dvar + (uint64_t)(400 << 2 * cs));
        const double den = sqrt((double)(20000 << 4 * cs) + (double)svar * (double)dvar);
        acc += (unsigned long long)floor(.5 + num / den);
        __builtin_amdgcn_wave_barrier();
    }
    accum[wave][lane] = acc;
    __syncthreads();
    if (tid < 64) {
        const unsigned long long t = accum[0][tid] + accum[1][tid] + accum[2][tid] + accum[3][tid];
        mse[(size_t)fb * 64 + tid] = t >> (2 * cs);  // compute_cdef_dist_*: sum >> 2*coeff_shift
    }
}

// ----------------------------------------------------------------------------- search, chroma ---
// Both chroma planes of one filter block; mse[1][fb][g] = dist(U) + dist(V) (EbCdefProcess.c:268-271).
template <typename PIX>
__global__ void __launch_bounds__(256)
cdef_search_chroma_kernel(const PIX* __restrict__ rec_u, const PIX* __restrict__ rec_v, int rec_stride, const PIX* __restrict__ src_u,
                          const PIX* __restrict__ src_v, int src_stride, int w, int h, const uint8_t* __restrict__ skip8,
                          int pri_damping, int cs, uint64_t* __restrict__ mse_uv, const uint8_t* __restrict__ dir_in) {
    constexpr int TS = 32 + 2 * kHB;
    __shared__ uint16_t tile[2][(32 + 2 * kVB) * TS];
    __shared__ __attribute__((aligned(16))) PIX ytab[4][64][64 + 16 / sizeof(PIX)];
    __shared__ __attribute__((aligned(16))) PIX stab[4][64];
    __shared__ unsigned long long accum[4][2][64];
    const int nhfb = (w + 63) >> 6, fb = svt_xcd_order(blockIdx.x, gridDim.x), fbr = fb / nhfb, fbc = fb - fbr * nhfb, c8 = w >> 3;
    const int nbx = min(8, c8 - 8 * fbc), nby = min(8, (h >> 3) - 8 * fbr);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int any = 0;
    for (int b = lane; b < 64; b += 64) { const int by = b >> 3, bx = b & 7; if (by < nby && bx < nbx && !skip8[(8 * fbr + by) * c8 + 8 * fbc + bx]) any = 1; }
    if (!__any(any)) return;
    stage_tile(tile[0], TS, rec_u, rec_stride, w >> 1, h >> 1, 32 * fbc, 32 * fbr, 32, 32, tid, 256);
    stage_tile(tile[1], TS, rec_v, rec_stride, w >> 1, h >> 1, 32 * fbc, 32 * fbr, 32, 32, tid, 256);
    __syncthreads();
    const int damping = pri_damping + cs - 1;  // pli != 0
    unsigned long long acc[2] = {0, 0};
    // a pass = 4 horizontally adjacent 4x4 blocks (one 4x16 strip); lane -> (block q, row i, col j)
    const int q = lane >> 4, i = (lane >> 2) & 3, j = lane & 3;
    for (int pass = wave; pass < 32; pass += 4) {
        const int pl = pass >> 4, strip = pass & 15;
        const int by = strip >> 1, bx = (strip & 1) * 4 + q;
        const bool live = by < nby && bx < nbx && !skip8[(8 * fbr + by) * c8 + 8 * fbc + bx];
        const PIX* sp = pl ? src_v : src_u;
        const int sy = min(32 * fbr + 4 * by + i, (h >> 1) - 1), sx = min(32 * fbc + 4 * bx + j, (w >> 1) - 1);
        const PIX s = sp[(size_t)sy * src_stride + sx];
        if (live) {
            const uint16_t* px = tile[pl] + (4 * by + i + kVB) * TS + 4 * bx + j + kHB;
            const int dir = dir_in[fb * 64 + by * 8 + bx];
            int t_of[16];
#pragma unroll
            for (int idx = 0; idx < 16; idx++) t_of[idx] = idx << cs;
            PixelTerms T;
            pixel_terms(px, TS, dir, t_of, cs, damping, T);
#pragma unroll
            for (int g = 0; g < 64; g += 2) {
                const s16x2 y = combine2(T, g >> 2, (g >> 1) & 1);
                ytab[wave][g][lane] = (PIX)y.x; ytab[wave][g + 1][lane] = (PIX)y.y;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 64; g++) ytab[wave][g][lane] = s;  // contributes (y - s)^2 = 0
        }
        stab[wave][lane] = s;
        __builtin_amdgcn_wave_barrier();
        uint32_t sy1, sy2, sys;
        Dots<PIX>::run(&ytab[wave][lane][0], &stab[wave][0], sy1, sy2, sys);
        const uint32_t ss2 = wave_sum_u32((uint32_t)s * (uint32_t)s);
        acc[pl] += (unsigned long long)sy2 + ss2 - 2ull * sys;  // sum (y - s)^2, mse_4_*: EbEncCdef.c:67-77,121-131
        __builtin_amdgcn_wave_barrier();
    }
    accum[wave][0][lane] = acc[0];
    accum[wave][1][lane] = acc[1];
    __syncthreads();
    if (tid < 64) {
        const unsigned long long u = accum[0][0][tid] + accum[1][0][tid] + accum[2][0][tid] + accum[3][0][tid];
        const unsigned long long v = accum[0][1][tid] + accum[1][1][tid] + accum[2][1][tid] + accum[3][1][tid];
        mse_uv[(size_t)fb * 64 + tid] = (u >> (2 * cs)) + (v >> (2 * cs));
    }
}

// ---------------------------------------------------------------------------------- apply -------
// One workgroup per (filter block, plane).  in = pre-CDEF plane, out = result plane: every sample of the picture is written (samples of
// unfiltered filter blocks and of skipped 8x8 blocks are passed through), so `out` needs no initial copy of `in`.
// strength[fb] = frame-header value pri*4 + sec_idx chosen for the fb.
template <typename PIX, int PLANE_KIND>  // 0 luma, 1 chroma
__global__ void __launch_bounds__(256)
cdef_apply_kernel(const PIX* __restrict__ in, PIX* __restrict__ out, int stride, int w, int h, const uint8_t* __restrict__ skip8,
                  const uint8_t* __restrict__ y_strength, const uint8_t* __restrict__ uv_strength, int damping_hdr, int cs,
                  uint8_t* __restrict__ dir_buf, const int32_t* __restrict__ var_in) {
    constexpr int DEC = PLANE_KIND, FBS = 64 >> DEC, TS = FBS + 2 * kHB;
    __shared__ uint16_t tile[(FBS + 2 * kVB) * TS];
    __shared__ int part[4][128];
    const int nhfb = (w + 63) >> 6, fb = svt_xcd_order(blockIdx.x, gridDim.x), fbr = fb / nhfb, fbc = fb - fbr * nhfb, c8 = w >> 3;
    const int nbx = min(8, c8 - 8 * fbc), nby = min(8, (h >> 3) - 8 * fbr);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ly = y_strength[fb] >> 2, sy = y_strength[fb] & 3, lu = uv_strength[fb] >> 2, su = uv_strength[fb] & 3;
    sy += sy == 3; su += su == 3;
    if (ly == 0 && sy == 0 && lu == 0 && su == 0) {  // EbEncCdef.c:434-441: the filter block keeps the pre-CDEF samples
        const int fw = min(FBS, (w >> DEC) - FBS * fbc), fh = min(FBS, (h >> DEC) - FBS * fbr);
        for (int i = tid; i < fw * fh; i += 256) {
            const int y = i / fw, x = i - y * fw;
            const size_t o = (size_t)(FBS * fbr + y) * stride + FBS * fbc + x;
            out[o] = in[o];
        }
        return;
    }
    const int level = PLANE_KIND ? lu : ly, sec = (PLANE_KIND ? su : sy) << cs;
    stage_tile(tile, TS, in, stride, w >> DEC, h >> DEC, FBS * fbc, FBS * fbr, FBS, FBS, tid, 256);
    __syncthreads();
    const int damping = damping_hdr + cs - (PLANE_KIND != 0);
    if (PLANE_KIND == 0) {
        const int i = lane >> 3, j = lane & 7;
        for (int b = wave; b < 64; b += 4) {
            const int by = b >> 3, bx = b & 7;
            if (by >= nby || bx >= nbx) continue;
            const uint16_t* px = tile + (8 * by + i + kVB) * TS + 8 * bx + j + kHB;
            if (skip8[(8 * fbr + by) * c8 + 8 * fbc + bx]) {   // skipped block: passed through
                out[(size_t)(64 * fbr + 8 * by + i) * stride + 64 * fbc + 8 * bx + j] = (PIX)px[0];
                continue;
            }
            int var, dir;
            if (var_in) {   // the strength search already ran svt_cdef_find_dir on this picture: reuse its direction / variance
                dir = dir_buf[fb * 64 + b]; var = var_in[fb * 64 + b];
            } else {
                dir = find_dir_wave(((int)px[0] >> cs), lane, part[wave], var);
                if (lane == 0) dir_buf[fb * 64 + b] = (uint8_t)dir;
            }
            const int t = level << cs;
            const int y = filter_px_single(px, TS, adjust_strength(t, var), sec, t ? dir : 0, cs, damping);
            out[(size_t)(64 * fbr + 8 * by + i) * stride + 64 * fbc + 8 * bx + j] = (PIX)y;
        }
    } else {
        const int q = lane >> 4, i = (lane >> 2) & 3, j = lane & 3;
        for (int strip = wave; strip < 16; strip += 4) {
            const int by = strip >> 1, bx = (strip & 1) * 4 + q;
            if (by >= nby || bx >= nbx) continue;
            const uint16_t* px = tile + (4 * by + i + kVB) * TS + 4 * bx + j + kHB;
            if (skip8[(8 * fbr + by) * c8 + 8 * fbc + bx]) {
                out[(size_t)(32 * fbr + 4 * by + i) * stride + 32 * fbc + 4 * bx + j] = (PIX)px[0];
                continue;
            }
            const int t = level << cs;
            const int dir = t ? dir_buf[fb * 64 + by * 8 + bx] : 0;
            const int y = filter_px_single(px, TS, t, sec, dir, cs, damping);
            out[(size_t)(32 * fbr + 4 * by + i) * stride + 32 * fbc + 4 * bx + j] = (PIX)y;
        }
    }
}

template <typename PIX>
int search_t(hipStream_t st, const void* const rec[3], const int rs[3], const void* const src[3], const int ss[3], int w, int h,
             const uint8_t* skip8, int pri_damping, int cs, uint64_t* mse, uint8_t* dir_buf, int32_t* var_buf) {
    const int nfb = ((w + 63) >> 6) * ((h + 63) >> 6);
    static const bool dedupe_off = getenv("SVT_HIP_CDEF_DEDUPE") && !atoi(getenv("SVT_HIP_CDEF_DEDUPE"));   // A/B: every primary index combined and stored (round 3)
    if (dedupe_off)
        hipLaunchKernelGGL((cdef_search_luma_kernel<PIX, true>), dim3(nfb), dim3(256), 0, st, (const PIX*)rec[0], rs[0], (const PIX*)src[0], ss[0], w, h,
                           skip8, pri_damping, cs, mse, dir_buf, var_buf);
    else
        hipLaunchKernelGGL((cdef_search_luma_kernel<PIX>), dim3(nfb), dim3(256), 0, st, (const PIX*)rec[0], rs[0], (const PIX*)src[0], ss[0], w, h,
                           skip8, pri_damping, cs, mse, dir_buf, var_buf);
    hipLaunchKernelGGL((cdef_search_chroma_kernel<PIX>), dim3(nfb), dim3(256), 0, st, (const PIX*)rec[1], (const PIX*)rec[2], rs[1],
                       (const PIX*)src[1], (const PIX*)src[2], ss[1], w, h, skip8, pri_damping, cs, mse + (size_t)nfb * 64, dir_buf);
    return (int)hipGetLastError();
}
template <typename PIX>
int apply_t(hipStream_t st, const void* const in[3], void* const out[3], const int stride[3], int w, int h, const uint8_t* skip8,
            const uint8_t* ys, const uint8_t* uvs, int damping, int cs, uint8_t* dir_buf, const int32_t* var_in) {
    const int nfb = ((w + 63) >> 6) * ((h + 63) >> 6);
    hipLaunchKernelGGL((cdef_apply_kernel<PIX, 0>), dim3(nfb), dim3(256), 0, st, (const PIX*)in[0], (PIX*)out[0], stride[0], w, h, skip8, ys, uvs, damping, cs, dir_buf, var_in);
    for (int p = 1; p < 3; p++)
        hipLaunchKernelGGL((cdef_apply_kernel<PIX, 1>), dim3(nfb), dim3(256), 0, st, (const PIX*)in[p], (PIX*)out[p], stride[p], w, h, skip8, ys, uvs, damping, cs, dir_buf, var_in);
    return (int)hipGetLastError();
}

// ---- per-call forms (include/svt_hip_rtcd.h): svt_cdef_find_dir for a list of 8x8 blocks, svt_cdef_filter_block for a list of blocks, on the
// 16-bit staging layout of the reference (CDEF_BSTRIDE rows, CDEF_VERY_LARGE outside the picture).  Same device functions as the frame kernels.
__global__ void __launch_bounds__(64)
cdef_find_dir_list_kernel(const uint16_t* __restrict__ img, const int32_t* __restrict__ offs, int stride, int coeff_shift, int32_t* __restrict__ dir_out, int32_t* __restrict__ var_out) {
    __shared__ int xs[64];
    const int lane = threadIdx.x, b = blockIdx.x;
    const uint16_t* p = img + offs[b];
    int var;
    const int dir = find_dir_wave((int)p[(lane >> 3) * stride + (lane & 7)] >> coeff_shift, lane, xs, var);
    if (lane == 0) { dir_out[b] = dir; var_out[b] = var; }
}
// dst8 != nullptr: 8-bit destination (stride dstride), else dst16
__global__ void __launch_bounds__(64)
cdef_filter_block_list_kernel(const uint16_t* __restrict__ in, int istride, const SvtHipCdefBlk* __restrict__ jobs, uint8_t* __restrict__ dst8, uint16_t* __restrict__ dst16, int dstride) {
    const SvtHipCdefBlk j = jobs[blockIdx.x];
    const int bw = 1 << j.bw_log2, bh = 1 << j.bh_log2, lane = threadIdx.x;
    if (lane >= bw * bh) return;
    const int y = lane >> j.bw_log2, x = lane & (bw - 1);
    const int v = filter_px_single(in + j.in_off + y * istride + x, istride, j.pri_strength, j.sec_strength, j.dir, j.coeff_shift, j.pri_damping, j.sec_damping);
    if (dst8) dst8[j.dst_off + y * dstride + x] = (uint8_t)v; else dst16[j.dst_off + y * dstride + x] = (uint16_t)v;
}

}  // namespace

extern "C" int svt_hip_launch_cdef_find_dir_list(hipStream_t st, const uint16_t* img, const int32_t* offs, int n, int stride, int coeff_shift, int32_t* dir_out, int32_t* var_out) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(cdef_find_dir_list_kernel, dim3(n), dim3(64), 0, st, img, offs, stride, coeff_shift, dir_out, var_out);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_cdef_filter_block_list(hipStream_t st, const uint16_t* in, int istride, const void* jobs, int n, uint8_t* dst8, uint16_t* dst16, int dstride) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(cdef_filter_block_list_kernel, dim3(n), dim3(64), 0, st, in, istride, (const SvtHipCdefBlk*)jobs, dst8, dst16, dstride);
    return (int)hipGetLastError();
}


extern "C" int svt_hip_launch_cdef_search(hipStream_t st, int pix_bytes, const void* const rec[3], const int rec_stride[3],
                                          const void* const src[3], const int src_stride[3], int w, int h, const uint8_t* skip8,
                                          int pri_damping, int bd, uint64_t* mse, uint8_t* dir_buf, int32_t* var_buf) {
    const int cs = bd - 8;
    if (pix_bytes == 1) return search_t<uint8_t>(st, rec, rec_stride, src, src_stride, w, h, skip8, pri_damping, cs, mse, dir_buf, var_buf);
    return search_t<uint16_t>(st, rec, rec_stride, src, src_stride, w, h, skip8, pri_damping, cs, mse, dir_buf, var_buf);
}
extern "C" int svt_hip_launch_cdef_apply(hipStream_t st, int pix_bytes, const void* const in[3], void* const out[3], const int stride[3],
                                         int w, int h, const uint8_t* skip8, const uint8_t* y_strength, const uint8_t* uv_strength,
                                         int damping, int bd, uint8_t* dir_buf, const int32_t* var_in) {
    const int cs = bd - 8;
    if (pix_bytes == 1) return apply_t<uint8_t>(st, in, out, stride, w, h, skip8, y_strength, uv_strength, damping, cs, dir_buf, var_in);
    return apply_t<uint16_t>(st, in, out, stride, w, h, skip8, y_strength, uv_strength, damping, cs, dir_buf, var_in);
}

SVT_HIP_TU_PROBE(cdef)
