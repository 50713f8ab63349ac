// sgr.hip — self-guided restoration (SGRPROJ): box filters, projection sums for the search, and the
// final application; gfx950.
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Common/Codec/EbRestoration.c:1012  svt_av1_selfguided_restoration_c  (:744 r = 2 "fast", :884 r = 1, box sums :541-705)
//   Common/Codec/EbRestoration.c:1047  svt_apply_selfguided_restoration_c (+ svt_decode_xq :707)
//   Encoder/Codec/EbRestorationPick.c:448  svt_get_proj_subspace_c (the integer sums; the 2x2 FP64 solve stays on the host)
//   Encoder/Codec/EbRestorationPick.c:554-671  apply_sgr / search_selfguided_restoration (per-unit, per-ep loops)
//
// The filter output at a pixel is a pure function of the 3-pixel-extended picture and of the row parity
// (r = 2 keeps A/B on odd rows only), and every processing-unit / stripe origin of the reference is even,
// so the picture is walked in 64x16 tiles regardless of the reference's 64x64 processing units.
// Per tile: (1) the (2r+1)^2 box sums of x and x^2 are built ONCE in LDS — they do not depend on the
// parameter set; (2) per parameter set the A'/B' pair of every position is packed in one LDS dword
// (A' <= 256: 9 bits, B' < 2^18); (3) each lane combines the 3x3 neighbourhood for its 4 pixels.
// The search kernel loops all 16 parameter sets over the same box sums and only emits five exact int64
// sums per (restoration unit, parameter set) — flt0/flt1 never leave the chip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "svt_hip_internal.h"
#include "lds_stage.h"

// the difference planes are written once and read back much later by the walk: SVT_SGR_NT (A/B) marks those accesses non-temporal
#ifdef SVT_SGR_NT
#define SGR_ST(p, v) __builtin_nontemporal_store((uint32_t)(v), (p))
#else
#define SGR_ST(p, v) (*(p) = (v))
#endif

namespace {

constexpr int TW = 64, TH = 16;            // output tile
constexpr int IW = TW + 6, IH = TH + 6;    // staged input (3-px halo)
constexpr int PW = TW + 2, PH1 = TH + 2;   // A/B positions incl. the 1-px border; r = 1 uses all rows
constexpr int PH2 = TH / 2 + 1;            // r = 2: rows -1, 1, ..., TH - 1

// eb_sgr_params {r0, r1, s0, s1} (EbRestoration.c:136-153)
__device__ __constant__ int kSgr[16][4] = {{2, 1, 140, 3236}, {2, 1, 112, 2158}, {2, 1, 93, 1618}, {2, 1, 80, 1438}, {2, 1, 70, 1295}, {2, 1, 58, 1177},
                                           {2, 1, 47, 1079},  {2, 1, 37, 996},   {2, 1, 30, 925},  {2, 1, 25, 863},  {0, 1, -1, 2589}, {0, 1, -1, 1618},
                                           {0, 1, -1, 1177},  {0, 1, -1, 925},   {2, 0, 56, -1},   {2, 0, 22, -1}};

struct TileLds {
    uint16_t in[IH * IW];
    uint32_t s1[PH1 * PW], q1[PH1 * PW];   // r = 1 box sums of x and x^2
    uint32_t s2[PH2 * PW], q2[PH2 * PW];   // r = 2 (odd rows)
    uint32_t ab1[PH1 * PW], ab2[PH2 * PW]; // packed A' | B' << 9 for the current parameter set
    uint16_t xtab[256];                    // eb_x_by_xplus1
};

__device__ __forceinline__ uint32_t rp2u(uint32_t v, int n) { return n == 0 ? v : ((v + (1u << (n - 1))) >> n); }

// Where the rows just outside a restoration stripe come from (svt_av1_loop_restoration_filter_unit, EbRestoration.c:1162-1249):
// the stripe [sy0, sy1) sees the DEBLOCKED picture in its 3 context rows above (rows sy0-2, sy0-2, sy0-1) and below
// (sy1, sy1+1, sy1+1, bottom-clamped) -- setup_processing_stripe_boundary :353-453 with the lines of
// save_deblock_boundary_lines :1645-1697 (edge-replicated) -- unless it touches the top / bottom of the frame.
template <typename PIX>
struct StripeCtx {
    const PIX* dbl;   // nullptr: no substitution (search, plain filter)
    int dbl_stride, sy0, sy1, above, below;
};

template <typename PIX>
__device__ __forceinline__ void stage_and_boxsum(TileLds& L, const PIX* __restrict__ plane, int stride, int pw, int ph, int x0, int y0, int tid,
                                                 const StripeCtx<PIX> sc = StripeCtx<PIX>{nullptr, 0, 0, 0, 0, 0}, bool box_sums = true) {
    if (tid < 256) L.xtab[tid] = tid == 0 ? 1 : (tid == 255 ? 256 : (uint16_t)((256 * tid + (tid + 1) / 2) / (tid + 1)));
    batched_stage<4, uint16_t>(IH * IW, tid, 256,
        [&](int i) {
            const int r = i / IW, c = i - r * IW;
            const int yy = y0 - 3 + r, xx = x0 - 3 + c;
            const bool up = sc.above && yy < sc.sy0, dn = sc.below && yy >= sc.sy1, ctx = up || dn;   // one unconditional load from a selected address
            const int  yd = up ? (yy == sc.sy0 - 1 ? sc.sy0 - 1 : sc.sy0 - 2) : min(yy == sc.sy1 ? sc.sy1 : sc.sy1 + 1, ph - 1);
            const int  x = ctx ? min(max(xx, 0), pw - 1) : min(max(xx, -3), pw + 2), y = ctx ? yd : min(max(yy, -3), ph + 2);   // the picture: never leave the 3-px extension
            const PIX* base = ctx ? sc.dbl : plane;
            return (uint16_t)base[(ptrdiff_t)y * (ctx ? sc.dbl_stride : stride) + x];
        },
        [&](int i, uint16_t v) { L.in[i] = v; });
    __syncthreads();
    if (!box_sums) return;
    for (int i = tid; i < PH1 * PW; i += 256) {            // r = 1: position (i/PW - 1, i%PW - 1)
        const int r = i / PW, c = i - r * PW;               // window centre in `in` coordinates: (r + 2, c + 2)
        uint32_t s = 0, q = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) { const uint32_t v = L.in[(r + 2 + dy) * IW + c + 2 + dx]; s += v; q += v * v; }
        L.s1[i] = s; L.q1[i] = q;
    }
    for (int i = tid; i < PH2 * PW; i += 256) {            // r = 2: rows -1, 1, 3, ...
        const int rr = i / PW, c = i - rr * PW, r = 2 * rr; // position row = r - 1 -> `in` row r + 2
        uint32_t s = 0, q = 0;
#pragma unroll
        for (int dy = -2; dy <= 2; dy++)
#pragma unroll
            for (int dx = -2; dx <= 2; dx++) { const uint32_t v = L.in[(r + 2 + dy) * IW + c + 2 + dx]; s += v; q += v * v; }
        L.s2[i] = s; L.q2[i] = q;
    }
    __syncthreads();
}

// A'/B' of one position: EbRestoration.c:787-858 / :926-985
template <int BD>
__device__ __forceinline__ uint32_t ab_pack(const TileLds& L, uint32_t sum, uint32_t sq, uint32_t n, uint32_t s, uint32_t one_by_n) {
    const uint32_t a = rp2u(sq, 2 * (BD - 8)), b = rp2u(sum, BD - 8);
    const uint32_t p = (a * n < b * b) ? 0u : a * n - b * b;
    const uint32_t z = rp2u(p * s, 20);
    const uint32_t A = L.xtab[min(z, 255u)];
    const uint32_t B = rp2u((256u - A) * sum * one_by_n, 12);
    return A | (B << 9);
}

template <int BD>
__device__ __forceinline__ void build_ab(TileLds& L, int ep, int tid) {
    const int r0 = kSgr[ep][0], r1 = kSgr[ep][1];
    if (r1 > 0)
        for (int i = tid; i < PH1 * PW; i += 256) L.ab1[i] = ab_pack<BD>(L, L.s1[i], L.q1[i], 9, (uint32_t)kSgr[ep][3], 455);
    if (r0 > 0)
        for (int i = tid; i < PH2 * PW; i += 256) L.ab2[i] = ab_pack<BD>(L, L.s2[i], L.q2[i], 25, (uint32_t)kSgr[ep][2], 164);
    __syncthreads();
}

// flt0 / flt1 of pixel (i, j) of the tile
__device__ __forceinline__ void filt_px(const TileLds& L, int ep, int i, int j, int32_t& f0, int32_t& f1) {
    const int x = L.in[(i + 3) * IW + j + 3];
#define A_(v) ((int32_t)((v) & 511u))
#define B_(v) ((int32_t)((v) >> 9))
    if (kSgr[ep][0] > 0) {
        int32_t a, b;
        if (!(i & 1)) {   // even row: rows i-1 and i+1 (ab2 rows (i)/2 and (i)/2 + 1), weights 6 / 5
            const uint32_t* u = L.ab2 + (i / 2) * PW + j + 1;
            const uint32_t* d = u + PW;
            a = (A_(u[0]) + A_(d[0])) * 6 + (A_(u[-1]) + A_(d[-1]) + A_(u[1]) + A_(d[1])) * 5;
            b = (B_(u[0]) + B_(d[0])) * 6 + (B_(u[-1]) + B_(d[-1]) + B_(u[1]) + B_(d[1])) * 5;
            f0 = (a * x + b + (1 << 8)) >> 9;
        } else {          // odd row: own row, weights 6 / 5
            const uint32_t* m = L.ab2 + ((i + 1) / 2) * PW + j + 1;
            a = A_(m[0]) * 6 + (A_(m[-1]) + A_(m[1])) * 5;
            b = B_(m[0]) * 6 + (B_(m[-1]) + B_(m[1])) * 5;
            f0 = (a * x + b + (1 << 7)) >> 8;
        }
    }
    if (kSgr[ep][1] > 0) {
        const uint32_t* m = L.ab1 + (i + 1) * PW + j + 1;
        const int32_t a = (A_(m[0]) + A_(m[-1]) + A_(m[1]) + A_(m[-PW]) + A_(m[PW])) * 4 + (A_(m[-PW - 1]) + A_(m[PW - 1]) + A_(m[-PW + 1]) + A_(m[PW + 1])) * 3;
        const int32_t b = (B_(m[0]) + B_(m[-1]) + B_(m[1]) + B_(m[-PW]) + B_(m[PW])) * 4 + (B_(m[-PW - 1]) + B_(m[PW - 1]) + B_(m[-PW + 1]) + B_(m[PW + 1])) * 3;
        f1 = (a * x + b + (1 << 8)) >> 9;
    }
#undef A_
#undef B_
}

// ---- svt_av1_selfguided_restoration over a whole plane, one parameter set: flt0 / flt1 planes (stride = flt_stride)
template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
sgr_filter_kernel(const PIX* __restrict__ plane, int stride, int pw, int ph, int ep, int32_t* __restrict__ flt0, int32_t* __restrict__ flt1, int flt_stride) {
    __shared__ TileLds L;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    stage_and_boxsum(L, plane, stride, pw, ph, x0, y0, tid);
    build_ab<BD>(L, ep, tid);
    for (int k = tid; k < TW * TH; k += 256) {
        const int i = k / TW, j = k - i * TW;
        if (x0 + j >= pw || y0 + i >= ph) continue;
        int32_t f0 = 0, f1 = 0;
        filt_px(L, ep, i, j, f0, f1);
        if (kSgr[ep][0] > 0) flt0[(size_t)(y0 + i) * flt_stride + x0 + j] = f0;
        if (kSgr[ep][1] > 0) flt1[(size_t)(y0 + i) * flt_stride + x0 + j] = f1;
    }
}

// ---------------------------------------------------------------------------------------------------
// Search kernel: projection sums of every (restoration unit, parameter set), sums[unit][16][5] += {H00, H01, H11, C0, C1}.
// Bit depth 8 (the 4K north-star path) and 10; ~1/3 of the instructions per pixel and parameter set of a per-pixel
// box-sum formulation:
//  * everything that does not depend on the parameter set is hoisted out of the 16-set loop and kept
//    in REGISTERS: p = max(n*sumsq - sum^2, 0) and m = sum * one_by_n of the ~13 A/B positions a thread
//    owns.  For 8-bit data p < 2^24 and the set's s < 2^12, so z and B are single v_mad_u32_u24;
//  * A' and B' are packed as A' << 20 | B' (A' <= 256, B' <= 65088): the un-weighted sum of a 3x3
//    neighbourhood (9 * 256 < 2^12, 9 * 65088 < 2^20) stays inside its field, so the neighbourhood is
//    gathered with PACKED adds and the 4/3 (r = 1) and 6/5 (r = 2) weights become 3*S9 + S5 and
//    5*S6 + S2 on the unpacked fields;
//  * a lane walks 8 rows of one column, so the horizontal triple sums are reused by three output rows;
//  * flt - u is formed inside the rounding shift ((a*x + b + 256 - (x << 13)) >> 9), the five projection
//    sums are v_mad_i32_i24 in int32 (|flt - u| <= 4084, 8 px * 16 lanes * 4084^2 < 2^31), reduced over
//    16-lane rows with DPP adds and accumulated per parameter set with 64-bit LDS atomics: one global
//    atomic per (tile, set, sum) at the very end;
//  * sets 11/12/13 (r0 = 0) share s1 with sets 2/5/8, so their H11/C1 are copies: 13 filter passes, not 16.
constexpr int S_TW = 64, S_TH = 32;
constexpr int S_IW = S_TW + 6, S_IH = S_TH + 6;
constexpr int S_PW = S_TW + 2, S_P1H = S_TH + 2, S_P2H = S_TH / 2 + 1;
constexpr int S_N1 = S_PW * S_P1H, S_N2 = S_PW * S_P2H, S_NP = S_N1 + S_N2;
constexpr int S_KP = (S_NP + 255) / 256;

__device__ __forceinline__ int32_t row16_sum(int32_t v) {   // every lane of a 16-lane row gets the row total
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);   // row_mirror
    return v;
}

// ---- building blocks shared by the 8-bit search and apply kernels (64 x 32 tiles) --------------------------------------------
// `in`: staged tile [S_IH][S_IW]; abmem: 2 * S_NP dwords of scratch that later hold A'/B' (two barriers inside)
template <int BD = 8>
__device__ __forceinline__ void sgr8_precompute(const uint16_t* __restrict__ in, uint32_t* __restrict__ abmem, int tid, uint32_t (&P)[S_KP], uint32_t (&M)[S_KP]) {
    // parameter-set independent part of A/B for the positions this thread owns.  The box sums are separable: (1) a lane takes one
    // column of the staged tile and a third of its rows and writes the VERTICAL 3- and 5-sums of x and x^2 (15 LDS reads), (2) a position
    // adds 3 (r = 1) or 5 (r = 2) neighbouring vertical sums.  ~27 instead of ~100 instructions per pixel.  The vertical sums live in
    // the memory of the A/B buffers, which are first written after the barrier below.
    uint16_t* vs3 = (uint16_t*)abmem;                       // [S_P1H][S_IW]
    uint32_t* vq3 = (uint32_t*)(vs3 + S_P1H * S_IW);            // [S_P1H][S_IW]   (S_P1H * S_IW is even: 4-byte aligned)
    uint16_t* vs5 = (uint16_t*)(vq3 + S_P1H * S_IW);            // [S_P2H][S_IW]
    uint32_t* vq5 = (uint32_t*)(vs5 + S_P2H * S_IW + (S_P2H * S_IW & 1));
    static_assert((S_P1H * S_IW * 6 + (S_P2H * S_IW + 1) * 6) <= (int)sizeof(uint32_t) * 2 * S_NP, "vertical sums must fit the A/B buffers");
    if (tid < 3 * S_IW) {
        const int c = tid % S_IW, b = tid / S_IW;               // column, row band: r = 1 position rows [12b, 12b + 12), r = 2 rows [6b, 6b + 6)
        uint32_t x[15];
#pragma unroll
        for (int k = 0; k < 15; k++) x[k] = in[min(12 * b + k, S_IH - 1) * S_IW + c];
#pragma unroll
        for (int k = 0; k < 12; k++) {                          // r = 1 position row pr: staged rows pr + 1 .. pr + 3
            const int pr = 12 * b + k;
            if (pr < S_P1H) {
                vs3[pr * S_IW + c] = (uint16_t)(x[k + 1] + x[k + 2] + x[k + 3]);
                vq3[pr * S_IW + c] = x[k + 1] * x[k + 1] + x[k + 2] * x[k + 2] + x[k + 3] * x[k + 3];
            }
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {                           // r = 2 position row rr: staged rows 2 rr .. 2 rr + 4
            const int rr = 6 * b + k;
            if (rr < S_P2H) {
                vs5[rr * S_IW + c] = (uint16_t)(x[2 * k] + x[2 * k + 1] + x[2 * k + 2] + x[2 * k + 3] + x[2 * k + 4]);
                vq5[rr * S_IW + c] = x[2 * k] * x[2 * k] + x[2 * k + 1] * x[2 * k + 1] + x[2 * k + 2] * x[2 * k + 2] + x[2 * k + 3] * x[2 * k + 3] + x[2 * k + 4] * x[2 * k + 4];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < S_KP; k++) {
        const int i = tid + 256 * k;
        uint32_t sm = 0, sq = 0, n = 9, obn = 455;
        if (i < S_N1) {                        // r = 1: position (r - 1, c - 1), window centre in[r + 2][c + 2]
            const int r = i / S_PW, c = i - r * S_PW;
            sm = (uint32_t)vs3[r * S_IW + c + 1] + vs3[r * S_IW + c + 2] + vs3[r * S_IW + c + 3];
            sq = vq3[r * S_IW + c + 1] + vq3[r * S_IW + c + 2] + vq3[r * S_IW + c + 3];
        } else if (i < S_NP) {                 // r = 2: picture rows -1, 1, 3, ...
            const int i2 = i - S_N1, rr = i2 / S_PW, c = i2 - rr * S_PW;
            n = 25; obn = 164;
            sm = (uint32_t)vs5[rr * S_IW + c] + vs5[rr * S_IW + c + 1] + vs5[rr * S_IW + c + 2] + vs5[rr * S_IW + c + 3] + vs5[rr * S_IW + c + 4];
            sq = vq5[rr * S_IW + c] + vq5[rr * S_IW + c + 1] + vq5[rr * S_IW + c + 2] + vq5[rr * S_IW + c + 3] + vq5[rr * S_IW + c + 4];
        }
        // EbRestoration.c:790-806 / :926-937: the sums are rounded down to 8-bit scale first (no-op at bit depth 8).  p stays below 2^24 and
        // p * s below 2^32 at bit depth 10 as well (r = 1: p <= 20 * 1023^2 / 16 + 2307 = 1 310 468, x 3236 < 2^32; r = 2: 10.21 M x 140).
        uint32_t a = sq, d = sm;
        if constexpr (BD > 8) { a = (sq + (1u << (2 * (BD - 8) - 1))) >> (2 * (BD - 8)); d = (sm + (1u << (BD - 9))) >> (BD - 8); }
        P[k] = (a * n < d * d) ? 0u : a * n - d * d;
        M[k] = sm * obn;                                     // <= 25 * 1023 * 164 < 2^24
    }
    __syncthreads();   // the vertical sums are dead: the first parameter set's A/B may overwrite them

}

// A'/B' of one parameter set for the positions this thread owns (EbRestoration.c:787-858 / :926-985), packed A' << 20 | B'
__device__ __forceinline__ void sgr8_build(uint32_t* __restrict__ abw, const uint32_t* __restrict__ xt, const uint32_t (&P)[S_KP], const uint32_t (&M)[S_KP],
                                           bool has0, bool has1, uint32_t s0, uint32_t s1, int tid) {
#pragma unroll
    for (int k = 0; k < S_KP; k++) {
        const int i = tid + 256 * k;
        if (i < S_N1 ? has1 : (has0 && i < S_NP)) {
            const uint32_t z = (__umul24(P[k], i < S_N1 ? s1 : s0) + (1u << 19)) >> 20;
            const uint32_t t = xt[min(z, 255u)];
            const uint32_t B = (__umul24(t & 0x1FFu, M[k]) + (1u << 11)) >> 12;
            abw[i] = (t & 0xFFF00000u) | B;
        }
    }
}

#define FA_(v) ((v) >> 20)
#define FB_(v) ((v) & 0xFFFFFu)
// flt0 - u (D0) and flt1 - u (D1) of the 8 pixels (column j, rows i0 .. i0 + 7) of a lane; X = pixel, CX = 256 - (X << 13)
__device__ __forceinline__ void sgr8_filter(const uint32_t* __restrict__ abw, int i0, int j, const uint32_t (&X)[8], const int32_t (&CX)[8], bool has0, bool has1,
                                            int32_t (&D0)[8], int32_t (&D1)[8]) {
        if (has1) {
        const uint32_t* a1 = abw + i0 * S_PW + j + 1;   // row index = picture row + 1
        uint32_t Rm, Cm, R0, C0;
        { const uint32_t l = a1[-1], c = a1[0], r = a1[1]; Rm = l + c + r; Cm = c; }
        { const uint32_t l = a1[S_PW - 1], c = a1[S_PW], r = a1[S_PW + 1]; R0 = l + c + r; C0 = c; }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t* q = a1 + (r + 2) * S_PW;
            const uint32_t l = q[-1], c = q[0], rt = q[1];
            const uint32_t Rp = l + c + rt;
            const uint32_t S9 = Rm + R0 + Rp, S5 = Cm + R0 + c;     // 4 * cross + 3 * corners = 3 * S9 + S5
            const uint32_t a = __umul24(FA_(S9), 3u) + FA_(S5), b = __umul24(FB_(S9), 3u) + FB_(S5);
            D1[r] = (int32_t)(__umul24(a, X[r]) + b + (uint32_t)CX[r]) >> 9;
            Rm = R0; Cm = C0; R0 = Rp; C0 = c;
        }
    }
    if (has0) {
        const uint32_t* a2 = abw + S_N1 + (i0 / 2) * S_PW + j + 1;   // ab2 row rr holds picture row 2 * rr - 1
        uint32_t H[5], C[5];
#pragma unroll
        for (int q = 0; q < 5; q++) { const uint32_t l = a2[q * S_PW - 1], c = a2[q * S_PW], r = a2[q * S_PW + 1]; H[q] = l + c + r; C[q] = c; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            {   // even row 2q: rows above/below, 6 * centres + 5 * sides = 5 * S6 + S2, >> 9
                const uint32_t S6 = H[q] + H[q + 1], S2 = C[q] + C[q + 1];
                const uint32_t a = __umul24(FA_(S6), 5u) + FA_(S2), b = __umul24(FB_(S6), 5u) + FB_(S2);
                D0[2 * q] = (int32_t)(__umul24(a, X[2 * q]) + b + (uint32_t)CX[2 * q]) >> 9;
            }
            {   // odd row 2q + 1: own row, >> 8 (rounding and u scale by one bit less: CX >> 1 is exact)
                const uint32_t a = __umul24(FA_(H[q + 1]), 5u) + FA_(C[q + 1]), b = __umul24(FB_(H[q + 1]), 5u) + FB_(C[q + 1]);
                D0[2 * q + 1] = (int32_t)(__umul24(a, X[2 * q + 1]) + b + (uint32_t)(CX[2 * q + 1] >> 1)) >> 8;
            }
        }
    }
}
#undef FA_
#undef FB_


// The same for bit depth 10: B' <= 2^18, so only the horizontal triple (3 B' < 2^20) stays packed; the vertical part of the
// neighbourhood sums is formed on the unpacked halves.
__device__ __forceinline__ void sgr10_filter(const uint32_t* __restrict__ abw, int i0, int j, const uint32_t (&X)[8], const int32_t (&CX)[8], bool has0, bool has1,
                                             int32_t (&D0)[8], int32_t (&D1)[8]) {
    if (has1) {
        const uint32_t* a1 = abw + i0 * S_PW + j + 1;
        uint32_t RmA, RmB, CmA, CmB, R0A, R0B, C0A, C0B;
        { const uint32_t l = a1[-1], c = a1[0], r = a1[1], t = l + c + r; RmA = t >> 20; RmB = t & 0xFFFFFu; CmA = c >> 20; CmB = c & 0xFFFFFu; }
        { const uint32_t l = a1[S_PW - 1], c = a1[S_PW], r = a1[S_PW + 1], t = l + c + r; R0A = t >> 20; R0B = t & 0xFFFFFu; C0A = c >> 20; C0B = c & 0xFFFFFu; }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t* q = a1 + (r + 2) * S_PW;
            const uint32_t l = q[-1], c = q[0], rt = q[1], t = l + c + rt;
            const uint32_t RpA = t >> 20, RpB = t & 0xFFFFFu, cA = c >> 20, cB = c & 0xFFFFFu;
            const uint32_t a = __umul24(RmA + R0A + RpA, 3u) + (CmA + R0A + cA), b = __umul24(RmB + R0B + RpB, 3u) + (CmB + R0B + cB);
            D1[r] = (int32_t)(__umul24(a, X[r]) + b + (uint32_t)CX[r]) >> 9;
            RmA = R0A; RmB = R0B; CmA = C0A; CmB = C0B; R0A = RpA; R0B = RpB; C0A = cA; C0B = cB;
        }
    }
    if (has0) {
        const uint32_t* a2 = abw + S_N1 + (i0 / 2) * S_PW + j + 1;
        uint32_t HA[5], HB[5], CA[5], CB[5];
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const uint32_t l = a2[q * S_PW - 1], c = a2[q * S_PW], r = a2[q * S_PW + 1], t = l + c + r;
            HA[q] = t >> 20; HB[q] = t & 0xFFFFFu; CA[q] = c >> 20; CB[q] = c & 0xFFFFFu;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            {
                const uint32_t a = __umul24(HA[q] + HA[q + 1], 5u) + CA[q] + CA[q + 1], b = __umul24(HB[q] + HB[q + 1], 5u) + CB[q] + CB[q + 1];
                D0[2 * q] = (int32_t)(__umul24(a, X[2 * q]) + b + (uint32_t)CX[2 * q]) >> 9;
            }
            {
                const uint32_t a = __umul24(HA[q + 1], 5u) + CA[q + 1], b = __umul24(HB[q + 1], 5u) + CB[q + 1];
                D0[2 * q + 1] = (int32_t)(__umul24(a, X[2 * q + 1]) + b + (uint32_t)(CX[2 * q + 1] >> 1)) >> 8;
            }
        }
    }
}

// 16-lane row total of p0 + p1 as a 64-bit value (|p0|, |p1| < 2^31): the low 16 bits and the signed upper halves are reduced separately
__device__ __forceinline__ long long row16_sum_wide(int32_t p0, int32_t p1) {
    const int32_t lo = row16_sum((p0 & 0xFFFF) + (p1 & 0xFFFF));
    const int32_t hi = row16_sum((p0 >> 16) + (p1 >> 16));
    return (long long)hi * 65536 + lo;
}

// One parameter set of a tile that lies completely inside the picture, bit depth 8, for the sum-only and the 6-byte STORE forms: the same arithmetic as the
// general loop body below as STRAIGHT-LINE code.  The general body asks `is this row inside the picture` and `does this set have filter 0 / 1` per row and per
// sum; the compiler turned each into a scalar branch (~30 per set) and serialised the five DPP reductions behind s_nops: the projection sums took 21 % and the
// stores 15 % of the launch for 11 % and 6 % of its instructions (tools/ubench/sgr_filter_probe.py on the MI355X).  Here the filter pair is a template
// parameter, every row is valid, and the five reductions advance in lockstep.
template <bool H0, bool H1, int STORE, int BD = 8>
__device__ __forceinline__ void sgr8_set_interior(uint32_t* __restrict__ abw, const uint32_t* __restrict__ xt, const uint32_t (&P)[S_KP], const uint32_t (&M)[S_KP], uint32_t s0,
                                                  uint32_t s1, int tid, int i0, int j, const uint32_t (&X)[8], const int32_t (&CX)[8], const int32_t (&SV)[8],
                                                  uint32_t* __restrict__ pairs_px, int dstride, int32_t* __restrict__ part_ep) {
#pragma unroll
    for (int k = 0; k < S_KP; k++) {
        const int i = tid + 256 * k;
        // positions [256 k, 256 k + 255]: all r = 1, all r = 2, or the one k that straddles S_N1 / the tail past S_NP (constants after unrolling)
        const bool all1 = 256 * k + 255 < S_N1, none1 = 256 * k >= S_N1, in_np = 256 * k + 255 < S_NP;
        const bool is1 = all1 ? true : (none1 ? false : i < S_N1);
        const bool live = (in_np ? true : i < S_NP) && (is1 ? H1 : H0);
        if (live) {
            const uint32_t z = (__umul24(P[k], is1 ? s1 : s0) + (1u << 19)) >> 20;
            const uint32_t t = xt[min(z, 255u)];
            const uint32_t B = (__umul24(t & 0x1FFu, M[k]) + (1u << 11)) >> 12;
            abw[i] = (t & 0xFFF00000u) | B;
        }
    }
    __syncthreads();   // also orders this set's build after every lane's reads of the set before last (double buffer)
    int32_t D0[8], D1[8];
    if (BD == 8) sgr8_filter(abw, i0, j, X, CX, H0, H1, D0, D1); else sgr10_filter(abw, i0, j, X, CX, H0, H1, D0, D1);
    if (STORE == 1) {
#pragma unroll
        for (int r = 0; r < 8; r++) SGR_ST(&pairs_px[(size_t)r * dstride], ((uint32_t)(H0 ? D0[r] : 0) & 0xFFFFu) | ((uint32_t)(H1 ? D1[r] : 0) << 16));
    }
    if (BD > 8) {
        // bit depth 10: |flt - u| < 2^14.1, a product < 2^28.1 -> four rows per int32 partial; the low 16 bits and the signed upper halves of the two partials are
        // reduced separately (ten values in lockstep), first inside the 16-lane rows, then over the four rows (row_bcast), and the wave's two totals per sum go to
        // its own LDS slots: [sum][wave][lo, hi], the sum is hi * 65536 + lo
        int32_t g[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (H0) { g[0][r >> 2] += __mul24(D0[r], D0[r]); g[3][r >> 2] += __mul24(D0[r], SV[r]); }
            if (H1) { g[2][r >> 2] += __mul24(D1[r], D1[r]); g[4][r >> 2] += __mul24(D1[r], SV[r]); }
            if (H0 && H1) g[1][r >> 2] += __mul24(D0[r], D1[r]);
        }
        constexpr bool use10[5] = {H0, H0 && H1, H1, H0, H1};
        int32_t v[10];
#pragma unroll
        for (int q = 0; q < 5; q++) { v[2 * q] = (g[q][0] & 0xFFFF) + (g[q][1] & 0xFFFF); v[2 * q + 1] = (g[q][0] >> 16) + (g[q][1] >> 16); }
#define SGR_RED10_(ctl)                                                                     \
        _Pragma("unroll") for (int q = 0; q < 10; q++) if (use10[q >> 1]) v[q] += __builtin_amdgcn_mov_dpp(v[q], ctl, 0xF, 0xF, true);
        SGR_RED10_(0xB1) SGR_RED10_(0x4E) SGR_RED10_(0x141) SGR_RED10_(0x140)
#undef SGR_RED10_
#pragma unroll
        for (int q = 0; q < 10; q++)
            if (use10[q >> 1]) {   // every lane of a row holds the row's total: rows 1 and 3 add the row before them, rows 2 and 3 add rows 0 + 1 -> the lanes of row 3 hold the wave's
                v[q] += __builtin_amdgcn_update_dpp(0, v[q], 0x142, 0xA, 0xF, false);   // row_bcast:15
                v[q] += __builtin_amdgcn_update_dpp(0, v[q], 0x143, 0xC, 0xF, false);   // row_bcast:31
            }
        if ((tid & 63) == 63) {
#pragma unroll
            for (int q = 0; q < 10; q++) if (use10[q >> 1]) part_ep[(q >> 1) * 8 + (tid >> 6) * 2 + (q & 1)] = v[q];
        }
        return;
    }
    int32_t h[5] = {0, 0, 0, 0, 0};   // H00, H01, H11, C0, C1
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (H0) { h[0] += __mul24(D0[r], D0[r]); h[3] += __mul24(D0[r], SV[r]); }
        if (H1) { h[2] += __mul24(D1[r], D1[r]); h[4] += __mul24(D1[r], SV[r]); }
        if (H0 && H1) h[1] += __mul24(D0[r], D1[r]);
    }
    constexpr bool use[5] = {H0, H0 && H1, H1, H0, H1};
    // the 16-lane row totals of the live sums, step by step over all of them: no reduction waits for its own previous step
#define SGR_RED_(ctl)                                                                       \
    _Pragma("unroll") for (int q = 0; q < 5; q++) if (use[q]) h[q] += __builtin_amdgcn_mov_dpp(h[q], ctl, 0xF, 0xF, true);
    SGR_RED_(0xB1) SGR_RED_(0x4E) SGR_RED_(0x141) SGR_RED_(0x140)
#undef SGR_RED_
    // the sixteen row totals of the workgroup go to slots of their own (plain stores): an LDS atomicAdd with a wave-uniform address is compiled into a
    // scalar loop over the active lanes (5 sums x 4 lanes x ~10 instructions per set); the tile's last step adds the sixteen slots
    if ((tid & 15) == 0) {
#pragma unroll
        for (int q = 0; q < 5; q++) if (use[q]) part_ep[q * 16 + (tid >> 4)] = h[q];
    }
}

// STORE: additionally leaves, per pixel, (flt0 - u) | (flt1 - u) << 16 (pairs[ep], int16 halves; sets 11 / 12 / 13 use the plane of 2 / 5 / 8), dat - src (sd,
// int16) and per unit the sum of (dat - src)^2 (d2) for the on-device unit search (sgr_walk.hip): a probe pass then re-reads 6 bytes per pixel instead of
// re-running the filters.
//
// STORE == 2 (bit depth 8 only): the PACKED form the walk of sgr_walk.hip reads since round 6 -- ONE 32-bit word per sample and filter pair that carries all three
// differences, [d1 : 11 | r_lo : 5 | d0 : 11 | r_hi : 5] with d0 = flt0 - u, d1 = flt1 - u (11-bit two's complement: |d| < 1024, i.e. the filter moved the sample by
// less than 64 levels) and r = dat - src (10-bit two's complement split in two five-bit halves, so that `word & 0xFFE0FFE0` IS the pair (32 d0, 32 d1) the walk's
// v_dot2_i32_i16 wants): 4 bytes per sample and set instead of 4 + 2 (shared), and no separate dat - src plane at all.  A sample whose d0 or d1 does not fit
// (never seen on coded pictures: a large |flt - u| needs a large local variance, which makes the filter pass the sample through; binary test pictures do produce
// them) is written as the zero word -- its error is then 0 for every candidate -- and appended, exactly, to the (unit, set)'s escape list (esc: [13 slots][dplane]
// entries of (d0 | d1 << 16, r), a unit's list starts at its first sample's offset so the lists can never collide; esc_cnt: [unit][16] counters, zeroed by the
// caller): the walk adds the listed samples one by one.  esc_lim (<= 1024) narrows the range for tests.
// The planes of a picture share ONE launch (a one-dimensional grid, the planes' tiles one after the other): a chroma plane of a 4K picture is 1035 tiles for 1024
// workgroup slots, so a launch of its own lasts one workgroup's whole latency (74 us against 56 us of throughput), and three launches have three tails.
struct SgrSearchPlaneArgs {
    const void* dgd; const void* src; unsigned long long* sums; uint32_t* pairs; int16_t* sd; unsigned long long* d2; uint2* esc; uint32_t* esc_cnt;
    size_t dplane;
    int stride, src_stride, pw, ph, unit_size, units_x, units_y, voff, dstride, tiles_x, n_tiles, esc_lim;
    uint32_t ep_mask;
};
constexpr int kSgrMaxPlanes = 12;   // the planes of up to four pictures (include/svt_hip.h: SVT_HIP_SGR_MAX_PLANES)
struct SgrSearchPic { SgrSearchPlaneArgs p[kSgrMaxPlanes]; int first_tile[kSgrMaxPlanes + 1]; };
template <typename PIX, int BD = 8, int STORE = 0>
__global__ void __launch_bounds__(256)
sgr_search8_kernel(const SgrSearchPic a) {
    // scalar copies of this workgroup's plane (a reference into the kernel-argument struct with a run-time index would force a private copy of the whole struct)
    int z = 0;
#pragma unroll
    for (int i = 1; i < kSgrMaxPlanes; i++) z += (int)blockIdx.x >= a.first_tile[i];   // first_tile of a plane that is not there = the grid size
    const PIX* __restrict__ dgd = (const PIX*)a.p[z].dgd; const PIX* __restrict__ src = (const PIX*)a.p[z].src;
    unsigned long long* __restrict__ sums = a.p[z].sums; uint32_t* __restrict__ pairs = a.p[z].pairs; int16_t* __restrict__ sd = a.p[z].sd;
    unsigned long long* __restrict__ d2 = a.p[z].d2; uint2* __restrict__ esc = a.p[z].esc; uint32_t* __restrict__ esc_cnt = a.p[z].esc_cnt;
    const size_t dplane = a.p[z].dplane;
    const int stride = a.p[z].stride, src_stride = a.p[z].src_stride, pw = a.p[z].pw, ph = a.p[z].ph, unit_size = a.p[z].unit_size, units_x = a.p[z].units_x,
              units_y = a.p[z].units_y, voff = a.p[z].voff, dstride = a.p[z].dstride, tiles_x = a.p[z].tiles_x, n_tiles = a.p[z].n_tiles, esc_lim = a.p[z].esc_lim;
    const uint32_t ep_mask = a.p[z].ep_mask;
    static_assert(STORE != 2 || BD == 8, "the packed difference words hold bit depth 8 only");
    __shared__ uint16_t in[S_IH * S_IW];
    __shared__ uint32_t ab[2][S_NP];             // [0, N1): r = 1 positions, [N1, NP): r = 2 positions (odd rows)
    __shared__ uint32_t xt[256];                 // eb_x_by_xplus1[z] << 20 | (256 - eb_x_by_xplus1[z])
    __shared__ unsigned long long acc[16][5];
    __shared__ unsigned long long acc_d2;
    const int tile = svt_xcd_order((int)blockIdx.x - a.first_tile[z], n_tiles), tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int x0 = tile_x * S_TW, y0 = tile_y * S_TH - voff, tid = threadIdx.x;   // grid shifted like the unit rows (foreach_rest_unit_in_tile, EbRestoration.c:1388-1391: unit rows start 8 >> ss_y above their nominal position, so a tile never straddles two units)
    const int unit = min((y0 + voff) / unit_size, units_y - 1) * units_x + min(x0 / unit_size, units_x - 1);
    size_t esc_base = 0;   // first sample of this tile's unit in list order: (first row) x dstride + (first column) x (rows): units of a row band share the band's rows
    if (STORE == 2) {
        const int ui = unit / units_x, uj = unit - ui * units_x, uy0 = ui * unit_size, uh = ui == units_y - 1 ? ph - uy0 : unit_size;
        const int v0 = max(uy0 - voff, 0), v1 = (uy0 + uh < ph) ? uy0 + uh - voff : uy0 + uh;
        esc_base = (size_t)v0 * dstride + (size_t)(uj * unit_size) * (size_t)(v1 - v0);
    }

    {
        const uint32_t A = tid == 0 ? 1u : (tid == 255 ? 256u : (uint32_t)((256 * tid + (tid + 1) / 2) / (tid + 1)));
        xt[tid] = (A << 20) | (256u - A);
        if (tid < 80) acc[tid / 5][tid % 5] = 0ull;
        if (tid == 80) acc_d2 = 0ull;
    }
    batched_stage<6, uint16_t>(S_IH * S_IW, tid, 256,
        [&](int i) {
            const int r = i / S_IW, c = i - r * S_IW;
            const int x = min(max(x0 - 3 + c, -3), pw + 2), y = min(max(y0 - 3 + r, -3), ph + 2);   // never leave the 3-px extension
            return (uint16_t)dgd[(ptrdiff_t)y * stride + x];
        },
        [&](int i, uint16_t v) { in[i] = v; });
    __syncthreads();

    uint32_t P[S_KP], M[S_KP];
    sgr8_precompute<BD>(in, &ab[0][0], tid, P, M);

    // ---- the 8 pixels (one column, 8 rows) this lane accumulates
    const int j = tid & 63, i0 = (tid >> 6) * 8;
    const int rlo = min(max(-(y0 + i0), 0), 8), rhi = min(max(ph - (y0 + i0), 0), 8);   // rows [rlo, rhi) of the 8 are inside the picture
    const bool colvalid = x0 + j < pw;
    uint32_t X[8]; int32_t SV[8], CX[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        X[r] = in[(i0 + r + 3) * S_IW + j + 3];
        const int yy = min(max(y0 + i0 + r, 0), ph - 1), xx = min(x0 + j, pw - 1);
        SV[r] = ((int32_t)src[(size_t)yy * src_stride + xx] - (int32_t)X[r]) << 4;     // (src << 4) - u
        CX[r] = 256 - (int32_t)(X[r] << 13);                                           // rounding - (u << 9)
        if (STORE == 1 && colvalid && r >= rlo && r < rhi) sd[(size_t)(y0 + i0 + r) * dstride + x0 + j] = (int16_t)(-(SV[r] >> 4));   // dat - src
    }
    if (STORE) {   // sum of (dat - src)^2 over the unit: the constant term of the quadratic error model the walk speculates on
        int32_t q = 0;
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (colvalid && r >= rlo && r < rhi) { const int32_t d = SV[r] >> 4; q += d * d; }   // <= 8 x 1023^2 < 2^24
        q = row16_sum(q);
        if ((tid & 15) == 0 && q) atomicAdd(&acc_d2, (unsigned long long)q);
    }

    // parameter sets that must be filtered: the masked ones, 11/12/13 folded onto 2/5/8
    uint32_t cmask = ep_mask & 0xC7FFu;
    if (ep_mask & (1u << 11)) cmask |= 1u << 2;
    if (ep_mask & (1u << 12)) cmask |= 1u << 5;
    if (ep_mask & (1u << 13)) cmask |= 1u << 8;

    const bool interior = x0 + S_TW <= pw && y0 >= 0 && y0 + S_TH <= ph;   // workgroup-uniform: every sample of the tile is a picture sample
    int32_t* part = (int32_t*)in;   // [16 sets][5 sums][16 row totals] of an interior tile: the staged tile is dead (every lane has read its X[] before the first set's barrier)
    static_assert(16 * 5 * 16 * sizeof(int32_t) <= sizeof(in), "the row totals must fit the staged tile");
    int buf = 0;
    for (int ep = 0; ep < 16; ep++) {
        if (!((cmask >> ep) & 1)) continue;
        const bool has0 = kSgr[ep][0] > 0, has1 = kSgr[ep][1] > 0;
        const uint32_t s0 = (uint32_t)kSgr[ep][2], s1 = (uint32_t)kSgr[ep][3];
        uint32_t* abw = ab[buf];
        if (STORE != 2 && interior) {   // the common case as straight-line code, one instance per filter pair
            uint32_t* ppx = STORE == 1 ? &pairs[(size_t)ep * dplane + (size_t)(y0 + i0) * dstride + x0 + j] : nullptr;
            if (has0 && has1) sgr8_set_interior<true, true, STORE, BD>(abw, xt, P, M, s0, s1, tid, i0, j, X, CX, SV, ppx, dstride, part + ep * 80);
            else if (has1) sgr8_set_interior<false, true, STORE, BD>(abw, xt, P, M, s0, s1, tid, i0, j, X, CX, SV, ppx, dstride, part + ep * 80);
            else sgr8_set_interior<true, false, STORE, BD>(abw, xt, P, M, s0, s1, tid, i0, j, X, CX, SV, ppx, dstride, part + ep * 80);
            buf ^= 1;
            continue;
        }
        sgr8_build(abw, xt, P, M, has0, has1, s0, s1, tid);
        __syncthreads();   // also orders this set's build after every lane's reads of the set before last (double buffer)

        if (BD == 8) {
            int32_t D0[8], D1[8];
            sgr8_filter(abw, i0, j, X, CX, has0, has1, D0, D1);
            if (STORE == 1 && colvalid) {   // (flt0 - u) | (flt1 - u) << 16: one dword per pixel and filter pair (64 lanes = 256 contiguous bytes per row)
#pragma unroll
                for (int r = 0; r < 8; r++)
                    if (r >= rlo && r < rhi)
                        SGR_ST(&pairs[(size_t)ep * dplane + (size_t)(y0 + i0 + r) * dstride + x0 + j], ((uint32_t)(has0 ? D0[r] : 0) & 0xFFFFu) | ((uint32_t)(has1 ? D1[r] : 0) << 16));
            }
            if (STORE == 2 && colvalid) {   // the packed word (see above); the rare sample that does not fit goes to the (unit, set)'s list
#pragma unroll
                for (int r = 0; r < 8; r++)
                    if (r >= rlo && r < rhi) {
                        const int32_t d0 = has0 ? D0[r] : 0, d1 = has1 ? D1[r] : 0, rr = -(SV[r] >> 4);
                        uint32_t wd = ((uint32_t)d1 << 21) | (((uint32_t)rr & 31u) << 16) | (((uint32_t)d0 & 0x7FFu) << 5) | (((uint32_t)rr >> 5) & 31u);
                        if ((uint32_t)(d0 + esc_lim) >= 2u * (uint32_t)esc_lim || (uint32_t)(d1 + esc_lim) >= 2u * (uint32_t)esc_lim) {
                            const uint32_t at = atomicAdd(&esc_cnt[(size_t)unit * 16 + ep], 1u);
                            esc[(size_t)(ep < 11 ? ep : ep - 3) * dplane + esc_base + at] = make_uint2(((uint32_t)d0 & 0xFFFFu) | ((uint32_t)d1 << 16), (uint32_t)rr);
                            wd = 0u;
                        }
                        SGR_ST(&pairs[(size_t)ep * dplane + (size_t)(y0 + i0 + r) * dstride + x0 + j], wd);
                    }
            }
            int32_t h00 = 0, h01 = 0, h11 = 0, c0 = 0, c1 = 0;
    #pragma unroll
            for (int r = 0; r < 8; r++) {
                if (r >= rlo && r < rhi) {   // wave-uniform
                    if (has0) { h00 += __mul24(D0[r], D0[r]); c0 += __mul24(D0[r], SV[r]); }
                    if (has1) { h11 += __mul24(D1[r], D1[r]); c1 += __mul24(D1[r], SV[r]); }
                    if (has0 && has1) h01 += __mul24(D0[r], D1[r]);
                }
            }
            if (!colvalid) { h00 = 0; h01 = 0; h11 = 0; c0 = 0; c1 = 0; }
            if (has0) { h00 = row16_sum(h00); c0 = row16_sum(c0); }
            if (has1) { h11 = row16_sum(h11); c1 = row16_sum(c1); }
            if (has0 && has1) h01 = row16_sum(h01);
            if ((tid & 15) == 0) {
                if (has0) { atomicAdd(&acc[ep][0], (unsigned long long)(long long)h00); atomicAdd(&acc[ep][3], (unsigned long long)(long long)c0); }
                if (has1) { atomicAdd(&acc[ep][2], (unsigned long long)(long long)h11); atomicAdd(&acc[ep][4], (unsigned long long)(long long)c1); }
                if (has0 && has1) atomicAdd(&acc[ep][1], (unsigned long long)(long long)h01);
            }
        } else {
            // bit depth 10: |flt - u| < 2^14.1, a product < 2^28.1 -> four rows per int32 partial, 64-bit from the row reduction on
            int32_t D0[8], D1[8];
            sgr10_filter(abw, i0, j, X, CX, has0, has1, D0, D1);
            if (STORE == 1 && colvalid) {   // (flt0 - u) | (flt1 - u) << 16: one dword per pixel and filter pair (64 lanes = 256 contiguous bytes per row)
#pragma unroll
                for (int r = 0; r < 8; r++)
                    if (r >= rlo && r < rhi)
                        SGR_ST(&pairs[(size_t)ep * dplane + (size_t)(y0 + i0 + r) * dstride + x0 + j], ((uint32_t)(has0 ? D0[r] : 0) & 0xFFFFu) | ((uint32_t)(has1 ? D1[r] : 0) << 16));
            }
            int32_t h00[2] = {0, 0}, h01[2] = {0, 0}, h11[2] = {0, 0}, c0[2] = {0, 0}, c1[2] = {0, 0};
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (r >= rlo && r < rhi) {   // wave-uniform
                    if (has0) { h00[r >> 2] += __mul24(D0[r], D0[r]); c0[r >> 2] += __mul24(D0[r], SV[r]); }
                    if (has1) { h11[r >> 2] += __mul24(D1[r], D1[r]); c1[r >> 2] += __mul24(D1[r], SV[r]); }
                    if (has0 && has1) h01[r >> 2] += __mul24(D0[r], D1[r]);
                }
            }
            if (!colvalid) { h00[0] = h00[1] = 0; h01[0] = h01[1] = 0; h11[0] = h11[1] = 0; c0[0] = c0[1] = 0; c1[0] = c1[1] = 0; }
            long long w00 = 0, w01 = 0, w11 = 0, wc0 = 0, wc1 = 0;
            if (has0) { w00 = row16_sum_wide(h00[0], h00[1]); wc0 = row16_sum_wide(c0[0], c0[1]); }
            if (has1) { w11 = row16_sum_wide(h11[0], h11[1]); wc1 = row16_sum_wide(c1[0], c1[1]); }
            if (has0 && has1) w01 = row16_sum_wide(h01[0], h01[1]);
            if ((tid & 15) == 0) {
                if (has0) { atomicAdd(&acc[ep][0], (unsigned long long)w00); atomicAdd(&acc[ep][3], (unsigned long long)wc0); }
                if (has1) { atomicAdd(&acc[ep][2], (unsigned long long)w11); atomicAdd(&acc[ep][4], (unsigned long long)wc1); }
                if (has0 && has1) atomicAdd(&acc[ep][1], (unsigned long long)w01);
            }
        }
        buf ^= 1;
    }
    __syncthreads();
    if (tid < 80) {
        const int ep = tid / 5, q = tid - ep * 5;
        if ((ep_mask >> ep) & 1) {
            const int ce = ep == 11 ? 2 : (ep == 12 ? 5 : (ep == 13 ? 8 : ep));
            const bool has0 = kSgr[ep][0] > 0, has1 = kSgr[ep][1] > 0;
            const bool used = q == 0 || q == 3 ? has0 : (q == 1 ? (has0 && has1) : has1);
            if (used) {
                unsigned long long v = acc[ce][q];
                if (STORE != 2 && interior) {
                    long long w = 0;
                    if (BD == 8) {
#pragma unroll
                        for (int k = 0; k < 16; k++) w += part[ce * 80 + q * 16 + k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) w += (long long)part[ce * 80 + q * 8 + 2 * k + 1] * 65536 + part[ce * 80 + q * 8 + 2 * k];
                    }
                    v = (unsigned long long)w;
                }
                atomicAdd(&sums[((size_t)unit * 16 + ep) * 5 + q], v);
            }
        }
    }
    if (STORE && tid == 80 && acc_d2) atomicAdd(&d2[unit], acc_d2);
}

// ---- projected error of xqd candidates: get_pixel_proj_error (EbRestorationPick.c:317-351 -> svt_av1_{lowbd,highbd}_pixel_proj_error, :174-316) for
// every (restoration unit, parameter set in ep_mask, candidate c < ncand): err[unit][16][ncand] += sum over the unit of
// (((u << 7) + xq0 (flt0 - u) + xq1 (flt1 - u) + 2^10) >> 11) - src)^2 with xq = svt_decode_xq(xqd[unit][16][ncand][2]).  Same tiling and filter
// passes as the search kernel; the candidates of a (unit, set) are workgroup-uniform, so their decode is scalar work.
constexpr int kSgrMaxCand = 24;
template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
sgr_proj_error_kernel(const PIX* __restrict__ dgd, int stride, const PIX* __restrict__ src, int src_stride, int pw, int ph, int unit_size,
                      int units_x, int units_y, int voff, uint32_t ep_mask, int ncand, const int32_t* __restrict__ xqd,
                      unsigned long long* __restrict__ err) {
    __shared__ uint16_t in[S_IH * S_IW];
    __shared__ uint32_t ab[2][S_NP];
    __shared__ uint32_t xt[256];
    __shared__ unsigned long long acc[16][kSgrMaxCand];
    const int x0 = blockIdx.x * S_TW, y0 = blockIdx.y * S_TH - voff, tid = threadIdx.x;
    const int unit = min((y0 + voff) / unit_size, units_y - 1) * units_x + min(x0 / unit_size, units_x - 1);
    // a first candidate of INT32_MIN skips the (unit, set): later rounds of the host's finer search only have a few walks still open
    for (int ep = 0; ep < 16; ep++)
        if (((ep_mask >> ep) & 1) && xqd[((size_t)unit * 16 + ep) * ncand * 2] == INT32_MIN) ep_mask &= ~(1u << ep);
    if (!ep_mask) return;
    {
        const uint32_t A = tid == 0 ? 1u : (tid == 255 ? 256u : (uint32_t)((256 * tid + (tid + 1) / 2) / (tid + 1)));
        xt[tid] = (A << 20) | (256u - A);
        for (int k = tid; k < 16 * kSgrMaxCand; k += 256) acc[k / kSgrMaxCand][k % kSgrMaxCand] = 0ull;
    }
    batched_stage<6, uint16_t>(S_IH * S_IW, tid, 256,
        [&](int i) {
            const int r = i / S_IW, c = i - r * S_IW;
            const int x = min(max(x0 - 3 + c, -3), pw + 2), y = min(max(y0 - 3 + r, -3), ph + 2);
            return (uint16_t)dgd[(ptrdiff_t)y * stride + x];
        },
        [&](int i, uint16_t v) { in[i] = v; });
    __syncthreads();
    uint32_t P[S_KP], M[S_KP];
    sgr8_precompute<BD>(in, &ab[0][0], tid, P, M);
    const int j = tid & 63, i0 = (tid >> 6) * 8;
    const int rlo = min(max(-(y0 + i0), 0), 8), rhi = min(max(ph - (y0 + i0), 0), 8);
    const bool colvalid = x0 + j < pw;
    uint32_t X[8]; int32_t BS[8], CX[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        X[r] = in[(i0 + r + 3) * S_IW + j + 3];
        const int yy = min(max(y0 + i0 + r, 0), ph - 1), xx = min(x0 + j, pw - 1);
        BS[r] = (int32_t)(X[r] << 11) + (1 << 10) - ((int32_t)src[(size_t)yy * src_stride + xx] << 11);   // (u << 7) + rounding - (src << 11): e = (BS + xq . D) >> 11
        CX[r] = 256 - (int32_t)(X[r] << 13);
    }
    int buf = 0;
    for (int ep = 0; ep < 16; ep++) {
        if (!((ep_mask >> ep) & 1)) continue;
        const bool has0 = kSgr[ep][0] > 0, has1 = kSgr[ep][1] > 0;
        uint32_t* abw = ab[buf];
        sgr8_build(abw, xt, P, M, has0, has1, (uint32_t)kSgr[ep][2], (uint32_t)kSgr[ep][3], tid);
        __syncthreads();
        int32_t D0[8], D1[8];
        if constexpr (BD == 8) sgr8_filter(abw, i0, j, X, CX, has0, has1, D0, D1);
        else sgr10_filter(abw, i0, j, X, CX, has0, has1, D0, D1);
        const int32_t* q = xqd + ((size_t)unit * 16 + ep) * ncand * 2;
        for (int c = 0; c < ncand; c++) {
            const int32_t xqd0 = q[2 * c], xqd1 = q[2 * c + 1];
            if (xqd0 == INT32_MIN) break;   // end of this (unit, set)'s candidate list (workgroup-uniform)
            const int32_t xq0 = has0 ? xqd0 : 0, xq1 = !has1 ? 0 : (has0 ? 128 - xqd0 - xqd1 : 128 - xqd1);   // svt_decode_xq (EbRestoration.c:707-718)
            int32_t p0 = 0, p1 = 0;   // rows 0-3 / 4-7: |e| < 2^13 at bit depth 10, four squares stay far below 2^31
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (r >= rlo && r < rhi) {
                    int32_t v = BS[r];
                    if (has0) v += __mul24(xq0, D0[r]);
                    if (has1) v += __mul24(xq1, D1[r]);
                    const int32_t e = v >> 11;
                    if (r < 4) p0 += __mul24(e, e); else p1 += __mul24(e, e);
                }
            }
            if (!colvalid) { p0 = 0; p1 = 0; }
            long long t;
            if constexpr (BD == 8) t = row16_sum(p0 + p1);   // 8 px x 16 lanes x 1300^2 < 2^31
            else t = row16_sum_wide(p0, p1);
            if ((tid & 15) == 0) atomicAdd(&acc[ep][c], (unsigned long long)t);
        }
        buf ^= 1;
    }
    __syncthreads();
    for (int k = tid; k < 16 * ncand; k += 256) {
        const int ep = k / ncand, c = k - ep * ncand;
        if ((ep_mask >> ep) & 1) atomicAdd(&err[((size_t)unit * 16 + ep) * ncand + c], acc[ep][c]);
    }
}

// ---- pieces shared by the frame apply and the Wiener tap walk: staging of one 64 x 32 restoration tile with its 3-sample surround (stripe rules: rows outside the
// tile's stripe come from the deblocked picture's two boundary rows, stretched to three), and svt_av1_wiener_convolve_add_src's two passes on the staged tile
template <typename PIX>
__device__ __forceinline__ StripeCtx<PIX> lr_stripe_of(const PIX* dbl, int dbl_stride, int y0, int voff, int stripe_h, int ph) {
    StripeCtx<PIX> sc{nullptr, 0, 0, 0, 0, 0};
    if (dbl) {
        const int s = (y0 + voff) / stripe_h;
        sc.dbl = dbl; sc.dbl_stride = dbl_stride;
        sc.sy0 = max(0, s * stripe_h - voff); sc.sy1 = min((s + 1) * stripe_h - voff, ph);
        sc.above = s > 0; sc.below = sc.sy1 < ph;
    }
    return sc;
}
// staged sample i of the tile at (x0, y0): the picture, or inside the stripe's context rows the deblocked picture (two saved rows stretched to three)
template <typename PIX>
__device__ __forceinline__ uint16_t lr_tile_sample(const PIX* __restrict__ dgd, int stride, int pw, int ph, int x0, int y0, const StripeCtx<PIX>& sc, int i) {
    const int r = i / S_IW, c = i - r * S_IW;
    const int yy = y0 - 3 + r, xx = x0 - 3 + c;
    // one unconditional load from a selected address (three loads under three conditions are three branches, each waiting for its own load)
    const bool up = sc.above && yy < sc.sy0, dn = sc.below && yy >= sc.sy1, ctx = up || dn;
    const int  yd = up ? (yy == sc.sy0 - 1 ? sc.sy0 - 1 : sc.sy0 - 2) : min(yy == sc.sy1 ? sc.sy1 : sc.sy1 + 1, ph - 1);
    const int  x = ctx ? min(max(xx, 0), pw - 1) : min(max(xx, -3), pw + 2), y = ctx ? yd : min(max(yy, -3), ph + 2);   // the picture: never leave the 3-px extension
    const PIX* base = ctx ? sc.dbl : dgd;
    return (uint16_t)base[(ptrdiff_t)y * (ctx ? sc.dbl_stride : stride) + x];
}
template <typename PIX>
__device__ __forceinline__ void lr_stage_tile(uint16_t* __restrict__ in, const PIX* __restrict__ dgd, int stride, int pw, int ph, int x0, int y0, const StripeCtx<PIX>& sc, int tid, int nt) {
    batched_stage<6, uint16_t>(S_IH * S_IW, tid, nt, [&](int i) { return lr_tile_sample<PIX>(dgd, stride, pw, ph, x0, y0, sc, i); }, [&](int i, uint16_t v) { in[i] = v; });
}
// horizontal pass (round 3) of the staged tile -> tmp [S_IH][S_TW]  (Common/Codec/convolve.c:60-145)
// A thread takes FOUR horizontally adjacent outputs: their ten inputs are five aligned dwords of the staged row (a row starts on a dword: S_IW is even) instead of 28
// 16-bit LDS reads, and the four results leave as two dwords.
template <int BD>
__device__ __forceinline__ void wiener_hpass(const uint16_t* __restrict__ in, uint16_t* __restrict__ tmp, const int (&fx)[8], int tid, int nt) {
    static_assert(S_IW % 2 == 0 && S_TW % 4 == 0, "dword-aligned rows");
    for (int g = tid; g < S_IH * (S_TW / 4); g += nt) {
        const int r = g / (S_TW / 4), c = (g - r * (S_TW / 4)) * 4;
        const uint32_t* row = (const uint32_t*)(in + r * S_IW + c);
        int32_t x[10];
#pragma unroll
        for (int k = 0; k < 5; k++) { const uint32_t v = row[k]; x[2 * k] = (int32_t)(v & 0xFFFFu); x[2 * k + 1] = (int32_t)(v >> 16); }
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int32_t sum = (x[q + 3] << 7) + (1 << (BD + 6));
#pragma unroll
            for (int t = 0; t < 7; t++) sum += x[q + t] * fx[t];
            o[q] = (uint32_t)min(max((sum + 4) >> 3, 0), (1 << (BD + 5)) - 1);   // WIENER_CLAMP_LIMIT(3, bd)
        }
        uint32_t* out = (uint32_t*)(tmp + r * S_TW + c);
        out[0] = o[0] | (o[1] << 16); out[1] = o[2] | (o[3] << 16);
    }
}
// vertical pass (round 11) for the eight rows i0 .. i0 + 7 of column j: the fourteen intermediate rows are read once (a sliding window in registers) instead of 56 times
template <int BD>
__device__ __forceinline__ void wiener_vcol8(const uint16_t* __restrict__ tmp, int i0, int j, const int (&fy)[8], int (&out)[8]) {
    int32_t t[14];
#pragma unroll
    for (int k = 0; k < 14; k++) t[k] = (int32_t)tmp[(i0 + k) * S_TW + j];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int32_t sum = (t[r + 3] << 7) - (1 << (BD + 10));
#pragma unroll
        for (int q = 0; q < 7; q++) sum += t[r + q] * fy[q];
        out[r] = min(max((sum + (1 << 10)) >> 11, 0), (1 << BD) - 1);
    }
}

// ---- frame apply (bit depth 8 and 10) on the search kernel's machinery (64 x 32 tiles, separable box sums, packed A'/B', one parameter set per unit):
// RESTORE_NONE units are copied, RESTORE_WIENER units take the 7-tap separable filter, RESTORE_SGRPROJ units the self-guided filter.
// A tile is one stripe high at most (stripes are 64 >> ss_y rows starting 8 >> ss_y above a multiple of that), so the StripeCtx rules
// of the generic kernel apply unchanged.
template <typename PIX, int BD = 8>
__global__ void __launch_bounds__(256)
lr_apply8_kernel(const PIX* __restrict__ dgd, int stride, PIX* __restrict__ dst, int dst_stride, int pw, int ph, int unit_size, int units_x,
                 int units_y, int voff, int stripe_h, const PIX* __restrict__ dbl, int dbl_stride, const uint8_t* __restrict__ unit_ep,
                 const int32_t* __restrict__ unit_xqd, const int16_t* __restrict__ unit_wiener, int tile_x0, int tile_y0) {
    __shared__ __attribute__((aligned(16))) uint16_t in[S_IH * S_IW];   // (rows are read as dwords by wiener_hpass)
    __shared__ uint32_t ab[2][S_NP];
    __shared__ uint32_t xt[256];
    const int tile = svt_xcd_order(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y), tile_y = tile / (int)gridDim.x, tile_x = tile - tile_y * (int)gridDim.x;
    const int x0 = (tile_x + tile_x0) * S_TW, y0 = (tile_y + tile_y0) * S_TH - voff, tid = threadIdx.x;   // (tile_x0, tile_y0): first tile of a partial launch
    const int unit = min((y0 + voff) / unit_size, units_y - 1) * units_x + min(x0 / unit_size, units_x - 1);
    const int ep = unit_ep[unit];
    const bool wiener = ep == 254 && unit_wiener != nullptr;
    if (ep > 15 && !wiener) {   // copy_tile (EbRestoration.c:1174-1177)
        for (int k = tid; k < S_TW * S_TH; k += 256) {
            const int i = k / S_TW, j = k - i * S_TW;
            if (x0 + j < pw && y0 + i < ph && y0 + i >= 0) dst[(size_t)(y0 + i) * dst_stride + x0 + j] = dgd[(ptrdiff_t)(y0 + i) * stride + x0 + j];
        }
        return;
    }
    const StripeCtx<PIX> sc = lr_stripe_of<PIX>(dbl, dbl_stride, y0, voff, stripe_h, ph);
    {
        const uint32_t A = tid == 0 ? 1u : (tid == 255 ? 256u : (uint32_t)((256 * tid + (tid + 1) / 2) / (tid + 1)));
        xt[tid] = (A << 20) | (256u - A);
    }
    lr_stage_tile<PIX>(in, dgd, stride, pw, ph, x0, y0, sc, tid, 256);
    __syncthreads();
    const int j = tid & 63, i0 = (tid >> 6) * 8;
    if (wiener) {   // svt_av1_wiener_convolve_add_src (Common/Codec/convolve.c:60-145): horizontal pass (round 3), vertical pass (round 11)
        int fx[8], fy[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { fy[k] = unit_wiener[16 * unit + k]; fx[k] = unit_wiener[16 * unit + 8 + k]; }
        uint16_t* tmp = (uint16_t*)&ab[0][0];   // [S_IH][S_TW]
        wiener_hpass<BD>(in, tmp, fx, tid, 256);
        __syncthreads();
        int v8[8];
        wiener_vcol8<BD>(tmp, i0, j, fy, v8);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int i = i0 + r;
            if (x0 + j >= pw || y0 + i >= ph || y0 + i < 0) continue;
            dst[(size_t)(y0 + i) * dst_stride + x0 + j] = (PIX)v8[r];
        }
        return;
    }
    uint32_t P[S_KP], M[S_KP];
    sgr8_precompute<BD>(in, &ab[0][0], tid, P, M);
    const bool has0 = kSgr[ep][0] > 0, has1 = kSgr[ep][1] > 0;
    sgr8_build(ab[0], xt, P, M, has0, has1, (uint32_t)kSgr[ep][2], (uint32_t)kSgr[ep][3], tid);
    __syncthreads();
    uint32_t X[8]; int32_t CX[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { X[r] = in[(i0 + r + 3) * S_IW + j + 3]; CX[r] = 256 - (int32_t)(X[r] << 13); }
    int32_t D0[8], D1[8];
    if constexpr (BD == 8) sgr8_filter(ab[0], i0, j, X, CX, has0, has1, D0, D1);
    else sgr10_filter(ab[0], i0, j, X, CX, has0, has1, D0, D1);
    // svt_decode_xq (EbRestoration.c:707-718)
    const int32_t xqd0 = unit_xqd[2 * unit], xqd1 = unit_xqd[2 * unit + 1];
    int32_t xq0, xq1;
    if (!has0) { xq0 = 0; xq1 = 128 - xqd1; }
    else if (!has1) { xq0 = xqd0; xq1 = 0; }
    else { xq0 = xqd0; xq1 = 128 - xq0 - xqd1; }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int i = i0 + r;
        if (x0 + j >= pw || y0 + i >= ph || y0 + i < 0) continue;
        int32_t v = (int32_t)(X[r] << 11);   // u << SGRPROJ_PRJ_BITS, u = x << SGRPROJ_RST_BITS
        if (has0) v += xq0 * D0[r];
        if (has1) v += xq1 * D1[r];
        const int32_t w = (int32_t)(int16_t)((v + (1 << 10)) >> 11);
        dst[(size_t)(y0 + i) * dst_stride + x0 + j] = (PIX)min(max(w, 0), (1 << BD) - 1);
    }
}

// ---- finer_tile_search_wiener_seg (Encoder/Codec/EbRestorationPick.c:1092-1200) for every restoration unit of a plane, ON THE DEVICE: one workgroup of four
// 256-thread teams per unit.  Thread 0 runs the reference's coordinate descent as a state machine (step sizes 4, 2, 1; the horizontal taps, then the vertical ones;
// a downward probe, then an upward one; at step 4 an accepted probe repeats — wn_issue / wn_result below); every probe = try_restoration_unit_seg (:137): the unit
// filtered with the probed taps (the apply kernel's staging and passes, tile by tile, a team per tile) and its squared error against the source, which never
// leaves the chip.  No host round trip per probe (round 3: 20 - 40 lockstep rounds of upload / launch / download per picture).
struct WnWalkState { int s, ph, p, up, skip, state; long long err; int16_t v[8], h[8]; };
__device__ __forceinline__ void wn_apply(int16_t* f, int p, int d) { f[p] += (int16_t)d; f[6 - p] += (int16_t)d; f[3] -= (int16_t)(2 * d); }   // WIENER_WIN 7, WIENER_HALFWIN 3
// moves to the next probe: true = one is outstanding (its taps are in w.v / w.h), false = the walk is over
__device__ inline bool wn_issue(WnWalkState& w, int off) {
    // WIENER_FILT_TAP{0,1,2}_{MINV,MAXV} (Common/Codec/EbRestoration.h:130-150): {-5, -23, -17} .. {10, 8, 46}
    auto tmin_of = [](int p) { return p == 0 ? -5 : (p == 1 ? -23 : -17); };
    auto tmax_of = [](int p) { return p == 0 ? 10 : (p == 1 ? 8 : 46); };
    for (;;) {
        if (w.s < 1) { w.state = 0; return false; }
        if (w.p >= 3) {           // this filter's taps are done: vertical after horizontal, then the next step size
            if (w.ph == 0) w.ph = 1; else { w.ph = 0; w.s >>= 1; }
            w.p = off; w.up = 0; w.skip = 0;
            continue;
        }
        int16_t* f = w.ph ? w.v : w.h;
        if (!w.up) {
            if (f[w.p] - w.s >= tmin_of(w.p)) { wn_apply(f, w.p, -w.s); w.state = 2; return true; }
            if (w.skip) w.p = 3; else w.up = 1;    // "if (skip) break" (:1126)
        } else {
            if (f[w.p] + w.s <= tmax_of(w.p)) { wn_apply(f, w.p, w.s); w.state = 2; return true; }
            w.p++; w.up = 0; w.skip = 0;
        }
    }
}
__device__ inline bool wn_result(WnWalkState& w, long long err2, int off) {
    if (w.state == 1) { w.err = err2; w.s = 4; w.ph = 0; w.p = off; w.up = 0; w.skip = 0; return wn_issue(w, off); }
    int16_t*  f = w.ph ? w.v : w.h;
    const int d = w.up ? w.s : -w.s;
    const bool accepted = !(err2 > w.err);
    if (!accepted) wn_apply(f, w.p, -d);
    else { w.err = err2; if (!w.up) w.skip = 1; }
    if (!(accepted && w.s == 4)) {             // at the highest step size an accepted probe keeps moving in the same direction
        if (!w.up) { if (w.skip) w.p = 3; else w.up = 1; }
        else { w.p++; w.up = 0; w.skip = 0; }
    }
    return wn_issue(w, off);
}
struct WnPlane { const void* dgd; const void* dbl; const void* src; int16_t* unit_wiener; const uint8_t* active; long long* err; uint32_t* probes;
                 int stride, pw, ph, unit_size, units_x, units_y, voff, stripe_h, dbl_stride, src_stride, win; };
struct WnPic { WnPlane p[3]; };   // blockIdx.y = plane: the planes of a picture walk side by side (a launch lasts as long as its longest walk)
template <typename PIX, int BD>
__global__ void __launch_bounds__(1024)
wiener_walk_kernel(const WnPic a) {
    // scalar copies of the plane's arguments (a reference into the kernel-argument struct with a run-time index would force a private copy of the whole struct)
    const int z = blockIdx.y;
    const PIX* __restrict__ dgd = (const PIX*)a.p[z].dgd; const PIX* __restrict__ dbl = (const PIX*)a.p[z].dbl; const PIX* __restrict__ src = (const PIX*)a.p[z].src;
    int16_t* __restrict__ unit_wiener = a.p[z].unit_wiener; const uint8_t* __restrict__ active = a.p[z].active;
    long long* __restrict__ err_out = a.p[z].err; uint32_t* __restrict__ probes_out = a.p[z].probes;
    const int stride = a.p[z].stride, pw = a.p[z].pw, ph = a.p[z].ph, unit_size = a.p[z].unit_size, units_x = a.p[z].units_x, units_y = a.p[z].units_y, voff = a.p[z].voff,
              stripe_h = a.p[z].stripe_h, dbl_stride = a.p[z].dbl_stride, src_stride = a.p[z].src_stride, win = a.p[z].win;
    if ((int)blockIdx.x >= units_x * units_y) return;   // a plane with fewer units than the widest one
    __shared__ __attribute__((aligned(16))) uint16_t in[4][S_IH * S_IW];    // (rows are read as dwords by wiener_hpass)
    __shared__ __attribute__((aligned(16))) uint16_t tmp[4][S_IH * S_TW];
    __shared__ int taps[16];            // the probe: [0..7] vertical, [8..15] horizontal
    __shared__ unsigned long long part[16];
    __shared__ int go;
    const int unit = blockIdx.x, tid = threadIdx.x, team = tid >> 8, tt = tid & 255;
    if (!active[unit]) return;
    const int ux = unit % units_x, uy = unit / units_x, off = (7 - win) >> 1;
    // the unit's rectangle (foreach_rest_unit_in_tile, EbRestoration.c:1369-1411): whole tiles of the apply kernel's grid
    const int rx0 = ux * unit_size, rx1 = ux == units_x - 1 ? pw : (ux + 1) * unit_size;
    const int ry0 = max(uy * unit_size - voff, 0), ry1 = uy == units_y - 1 ? ph : (uy + 1) * unit_size - voff;
    const int tiles_x = (rx1 - rx0 + S_TW - 1) / S_TW, ty_first = (ry0 + voff) / S_TH, tiles_y = (ry1 + voff + S_TH - 1) / S_TH - ty_first, n_tiles = tiles_x * tiles_y;
    __shared__ WnWalkState w_lds;   // thread 0's walk state: its tap arrays are indexed at run time, which as a private object means scratch memory (a memory round trip per access)
    WnWalkState& w = w_lds;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { w.v[k] = unit_wiener[16 * unit + k]; w.h[k] = unit_wiener[16 * unit + 8 + k]; }
        w.state = 1; w.err = 0; w.s = 0; w.ph = 0; w.p = 0; w.up = 0; w.skip = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { taps[k] = w.v[k]; taps[8 + k] = w.h[k]; }
        go = 1;
    }
    uint32_t n_probes = 0;
    __syncthreads();
    while (go) {
        int fx[8], fy[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { fy[k] = taps[k]; fx[k] = taps[8 + k]; }
        unsigned long long sse = 0;
        // a team's tiles are software-pipelined: the samples of the NEXT tile are loaded into registers while this tile is filtered (a probe is a chain of dependent
        // tile rounds on one compute unit; the staging loads were the longest wait of a round)
        constexpr int kPre = (S_IH * S_IW + 255) / 256;
        uint16_t pre[kPre];
        auto load_tile = [&](int t) {
            const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
            const int x0 = rx0 + txi * S_TW, y0 = (ty_first + tyi) * S_TH - voff;
            const StripeCtx<PIX> sc = lr_stripe_of<PIX>(dbl, dbl_stride, y0, voff, stripe_h, ph);
#pragma unroll
            for (int k = 0; k < kPre; k++) { const int i = tt + 256 * k; pre[k] = i < S_IH * S_IW ? lr_tile_sample<PIX>(dgd, stride, pw, ph, x0, y0, sc, i) : (uint16_t)0; }
        };
        if (team < n_tiles) load_tile(team);
        for (int t = team; t < n_tiles; t += 4) {   // uniform per team: its four waves pass the same barriers
            const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
            const int x0 = rx0 + txi * S_TW, y0 = (ty_first + tyi) * S_TH - voff;
#pragma unroll
            for (int k = 0; k < kPre; k++) { const int i = tt + 256 * k; if (i < S_IH * S_IW) in[team][i] = pre[k]; }
            __syncthreads();   // the teams' trip counts differ by at most one tile; every wave of the workgroup reaches the same number of barriers (see below)
            if (t + 4 < n_tiles) load_tile(t + 4);
            wiener_hpass<BD>(in[team], tmp[team], fx, tt, 256);
            __syncthreads();
            const int j = tt & 63, i0 = (tt >> 6) * 8;
            uint32_t e = 0;
            int sv[8];   // the source samples first: their loads are in flight while the column is filtered
#pragma unroll
            for (int r = 0; r < 8; r++) {
                // (branch-free: a load under a condition becomes a branch with its own s_waitcnt vmcnt(0) — eight serialised L2 round trips per tile)
                const int x = x0 + j, y = y0 + i0 + r;
                const int s = (int)src[(size_t)min(max(y, ry0), ry1 - 1) * src_stride + min(x, rx1 - 1)];
                sv[r] = (x >= rx1 || y >= ry1 || y < ry0) ? -1 : s;
            }
            int v8[8];
            wiener_vcol8<BD>(tmp[team], i0, j, fy, v8);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (sv[r] < 0) continue;
                const int d = v8[r] - sv[r];
                e += (uint32_t)(d * d);
            }
            sse += e;
            __syncthreads();   // tmp / in are rewritten by the team's next tile
        }
        // teams that ran one tile fewer make up their three barriers, so that the workgroup's barrier count is uniform
        if (n_tiles % 4 && team >= n_tiles % 4) { __syncthreads(); __syncthreads(); __syncthreads(); }
        // 64-lane sums, then the sixteen waves' partials
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) sse += ((unsigned long long)(uint32_t)__shfl_xor((int)(sse >> 32), m, 64) << 32) | (uint32_t)__shfl_xor((int)sse, m, 64);
        if ((tid & 63) == 0) part[tid >> 6] = sse;
        __syncthreads();
        if (tid == 0) {
            unsigned long long tot = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) tot += part[k];
            n_probes++;
            go = wn_result(w, (long long)tot, off) ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 8; k++) { taps[k] = w.v[k]; taps[8 + k] = w.h[k]; }
        }
        __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { unit_wiener[16 * unit + k] = w.v[k]; unit_wiener[16 * unit + 8 + k] = w.h[k]; }
        err_out[unit] = w.err;
        if (probes_out) probes_out[unit] = n_probes;
    }
}

// A workgroup barrier that orders LDS only.  __syncthreads() also waits for every outstanding GLOBAL load (s_waitcnt vmcnt(0) before s_barrier), which puts the full
// L2 latency of loads that were issued early on purpose — the source samples of a tile's error — in front of every barrier of the round.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ---- the same walk with the unit's input RESIDENT in LDS (8-bit planes; round 5).  A probe only changes the taps: the unit's samples — the CDEF output with the stripes'
// context rows from the deblocked picture — are the same for all ~30 probes of a walk, yet the form above stages every 64 x 32 tile from memory for every probe (eleven
// 1-byte loads per thread and tile, each behind ~25 instructions of index / clamp / stripe arithmetic: about half of a probe's instructions and all of its memory
// latency).  Here the workgroup stages the whole unit ONCE, as bytes, one 38-row band per tile row (a band = the tile row's 32 rows + 3 context rows above and below, so the
// stripe rules are applied once, at staging time), and a probe reads LDS only (plus the source samples of the error, which stay in the L2).  The horizontal pass takes its
// seven taps as two v_dot4_i32_i8: the taps fit int8 (|f| <= 128, WIENER_FILT_TAP*_{MINV,MAXV}, centre = -2 x the others) and sum to zero, so the samples can be biased by
// -128 into int8 without changing the sum (the launcher's kernels add 128 x sum(f) anyway: exact for any taps).  Units whose bands do not fit (above ~256 x 400 samples)
// and 16-bit planes keep the form above; the launcher decides per launch.
constexpr int kWnBandBytes = 124 * 1024;
__global__ void __launch_bounds__(1024)
wiener_walk8r_kernel(const WnPic a) {
    const int z = blockIdx.y;
    const uint8_t* __restrict__ dgd = (const uint8_t*)a.p[z].dgd; const uint8_t* __restrict__ dbl = (const uint8_t*)a.p[z].dbl; const uint8_t* __restrict__ src = (const uint8_t*)a.p[z].src;
    int16_t* __restrict__ unit_wiener = a.p[z].unit_wiener; const uint8_t* __restrict__ active = a.p[z].active;
    long long* __restrict__ err_out = a.p[z].err; uint32_t* __restrict__ probes_out = a.p[z].probes;
    const int stride = a.p[z].stride, pw = a.p[z].pw, ph = a.p[z].ph, unit_size = a.p[z].unit_size, units_x = a.p[z].units_x, units_y = a.p[z].units_y, voff = a.p[z].voff,
              stripe_h = a.p[z].stripe_h, dbl_stride = a.p[z].dbl_stride, src_stride = a.p[z].src_stride, win = a.p[z].win;
    if ((int)blockIdx.x >= units_x * units_y) return;
    __shared__ __attribute__((aligned(16))) uint8_t bands[kWnBandBytes];          // [tile row][S_IH][pitch]; byte column cb of a band row = sample x = rx0 - 4 + cb
    __shared__ __attribute__((aligned(16))) uint16_t tmp[4][S_IH * S_TW];
    __shared__ int taps[16];
    __shared__ unsigned long long part[16];
    __shared__ int go;
    const int unit = blockIdx.x, tid = threadIdx.x, team = tid >> 8, tt = tid & 255;
    if (!active[unit]) return;
    const int ux = unit % units_x, uy = unit / units_x, off = (7 - win) >> 1;
    const int rx0 = ux * unit_size, rx1 = ux == units_x - 1 ? pw : (ux + 1) * unit_size;
    const int ry0 = max(uy * unit_size - voff, 0), ry1 = uy == units_y - 1 ? ph : (uy + 1) * unit_size - voff;
    const int tiles_x = (rx1 - rx0 + S_TW - 1) / S_TW, ty_first = (ry0 + voff) / S_TH, tiles_y = (ry1 + voff + S_TH - 1) / S_TH - ty_first, n_tiles = tiles_x * tiles_y;
    const int pitch = tiles_x * S_TW + 8, pitch_dw = pitch >> 2;
    // ---- stage the unit once: a dword (four samples of one band row) per step
    for (int i = tid; i < tiles_y * S_IH * pitch_dw; i += 1024) {
        const int b = i / (S_IH * pitch_dw), rem = i - b * (S_IH * pitch_dw), r = rem / pitch_dw, cd = rem - r * pitch_dw;
        const int y0 = (ty_first + b) * S_TH - voff, yy = y0 - 3 + r, xx0 = rx0 - 4 + 4 * cd;
        const StripeCtx<uint8_t> sc = lr_stripe_of<uint8_t>(dbl, dbl_stride, y0, voff, stripe_h, ph);
        const uint8_t* row; int lo, hi;
        if (sc.above && yy < sc.sy0) { row = dbl + (ptrdiff_t)(yy == sc.sy0 - 1 ? sc.sy0 - 1 : sc.sy0 - 2) * dbl_stride; lo = 0; hi = pw - 1; }
        else if (sc.below && yy >= sc.sy1) { row = dbl + (ptrdiff_t)min(yy == sc.sy1 ? sc.sy1 : sc.sy1 + 1, ph - 1) * dbl_stride; lo = 0; hi = pw - 1; }
        else { row = dgd + (ptrdiff_t)min(max(yy, -3), ph + 2) * stride; lo = -3; hi = pw + 2; }
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)row[min(max(xx0 + k, lo), hi)] << (8 * k);
        ((uint32_t*)bands)[i] = v;
    }
    __shared__ WnWalkState w_lds;   // thread 0's walk state: its tap arrays are indexed at run time, which as a private object means scratch memory (a memory round trip per access)
    WnWalkState& w = w_lds;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { w.v[k] = unit_wiener[16 * unit + k]; w.h[k] = unit_wiener[16 * unit + 8 + k]; }
        w.state = 1; w.err = 0; w.s = 0; w.ph = 0; w.p = 0; w.up = 0; w.skip = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { taps[k] = w.v[k]; taps[8 + k] = w.h[k]; }
        go = 1;
    }
    uint32_t n_probes = 0;
    __syncthreads();
    while (go) {
        int fy[8];
#pragma unroll
        for (int k = 0; k < 8; k++) fy[k] = taps[k];
        // horizontal taps as two int8 quadruples; hbase = the rounding offset + 128 x their sum (zero for the symmetric filters the walk produces)
        const int f0 = taps[8], f1 = taps[9], f2 = taps[10], f3 = taps[11], f4 = taps[12], f5 = taps[13], f6 = taps[14];
        const int TA = (f0 & 0xff) | (f1 & 0xff) << 8 | (f2 & 0xff) << 16 | (f3 & 0xff) << 24, TB = (f4 & 0xff) | (f5 & 0xff) << 8 | (f6 & 0xff) << 16;
        const int hbase = (1 << (8 + 6)) + 4 + 128 * (f0 + f1 + f2 + f3 + f4 + f5 + f6);
        unsigned long long sse = 0;
        for (int t = team; t < n_tiles; t += 4) {   // uniform per team: its four waves pass the same barriers
            const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
            const int x0 = rx0 + txi * S_TW, y0 = (ty_first + tyi) * S_TH - voff;
            const uint8_t* band = bands + (size_t)tyi * S_IH * pitch + txi * S_TW;
            const int j = tt & 63, i0 = (tt >> 6) * 8;
            int sv[8];   // the source samples of the error first: their loads are in flight during the horizontal pass
#pragma unroll
            for (int r = 0; r < 8; r++) {
                // (branch-free: a load under a condition becomes a branch with its own s_waitcnt vmcnt(0) — eight serialised L2 round trips per tile)
                const int x = x0 + j, y = y0 + i0 + r;
                const int s = (int)src[(size_t)min(max(y, ry0), ry1 - 1) * src_stride + min(x, rx1 - 1)];
                sv[r] = (x >= rx1 || y >= ry1 || y < ry0) ? -1 : s;
            }
            // horizontal pass (round 3): four outputs from twelve bytes (three aligned dwords; the ten inputs are bytes 1..10)
            for (int g = tt; g < S_IH * (S_TW / 4); g += 256) {
                const int r = g >> 4, c = (g & 15) << 2;
                const uint32_t* wd = (const uint32_t*)(band + r * pitch + c);
                const uint32_t d0 = wd[0], d1 = wd[1], d2 = wd[2];
                const uint32_t e0 = d0 ^ 0x80808080u, e1 = d1 ^ 0x80808080u, e2 = d2 ^ 0x80808080u;
                uint32_t o[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t A = q == 3 ? e1 : __builtin_amdgcn_alignbyte(e1, e0, (uint32_t)(q + 1)), B = q == 3 ? e2 : __builtin_amdgcn_alignbyte(e2, e1, (uint32_t)(q + 1));
                    int sum = (int)(((d1 >> (8 * q)) & 0xffu) << 7) + hbase;
                    sum = __builtin_amdgcn_sdot4((int)A, TA, sum, false);
                    sum = __builtin_amdgcn_sdot4((int)B, TB, sum, false);
                    o[q] = (uint32_t)min(max(sum >> 3, 0), (1 << (8 + 5)) - 1);   // WIENER_CLAMP_LIMIT(3, 8)
                }
                uint32_t* out = (uint32_t*)(tmp[team] + r * S_TW + c);
                out[0] = o[0] | (o[1] << 16); out[1] = o[2] | (o[3] << 16);
            }
            lds_barrier();
            uint32_t e = 0;
            int v8[8];
            wiener_vcol8<8>(tmp[team], i0, j, fy, v8);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (sv[r] < 0) continue;
                const int d = v8[r] - sv[r];
                e += (uint32_t)(d * d);
            }
            sse += e;
            lds_barrier();   // tmp is rewritten by the team's next tile
        }
        if (n_tiles % 4 && team >= n_tiles % 4) { lds_barrier(); lds_barrier(); }   // teams that ran one tile fewer make up their two barriers
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) sse += ((unsigned long long)(uint32_t)__shfl_xor((int)(sse >> 32), m, 64) << 32) | (uint32_t)__shfl_xor((int)sse, m, 64);
        if ((tid & 63) == 0) part[tid >> 6] = sse;
        lds_barrier();
        if (tid == 0) {
            unsigned long long tot = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) tot += part[k];
            n_probes++;
            go = wn_result(w, (long long)tot, off) ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 8; k++) { taps[k] = w.v[k]; taps[8 + k] = w.h[k]; }
        }
        lds_barrier();
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { unit_wiener[16 * unit + k] = w.v[k]; unit_wiener[16 * unit + 8 + k] = w.h[k]; }
        err_out[unit] = w.err;
        if (probes_out) probes_out[unit] = n_probes;
    }
}

// ---- round 5, second form: a WAVE per tile, no intermediate in LDS, no barrier inside a probe.  Measured on the form above (profiles/r05/wiener_walk_phases.txt): of the
// 34.5 us of a probe of a 48-tile unit the horizontal pass is 11, the vertical pass 11.6, the source loads 4 and the rest (barriers, the reduction, the walk's state
// machine) 8 — each pass ~4-5x its issue time, because a team's three phases per tile are separated by workgroup barriers and every phase starts with an LDS round trip.
// Here a lane owns four adjacent columns x eight rows of a 64 x 32 tile: it filters the fourteen rows it needs horizontally straight from the resident bytes (1.75x
// the horizontal work: the six context rows are shared with the lane above / below), keeps the results as vertically packed 16-bit pairs in registers, and the
// vertical pass is four v_dot2_i32_i16 per output on those pairs (even rows: taps (f0 f1)(f2 f3)(f4 f5)(f6 0); odd rows: (0 f0)(f1 f2)(f3 f4)(f5 f6) on the same aligned
// pairs; the centre tap carries the "+ 128 x sample" of the reference's rounding form).  The source is read as one dword per row.  Sixteen waves walk the unit's tiles
// independently; the only barriers left are the two around the walk's decision.  LDS reads: lanes of a row group read consecutive dwords, the four row groups are
// 8 rows = 8 x (64 k + 8) bytes = 16 banks apart: conflict-free.
__global__ void __launch_bounds__(1024)
wiener_walk8w_kernel(const WnPic a) {
    const int z = blockIdx.y;
    const uint8_t* __restrict__ dgd = (const uint8_t*)a.p[z].dgd; const uint8_t* __restrict__ dbl = (const uint8_t*)a.p[z].dbl; const uint8_t* __restrict__ src = (const uint8_t*)a.p[z].src;
    int16_t* __restrict__ unit_wiener = a.p[z].unit_wiener; const uint8_t* __restrict__ active = a.p[z].active;
    long long* __restrict__ err_out = a.p[z].err; uint32_t* __restrict__ probes_out = a.p[z].probes;
    const int stride = a.p[z].stride, pw = a.p[z].pw, ph = a.p[z].ph, unit_size = a.p[z].unit_size, units_x = a.p[z].units_x, units_y = a.p[z].units_y, voff = a.p[z].voff,
              stripe_h = a.p[z].stripe_h, dbl_stride = a.p[z].dbl_stride, src_stride = a.p[z].src_stride, win = a.p[z].win;
    if ((int)blockIdx.x >= units_x * units_y) return;
    __shared__ __attribute__((aligned(16))) uint8_t bands[kWnBandBytes + 16];     // [tile row][S_IH][pitch]; byte column cb of a band row = sample x = rx0 - 4 + cb
    __shared__ int taps[16];
    __shared__ unsigned long long part[16];
    __shared__ int go;
    const int unit = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (!active[unit]) return;
    const int ux = unit % units_x, uy = unit / units_x, off = (7 - win) >> 1;
    const int rx0 = ux * unit_size, rx1 = ux == units_x - 1 ? pw : (ux + 1) * unit_size;
    const int ry0 = max(uy * unit_size - voff, 0), ry1 = uy == units_y - 1 ? ph : (uy + 1) * unit_size - voff;
    const int tiles_x = (rx1 - rx0 + S_TW - 1) / S_TW, ty_first = (ry0 + voff) / S_TH, tiles_y = (ry1 + voff + S_TH - 1) / S_TH - ty_first, n_tiles = tiles_x * tiles_y;
    const int pitch = tiles_x * S_TW + 8, pitch_dw = pitch >> 2;
    // stage the unit once (as in the form above), four dwords = sixteen byte loads in flight per thread (one dword per iteration is ~30 dependent memory round trips)
    batched_stage<4, uint32_t>(tiles_y * S_IH * pitch_dw, tid, 1024,
        [&](int i) {
            const int b = i / (S_IH * pitch_dw), rem = i - b * (S_IH * pitch_dw), r = rem / pitch_dw, cd = rem - r * pitch_dw;
            const int y0 = (ty_first + b) * S_TH - voff, yy = y0 - 3 + r, xx0 = rx0 - 4 + 4 * cd;
            const StripeCtx<uint8_t> sc = lr_stripe_of<uint8_t>(dbl, dbl_stride, y0, voff, stripe_h, ph);
            const bool up = sc.above && yy < sc.sy0, dn = sc.below && yy >= sc.sy1, ctx = up || dn;
            const int  yd = up ? (yy == sc.sy0 - 1 ? sc.sy0 - 1 : sc.sy0 - 2) : min(yy == sc.sy1 ? sc.sy1 : sc.sy1 + 1, ph - 1);
            const uint8_t* row = ctx ? dbl + (ptrdiff_t)yd * dbl_stride : dgd + (ptrdiff_t)min(max(yy, -3), ph + 2) * stride;
            const int lo = ctx ? 0 : -3, hi = ctx ? pw - 1 : pw + 2;
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= (uint32_t)row[min(max(xx0 + k, lo), hi)] << (8 * k);
            return v;
        },
        [&](int i, uint32_t v) { ((uint32_t*)bands)[i] = v; });
    __shared__ WnWalkState w_lds;   // thread 0's walk state: its tap arrays are indexed at run time, which as a private object means scratch memory (a memory round trip per access)
    WnWalkState& w = w_lds;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { w.v[k] = unit_wiener[16 * unit + k]; w.h[k] = unit_wiener[16 * unit + 8 + k]; }
        w.state = 1; w.err = 0; w.s = 0; w.ph = 0; w.p = 0; w.up = 0; w.skip = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { taps[k] = w.v[k]; taps[8 + k] = w.h[k]; }
        go = 1;
    }
    uint32_t n_probes = 0;
    const int c4 = (lane & 15) << 2, rg = lane >> 4;   // the lane's four columns and its eight rows 8 rg .. 8 rg + 7 of a tile
    const bool src_dwords = ((uintptr_t)src & 3) == 0 && (src_stride & 3) == 0;
    __syncthreads();
    while (go) {
        // (the taps are workgroup-uniform: v_readfirstlane puts everything derived from them into scalar registers)
        auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        const int f0 = uni(taps[8]), f1 = uni(taps[9]), f2 = uni(taps[10]), f3 = uni(taps[11]), f4 = uni(taps[12]), f5 = uni(taps[13]), f6 = uni(taps[14]);
        // The reference's "+ (centre sample << 7)" rides in the dot products: 128 = 1 + 127, the 1 on the centre tap (f3 + 1 is in [-127, 91]) and the 127 in the second
        // quadruple's free slot, whose fourth byte is made the centre sample by the byte permute that builds it.  Samples are biased by -128: + 128 x (sum of all eight taps).
        const int TA = (f0 & 0xff) | (f1 & 0xff) << 8 | (f2 & 0xff) << 16 | ((f3 + 1) & 0xff) << 24, TB = (f4 & 0xff) | (f5 & 0xff) << 8 | (f6 & 0xff) << 16 | 127 << 24;
        const int hbase = (1 << (8 + 6)) + 4 + 128 * (f0 + f1 + f2 + f3 + f4 + f5 + f6 + 128);
        // vertical taps as 16-bit pairs (low half = the upper row of a pair); g3 = the centre tap + 128 (the reference adds sample << 7)
        const int g0 = uni(taps[0]), g1 = uni(taps[1]), g2 = uni(taps[2]), g3 = uni(taps[3]) + 128, g4 = uni(taps[4]), g5 = uni(taps[5]), g6 = uni(taps[6]);
        auto pk = [](int lo, int hi) { return (uint32_t)(lo & 0xffff) | (uint32_t)hi << 16; };
        const uint32_t E0 = pk(g0, g1), E1 = pk(g2, g3), E2 = pk(g4, g5), E3 = pk(g6, 0), O0 = pk(0, g0), O1 = pk(g1, g2), O2 = pk(g3, g4), O3 = pk(g5, g6);
        uint32_t sse = 0;   // a lane's share of a probe: at most a few tiles x 32 samples x 255^2; a wave's total stays below 2^32 as well (64 lanes x 6 tiles x 32 x 65 025)
        for (int t = wave; t < n_tiles; t += 16) {
            const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
            const int x0 = rx0 + txi * S_TW + c4, y0 = (ty_first + tyi) * S_TH - voff + 8 * rg;   // the lane's first sample
            // the source first: its loads are in flight during both passes (branch-free, clamped addresses)
            uint32_t sv[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint8_t* sp = src + (size_t)min(max(y0 + r, ry0), ry1 - 1) * src_stride;
                if (src_dwords && x0 + 3 < pw) sv[r] = *(const uint32_t*)(sp + x0);
                else {
                    sv[r] = 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) sv[r] |= (uint32_t)sp[min(x0 + c, pw - 1)] << (8 * c);
                }
            }
            const uint8_t* band = bands + (size_t)tyi * S_IH * pitch + txi * S_TW + (8 * rg) * pitch + c4;
            uint32_t P[4][7];   // [column][pair of rows]: the horizontally filtered rows 2 k (low half) and 2 k + 1 of the lane's fourteen
#pragma unroll
            for (int rr = 0; rr < 14; rr++) {
                const uint32_t* wd = (const uint32_t*)(band + rr * pitch);
                const uint32_t e0 = wd[0] ^ 0x80808080u, e1 = wd[1] ^ 0x80808080u, e2 = wd[2] ^ 0x80808080u;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // window bytes 0 .. 11 = e0 e1 e2; output q: taps on bytes q + 1 .. q + 7, centre = byte q + 4.  A = bytes q + 1 .. q + 4; B = bytes q + 5, q + 6, q + 7, q + 4
                    const uint32_t A = q == 3 ? e1 : __builtin_amdgcn_alignbyte(e1, e0, (uint32_t)(q + 1));
                    const uint32_t B = __builtin_amdgcn_perm(e2, e1, (uint32_t)((q + 1) | (q + 2) << 8 | (q + 3) << 16 | q << 24));
                    int sum = __builtin_amdgcn_sdot4((int)A, TA, hbase, false);
                    sum = __builtin_amdgcn_sdot4((int)B, TB, sum, false);
                    const uint32_t o = (uint32_t)min(max(sum >> 3, 0), (1 << (8 + 5)) - 1);   // WIENER_CLAMP_LIMIT(3, 8)
                    if (rr & 1) P[q][rr >> 1] |= o << 16; else P[q][rr >> 1] = o;
                }
                if (rr & 1) __builtin_amdgcn_sched_barrier(0);   // two rows' reads in flight at a time: hoisting all 42 dwords to the top costs the registers of the pairs
            }
            uint32_t e = 0;
            const bool inside = y0 >= ry0 && y0 + 8 <= ry1 && x0 + 4 <= rx1;   // all 32 samples of the lane count (every lane of an interior tile)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int  y = y0 + r;
                const bool row_ok = inside || (y >= ry0 && y < ry1);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    typedef short v2s __attribute__((ext_vector_type(2)));
                    auto dot = [](uint32_t x, uint32_t y2, int acc) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, x), __builtin_bit_cast(v2s, y2), acc, false); };
                    const int k = r >> 1;
                    int sum = (1 << 10) - (1 << (8 + 10));
                    if (r & 1) { sum = dot(P[c][k], O0, sum); sum = dot(P[c][k + 1], O1, sum); sum = dot(P[c][k + 2], O2, sum); sum = dot(P[c][k + 3], O3, sum); }
                    else       { sum = dot(P[c][k], E0, sum); sum = dot(P[c][k + 1], E1, sum); sum = dot(P[c][k + 2], E2, sum); sum = dot(P[c][k + 3], E3, sum); }
                    const int v = min(max(sum >> 11, 0), 255);
                    const int d = v - (int)((sv[r] >> (8 * c)) & 0xffu);
                    if (inside || (row_ok && x0 + c < rx1)) e += (uint32_t)(d * d);
                }
            }
            sse += e;
        }
        // 64-lane sums, then the sixteen waves' partials
        {   // DPP inside the 16-lane rows, the four row totals through scalar registers
            int v = (int)sse;
            v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
            v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(v, 0) + (uint32_t)__builtin_amdgcn_readlane(v, 16) + (uint32_t)__builtin_amdgcn_readlane(v, 32) + (uint32_t)__builtin_amdgcn_readlane(v, 48);
            if (lane == 0) part[wave] = tot;
        }
        lds_barrier();
        if (tid == 0) {
            unsigned long long tot = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) tot += part[k];
            n_probes++;
            go = wn_result(w, (long long)tot, off) ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 8; k++) { taps[k] = w.v[k]; taps[8 + k] = w.h[k]; }
        }
        lds_barrier();
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { unit_wiener[16 * unit + k] = w.v[k]; unit_wiener[16 * unit + 8 + k] = w.h[k]; }
        err_out[unit] = w.err;
        if (probes_out) probes_out[unit] = n_probes;
    }
}

}  // namespace

extern "C" int svt_hip_launch_wiener_walk_multi(hipStream_t st, int pix_bytes, int bd, int n_planes, const SvtHipWienerWalkPlane* planes) {
    if (n_planes < 1 || n_planes > 3) return (int)hipErrorInvalidValue;
    WnPic a = {};
    int n = 0;
    for (int i = 0; i < n_planes; i++) {
        const SvtHipWienerWalkPlane& P = planes[i];
        const int ux = max((P.pw + P.unit_size / 2) / P.unit_size, 1), uy = max((P.ph + P.unit_size / 2) / P.unit_size, 1);   // av1_lr_count_units_in_tile (EbRestoration.c:1445)
        a.p[i] = WnPlane{P.d_dgd, P.d_dbl, P.d_src, P.d_unit_wiener, P.d_active, (long long*)P.d_err, P.d_probes, P.stride, P.pw, P.ph, P.unit_size, ux, uy, 8 >> P.ss_y, 64 >> P.ss_y,
                         P.dbl_stride, P.src_stride, P.wiener_win};
        n = max(n, ux * uy);
    }
    if (n <= 0) return 0;
    const dim3 grid(n, n_planes);
    // the resident-input form (8-bit): every unit's bands must fit its LDS window; SVT_HIP_WIENER_WALK=tiles keeps the form that stages tile by tile (A/B runs)
    static int form = -1;   // 0 = "tiles" (round 4), 1 = "teams" (resident unit, a 256-thread team per tile), 2 = a wave per tile (default)
    if (form < 0) { const char* e = getenv("SVT_HIP_WIENER_WALK"); form = e && !strcmp(e, "tiles") ? 0 : (e && !strcmp(e, "teams") ? 1 : 2); }
    bool fits = pix_bytes == 1 && form >= 1;
    for (int i = 0; i < n_planes && fits; i++) {
        const WnPlane& q = a.p[i];
        for (int uy = 0; uy < q.units_y && fits; uy++)
            for (int ux = 0; ux < q.units_x && fits; ux += (q.units_x > 2 && ux == 0) ? q.units_x - 2 : 1) {   // the first and the last two columns cover every width
                const int rx0 = ux * q.unit_size, rx1 = ux == q.units_x - 1 ? q.pw : (ux + 1) * q.unit_size;
                const int ry0 = max(uy * q.unit_size - q.voff, 0), ry1 = uy == q.units_y - 1 ? q.ph : (uy + 1) * q.unit_size - q.voff;
                const int tiles_x = (rx1 - rx0 + S_TW - 1) / S_TW, tiles_y = (ry1 + q.voff + S_TH - 1) / S_TH - (ry0 + q.voff) / S_TH;
                fits = (size_t)tiles_y * S_IH * (tiles_x * S_TW + 8) <= (size_t)kWnBandBytes;
            }
    }
    if (fits && form == 2) hipLaunchKernelGGL(wiener_walk8w_kernel, grid, dim3(1024), 0, st, a);
    else if (fits) hipLaunchKernelGGL(wiener_walk8r_kernel, grid, dim3(1024), 0, st, a);
    else if (pix_bytes == 1) hipLaunchKernelGGL((wiener_walk_kernel<uint8_t, 8>), grid, dim3(1024), 0, st, a);
    else if (bd == 8) hipLaunchKernelGGL((wiener_walk_kernel<uint16_t, 8>), grid, dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((wiener_walk_kernel<uint16_t, 10>), grid, dim3(1024), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_sgr_filter(hipStream_t st, int pix_bytes, int bd, const void* plane, int stride, int pw, int ph, int ep,
                                         int32_t* flt0, int32_t* flt1, int flt_stride) {
    dim3 grid((pw + 63) / 64, (ph + 15) / 16);
    if (pix_bytes == 1) hipLaunchKernelGGL((sgr_filter_kernel<uint8_t, 8>), grid, dim3(256), 0, st, (const uint8_t*)plane, stride, pw, ph, ep, flt0, flt1, flt_stride);
    else if (bd == 8) hipLaunchKernelGGL((sgr_filter_kernel<uint16_t, 8>), grid, dim3(256), 0, st, (const uint16_t*)plane, stride, pw, ph, ep, flt0, flt1, flt_stride);
    else hipLaunchKernelGGL((sgr_filter_kernel<uint16_t, 10>), grid, dim3(256), 0, st, (const uint16_t*)plane, stride, pw, ph, ep, flt0, flt1, flt_stride);
    return (int)hipGetLastError();
}
namespace {
SgrSearchPlaneArgs sgr_search_plane_args(const void* dgd, int stride, const void* src, int src_stride, int pw, int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask,
                                         int64_t* sums, uint32_t* pairs, int16_t* sd, int dstride, size_t dplane, int64_t* d2, void* esc, uint32_t* esc_cnt, int esc_lim) {
    SgrSearchPlaneArgs A = {};
    A.dgd = dgd; A.src = src; A.sums = (unsigned long long*)sums; A.pairs = pairs; A.sd = sd; A.d2 = (unsigned long long*)d2; A.esc = (uint2*)esc; A.esc_cnt = esc_cnt;
    A.dplane = dplane; A.stride = stride; A.src_stride = src_stride; A.pw = pw; A.ph = ph; A.unit_size = unit_size; A.units_x = units_x; A.units_y = units_y;
    A.voff = 8 >> ss_y; A.dstride = dstride; A.tiles_x = (pw + S_TW - 1) / S_TW; A.n_tiles = A.tiles_x * ((ph + A.voff + S_TH - 1) / S_TH); A.esc_lim = esc_lim; A.ep_mask = ep_mask;
    return A;
}
template <int STORE>
int sgr_search_launch(hipStream_t st, int pix_bytes, int bd, const SgrSearchPic& a) {
    const dim3 grid((unsigned)a.first_tile[kSgrMaxPlanes]);
    if (STORE == 2) {
        if (pix_bytes == 1) hipLaunchKernelGGL((sgr_search8_kernel<uint8_t, 8, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((sgr_search8_kernel<uint16_t, 8, 2>), grid, dim3(256), 0, st, a);
    } else if (pix_bytes == 1) hipLaunchKernelGGL((sgr_search8_kernel<uint8_t, 8, STORE == 2 ? 1 : STORE>), grid, dim3(256), 0, st, a);
    else if (bd == 8) hipLaunchKernelGGL((sgr_search8_kernel<uint16_t, 8, STORE == 2 ? 1 : STORE>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((sgr_search8_kernel<uint16_t, 10, STORE == 2 ? 1 : STORE>), grid, dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
int sgr_esc_lim() {
    const char* lim_env = getenv("SVT_HIP_SGR_ESC_LIM");   // read per launch: tests narrow the range so that ordinary content reaches the escape lists
    const int lim = lim_env ? atoi(lim_env) : 1024;
    return lim < 1 ? 1 : (lim > 1024 ? 1024 : lim);
}
}  // namespace
extern "C" int svt_hip_launch_sgr_search(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, const void* src, int src_stride,
                                         int pw, int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask, int64_t* sums) {
    SgrSearchPic a = {};
    a.p[0] = sgr_search_plane_args(dgd, stride, src, src_stride, pw, ph, unit_size, units_x, units_y, ss_y, ep_mask, sums, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 1024);
    for (int i = 1; i <= kSgrMaxPlanes; i++) a.first_tile[i] = a.p[0].n_tiles;
    return sgr_search_launch<0>(st, pix_bytes, bd, a);
}
// the search kernel with the int16 difference planes of the on-device unit search (sgr_walk.hip): every plane of the picture in one launch
extern "C" int svt_hip_launch_sgr_search_store_multi(hipStream_t st, int pix_bytes, int bd, int n_planes, const SvtHipSgrSearchStorePlane* pl) {
    if (n_planes < 1 || n_planes > kSgrMaxPlanes) return (int)hipErrorInvalidValue;
    SgrSearchPic a = {};
    bool packed = false;
    int at = 0;
    for (int i = 0; i < n_planes; i++) {
        a.first_tile[i] = at;
        const SvtHipSgrSearchStorePlane& P = pl[i];
        if (i == 0) packed = P.esc != nullptr;
        if ((P.esc != nullptr) != packed || (packed && (bd != 8 || !P.esc_cnt))) return (int)hipErrorInvalidValue;
        a.p[i] = sgr_search_plane_args(P.dgd, P.stride, P.src, P.src_stride, P.pw, P.ph, P.unit_size, P.units_x, P.units_y, P.ss_y, P.ep_mask, P.sums, P.pairs, P.sd, P.dstride,
                                       P.dplane, P.d2, P.esc, P.esc_cnt, packed ? sgr_esc_lim() : 1024);
        at += a.p[i].n_tiles;
    }
    for (int i = n_planes; i <= kSgrMaxPlanes; i++) a.first_tile[i] = at;   // the grid size; no tile belongs to a plane that is not there
    return packed ? sgr_search_launch<2>(st, pix_bytes, bd, a) : sgr_search_launch<1>(st, pix_bytes, bd, a);
}
extern "C" int svt_hip_launch_sgr_search_store(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, const void* src, int src_stride,
                                               int pw, int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask, int64_t* sums,
                                               uint32_t* pairs, int16_t* sd, int dstride, size_t dplane, int64_t* d2, void* esc, uint32_t* esc_cnt) {
    const SvtHipSgrSearchStorePlane P = {dgd, src, sums, pairs, sd, d2, esc, esc_cnt, dplane, stride, src_stride, pw, ph, unit_size, units_x, units_y, ss_y, dstride, ep_mask};
    return svt_hip_launch_sgr_search_store_multi(st, pix_bytes, bd, 1, &P);
}
extern "C" int svt_hip_launch_sgr_proj_error(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, const void* src, int src_stride, int pw, int ph,
                                             int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask, int ncand, const int32_t* xqd, int64_t* err) {
    const int voff = 8 >> ss_y;
    dim3 grid8((pw + S_TW - 1) / S_TW, (ph + voff + S_TH - 1) / S_TH);
    unsigned long long* e = (unsigned long long*)err;
    if (ncand < 1 || ncand > kSgrMaxCand) return (int)hipErrorInvalidValue;
    if (pix_bytes == 1) hipLaunchKernelGGL((sgr_proj_error_kernel<uint8_t, 8>), grid8, dim3(256), 0, st, (const uint8_t*)dgd, stride, (const uint8_t*)src, src_stride, pw, ph, unit_size, units_x, units_y, voff, ep_mask, ncand, xqd, e);
    else if (bd == 8) hipLaunchKernelGGL((sgr_proj_error_kernel<uint16_t, 8>), grid8, dim3(256), 0, st, (const uint16_t*)dgd, stride, (const uint16_t*)src, src_stride, pw, ph, unit_size, units_x, units_y, voff, ep_mask, ncand, xqd, e);
    else hipLaunchKernelGGL((sgr_proj_error_kernel<uint16_t, 10>), grid8, dim3(256), 0, st, (const uint16_t*)dgd, stride, (const uint16_t*)src, src_stride, pw, ph, unit_size, units_x, units_y, voff, ep_mask, ncand, xqd, e);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_sgr_apply_tiles(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, void* dst, int dst_stride, int pw,
                                              int ph, int unit_size, int units_x, int units_y, int ss_y, const void* dbl, int dbl_stride,
                                              const uint8_t* unit_ep, const int32_t* unit_xqd, const int16_t* unit_wiener, int tx0, int ty0, int ntx, int nty) {
    const int voff = 8 >> ss_y, sh = 64 >> ss_y;
    if (ntx <= 0 || nty <= 0) return 0;
    dim3 grid8(ntx, nty);
    if (pix_bytes == 1) hipLaunchKernelGGL((lr_apply8_kernel<uint8_t>), grid8, dim3(256), 0, st, (const uint8_t*)dgd, stride, (uint8_t*)dst, dst_stride, pw, ph, unit_size, units_x, units_y, voff, sh, (const uint8_t*)dbl, dbl_stride, unit_ep, unit_xqd, unit_wiener, tx0, ty0);
    else if (bd == 8) hipLaunchKernelGGL((lr_apply8_kernel<uint16_t>), grid8, dim3(256), 0, st, (const uint16_t*)dgd, stride, (uint16_t*)dst, dst_stride, pw, ph, unit_size, units_x, units_y, voff, sh, (const uint16_t*)dbl, dbl_stride, unit_ep, unit_xqd, unit_wiener, tx0, ty0);
    else hipLaunchKernelGGL((lr_apply8_kernel<uint16_t, 10>), grid8, dim3(256), 0, st, (const uint16_t*)dgd, stride, (uint16_t*)dst, dst_stride, pw, ph, unit_size, units_x, units_y, voff, sh, (const uint16_t*)dbl, dbl_stride, unit_ep, unit_xqd, unit_wiener, tx0, ty0);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_sgr_apply(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, void* dst, int dst_stride, int pw,
                                        int ph, int unit_size, int units_x, int units_y, int ss_y, const void* dbl, int dbl_stride,
                                        const uint8_t* unit_ep, const int32_t* unit_xqd, const int16_t* unit_wiener) {
    const int voff = 8 >> ss_y;
    return svt_hip_launch_sgr_apply_tiles(st, pix_bytes, bd, dgd, stride, dst, dst_stride, pw, ph, unit_size, units_x, units_y, ss_y, dbl, dbl_stride, unit_ep, unit_xqd,
                                          unit_wiener, 0, 0, (pw + S_TW - 1) / S_TW, (ph + voff + S_TH - 1) / S_TH);
}

SVT_HIP_TU_PROBE(sgr)
