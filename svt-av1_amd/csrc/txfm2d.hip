// txfm2d.hip — batched residual -> forward 2-D transform -> quantize, and dequantized coefficients ->
// inverse 2-D transform -> reconstruction, for all 19 AV1 transform sizes; gfx950 (wave64).
//
// Replaces, for a whole list of blocks per launch (file:line under /root/reference/Source/Lib):
//   Common/C_DEFAULT/EbPictureOperators_C.c  svt_residual_kernel8bit/16bit_c          (fused prologue)
//   Encoder/Codec/EbTransforms.c:2301        av1_tranform_two_d_core_c  (+ :2732 av1_transform_config)
//   Encoder/Codec/EbTransforms.c:2763-2931   svt_handle_transform64x64/64x32/32x64/64x16/16x64_c
//   Encoder/Codec/EbFullLoop.c:37,171,314,467 quantize_b / highbd_quantize_b / quantize_fp / highbd_quantize_fp
//   Encoder/Codec/EbFullLoop.c:1595-1608     cul_level + dc sign
//   Common/Codec/EbInvTransforms.c:2455      inv_txfm2d_add_c (+ :2432 cfg, :23 stage ranges, :2648 64-pt remap)
//
// Mapping: one lane = one 1-D transform (a column, then a row) held entirely in VGPRs (txfm_1d.h);
// max(W,H) lanes form a block team, 256/max(W,H) teams per workgroup; the column->row hand-over is a
// padded LDS tile.  Quantisation is per lane on the row it just produced; eob / cul_level / energy
// are team reductions with wave shuffles.  Integer only — no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"
#include "txfm_1d.h"
#include "quant_dev.h"

namespace {

using namespace tx1d;

// ---- per-size configuration (Encoder/Codec/EbTransforms.h:26-57, Common/Codec/EbInvTransforms.h:39-69)
__host__ __device__ constexpr int fwd_shift_of(int W, int H, int i) {
    // {shift0, shift1, shift2}
    if (W == 64 && H == 64) return i == 0 ? 0 : -2;
    if (W == 32 && H == 64) return i == 0 ? 0 : -2;
    if (W == 64 && H == 32) return i == 0 ? 2 : (i == 1 ? -4 : -2);
    if (W == 16 && H == 64) return i == 0 ? 0 : (i == 1 ? -2 : 0);
    if (W == 64 && H == 16) return i == 0 ? 2 : (i == 1 ? -4 : 0);
    if (i == 0) return 2;
    if (i == 2) return 0;
    const int m = W > H ? W : H, n = W < H ? W : H;
    if (m == 4) return 0;
    if (m == 8) return -1;
    if (m == 16) return n == 4 ? -1 : -2;  // 4x16/16x4: -1 ; 8x16/16x8/16x16: -2
    /* m == 32 */ return n == 8 ? -2 : -4; // 8x32/32x8: -2 ; 16x32/32x16/32x32: -4
}
// 64x64 keeps two block teams per workgroup so that the padded LDS tile stays under 64 KB
__host__ __device__ constexpr int threads_of(int W, int H) { return (W == 64 && H == 64) ? 128 : 256; }
__host__ __device__ constexpr int lg(int n) { return n == 4 ? 0 : n == 8 ? 1 : n == 16 ? 2 : n == 32 ? 3 : 4; }
__host__ __device__ constexpr int fwd_cos_col_of(int W, int H) {
    constexpr int t[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
    return t[lg(W)][lg(H)];
}
__host__ __device__ constexpr int fwd_cos_row_of(int W, int H) {
    constexpr int t[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
    return t[lg(W)][lg(H)];
}
__host__ __device__ constexpr int inv_shift0_of(int W, int H) {
    const int m = W > H ? W : H, n = W < H ? W : H;
    if (m == 4) return 0;
    if (m == 8) return n == 4 ? 0 : -1;
    if (m == 16) return n == 16 ? -2 : -1;
    if (m == 32) return (n == 32 || n == 8) ? -2 : -1;
    /* m == 64 */ return n == 32 ? -1 : -2;
}
// 1-D kind per TxType (vtx_tab / htx_tab, EbInvTransforms.h:71-106): 0 DCT 1 ADST 2 FLIPADST 3 IDTX
__device__ __constant__ uint8_t kVtx[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
__device__ __constant__ uint8_t kHtx[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};

template <int N, int BIT> __device__ __forceinline__ void fwd_1d(int kind, const int32_t (&in)[N], int32_t (&out)[N]) {
    if constexpr (N <= 16) {
        if (kind == 0) fwd_dct<N, BIT>(in, out);
        else if (kind == 3) identity<N>(in, out);
        else fwd_adst<N, BIT>(in, out);
    } else {
        if (kind == 3) identity<N>(in, out);
        else fwd_dct<N, BIT>(in, out);
    }
}
template <int N, int BIT, int CB> __device__ __forceinline__ void inv_1d(int kind, const int32_t (&in)[N], int32_t (&out)[N]) {
    if constexpr (N <= 16) {
        if (kind == 0) inv_dct<N, BIT, CB>(in, out);
        else if (kind == 3) identity<N>(in, out);
        else inv_adst<N, BIT, CB>(in, out);
    } else {
        if (kind == 3) identity<N>(in, out);
        else inv_dct<N, BIT, CB>(in, out);
    }
}

template <int L> __device__ __forceinline__ uint32_t team_max(uint32_t v) {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m, 64));
    return v;
}
template <int L> __device__ __forceinline__ uint32_t team_add(uint32_t v) {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}
template <int L> __device__ __forceinline__ uint64_t team_add64(uint64_t v) {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// ================================================================== inverse passes (shared by inv_block and the fused encode block) =====
// Row pass of the inverse for row t of a block: `row` = the KW kept dequantised coefficients of the row (registers), written transformed to `tile_row`.
template <int W, int H, int BD>
__device__ __forceinline__ void inv_row_pass(int kr, int t, const int32_t* row, int32_t* __restrict__ tile_row) {
    constexpr int KW = W > 32 ? 32 : W, KH = H > 32 ? 32 : H;
    constexpr int S0 = -inv_shift0_of(W, H);
    constexpr bool RECT2 = (W == 2 * H) || (H == 2 * W);
    constexpr int RNG_ROW = BD == 8 ? 16 : 18;     // svt_av1_gen_inv_stage_range, EbInvTransforms.c:23-60
    constexpr int IN_CLAMP = BD + 8;
    int32_t out[W];
    if (t < KH) {
        int32_t in[W];
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t v = 0;
            if (c < KW) v = row[c];
            if (RECT2) v = mul_inv_sqrt2(v);
            in[c] = clampv<IN_CLAMP>(v);
        }
        inv_1d<W, 12, RNG_ROW>(kr, in, out);
#pragma unroll
        for (int c = 0; c < W; c++) out[c] = S0 ? rshift_round(out[c], S0 ? S0 : 1) : out[c];
    } else {
#pragma unroll
        for (int c = 0; c < W; c++) out[c] = 0;  // rows beyond the kept 32 are zero in, zero out
    }
#pragma unroll
    for (int c = 0; c < W; c++) tile_row[c] = out[c];
}
// Column pass for column t + add to the prediction + clip: tile_team = the team's H x (W + 1) tile of row-pass outputs.
template <int W, int H, int BD, typename PIX>
__device__ __forceinline__ void inv_col_pass(const int32_t* __restrict__ tile_team, int t, int kc, int kr, const PIX* pred, int pred_stride, PIX* recon,
                                             int recon_stride, int bx, int by) {
    constexpr int RNG_COL = 16, COL_CLAMP = (BD + 6 > 16) ? BD + 6 : 16, S1 = 4, LS = W + 1;
    int32_t in[H], out[H];
    const int cs = (kr == 2) ? W - 1 - t : t;
#pragma unroll
    for (int r = 0; r < H; r++) in[r] = clampv<COL_CLAMP>(tile_team[r * LS + cs]);
    inv_1d<H, 12, RNG_COL>(kc, in, out);
    constexpr int32_t res_max = (1 << (7 + BD)) - 1 + (914 << (BD - 7)), res_min = -res_max - 1;  // check_range, :2398-2411
    constexpr int32_t pix_max = (1 << BD) - 1;
    const PIX* p = pred + (size_t)by * pred_stride + bx + t;
    PIX* w = recon + (size_t)by * recon_stride + bx + t;
#pragma unroll
    for (int r = 0; r < H; r++) {
        const int rr = (kc == 2) ? H - 1 - r : r;
        int32_t v = rshift_round(out[rr], S1);
        v = min(max(v, res_min), res_max);
        const int32_t px = (int32_t)p[(size_t)r * pred_stride] + v;
        w[(size_t)r * recon_stride] = (PIX)min(max(px, 0), pix_max);
    }
}

// ================================================================== forward + quantize ==========
// One workgroup's share of a forward launch: blocks [wg * TEAMS, (wg + 1) * TEAMS) of the list.  `tile` is TEAMS * H * (W + 1) dwords of
// LDS.  Called by the single-size kernel and by the mixed-size kernel (where the workgroup always has 256 threads: threads beyond
// threads_of(W, H) only take part in the barrier and the shuffles).
// FBD = 0: forward + quantize only.  FBD = 8 / 10: the block is also reconstructed (the encode loop's fwd -> quant -> inverse -> recon per block,
// EbCodingLoop.c:379-596): the forward pass ends with thread t holding row t of the dequantised coefficients, which is exactly what the inverse's row
// pass of thread t starts from, so they never leave registers; qcoeff is required, dqcoeff may be NULL.
template <int W, int H, typename PIX, int FBD = 0>
__device__ __forceinline__ void fwd_block(int32_t* __restrict__ tile, int wg, int tid, const PIX* __restrict__ src, int src_stride,
                                          const PIX* __restrict__ pred, int pred_stride, const uint32_t* __restrict__ descs, int nblk,
                                          const SvtHipQuantParams& qp, const SvtHipScanTables& scans, int32_t* __restrict__ coeff_out,
                                          int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob_out,
                                          int32_t* __restrict__ cul_out, uint64_t* __restrict__ energy_out, PIX* recon = nullptr, int recon_stride = 0) {
    constexpr int L = W > H ? W : H, TEAMS = threads_of(W, H) / L;
    constexpr int KW = W > 32 ? 32 : W, KH = H > 32 ? 32 : H, NK = KW * KH;
    constexpr int S0 = fwd_shift_of(W, H, 0), S1 = -fwd_shift_of(W, H, 1), S2 = -fwd_shift_of(W, H, 2);
    constexpr int CBC = fwd_cos_col_of(W, H), CBR = fwd_cos_row_of(W, H);
    constexpr bool RECT2 = (W == 2 * H) || (H == 2 * W);
    constexpr int LS = W + 1;  // padded LDS row stride (dwords)
    constexpr int TS = H * LS; // dwords per team

    const int team = tid / L, t = tid % L;
    const int blk = wg * TEAMS + team;
    const bool live = blk < nblk && team < TEAMS;
    uint32_t d = live ? descs[blk] : 0u;
    const int bx = d & 0x3FFF, by = (d >> 14) & 0x3FFF, tt = d >> 28;
    const int kc = kVtx[tt], kr = kHtx[tt];

    if (live && t < W) {
        int32_t in[H], out[H];
        const PIX* s = src + (size_t)by * src_stride + bx + t;
        const PIX* p = pred + (size_t)by * pred_stride + bx + t;
#pragma unroll
        for (int r = 0; r < H; r++) {
            const int rr = (kc == 2) ? H - 1 - r : r;  // FLIPADST: upside-down input (EbTransforms.c:2336-2343)
            in[r] = ((int32_t)s[(size_t)rr * src_stride] - (int32_t)p[(size_t)rr * pred_stride]) * (1 << S0);
        }
        fwd_1d<H, CBC>(kc, in, out);
        const int cc = (kr == 2) ? W - 1 - t : t;      // left-right flip on store (:2351-2356)
#pragma unroll
        for (int r = 0; r < H; r++) tile[team * TS + r * LS + cc] = S1 ? rshift_round(out[r], S1 ? S1 : 1) : out[r];
    }
    __syncthreads();

    uint32_t eobmax = 0, sumabs = 0;
    uint64_t energy = 0;
    int32_t q0 = 0;
    if (live && t < H) {
        int32_t in[W], out[W];
#pragma unroll
        for (int c = 0; c < W; c++) in[c] = tile[team * TS + t * LS + c];
        fwd_1d<W, CBR>(kr, in, out);
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t v = S2 ? rshift_round(out[c], S2 ? S2 : 1) : out[c];
            if (RECT2) v = mul_sqrt2(v);
            out[c] = v;
        }
        if (qp.coeff_shape) {
            // N2 / N4 / ONLY_DC (av1_estimate_transform_N2 / _N4 / _ONLY_DC, EbTransforms.c:3055-3431): the pruned transform families produce
            // the default coefficients of the top-left corner and zeros elsewhere; no energy of the discarded region (:2933-2964)
            const int cw = qp.coeff_shape == 3 ? 1 : W >> qp.coeff_shape, ch = qp.coeff_shape == 3 ? 1 : H >> qp.coeff_shape;
#pragma unroll
            for (int c = 0; c < W; c++)
                if (c >= cw || t >= ch) out[c] = 0;
        } else if constexpr (KW != W || KH != H) {
            // 64-pt sizes keep the top-left 32x32 / 32x16 / 16x32 only (svt_handle_transform*, EbTransforms.c:2763-2931)
#pragma unroll
            for (int c = 0; c < W; c++)
                if (c >= KW || t >= KH) energy += (uint64_t)((int64_t)out[c] * (int64_t)out[c]);
        }
        if (t < KH) {
            const size_t base = (size_t)blk * NK + (size_t)t * KW;
            if (coeff_out) {
#pragma unroll
                for (int c = 0; c < KW; c += 4) *(int4*)(coeff_out + base + c) = make_int4(out[c], out[c + 1], out[c + 2], out[c + 3]);
            }
            if (qcoeff) {
                const int16_t* iscan = scans.iscan[(W <= 16 && H <= 16) ? (tt < 10 ? 0 : ((tt & 1) ? 2 : 1)) : 0];
                int32_t qv[KW], dv[KW];
#pragma unroll
                for (int c = 0; c < KW; c++) {
                    const int rc = t * KW + c;
                    int32_t dq_abs;
                    const int32_t lvl = quant_one(qp, out[c], rc != 0, dq_abs);
                    const bool neg = out[c] < 0;
                    qv[c] = neg ? -lvl : lvl;
                    dv[c] = neg ? -dq_abs : dq_abs;
                    if (lvl) eobmax = max(eobmax, (uint32_t)iscan[rc] + 1u);
                    sumabs += (uint32_t)lvl;
                }
                q0 = qv[0];
#pragma unroll
                for (int c = 0; c < KW; c += 4) {
                    *(int4*)(qcoeff + base + c)  = make_int4(qv[c], qv[c + 1], qv[c + 2], qv[c + 3]);
                    if (!FBD || dqcoeff) *(int4*)(dqcoeff + base + c) = make_int4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
                }
                if constexpr (FBD != 0) inv_row_pass<W, H, FBD>(kr, t, dv, tile + team * TS + t * LS);   // row t of the tile was only ever read by this thread
            }
        }
        if constexpr (FBD != 0) {
            if (t >= KH) { const int32_t none[1] = {0}; inv_row_pass<W, H, FBD>(kr, t, none, tile + team * TS + t * LS); }
        }
    }
    // team reductions (all 64 lanes of each wave execute the shuffles)
    eobmax = team_max<L>(eobmax);
    sumabs = team_add<L>(sumabs);
    if constexpr (KW != W || KH != H) energy = team_add64<L>(energy);
    if (live && t == 0) {
        if (eob_out && qcoeff) eob_out[blk] = (uint16_t)eobmax;
        if (cul_out && qcoeff) {  // EbFullLoop.c:1595-1608
            int32_t cul = (int32_t)min(sumabs, 63u);
            if (q0 < 0) cul |= 1 << 6; else if (q0 > 0) cul += 2 << 6;
            cul_out[blk] = cul;
        }
        if (energy_out) energy_out[blk] = energy;
    }
    if constexpr (FBD != 0) {
        __syncthreads();
        if (live && t < W) inv_col_pass<W, H, FBD, PIX>(tile + team * TS, t, kc, kr, pred, pred_stride, recon, recon_stride, bx, by);
    }
}

template <int W, int H, typename PIX>
__global__ void __launch_bounds__(256)
fwd_txfm_quant_kernel(const PIX* __restrict__ src, int src_stride, const PIX* __restrict__ pred, int pred_stride,
                      const uint32_t* __restrict__ descs, int nblk, SvtHipQuantParams qp, SvtHipScanTables scans,
                      int32_t* __restrict__ coeff_out, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
                      uint16_t* __restrict__ eob_out, int32_t* __restrict__ cul_out, uint64_t* __restrict__ energy_out) {
    constexpr int L = W > H ? W : H, TEAMS = threads_of(W, H) / L;
    __shared__ int32_t tile[TEAMS * H * (W + 1)];
    fwd_block<W, H, PIX>(tile, blockIdx.x, threadIdx.x, src, src_stride, pred, pred_stride, descs, nblk, qp, scans, coeff_out, qcoeff, dqcoeff,
                         eob_out, cul_out, energy_out);
}

// ================================================================== inverse + reconstruction =====
template <int W, int H, int BD, typename PIX>
__device__ __forceinline__ void inv_block(int32_t* __restrict__ tile, int wg, int tid, const int32_t* __restrict__ dqcoeff, const PIX* pred,
                                          int pred_stride, PIX* recon, int recon_stride, const uint32_t* __restrict__ descs, int nblk) {
    constexpr int L = W > H ? W : H, TEAMS = threads_of(W, H) / L;
    constexpr int KW = W > 32 ? 32 : W, KH = H > 32 ? 32 : H, NK = KW * KH;
    constexpr int LS = W + 1;
    constexpr int TS = H * LS;

    const int team = tid / L, t = tid % L;
    const int blk = wg * TEAMS + team;
    const bool live = blk < nblk && team < TEAMS;
    uint32_t d = live ? descs[blk] : 0u;
    const int bx = d & 0x3FFF, by = (d >> 14) & 0x3FFF, tt = d >> 28;
    const int kc = kVtx[tt], kr = kHtx[tt];

    if (live && t < H) {
        int32_t row[KW];
        if (t < KH) {
            const int32_t* src = dqcoeff + (size_t)blk * NK + (size_t)t * KW;
#pragma unroll
            for (int c = 0; c < KW; c++) row[c] = src[c];
        }
        inv_row_pass<W, H, BD>(kr, t, row, tile + team * TS + t * LS);
    }
    __syncthreads();
    if (live && t < W) inv_col_pass<W, H, BD, PIX>(tile + team * TS, t, kc, kr, pred, pred_stride, recon, recon_stride, bx, by);
}

template <int W, int H, int BD, typename PIX>
__global__ void __launch_bounds__(256)
inv_txfm_add_kernel(const int32_t* __restrict__ dqcoeff, const PIX* pred, int pred_stride, PIX* recon, int recon_stride,
                    const uint32_t* __restrict__ descs, int nblk) {
    constexpr int L = W > H ? W : H, TEAMS = threads_of(W, H) / L;
    __shared__ int32_t tile[TEAMS * H * (W + 1)];
    inv_block<W, H, BD, PIX>(tile, blockIdx.x, threadIdx.x, dqcoeff, pred, pred_stride, recon, recon_stride, descs, nblk);
}

// ================================================================== mixed-size launches ==========
// One launch over several (transform size, plane) job lists: a frame's transform work is 15-20 short lists (a few hundred to a few
// thousand blocks each), too small to fill 256 CUs one at a time.  Every workgroup finds its job by its index (uniform scalar scan
// over <= 16 prefix sums held in the kernel argument), then runs the size's body; LDS and registers are those of the largest size.
constexpr int kMultiJobs = 16, kMultiTileDw = 8448;   // max over sizes of TEAMS * H * (W + 1)
struct FwdMulti { int njobs; int first_wg[kMultiJobs + 1]; SvtHipFwdTxJob job[kMultiJobs]; };
struct InvMulti { int njobs; int first_wg[kMultiJobs + 1]; SvtHipInvTxJob job[kMultiJobs]; };

#define FOR_ALL_TX_SIZES_DEV(X) \
    X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(4, 64, 64) X(5, 4, 8) X(6, 8, 4) X(7, 8, 16) X(8, 16, 8) X(9, 16, 32) \
    X(10, 32, 16) X(11, 32, 64) X(12, 64, 32) X(13, 4, 16) X(14, 16, 4) X(15, 8, 32) X(16, 32, 8) X(17, 16, 64) X(18, 64, 16)

template <typename PIX>
__global__ void __launch_bounds__(256)
fwd_txfm_quant_multi_kernel(const FwdMulti a) {
    __shared__ int32_t tile[kMultiTileDw];
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.first_wg[j + 1]) j++;
    const int wg = (int)blockIdx.x - a.first_wg[j];
    // scalar copies of the job (references into the kernel-argument struct would force a private-memory copy of the whole struct)
    const int tx_size = a.job[j].tx_size, nblk = a.job[j].nblk, src_stride = a.job[j].src_stride, pred_stride = a.job[j].pred_stride;
    const PIX* src = (const PIX*)a.job[j].d_src; const PIX* pred = (const PIX*)a.job[j].d_pred;
    const uint32_t* descs = a.job[j].d_descs;
    const SvtHipQuantParams qp = a.job[j].qp;
    const SvtHipScanTables scans = a.job[j].scans;
    int32_t *coeff = a.job[j].d_coeff, *qcoeff = a.job[j].d_qcoeff, *dqcoeff = a.job[j].d_dqcoeff, *cul = a.job[j].d_cul_level;
    uint16_t* eob = a.job[j].d_eob; uint64_t* energy = a.job[j].d_energy;
    switch (tx_size) {
#define X(id, w, h)                                                                                                                            \
    case id:                                                                                                                                   \
        fwd_block<w, h, PIX>(tile, wg, threadIdx.x, src, src_stride, pred, pred_stride, descs, nblk, qp, scans, coeff, qcoeff, dqcoeff, eob,  \
                             cul, energy);                                                                                                     \
        break;
        FOR_ALL_TX_SIZES_DEV(X)
#undef X
    default: break;
    }
}
// the same job lists with the reconstruction fused in (FBD = bit depth): recon[j] / recon_stride[j] per job
struct EncMulti { FwdMulti f; void* recon[kMultiJobs]; int recon_stride[kMultiJobs]; };
template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
enc_txfm_multi_kernel(const EncMulti e) {
    __shared__ int32_t tile[kMultiTileDw];
    int j = 0;
    while (j + 1 < e.f.njobs && (int)blockIdx.x >= e.f.first_wg[j + 1]) j++;
    const int wg = (int)blockIdx.x - e.f.first_wg[j];
    const int tx_size = e.f.job[j].tx_size, nblk = e.f.job[j].nblk, src_stride = e.f.job[j].src_stride, pred_stride = e.f.job[j].pred_stride, recon_stride = e.recon_stride[j];
    const PIX* src = (const PIX*)e.f.job[j].d_src; const PIX* pred = (const PIX*)e.f.job[j].d_pred; PIX* recon = (PIX*)e.recon[j];
    const uint32_t* descs = e.f.job[j].d_descs;
    const SvtHipQuantParams qp = e.f.job[j].qp;
    const SvtHipScanTables scans = e.f.job[j].scans;
    int32_t *coeff = e.f.job[j].d_coeff, *qcoeff = e.f.job[j].d_qcoeff, *dqcoeff = e.f.job[j].d_dqcoeff, *cul = e.f.job[j].d_cul_level;
    uint16_t* eob = e.f.job[j].d_eob; uint64_t* energy = e.f.job[j].d_energy;
    switch (tx_size) {
#define X(id, w, h)                                                                                                                                  \
    case id:                                                                                                                                         \
        fwd_block<w, h, PIX, BD>(tile, wg, threadIdx.x, src, src_stride, pred, pred_stride, descs, nblk, qp, scans, coeff, qcoeff, dqcoeff, eob, cul, \
                                 energy, recon, recon_stride);                                                                                       \
        break;
        FOR_ALL_TX_SIZES_DEV(X)
#undef X
    default: break;
    }
}
template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
inv_txfm_add_multi_kernel(const InvMulti a) {
    __shared__ int32_t tile[kMultiTileDw];
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.first_wg[j + 1]) j++;
    const SvtHipInvTxJob& J = a.job[j];
    const int wg = (int)blockIdx.x - a.first_wg[j];
    switch (J.tx_size) {
#define X(id, w, h)                                                                                                                     \
    case id:                                                                                                                            \
        inv_block<w, h, BD, PIX>(tile, wg, threadIdx.x, J.d_dqcoeff, (const PIX*)J.d_pred, J.pred_stride, (PIX*)J.d_recon, J.recon_stride, \
                                 J.d_descs, J.nblk);                                                                                     \
        break;
        FOR_ALL_TX_SIZES_DEV(X)
#undef X
    default: break;
    }
}
__host__ constexpr int teams_of(int ts) {
    constexpr int w[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
    constexpr int h[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};
    return threads_of(w[ts], h[ts]) / (w[ts] > h[ts] ? w[ts] : h[ts]);
}

template <int W, int H>
int launch_fwd(hipStream_t st, int pix_bytes, const void* src, int ss, const void* pred, int ps, const uint32_t* descs, int n,
               const SvtHipQuantParams& qp, const SvtHipScanTables& sc, int32_t* coeff, int32_t* q, int32_t* dq, uint16_t* eob,
               int32_t* cul, uint64_t* energy) {
    constexpr int L = W > H ? W : H, TEAMS = threads_of(W, H) / L;
    dim3 grid((n + TEAMS - 1) / TEAMS), block(threads_of(W, H));
    if (pix_bytes == 1)
        hipLaunchKernelGGL((fwd_txfm_quant_kernel<W, H, uint8_t>), grid, block, 0, st, (const uint8_t*)src, ss, (const uint8_t*)pred, ps,
                           descs, n, qp, sc, coeff, q, dq, eob, cul, energy);
    else
        hipLaunchKernelGGL((fwd_txfm_quant_kernel<W, H, uint16_t>), grid, block, 0, st, (const uint16_t*)src, ss, (const uint16_t*)pred, ps,
                           descs, n, qp, sc, coeff, q, dq, eob, cul, energy);
    return (int)hipGetLastError();
}
template <int W, int H>
int launch_inv(hipStream_t st, int pix_bytes, int bd, const int32_t* dq, const void* pred, int ps, void* recon, int rs,
               const uint32_t* descs, int n) {
    constexpr int L = W > H ? W : H, TEAMS = threads_of(W, H) / L;
    dim3 grid((n + TEAMS - 1) / TEAMS), block(threads_of(W, H));
    if (pix_bytes == 1)
        hipLaunchKernelGGL((inv_txfm_add_kernel<W, H, 8, uint8_t>), grid, block, 0, st, dq, (const uint8_t*)pred, ps, (uint8_t*)recon, rs, descs, n);
    else if (bd == 8)
        hipLaunchKernelGGL((inv_txfm_add_kernel<W, H, 8, uint16_t>), grid, block, 0, st, dq, (const uint16_t*)pred, ps, (uint16_t*)recon, rs, descs, n);
    else
        hipLaunchKernelGGL((inv_txfm_add_kernel<W, H, 10, uint16_t>), grid, block, 0, st, dq, (const uint16_t*)pred, ps, (uint16_t*)recon, rs, descs, n);
    return (int)hipGetLastError();
}

}  // namespace

#define FOR_ALL_TX_SIZES(X) \
    X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(4, 64, 64) X(5, 4, 8) X(6, 8, 4) X(7, 8, 16) X(8, 16, 8) X(9, 16, 32) \
    X(10, 32, 16) X(11, 32, 64) X(12, 64, 32) X(13, 4, 16) X(14, 16, 4) X(15, 8, 32) X(16, 32, 8) X(17, 16, 64) X(18, 64, 16)

extern "C" int svt_hip_launch_fwd_txfm_quant(hipStream_t st, int tx_size, int pix_bytes, const void* src, int src_stride,
                                             const void* pred, int pred_stride, const uint32_t* descs, int nblk,
                                             const SvtHipQuantParams* qp, const SvtHipScanTables* scans, int32_t* coeff,
                                             int32_t* qcoeff, int32_t* dqcoeff, uint16_t* eob, int32_t* cul_level,
                                             uint64_t* energy) {
    if (nblk <= 0) return 0;
    SvtHipQuantParams q0 = {};
    SvtHipScanTables s0 = {};
    if (qp) q0 = *qp;
    if (scans) s0 = *scans;
    switch (tx_size) {
#define X(id, w, h) case id: return launch_fwd<w, h>(st, pix_bytes, src, src_stride, pred, pred_stride, descs, nblk, q0, s0, coeff, qcoeff, dqcoeff, eob, cul_level, energy);
        FOR_ALL_TX_SIZES(X)
#undef X
    default: return (int)hipErrorInvalidValue;
    }
}

extern "C" int svt_hip_launch_inv_txfm_add(hipStream_t st, int tx_size, int pix_bytes, int bd, const int32_t* dqcoeff,
                                           const void* pred, int pred_stride, void* recon, int recon_stride,
                                           const uint32_t* descs, int nblk) {
    if (nblk <= 0) return 0;
    switch (tx_size) {
#define X(id, w, h) case id: return launch_inv<w, h>(st, pix_bytes, bd, dqcoeff, pred, pred_stride, recon, recon_stride, descs, nblk);
        FOR_ALL_TX_SIZES(X)
#undef X
    default: return (int)hipErrorInvalidValue;
    }
}

extern "C" int svt_hip_launch_fwd_txfm_quant_multi(hipStream_t st, int pix_bytes, const SvtHipFwdTxJob* jobs, int njobs) {
    for (int j0 = 0; j0 < njobs; j0 += kMultiJobs) {
        FwdMulti a = {};
        int wg = 0;
        for (int j = j0; j < njobs && j < j0 + kMultiJobs; j++) {
            if (jobs[j].nblk <= 0) continue;
            const int t = teams_of(jobs[j].tx_size);
            a.first_wg[a.njobs] = wg;
            a.job[a.njobs++] = jobs[j];
            wg += (jobs[j].nblk + t - 1) / t;
        }
        a.first_wg[a.njobs] = wg;
        if (!wg) continue;
        if (pix_bytes == 1) hipLaunchKernelGGL((fwd_txfm_quant_multi_kernel<uint8_t>), dim3(wg), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((fwd_txfm_quant_multi_kernel<uint16_t>), dim3(wg), dim3(256), 0, st, a);
    }
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_enc_txfm_multi(hipStream_t st, int pix_bytes, int bd, const SvtHipEncTxJob* jobs, int njobs) {
    for (int j0 = 0; j0 < njobs; j0 += kMultiJobs) {
        EncMulti e = {};
        int wg = 0;
        for (int j = j0; j < njobs && j < j0 + kMultiJobs; j++) {
            if (jobs[j].fwd.nblk <= 0) continue;
            const int t = teams_of(jobs[j].fwd.tx_size);
            e.f.first_wg[e.f.njobs] = wg;
            e.recon[e.f.njobs] = jobs[j].d_recon; e.recon_stride[e.f.njobs] = jobs[j].recon_stride;
            e.f.job[e.f.njobs++] = jobs[j].fwd;
            wg += (jobs[j].fwd.nblk + t - 1) / t;
        }
        e.f.first_wg[e.f.njobs] = wg;
        if (!wg) continue;
        if (pix_bytes == 1) hipLaunchKernelGGL((enc_txfm_multi_kernel<uint8_t, 8>), dim3(wg), dim3(256), 0, st, e);
        else if (bd == 8) hipLaunchKernelGGL((enc_txfm_multi_kernel<uint16_t, 8>), dim3(wg), dim3(256), 0, st, e);
        else hipLaunchKernelGGL((enc_txfm_multi_kernel<uint16_t, 10>), dim3(wg), dim3(256), 0, st, e);
    }
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_inv_txfm_add_multi(hipStream_t st, int pix_bytes, int bd, const SvtHipInvTxJob* jobs, int njobs) {
    for (int j0 = 0; j0 < njobs; j0 += kMultiJobs) {
        InvMulti a = {};
        int wg = 0;
        for (int j = j0; j < njobs && j < j0 + kMultiJobs; j++) {
            if (jobs[j].nblk <= 0) continue;
            const int t = teams_of(jobs[j].tx_size);
            a.first_wg[a.njobs] = wg;
            a.job[a.njobs++] = jobs[j];
            wg += (jobs[j].nblk + t - 1) / t;
        }
        a.first_wg[a.njobs] = wg;
        if (!wg) continue;
        if (pix_bytes == 1) hipLaunchKernelGGL((inv_txfm_add_multi_kernel<uint8_t, 8>), dim3(wg), dim3(256), 0, st, a);
        else if (bd == 8) hipLaunchKernelGGL((inv_txfm_add_multi_kernel<uint16_t, 8>), dim3(wg), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((inv_txfm_add_multi_kernel<uint16_t, 10>), dim3(wg), dim3(256), 0, st, a);
    }
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(txfm2d)
