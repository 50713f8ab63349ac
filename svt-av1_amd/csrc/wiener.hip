// wiener.hip — Wiener restoration statistics (the autocorrelation / cross-correlation sums of the filter search) on the matrix
// cores; gfx950.  (The Wiener *filter* pass lives in sgr.hip's loop-restoration apply kernel.)
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Encoder/Codec/EbRestorationPick.c:704  svt_av1_compute_stats_c  (+ find_average, EbRestorationPick.h:24) as called per restoration
//   unit by search_wiener_seg (:1347): M[k] = sum (y_k - avg)(x - avg), H[k][l] = sum (y_k - avg)(y_l - avg), k, l over the win x win
//   window offsets -- a Gram matrix of the (pixels x features) matrix Z = [window samples | source sample | 1].  This is the one
//   genuinely GEMM-shaped reduction on the hot path, so it runs on MFMA:
//     * samples are biased to int8 (d' = d - 128) and fed to v_mfma_i32_32x32x32_i8: exact int32 accumulation (|products| <= 2^14,
//       a wave sees < 2^17 pixels), A and B operands are THE SAME registers (G = Z^T Z), so any k-ordering inside the instruction
//       cancels out; the constant-1 feature yields sum d'_k and the pixel count, from which the kernel recovers the unit average
//       (find_average) and removes it algebraically:  H = G_kl - a (S_k + S_l) + N a^2,  M = G_kx - a (S_k + S_x) + N a^2,  a = avg - 128;
//     * a lane's operand = 16 consecutive pixels of one window offset (dx, dy): one aligned ds_read_b128 out of one of 7 LDS copies of
//       the row ring, copy dx being the rows shifted by dx bytes;
//     * one workgroup (4 waves) per restoration unit, wave w takes rows v_start + w, + 4, ...; 7x7: features 0..50 -> 2 blocks of 32 ->
//       3 MFMA tiles per 32 pixels; 5x5 / 3x3: one block, one MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kRing = 16;        // rows kept in LDS (10 needed per 4-row step)
// bytes per row copy (template parameter PITCH): 16 + unit width (< 1.5 x unit size) + 3 + 16-B over-read, multiple of 16: 144 / 240 / 432

// M, H of a unit from its Gram matrix G over the features [window samples (biased by -128) | source sample | 1]
// `bias` = what was subtracted from every sample before the products (128 for 8-bit planes, 2^(bd-1) for the 16-bit path), `divider` =
// svt_av1_compute_stats_highbd's bit_depth_divider (EbRestorationPick.c:753-757, :783-790; C division, truncating toward zero)
template <int WIN, typename GF>
__device__ __forceinline__ void wiener_finish(GF G, int tid, long long* __restrict__ Mo, long long* __restrict__ Ho, long long bias = 128,
                                              long long divider = 1) {
    constexpr int HALF = WIN / 2, NF = WIN * WIN, XF = NF, ONE = NF + 1;
    const long long N = G(ONE, ONE);
    const long long sum_d = G(HALF * WIN + HALF, ONE) + bias * N;
    const long long a = (long long)((unsigned long long)sum_d / (unsigned long long)N) - bias;   // find_average() - bias
    const long long Sx = G(XF, ONE);
    for (int i = tid; i < NF * NF + NF; i += 256) {
        if (i < NF * NF) {
            const int k = i / NF, l = i - k * NF;
            Ho[i] = (G(k, l) - a * (G(k, ONE) + G(l, ONE)) + N * a * a) / divider;
        } else {
            const int k = i - NF * NF;
            Mo[k] = (G(k, XF) - a * (G(k, ONE) + Sx) + N * a * a) / divider;
        }
    }
}

template <int WIN, int kPitch, bool BANDED>
__global__ void __launch_bounds__(256)
wiener_stats8_kernel(const uint8_t* __restrict__ dgd, int dgd_stride, const uint8_t* __restrict__ src, int src_stride, int pw, int ph,
                     int unit_size, int units_x, int units_y, int voff, long long* __restrict__ M_out, long long* __restrict__ H_out,
                     long long* __restrict__ G_raw) {
    constexpr int HALF = WIN / 2, NF = WIN * WIN, XF = NF, ONE = NF + 1, NBLK = (NF + 2 + 31) / 32, NT = NBLK == 1 ? 1 : 3;
    // copy c of the row ring holds d'[col + (c - 3)] at byte col + 16; the same memory later holds the [NT][32 x 32] int64 tile sums
    constexpr int kRingBytes = 7 * kRing * kPitch, kAccBytes = NT * 1024 * 8, kMainBytes = kRingBytes > kAccBytes ? kRingBytes : kAccBytes;
    __shared__ __attribute__((aligned(16))) int8_t lds_main[kMainBytes];
    int8_t (*ring)[kRing][kPitch] = (int8_t (*)[kRing][kPitch])lds_main;
    __shared__ __attribute__((aligned(16))) int8_t sring[2][4][kPitch];      // source rows of the current / next step (x' = s - 128)
    __shared__ __attribute__((aligned(16))) uint8_t raw[kRing][kPitch + 32];   // unbiased picture rows, byte 16 = column h0 - 3
    const int unit = blockIdx.x, ui = unit / units_x, uj = unit - ui * units_x;
    const int h0 = uj * unit_size, h1 = uj == units_x - 1 ? pw : h0 + unit_size;
    const int uv0 = max(0, ui * unit_size - voff), uv1 = ui == units_y - 1 ? ph : (ui + 1) * unit_size - voff;
    // BANDED (unit sizes 128 / 256: fewer units than CUs): blockIdx.y picks a band of 64 rows of the unit; the bands add their Gram
    // matrices into the unit's H buffer (packed upper triangle, zeroed by the launcher) and wiener_finalize_kernel turns it into M / H
    const int v0 = BANDED ? uv0 + 64 * (int)blockIdx.y : uv0, v1 = BANDED ? min(uv1, v0 + 64) : uv1;
    if (v0 >= v1) return;
    const int uw = h1 - h0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // lane -> features (block 0: lane & 31, block 1: 32 + (lane & 31)), k-group g = lane >> 5 (16 pixels each)
    const int g = lane >> 5;
    int copy[2], dy[2], kind[2];   // kind 0: window sample, 1: source sample, 2: ones, 3: unused
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int f = 32 * b + (lane & 31);
        kind[b] = f < NF ? 0 : (f == XF ? 1 : (f == ONE ? 2 : 3));
        copy[b] = f < NF ? f / WIN - HALF + 3 : 3;
        dy[b] = f < NF ? f % WIN - HALF : 0;
    }
    v16i C[3];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) C[t][r] = 0;

    // Row staging, two on-chip phases.  (A) picture rows -> raw[slot][] with plain dword loads and the byte misalignment of the row start
    // removed (raw byte 16 + j = picture column h0 - 3 + j, j in [0, uw + 6)); (B) raw -> the seven shifted, int8-biased copies.
    auto load_rows = [&](int ya, int yb) {          // picture rows [ya, yb] -> raw
        const int row_dw = (uw + 6 + 3) >> 2, n = (yb - ya + 1) * row_dw;
        for (int i = tid; i < n; i += 256) {
            const int r = i / row_dw, j = i - r * row_dw;
            const uint8_t* b = dgd + (ptrdiff_t)(ya + r) * dgd_stride + h0 - 3;
            const uint32_t sh = (uint32_t)((uintptr_t)b & 3);
            const uint32_t* gp = (const uint32_t*)(b - sh) + j;
            const int last_dw = (uw + 6 + (int)sh + 3) >> 2;
            const uint32_t lo = gp[0], hi = (sh && j + 1 < last_dw) ? gp[1] : 0u;
            *(uint32_t*)&raw[(ya + r) & (kRing - 1)][16 + 4 * j] = __builtin_amdgcn_alignbyte(hi, lo, sh);
        }
    };
    auto build_copies = [&](int ya, int yb) {       // raw rows [ya, yb] -> ring[0..6]
        const int row_dw = (16 + uw + 3 + 3) >> 2, n = (yb - ya + 1) * 7 * row_dw;   // copy bytes [0, 16 + uw + 3) hold every window sample of a valid pixel
        for (int i = tid; i < n; i += 256) {
            const int q = i % row_dw, c = (i / row_dw) % 7, r = i / (7 * row_dw);
            const int slot = (ya + r) & (kRing - 1);
            // copy byte jj <-> picture column h0 + jj - 16 + (c - 3) <-> raw byte 16 + (jj - 16 + c - 3) + 3 = jj + c
            const int rb = 4 * q + c;                   // raw byte index of the copy dword's first byte
            const uint32_t* rp = (const uint32_t*)&raw[slot][rb & ~3];
            *(uint32_t*)&ring[c][slot][4 * q] = __builtin_amdgcn_alignbyte(rp[1], rp[0], (uint32_t)(rb & 3)) ^ 0x80808080u;
        }
    };

    auto load_src = [&](int r0, int buf) {          // source rows r0 .. r0+3 (x' = s - 128), alignment removed the same way
        const int row_dw = (uw + 3) >> 2;
        for (int i = tid; i < 4 * row_dw; i += 256) {
            const int rr = i / row_dw, j = i - rr * row_dw;
            uint32_t v = 0;
            if (r0 + rr < v1) {
                const uint8_t* b = src + (size_t)(r0 + rr) * src_stride + h0;
                const uint32_t sh = (uint32_t)((uintptr_t)b & 3);
                const uint32_t* gp = (const uint32_t*)(b - sh) + j;
                const int last_dw = (uw + (int)sh + 3) >> 2;
                const uint32_t lo = gp[0], hi = (sh && j + 1 < last_dw) ? gp[1] : 0u;
                v = __builtin_amdgcn_alignbyte(hi, lo, sh) ^ 0x80808080u;
            }
            *(uint32_t*)&sring[buf][rr][4 * j] = v;
        }
    };

    // software pipeline over 4-row steps: while the MFMAs of step s run, the rows of step s+1 are loaded into ring slots the current step
    // does not read (16 slots >= 10 in use + 4 new) and into the other source buffer; two barriers per step
    int staged_to = min(v0 + 6, v1 + 2);
    load_rows(v0 - 3, staged_to);
    load_src(v0, 0);
    __syncthreads();
    build_copies(v0 - 3, staged_to);
    int sbuf = 0;
    for (int r0 = v0; r0 < v1; r0 += 4, sbuf ^= 1) {
        __syncthreads();   // copies of this step are complete; the previous step's operand reads are done
        const int ya = staged_to + 1, yb = min(r0 + 10, v1 + 2);
        const bool more = r0 + 4 < v1;
        if (more && yb >= ya) load_rows(ya, yb);
        if (more) load_src(r0 + 4, sbuf ^ 1);
        const int row = r0 + wave;
        if (row < v1) {
            for (int x0 = 0; x0 < uw; x0 += 32) {
                const int p0 = x0 + 16 * g;                       // first pixel of this lane's 16
                const int nvalid = min(max(uw - p0, 0), 16);
                v4i A[2];
#pragma unroll
                for (int b = 0; b < NBLK; b++) {
                    v4i a = {0, 0, 0, 0};
                    if (kind[b] == 0) a = *(const v4i*)&ring[copy[b]][(row + dy[b]) & (kRing - 1)][16 + p0];
                    else if (kind[b] == 1) a = *(const v4i*)&sring[sbuf][wave][p0];
                    else if (kind[b] == 2) a = v4i{0x01010101, 0x01010101, 0x01010101, 0x01010101};
                    if (nvalid < 16) {   // pixels past the unit's right edge contribute nothing (only the last 32-pixel step)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int nb = min(max(nvalid - 4 * q, 0), 4);
                            a[q] &= nb == 4 ? -1 : (int)((1u << (8 * nb)) - 1u);
                        }
                    }
                    A[b] = a;
                }
                C[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], A[0], C[0], 0, 0, 0);
                if (NBLK == 2) {
                    C[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], A[1], C[1], 0, 0, 0);
                    C[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], A[1], C[2], 0, 0, 0);
                }
            }
        }
        if (more && yb >= ya) {
            __syncthreads();   // raw rows of the next step have landed
            build_copies(ya, yb);
            staged_to = yb;
        }
    }
    // ---- add the four waves' tiles (int32 -> int64) in LDS; element (row, col) of a tile: lane = col + 32 * ((row >> 2) & 1),
    //      register = (row & 3) + 4 * (row >> 3)   (C/D layout of the 32x32 MFMAs)
    __syncthreads();
    unsigned long long* acc = (unsigned long long*)lds_main;   // [NT][32 * 32]
    for (int i = tid; i < NT * 1024; i += 256) acc[i] = 0ull;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), tcol = lane & 31;
            atomicAdd(&acc[t * 1024 + trow * 32 + tcol], (unsigned long long)(long long)C[t][r]);
        }
    __syncthreads();
    auto G = [&](int f1, int f2) -> long long {
        if (f1 > f2) { const int t = f1; f1 = f2; f2 = t; }
        const int t = f2 < 32 ? 0 : (f1 < 32 ? 1 : 2);
        return (long long)acc[t * 1024 + (f1 & 31) * 32 + (f2 & 31)];
    };
    if (BANDED || G_raw) {   // G_raw: the caller wants the packed Gram matrix itself (16-bit path: three 8-bit component planes, combined later)
        constexpr int F = NF + 2;
        unsigned long long* P = G_raw ? (unsigned long long*)(G_raw + (size_t)unit * (F * (F + 1) / 2)) : (unsigned long long*)(H_out + (size_t)unit * NF * NF);
        for (int i = tid; i < F * F; i += 256) {
            const int f1 = i / F, f2 = i - f1 * F;
            if (f1 <= f2) atomicAdd(&P[f1 * F - f1 * (f1 - 1) / 2 + (f2 - f1)], (unsigned long long)G(f1, f2));
        }
        return;
    }
    wiener_finish<WIN>(G, tid, M_out + (size_t)unit * NF, H_out + (size_t)unit * NF * NF);
}

// packed Gram matrix of a unit (written by the banded kernel into the unit's H buffer) -> M, H
template <int WIN>
__global__ void __launch_bounds__(256)
wiener_finalize_kernel(long long* __restrict__ M_out, long long* __restrict__ H_out) {
    constexpr int NF = WIN * WIN, F = NF + 2, NP = F * (F + 1) / 2;
    __shared__ long long P[NP];
    const int unit = blockIdx.x, tid = threadIdx.x;
    long long* Ho = H_out + (size_t)unit * NF * NF;
    for (int i = tid; i < NP; i += 256) P[i] = Ho[i];
    __syncthreads();   // everything is read before the first H entry is overwritten
    auto G = [&](int f1, int f2) -> long long {
        if (f1 > f2) { const int t = f1; f1 = f2; f2 = t; }
        return P[f1 * F - f1 * (f1 - 1) / 2 + (f2 - f1)];
    };
    wiener_finish<WIN>(G, tid, M_out + (size_t)unit * NF, Ho);
}

// ---- 16-bit planes.  v = d - 2^(bd-1) is split as v = 32 h + l (h = v >> 5, l = v & 31): three 8-bit component planes H, L and S = H + L
// (|h + l| < 128 up to 12 bits) go through the exact int8 Gram kernel above, and
//   sum v_k v_l = 1024 G_hh + 32 (G_ss - G_hh - G_ll) + G_ll,   sum v_k = 32 S_h + S_l          (all int64, exact)
// because (h_k + l_k)(h_l + l_l) - h_k h_l - l_k l_l = h_k l_l + l_k h_l.  Components are stored + 128 so that the 8-bit kernel's own
// bias removal (x ^ 0x80) returns them.
__global__ void __launch_bounds__(256)
wiener_split16_kernel(const uint16_t* __restrict__ in, int in_stride, int w, int h, int bias, uint8_t* __restrict__ oh, uint8_t* __restrict__ ol,
                      uint8_t* __restrict__ os, int pitch) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const int v = (int)in[(size_t)y * in_stride + x] - bias;
    const int hh = v >> 5, ll = v & 31;
    oh[(size_t)y * pitch + x] = (uint8_t)(hh + 128); ol[(size_t)y * pitch + x] = (uint8_t)(ll + 128); os[(size_t)y * pitch + x] = (uint8_t)(hh + ll + 128);
}

template <int WIN>
__global__ void __launch_bounds__(256)
wiener_finalize16_kernel(const long long* __restrict__ Ghh, const long long* __restrict__ Gll, const long long* __restrict__ Gss, long long bias,
                         long long divider, long long* __restrict__ M_out, long long* __restrict__ H_out) {
    constexpr int NF = WIN * WIN, F = NF + 2, NP = F * (F + 1) / 2, ONE = NF + 1;
    __shared__ long long P[NP];
    const int unit = blockIdx.x, tid = threadIdx.x;
    const long long *hh = Ghh + (size_t)unit * NP, *ll = Gll + (size_t)unit * NP, *ss = Gss + (size_t)unit * NP;
    for (int f1 = 0; f1 < F; f1++)
        for (int f2 = f1 + tid; f2 < F; f2 += 256) {
            const int i = f1 * F - f1 * (f1 - 1) / 2 + (f2 - f1);
            long long g;
            if (f2 == ONE) g = f1 == ONE ? hh[i] : 32 * hh[i] + ll[i];
            else g = 1024 * hh[i] + 32 * (ss[i] - hh[i] - ll[i]) + ll[i];
            P[i] = g;
        }
    __syncthreads();
    auto G = [&](int f1, int f2) -> long long {
        if (f1 > f2) { const int t = f1; f1 = f2; f2 = t; }
        return P[f1 * F - f1 * (f1 - 1) / 2 + (f2 - f1)];
    };
    wiener_finish<WIN>(G, tid, M_out + (size_t)unit * NF, H_out + (size_t)unit * NF * NF, bias, divider);
}

template <int WIN>
int launch_raw(hipStream_t st, const uint8_t* dgd, int dgd_stride, const uint8_t* src, int src_stride, int pw, int ph, int unit_size, int units_x,
               int units_y, int voff, long long* G) {
    const int n = units_x * units_y;
    if (unit_size <= 64) hipLaunchKernelGGL((wiener_stats8_kernel<WIN, 144, false>), dim3(n), dim3(256), 0, st, dgd, dgd_stride, src, src_stride, pw, ph, unit_size,
                                            units_x, units_y, voff, (long long*)nullptr, (long long*)nullptr, G);
    else {
        const dim3 grid(n, (unit_size * 3 / 2 + 63) / 64);
        if (unit_size <= 128) hipLaunchKernelGGL((wiener_stats8_kernel<WIN, 240, true>), grid, dim3(256), 0, st, dgd, dgd_stride, src, src_stride, pw, ph, unit_size,
                                                 units_x, units_y, voff, (long long*)nullptr, (long long*)nullptr, G);
        else hipLaunchKernelGGL((wiener_stats8_kernel<WIN, 432, true>), grid, dim3(256), 0, st, dgd, dgd_stride, src, src_stride, pw, ph, unit_size, units_x,
                                units_y, voff, (long long*)nullptr, (long long*)nullptr, G);
    }
    return (int)hipGetLastError();
}

// ---- initial Wiener filter of every restoration unit from its statistics: search_wiener_seg between svt_av1_compute_stats and the tap refinement
// (Encoder/Codec/EbRestorationPick.c:1388-1407): wiener_decompose_sep_sym (:946-979 — four rounds of update_a_sep_sym :841 / update_b_sep_sym :895, each a folded
// (half + 1)-tap normal-equation system solved by linsolve_wiener :800), finalize_sym_filter (:1022-1052), compute_score (:980-1020).  64-bit INTEGER arithmetic with
// truncating divisions throughout, so the device result is the host's bit for bit (the tests compare it with a CPU restatement that is itself pinned to the
// reference's functions).
// One wave per unit.  The statistics (M: win^2, H: win^4 int64 = 19 KB for win 7) sit in LDS for the eight half-rounds and the score.  A half-round's 2401-term sum is
// split by destination: lane (p, q) owns the terms that land on B[fold(q)][fold(p)] with (p, q) as the un-folded index pair, i.e. 49 lanes x 49 terms; the folds (<= 4
// lanes per entry) and the 3 x 3 solve are lane 0's, a few hundred serial operations per half-round.
constexpr long long kWnScale = 1ll << 16;   // WIENER_TAP_SCALE_FACTOR (:42)
constexpr int kWnStep = 128;                // WIENER_FILT_STEP (Common/Codec/EbRestoration.h:125)

struct WnInitLds {
    long long H[49 * 49], M[49];
    long long partB[64], partA[64];
    int a[8], b[8];          // vertical / horizontal taps, scaled by kWnScale
    short v[8], h[8];
    long long score;
};

__device__ __forceinline__ int wn_fold(int i, int win) { return i > (win >> 1) ? win - 1 - i : i; }

// linsolve_wiener (:800-838): neighbour-swap pivoting, the reference's truncations; A is n x n with row stride st
__device__ int wn_solve(int n, long long* A, int st, long long* b, int* x) {
    for (int k = 0; k + 1 < n; k++) {
        for (int i = n - 1; i > k; i--) {
            const long long lo = A[(i - 1) * st + k], hi = A[i * st + k];
            if ((lo < 0 ? -lo : lo) >= (hi < 0 ? -hi : hi)) continue;
            for (int j = 0; j < n; j++) { const long long t = A[i * st + j]; A[i * st + j] = A[(i - 1) * st + j]; A[(i - 1) * st + j] = t; }
            const long long t = b[i]; b[i] = b[i - 1]; b[i - 1] = t;
        }
        for (int i = k + 1; i < n; i++) {
            const long long piv = A[k * st + k], c = A[i * st + k];
            if (piv == 0) return 0;
            for (int j = 0; j < n; j++) A[i * st + j] -= c / 256 * A[k * st + j] / piv * 256;
            b[i] -= c * b[k] / piv;
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        if (A[i * st + i] == 0) return 0;
        long long c = 0;
        for (int j = i + 1; j < n; j++) c += A[i * st + j] * x[j] / kWnScale;
        x[i] = (int)(kWnScale * (b[i] - c) / A[i * st + i]);
    }
    return 1;
}

// which = 0: update_a_sep_sym (solve the vertical taps L.a with L.b fixed), 1: update_b_sep_sym
__device__ void wn_half_round(WnInitLds& L, int win, int which, int lane) {
    const int win2 = win * win, h1 = (win >> 1) + 1;
    const int* fixed = which == 0 ? L.b : L.a;
    long long sb = 0, sa = 0;
    if (lane < win2) {
        const int p = lane / win, q = lane - p * win;
        // which 0 (:856-867): lane (k, l) = (p, q) sums hc[j * win + i][k * win2 + l] * b[i] / S * b[j] / S over (i, j) = (r, t);
        // which 1 (:909-921): lane (i, j) = (p, q) sums hc[i * win + j][k * win2 + l] * a[k] / S * a[l] / S over (k, l) = (r, t);
        // hc[x * win + y][z * win2 + w] = H[(x * win + z) * win2 + y * win + w] (:962-968)
        for (int r = 0; r < win; r++)
            for (int t = 0; t < win; t++) {
                const long long hv = which == 0 ? L.H[(t * win + p) * win2 + r * win + q] : L.H[(p * win + r) * win2 + q * win + t];
                sb += hv * fixed[r] / kWnScale * fixed[t] / kWnScale;
            }
        // A (:850-855 / :904-907): which 0: M[i][j] * b[i] lands on fold(j); which 1: M[i][j] * a[j] lands on fold(i); lane (i, j) = (p, q)
        sa = L.M[lane] * fixed[which == 0 ? p : q] / kWnScale;
    }
    L.partB[lane] = sb; L.partA[lane] = sa;
    __syncthreads();
    if (lane == 0) {
        long long A[4] = {0, 0, 0, 0}, B[16];
        for (int e = 0; e < 16; e++) B[e] = 0;
        for (int p = 0; p < win; p++)
            for (int q = 0; q < win; q++) {
                B[wn_fold(q, win) * h1 + wn_fold(p, win)] += L.partB[p * win + q];
                A[wn_fold(which == 0 ? q : p, win)] += L.partA[p * win + q];
            }
        // the taps sum to one: the centre tap is eliminated from the system (:868-882 / :923-937)
        const long long a_c = A[h1 - 1], b_cc = B[(h1 - 1) * h1 + h1 - 1];
        for (int i = 0; i < h1 - 1; i++) A[i] -= a_c * 2 + B[i * h1 + h1 - 1] - 2 * b_cc;
        for (int i = 0; i < h1 - 1; i++)
            for (int j = 0; j < h1 - 1; j++) B[i * h1 + j] -= 2 * (B[i * h1 + h1 - 1] + B[(h1 - 1) * h1 + j] - 2 * b_cc);
        int S[7];
        if (wn_solve(h1 - 1, B, h1, A, S)) {   // singular: the vector keeps its value (:883, :938)
            S[h1 - 1] = (int)kWnScale;
            for (int i = h1; i < win; i++) { S[i] = S[win - 1 - i]; S[h1 - 1] -= 2 * S[i]; }
            int* upd = which == 0 ? L.a : L.b;
            for (int i = 0; i < win; i++) upd[i] = S[i];
        }
    }
    __syncthreads();
}

// finalize_sym_filter (:1022-1052); fi[8] starts zeroed (the 3-tap branch reads fi[1] before anything wrote it; the host hook hands the reference a zeroed WienerInfo)
__device__ void wn_finalize(int win, const int* f, short* fi) {
    for (int i = 0; i < 8; i++) fi[i] = 0;
    for (int i = 0; i < (win >> 1); i++) {
        const long long v = (long long)f[i] * kWnStep;
        fi[i] = (short)((v < 0 ? v - kWnScale / 2 : v + kWnScale / 2) / kWnScale);
    }
    // WIENER_FILT_TAPn_{MIN,MAX}V (EbRestoration.h:130-149): tap 0 in [-5, 10], tap 1 in [-23, 8], tap 2 in [-17, 46]
    if (win == 7) {
        fi[0] = (short)min(max((int)fi[0], -5), 10); fi[1] = (short)min(max((int)fi[1], -23), 8); fi[2] = (short)min(max((int)fi[2], -17), 46);
    } else {
        fi[2] = (short)min(max((int)fi[1], -17), 46); fi[1] = (short)min(max((int)fi[0], -23), 8); fi[0] = 0;
    }
    fi[6] = fi[0]; fi[5] = fi[1]; fi[4] = fi[2];
    fi[3] = (short)(-2 * (fi[0] + fi[1] + fi[2]));
}

__global__ void __launch_bounds__(64)
wiener_init_kernel(const long long* __restrict__ M, const long long* __restrict__ H, int win, int n_units, short* __restrict__ unit_wiener,
                   unsigned char* __restrict__ active, signed char* __restrict__ status) {
    __shared__ WnInitLds L;
    const int u = blockIdx.x, lane = threadIdx.x, win2 = win * win, off = (7 - win) >> 1;
    if (u >= n_units) return;
    for (int i = lane; i < win2 * win2; i += 64) L.H[i] = H[(size_t)u * win2 * win2 + i];
    for (int i = lane; i < win2; i += 64) L.M[i] = M[(size_t)u * win2 + i];
    if (lane < win) {
        const int mid[7] = {3, -7, 15, 128 - 2 * (3 - 7 + 15), 15, -7, 3};   // WIENER_FILT_TAPn_MIDV
        L.a[lane] = L.b[lane] = (int)(kWnScale / kWnStep) * mid[lane + off];
    }
    __syncthreads();
    for (int round = 1; round < 5; round++) {   // NUM_WIENER_ITERS = 5 (:40): four rounds
        wn_half_round(L, win, 0, lane);
        wn_half_round(L, win, 1, lane);
    }
    if (lane == 0) { wn_finalize(win, L.a, L.v); wn_finalize(win, L.b, L.h); }
    __syncthreads();
    // compute_score (:980-1020): lane k sums row k of ab^T H ab and its term of ab . M
    long long q = 0, pterm = 0;
    if (lane < win2) {
        short a[7], b[7];
        a[3] = b[3] = (short)kWnStep;
        for (int i = 0; i < 3; i++) { a[i] = a[6 - i] = L.v[i]; b[i] = b[6 - i] = L.h[i]; a[3] -= 2 * a[i]; b[3] -= 2 * b[i]; }
        const int kk = lane / win, kl = lane - kk * win;
        const int abk = a[kl + off] * b[kk + off];
        pterm = abk * L.M[lane] / kWnStep / kWnStep;
        for (int l = 0; l < win2; l++) {
            const int lk = l / win, ll = l - lk * win;
            const int abl = a[ll + off] * b[lk + off];
            q += abk * L.H[lane * win2 + l] * abl / kWnStep / kWnStep / kWnStep / kWnStep;
        }
    }
    L.partB[lane] = q; L.partA[lane] = pterm;
    __syncthreads();
    if (lane == 0) {
        long long Q = 0, P = 0;
        for (int k = 0; k < win2; k++) { Q += L.partB[k]; P += L.partA[k]; }
        const long long ident = L.H[(win2 >> 1) * win2 + (win2 >> 1)] - 2 * L.M[win2 >> 1];
        const int st = (Q - 2 * P - ident) > 0 ? 2 : 1;
        status[u] = (signed char)st;
        active[u] = (unsigned char)(st == 1);
    }
    if (lane < 8) { unit_wiener[16 * (size_t)u + lane] = L.v[lane]; unit_wiener[16 * (size_t)u + 8 + lane] = L.h[lane]; }
}

}  // namespace

// scratch layout (bytes) of the 16-bit path; the API allocates it
extern "C" size_t svt_hip_wiener_stats16_scratch(int win, int pw, int ph, int n_units) {
    const size_t dp = ((size_t)pw + 6 + 63) & ~(size_t)63, sp = ((size_t)pw + 63) & ~(size_t)63;
    const size_t F = (size_t)win * win + 2, np = F * (F + 1) / 2;
    return 3 * (64 + dp * ((size_t)ph + 6) + 64) + 3 * (sp * (size_t)ph + 64) + 3 * np * 8 * (size_t)n_units + 256;
}

extern "C" int svt_hip_launch_wiener_stats16(hipStream_t st, int win, int bd, const uint16_t* dgd, int dgd_stride, const uint16_t* src, int src_stride,
                                             int pw, int ph, int unit_size, int units_x, int units_y, int ss_y, int64_t* M, int64_t* H, uint8_t* scratch) {
    const int voff = 8 >> ss_y, n = units_x * units_y;
    if (n <= 0) return 0;
    const size_t dp = ((size_t)pw + 6 + 63) & ~(size_t)63, sp = ((size_t)pw + 63) & ~(size_t)63;
    const size_t F = (size_t)win * win + 2, np = F * (F + 1) / 2;
    uint8_t* p = (uint8_t*)(((uintptr_t)scratch + 63) & ~(uintptr_t)63);
    uint8_t *dpl[3], *spl[3];
    for (int i = 0; i < 3; i++) { dpl[i] = p + 64; p += 64 + dp * ((size_t)ph + 6) + 64; }   // 64 B of slack in front: the stats kernel reads aligned dwords
    for (int i = 0; i < 3; i++) { spl[i] = p; p += sp * (size_t)ph + 64; }
    p = (uint8_t*)(((uintptr_t)p + 63) & ~(uintptr_t)63);
    long long* G[3];
    for (int i = 0; i < 3; i++) { G[i] = (long long*)p; p += np * 8 * (size_t)n; }
    if (hipMemsetAsync(G[0], 0, 3 * np * 8 * (size_t)n, st) != hipSuccess) return (int)hipGetLastError();
    const int bias = 1 << (bd - 1);
    hipLaunchKernelGGL(wiener_split16_kernel, dim3((pw + 6 + 255) / 256, ph + 6), dim3(256), 0, st, dgd - 3 * (ptrdiff_t)dgd_stride - 3, dgd_stride, pw + 6, ph + 6, bias,
                       dpl[0], dpl[1], dpl[2], (int)dp);
    hipLaunchKernelGGL(wiener_split16_kernel, dim3((pw + 255) / 256, ph), dim3(256), 0, st, src, src_stride, pw, ph, bias, spl[0], spl[1], spl[2], (int)sp);
    for (int i = 0; i < 3; i++) {
        const uint8_t* d0 = dpl[i] + 3 * dp + 3;
        int e;
        if (win == 7) e = launch_raw<7>(st, d0, (int)dp, spl[i], (int)sp, pw, ph, unit_size, units_x, units_y, voff, G[i]);
        else if (win == 5) e = launch_raw<5>(st, d0, (int)dp, spl[i], (int)sp, pw, ph, unit_size, units_x, units_y, voff, G[i]);
        else e = launch_raw<3>(st, d0, (int)dp, spl[i], (int)sp, pw, ph, unit_size, units_x, units_y, voff, G[i]);
        if (e) return e;
    }
    const long long divider = bd == 12 ? 16 : (bd == 10 ? 4 : 1);
    if (win == 7) hipLaunchKernelGGL((wiener_finalize16_kernel<7>), dim3(n), dim3(256), 0, st, G[0], G[1], G[2], (long long)bias, divider, (long long*)M, (long long*)H);
    else if (win == 5) hipLaunchKernelGGL((wiener_finalize16_kernel<5>), dim3(n), dim3(256), 0, st, G[0], G[1], G[2], (long long)bias, divider, (long long*)M, (long long*)H);
    else hipLaunchKernelGGL((wiener_finalize16_kernel<3>), dim3(n), dim3(256), 0, st, G[0], G[1], G[2], (long long)bias, divider, (long long*)M, (long long*)H);
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_wiener_stats8(hipStream_t st, int win, const uint8_t* dgd, int dgd_stride, const uint8_t* src, int src_stride, int pw,
                                            int ph, int unit_size, int units_x, int units_y, int ss_y, int64_t* M, int64_t* H) {
    const int voff = 8 >> ss_y, n = units_x * units_y;
    if (n <= 0) return 0;
#define LAUNCH(W, P, B, GRID) hipLaunchKernelGGL((wiener_stats8_kernel<W, P, B>), GRID, dim3(256), 0, st, dgd, dgd_stride, src, src_stride, pw, ph, \
                                                 unit_size, units_x, units_y, voff, (long long*)M, (long long*)H, (long long*)nullptr)
#define BY_WIN(P, B, GRID) do { if (win == 7) LAUNCH(7, P, B, GRID); else if (win == 5) LAUNCH(5, P, B, GRID); else LAUNCH(3, P, B, GRID); } while (0)
    if (unit_size <= 64) {
        BY_WIN(144, false, dim3(n));
    } else {
        // a unit is up to 1.5 x unit_size rows: ceil(1.5 * unit_size / 64) bands
        const dim3 grid(n, (unit_size * 3 / 2 + 63) / 64);
        if (hipMemsetAsync(H, 0, (size_t)n * win * win * win * win * sizeof(int64_t), st) != hipSuccess) return (int)hipGetLastError();
        if (unit_size <= 128) BY_WIN(240, true, grid); else BY_WIN(432, true, grid);
        if (win == 7) hipLaunchKernelGGL((wiener_finalize_kernel<7>), dim3(n), dim3(256), 0, st, (long long*)M, (long long*)H);
        else if (win == 5) hipLaunchKernelGGL((wiener_finalize_kernel<5>), dim3(n), dim3(256), 0, st, (long long*)M, (long long*)H);
        else hipLaunchKernelGGL((wiener_finalize_kernel<3>), dim3(n), dim3(256), 0, st, (long long*)M, (long long*)H);
    }
#undef BY_WIN
#undef LAUNCH
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_wiener_init(hipStream_t st, int win, int n_units, const int64_t* M, const int64_t* H, int16_t* unit_wiener, uint8_t* active, int8_t* status) {
    if (n_units <= 0) return 0;
    hipLaunchKernelGGL(wiener_init_kernel, dim3(n_units), dim3(64), 0, st, (const long long*)M, (const long long*)H, win, n_units, (short*)unit_wiener, active, (signed char*)status);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(wiener)
