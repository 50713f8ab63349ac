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
#include "svt_hip_internal.h"

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

template <typename PIX>
__device__ __forceinline__ void stage_and_boxsum(TileLds& L, const PIX* __restrict__ plane, int stride, int pw, int ph, int x0, int y0, int tid) {
    if (tid < 256) L.xtab[tid] = tid == 0 ? 1 : (tid == 255 ? 256 : (uint16_t)((256 * tid + (tid + 1) / 2) / (tid + 1)));
    for (int i = tid; i < IH * IW; i += 256) {
        const int r = i / IW, c = i - r * IW;
        const int x = min(max(x0 - 3 + c, -3), pw + 2), y = min(max(y0 - 3 + r, -3), ph + 2);   // never leave the 3-px extension
        L.in[i] = (uint16_t)plane[(ptrdiff_t)y * stride + x];
    }
    __syncthreads();
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

// ---- projection sums of every (restoration unit, parameter set): sums[unit][16][5] += {H00, H01, H11, C0, C1}
template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
sgr_search_kernel(const PIX* __restrict__ dgd, int stride, const PIX* __restrict__ src, int src_stride, int pw, int ph, int unit_size,
                  int units_x, int units_y, uint32_t ep_mask, unsigned long long* __restrict__ sums) {
    __shared__ TileLds L;
    __shared__ long long red[4][5];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    const int unit = min(y0 / unit_size, units_y - 1) * units_x + min(x0 / unit_size, units_x - 1);
    stage_and_boxsum(L, dgd, stride, pw, ph, x0, y0, tid);
    int32_t sv[4];   // (src << 4) - u per owned pixel; 0 for out-of-picture pixels
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int k = tid + 256 * t, i = k / TW, j = k - i * TW;
        sv[t] = (x0 + j < pw && y0 + i < ph) ? ((int32_t)src[(size_t)(y0 + i) * src_stride + x0 + j] << 4) - ((int32_t)L.in[(i + 3) * IW + j + 3] << 4) : 0;
    }
    for (int ep = 0; ep < 16; ep++) {
        if (!((ep_mask >> ep) & 1)) continue;
        __syncthreads();
        build_ab<BD>(L, ep, tid);
        long long h00 = 0, h01 = 0, h11 = 0, c0 = 0, c1 = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int k = tid + 256 * t, i = k / TW, j = k - i * TW;
            if (x0 + j >= pw || y0 + i >= ph) continue;
            int32_t f0 = 0, f1 = 0;
            filt_px(L, ep, i, j, f0, f1);
            const int32_t u = (int32_t)L.in[(i + 3) * IW + j + 3] << 4;
            const long long a = kSgr[ep][0] > 0 ? f0 - u : 0, b = kSgr[ep][1] > 0 ? f1 - u : 0;
            h00 += a * a; h01 += a * b; h11 += b * b; c0 += a * sv[t]; c1 += b * sv[t];
        }
        long long v[5] = {h00, h01, h11, c0, c1};
#pragma unroll
        for (int q = 0; q < 5; q++) {
#pragma unroll
            for (int m = 1; m < 64; m <<= 1)
                v[q] += ((long long)__shfl_xor((int)(v[q] >> 32), m, 64) << 32) | (unsigned)__shfl_xor((int)v[q], m, 64);
            if ((tid & 63) == 0) red[tid >> 6][q] = v[q];
        }
        __syncthreads();
        if (tid < 5) atomicAdd(&sums[((size_t)unit * 16 + ep) * 5 + tid], (unsigned long long)(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]));
    }
}

// ---- svt_apply_selfguided_restoration over a plane: per-unit parameter set (255 = unit not restored) and xqd
template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
sgr_apply_kernel(const PIX* __restrict__ dgd, int stride, PIX* __restrict__ dst, int dst_stride, int pw, int ph, int unit_size, int units_x,
                 int units_y, const uint8_t* __restrict__ unit_ep, const int32_t* __restrict__ unit_xqd) {
    __shared__ TileLds L;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    const int unit = min(y0 / unit_size, units_y - 1) * units_x + min(x0 / unit_size, units_x - 1);
    const int ep = unit_ep[unit];
    if (ep > 15) return;
    stage_and_boxsum(L, dgd, stride, pw, ph, x0, y0, tid);
    build_ab<BD>(L, ep, tid);
    // svt_decode_xq (EbRestoration.c:707-718)
    const int32_t xqd0 = unit_xqd[2 * unit], xqd1 = unit_xqd[2 * unit + 1];
    int32_t xq0, xq1;
    if (kSgr[ep][0] == 0) { xq0 = 0; xq1 = 128 - xqd1; }
    else if (kSgr[ep][1] == 0) { xq0 = xqd0; xq1 = 0; }
    else { xq0 = xqd0; xq1 = 128 - xq0 - xqd1; }
    for (int k = tid; k < TW * TH; k += 256) {
        const int i = k / TW, j = k - i * TW;
        if (x0 + j >= pw || y0 + i >= ph) continue;
        int32_t f0 = 0, f1 = 0;
        filt_px(L, ep, i, j, f0, f1);
        const int32_t u = (int32_t)L.in[(i + 3) * IW + j + 3] << 4;
        int32_t v = u << 7;
        if (kSgr[ep][0] > 0) v += xq0 * (f0 - u);
        if (kSgr[ep][1] > 0) v += xq1 * (f1 - u);
        const int32_t w = (int32_t)(int16_t)((v + (1 << 10)) >> 11);
        dst[(size_t)(y0 + i) * dst_stride + x0 + j] = (PIX)min(max(w, 0), (1 << BD) - 1);
    }
}

}  // namespace

extern "C" int svt_hip_launch_sgr_filter(hipStream_t st, int pix_bytes, int bd, const void* plane, int stride, int pw, int ph, int ep,
                                         int32_t* flt0, int32_t* flt1, int flt_stride) {
    dim3 grid((pw + 63) / 64, (ph + 15) / 16);
    if (pix_bytes == 1) hipLaunchKernelGGL((sgr_filter_kernel<uint8_t, 8>), grid, dim3(256), 0, st, (const uint8_t*)plane, stride, pw, ph, ep, flt0, flt1, flt_stride);
    else if (bd == 8) hipLaunchKernelGGL((sgr_filter_kernel<uint16_t, 8>), grid, dim3(256), 0, st, (const uint16_t*)plane, stride, pw, ph, ep, flt0, flt1, flt_stride);
    else hipLaunchKernelGGL((sgr_filter_kernel<uint16_t, 10>), grid, dim3(256), 0, st, (const uint16_t*)plane, stride, pw, ph, ep, flt0, flt1, flt_stride);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_sgr_search(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, const void* src, int src_stride,
                                         int pw, int ph, int unit_size, int units_x, int units_y, uint32_t ep_mask, int64_t* sums) {
    dim3 grid((pw + 63) / 64, (ph + 15) / 16);
    unsigned long long* s = (unsigned long long*)sums;
    if (pix_bytes == 1) hipLaunchKernelGGL((sgr_search_kernel<uint8_t, 8>), grid, dim3(256), 0, st, (const uint8_t*)dgd, stride, (const uint8_t*)src, src_stride, pw, ph, unit_size, units_x, units_y, ep_mask, s);
    else if (bd == 8) hipLaunchKernelGGL((sgr_search_kernel<uint16_t, 8>), grid, dim3(256), 0, st, (const uint16_t*)dgd, stride, (const uint16_t*)src, src_stride, pw, ph, unit_size, units_x, units_y, ep_mask, s);
    else hipLaunchKernelGGL((sgr_search_kernel<uint16_t, 10>), grid, dim3(256), 0, st, (const uint16_t*)dgd, stride, (const uint16_t*)src, src_stride, pw, ph, unit_size, units_x, units_y, ep_mask, s);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_sgr_apply(hipStream_t st, int pix_bytes, int bd, const void* dgd, int stride, void* dst, int dst_stride, int pw,
                                        int ph, int unit_size, int units_x, int units_y, const uint8_t* unit_ep, const int32_t* unit_xqd) {
    dim3 grid((pw + 63) / 64, (ph + 15) / 16);
    if (pix_bytes == 1) hipLaunchKernelGGL((sgr_apply_kernel<uint8_t, 8>), grid, dim3(256), 0, st, (const uint8_t*)dgd, stride, (uint8_t*)dst, dst_stride, pw, ph, unit_size, units_x, units_y, unit_ep, unit_xqd);
    else if (bd == 8) hipLaunchKernelGGL((sgr_apply_kernel<uint16_t, 8>), grid, dim3(256), 0, st, (const uint16_t*)dgd, stride, (uint16_t*)dst, dst_stride, pw, ph, unit_size, units_x, units_y, unit_ep, unit_xqd);
    else hipLaunchKernelGGL((sgr_apply_kernel<uint16_t, 10>), grid, dim3(256), 0, st, (const uint16_t*)dgd, stride, (uint16_t*)dst, dst_stride, pw, ph, unit_size, units_x, units_y, unit_ep, unit_xqd);
    return (int)hipGetLastError();
}
