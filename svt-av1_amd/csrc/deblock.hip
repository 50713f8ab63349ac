// deblock.hip — AV1 deblocking loop filter, whole plane per launch, gfx950.
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Common/Codec/EbDeblockingCommon.c:251-393, :882-921   svt_aom_lpf_{horizontal,vertical}_{4,6,8,14}_c
//   Common/Codec/EbDeblockingCommon.c:483-582, :698-805   svt_aom_highbd_lpf_*_c
//   Encoder/Codec/EbDeblockingFilter.c:321-611            svt_av1_filter_block_plane_vert / _horz (the per-SB walk)
//   Encoder/Codec/EbDeblockingFilter.c:711                svt_av1_loop_filter_frame
// The reference walks SB by SB and interleaves "vertical edges of SB(x), horizontal edges of
// SB(x-1)"; that is order-equivalent to the normative "all vertical edges of the picture, then all
// horizontal edges", and inside one direction every edge segment is independent because the filter
// length is bounded by the smaller transform next to the edge (set_lpf_parameters, :286-300).
// So: two launches per plane, one lane per (edge, sample), the per-4x4 edge descriptors
// ((level << 8) | length) are produced on the host (svt_hip_dlf_build_edges).
// Lanes run along x in both passes, so plane reads/writes are coalesced rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"

namespace {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// px[0..13] = p6..p0 q0..q6; returns how many samples on each side may have changed
// lim8 / blim8 / thr8: the 8-bit-scale thresholds the reference's filters take (*limit, *blimit, *thresh); scaled by the bit depth here
template <int BD>
__device__ __forceinline__ int lpf_core_thr(int (&px)[14], int len, int lim8, int blim8, int thr8) {
    constexpr int sh = BD - 8, t80 = 0x80 << sh, lo = -t80, hi = t80 - 1, one = 1 << sh;
    const int lim = lim8 << sh, blim = blim8 << sh, thr = thr8 << sh;
#define p(i) px[6 - (i)]
#define q(i) px[7 + (i)]
    bool m = (iabs(p(1) - p(0)) > lim) | (iabs(q(1) - q(0)) > lim) | ((iabs(p(0) - q(0)) * 2 + iabs(p(1) - q(1)) / 2) > blim);
    if (len >= 6) m |= (iabs(p(2) - p(1)) > lim) | (iabs(q(2) - q(1)) > lim);
    if (len >= 8) m |= (iabs(p(3) - p(2)) > lim) | (iabs(q(3) - q(2)) > lim);
    const bool mask = !m;
    bool flat = false, flat2 = false;
    if (len >= 6) {
        flat = !((iabs(p(1) - p(0)) > one) | (iabs(q(1) - q(0)) > one) | (iabs(p(2) - p(0)) > one) | (iabs(q(2) - q(0)) > one));
        if (len >= 8) flat &= !((iabs(p(3) - p(0)) > one) | (iabs(q(3) - q(0)) > one));
    }
    if (len == 14)
        flat2 = !((iabs(p(4) - p(0)) > one) | (iabs(q(4) - q(0)) > one) | (iabs(p(5) - p(0)) > one) | (iabs(q(5) - q(0)) > one) |
                  (iabs(p(6) - p(0)) > one) | (iabs(q(6) - q(0)) > one));
#define RP2(v, n) (((v) + (1 << ((n)-1))) >> (n))
    if (len == 14 && flat2 && flat && mask) {
        const int p6 = p(6), p5 = p(5), p4 = p(4), p3 = p(3), p2 = p(2), p1 = p(1), p0 = p(0);
        const int q0 = q(0), q1 = q(1), q2 = q(2), q3 = q(3), q4 = q(4), q5 = q(5), q6 = q(6);
        p(5) = RP2(p6 * 7 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0, 4);
        p(4) = RP2(p6 * 5 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1, 4);
        p(3) = RP2(p6 * 4 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2, 4);
        p(2) = RP2(p6 * 3 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3, 4);
        p(1) = RP2(p6 * 2 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4, 4);
        p(0) = RP2(p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5, 4);
        q(0) = RP2(p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6, 4);
        q(1) = RP2(p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 * 2, 4);
        q(2) = RP2(p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 * 3, 4);
        q(3) = RP2(p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 * 4, 4);
        q(4) = RP2(p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 * 5, 4);
        q(5) = RP2(p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 7, 4);
        return 6;
    }
    if (len >= 8 && flat && mask) {
        const int p3 = p(3), p2 = p(2), p1 = p(1), p0 = p(0), q0 = q(0), q1 = q(1), q2 = q(2), q3 = q(3);
        p(2) = RP2(p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0, 3);
        p(1) = RP2(p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1, 3);
        p(0) = RP2(p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2, 3);
        q(0) = RP2(p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3, 3);
        q(1) = RP2(p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3, 3);
        q(2) = RP2(p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3, 3);
        return 3;
    }
    if (len == 6 && flat && mask) {
        const int p2 = p(2), p1 = p(1), p0 = p(0), q0 = q(0), q1 = q(1), q2 = q(2);
        p(1) = RP2(p2 * 3 + p1 * 2 + p0 * 2 + q0, 3);
        p(0) = RP2(p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1, 3);
        q(0) = RP2(p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2, 3);
        q(1) = RP2(p0 + q0 * 2 + q1 * 2 + q2 * 3, 3);
        return 2;
    }
    {
        const int ps1 = p(1) - t80, ps0 = p(0) - t80, qs0 = q(0) - t80, qs1 = q(1) - t80;
        const bool hev = (iabs(p(1) - p(0)) > thr) | (iabs(q(1) - q(0)) > thr);
        int f = hev ? clampi(ps1 - qs1, lo, hi) : 0;
        f = mask ? clampi(f + 3 * (qs0 - ps0), lo, hi) : 0;
        const int f1 = clampi(f + 4, lo, hi) >> 3, f2 = clampi(f + 3, lo, hi) >> 3;
        q(0) = clampi(qs0 - f1, lo, hi) + t80;
        p(0) = clampi(ps0 + f2, lo, hi) + t80;
        const int f3 = hev ? 0 : ((f1 + 1) >> 1);
        q(1) = clampi(qs1 - f3, lo, hi) + t80;
        p(1) = clampi(ps1 + f3, lo, hi) + t80;
        return 2;
    }
#undef p
#undef q
#undef RP2
}
template <int BD>
__device__ __forceinline__ int lpf_core(int (&px)[14], int len, int level, int sharpness) {
    // limits of the level: update_sharpness (EbDeblockingCommon.c:587-606), hev_thr = lvl >> 4 (EbDeblockingFilter.c:38)
    int inside = level >> ((sharpness > 0) + (sharpness > 4));
    if (sharpness > 0) inside = min(inside, 9 - sharpness);
    inside = max(inside, 1);
    return lpf_core_thr<BD>(px, len, inside, 2 * (level + 2) + inside, level >> 4);
}

// ---- per-call form (include/svt_hip_rtcd.h): a list of 4-sample edge segments with explicit thresholds = svt_aom_[highbd_]lpf_{horizontal,vertical}_{4,6,8,14}
template <typename PIX, int BD>
__global__ void __launch_bounds__(64)
lpf_edge_list_kernel(PIX* __restrict__ plane, int stride, const SvtHipLpfEdge* __restrict__ jobs, int n) {
    const int e = blockIdx.x * 16 + (threadIdx.x >> 2), k = threadIdx.x & 3;
    if (e >= n) return;
    const SvtHipLpfEdge j = jobs[e];
    const int half = j.len == 4 ? 2 : (j.len == 6 ? 3 : (j.len == 8 ? 4 : 7));
    const ptrdiff_t tap = j.dir == 0 ? 1 : stride, step = j.dir == 0 ? stride : 1;   // dir 0: vertical edge (taps along x, the 4 samples along y)
    PIX* s = plane + j.off + (ptrdiff_t)k * step;
    int px[14];
#pragma unroll
    for (int t = 1; t <= 7; t++) {
        px[7 - t] = (t <= half) ? (int)s[-(ptrdiff_t)t * tap] : 0;
        px[6 + t] = (t <= half) ? (int)s[(ptrdiff_t)(t - 1) * tap] : 0;
    }
    const int changed = lpf_core_thr<BD>(px, j.len, j.limit, j.blimit, j.thresh);
#pragma unroll
    for (int t = 1; t <= 6; t++)
        if (t <= changed) { s[-(ptrdiff_t)t * tap] = (PIX)px[7 - t]; s[(ptrdiff_t)(t - 1) * tap] = (PIX)px[6 + t]; }
}


// DIR 0: vertical edges (taps along x); DIR 1: horizontal edges (taps along y)
template <typename PIX, int BD, int DIR>
__global__ void __launch_bounds__(256)
deblock_pass_kernel(PIX* __restrict__ plane, int stride, const uint16_t* __restrict__ edges, int units_w, int units_h, int sharpness,
                    int level_override) {
    int ux, uy, sx, sy;  // unit, sample position of q0
    if (DIR == 0) {
        ux = blockIdx.x * 256 + threadIdx.x;  sy = blockIdx.y;  uy = sy >> 2;  sx = 4 * ux;
        if (ux >= units_w) return;
    } else {
        sx = blockIdx.x * 256 + threadIdx.x;  uy = blockIdx.y;  ux = sx >> 2;  sy = 4 * uy;
        if (ux >= units_w) return;
    }
    const uint32_t e = edges[uy * units_w + ux];
    // level_override >= 0: frame-level probe of svt_av1_pick_filter_level (every block at the probed level, EbDeblockingFilter.c:966-1024)
    const int len = e & 0xff, level = level_override >= 0 ? level_override : (int)(e >> 8);
    if (!len || !level) return;
    const int half = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
    const ptrdiff_t tap = DIR == 0 ? 1 : stride;
    PIX* s = plane + (size_t)sy * stride + sx;
    int px[14];
#pragma unroll
    for (int k = 1; k <= 7; k++) {
        px[7 - k] = (k <= half) ? (int)s[-(ptrdiff_t)k * tap] : 0;
        px[6 + k] = (k <= half) ? (int)s[(ptrdiff_t)(k - 1) * tap] : 0;
    }
    const int changed = lpf_core<BD>(px, len, level, sharpness);
#pragma unroll
    for (int k = 1; k <= 6; k++)
        if (k <= changed) {
            s[-(ptrdiff_t)k * tap]     = (PIX)px[7 - k];
            s[(ptrdiff_t)(k - 1) * tap] = (PIX)px[6 + k];
        }
}

// all planes of a picture in one launch per direction (blockIdx.z = plane): svt_av1_loop_filter_frame(frame, pcs, 0, 3)
struct DeblockFrame { void* plane[3]; int stride[3]; const uint16_t* ev[3]; const uint16_t* eh[3]; int units_w[3], units_h[3]; };
template <typename PIX, int BD, int DIR>
__global__ void __launch_bounds__(256)
deblock_frame_pass_kernel(const DeblockFrame f, int sharpness) {
    const int p = blockIdx.z;
    // scalar copies (indexing the kernel-argument struct by reference would force a private-memory copy of it)
    PIX* plane = (PIX*)(p == 0 ? f.plane[0] : (p == 1 ? f.plane[1] : f.plane[2]));
    const int stride = p == 0 ? f.stride[0] : (p == 1 ? f.stride[1] : f.stride[2]);
    const uint16_t* edges = DIR == 0 ? (p == 0 ? f.ev[0] : (p == 1 ? f.ev[1] : f.ev[2])) : (p == 0 ? f.eh[0] : (p == 1 ? f.eh[1] : f.eh[2]));
    const int units_w = p == 0 ? f.units_w[0] : (p == 1 ? f.units_w[1] : f.units_w[2]);
    const int units_h = p == 0 ? f.units_h[0] : (p == 1 ? f.units_h[1] : f.units_h[2]);
    if (!plane || !edges) return;
    int ux, uy, sx, sy;
    if (DIR == 0) {
        ux = blockIdx.x * 256 + threadIdx.x;  sy = blockIdx.y;  uy = sy >> 2;  sx = 4 * ux;
        if (ux >= units_w || uy >= units_h) return;
    } else {
        sx = blockIdx.x * 256 + threadIdx.x;  uy = blockIdx.y;  ux = sx >> 2;  sy = 4 * uy;
        if (ux >= units_w || uy >= units_h) return;
    }
    const uint32_t e = edges[uy * units_w + ux];
    const int len = e & 0xff, level = (int)(e >> 8);
    if (!len || !level) return;
    const int half = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
    const ptrdiff_t tap = DIR == 0 ? 1 : stride;
    PIX* s = plane + (size_t)sy * stride + sx;
    int px[14];
#pragma unroll
    for (int k = 1; k <= 7; k++) {
        px[7 - k] = (k <= half) ? (int)s[-(ptrdiff_t)k * tap] : 0;
        px[6 + k] = (k <= half) ? (int)s[(ptrdiff_t)(k - 1) * tap] : 0;
    }
    const int changed = lpf_core<BD>(px, len, level, sharpness);
#pragma unroll
    for (int k = 1; k <= 6; k++)
        if (k <= changed) {
            s[-(ptrdiff_t)k * tap]     = (PIX)px[7 - k];
            s[(ptrdiff_t)(k - 1) * tap] = (PIX)px[6 + k];
        }
}
template <typename PIX, int BD>
int launch_frame(hipStream_t st, const DeblockFrame& f, int sharp) {
    int uw = 0, uh = 0;
    for (int p = 0; p < 3; p++) { if (f.plane[p]) { uw = f.units_w[p] > uw ? f.units_w[p] : uw; uh = f.units_h[p] > uh ? f.units_h[p] : uh; } }
    if (uw <= 0 || uh <= 0) return 0;
    hipLaunchKernelGGL((deblock_frame_pass_kernel<PIX, BD, 0>), dim3((uw + 255) / 256, 4 * uh, 3), dim3(256), 0, st, f, sharp);
    hipLaunchKernelGGL((deblock_frame_pass_kernel<PIX, BD, 1>), dim3((4 * uw + 255) / 256, uh, 3), dim3(256), 0, st, f, sharp);
    return (int)hipGetLastError();
}

// ---- both directions of all three planes in ONE launch, out of place (EbDeblockingFilter.c:614 loop_filter_sb does both per superblock).
// A workgroup owns a 128 x 64 tile of a plane.  Which samples decide a tile's result: a horizontal edge at row e reads the vertically filtered rows
// e - 7 .. e + 6 and changes e - 6 .. e + 5, and the tile's rows are changed by the edges y0 .. y1 (y1 = the tile below's top edge), so the horizontal
// phase needs the VERTICALLY FILTERED rows y0 - 7 .. y1 + 6 of the tile's columns; a vertical edge at column c reads the unfiltered columns c - 7 .. c + 6,
// so those rows need the unfiltered columns x0 - 7 .. x1 + 6.  The region (142 x 78 samples) is staged in LDS once, the vertical edges x0 .. x1 of all 78
// rows are filtered there (edges of one direction are independent: the filter length is bounded by the smaller transform next to the edge), then the horizontal
// edges y0 .. y1 of the tile's 128 columns, then the tile is written: 1.37 x the picture read once + the picture written once, instead of two read-modify-write
// passes.  The staged region's samples outside the tile are filtered redundantly (they are some neighbour's to write).
struct DeblockFused { const void* src[3]; void* dst[3]; int stride[3]; const uint16_t* ev[3]; const uint16_t* eh[3]; int units_w[3], units_h[3], pw[3], ph[3]; int tiles_x[3], tiles_y[3]; };
constexpr int kFW = 128, kFH = 64, kFHalo = 7, kFPad = 8, kFRH = kFH + 2 * kFHalo;   // staged columns x0 - 8 .. x0 + 135 (dword-aligned for both sample sizes)
constexpr int kFRowB = 144;   // staged samples per row
template <typename PIX, int BD>
__global__ void __launch_bounds__(256)
deblock_fused_kernel(const DeblockFused f, int sharpness) {
    // samples stay in their own type on chip (global <-> LDS moves are whole dwords); row stride = 37 dwords of 8-bit / 73 of 16-bit samples: odd, so the rows of
    // one wave's lanes start in different banks
    constexpr int SPD = 4 / (int)sizeof(PIX), RD = kFRowB / SPD, RS = (RD + 1) * SPD;   // samples per dword, dwords per staged row, row stride in samples
    __shared__ __attribute__((aligned(16))) PIX t[kFRH * RS];
    const int p = blockIdx.z;
    const PIX* src = (const PIX*)(p == 0 ? f.src[0] : (p == 1 ? f.src[1] : f.src[2]));
    PIX* dst = (PIX*)(p == 0 ? f.dst[0] : (p == 1 ? f.dst[1] : f.dst[2]));
    const int tiles_x = p == 0 ? f.tiles_x[0] : (p == 1 ? f.tiles_x[1] : f.tiles_x[2]), tiles_y = p == 0 ? f.tiles_y[0] : (p == 1 ? f.tiles_y[1] : f.tiles_y[2]);
    if (!src || (int)blockIdx.x >= tiles_x * tiles_y) return;
    const int stride = p == 0 ? f.stride[0] : (p == 1 ? f.stride[1] : f.stride[2]);
    const uint16_t* ev = p == 0 ? f.ev[0] : (p == 1 ? f.ev[1] : f.ev[2]);
    const uint16_t* eh = p == 0 ? f.eh[0] : (p == 1 ? f.eh[1] : f.eh[2]);
    const int units_w = p == 0 ? f.units_w[0] : (p == 1 ? f.units_w[1] : f.units_w[2]), units_h = p == 0 ? f.units_h[0] : (p == 1 ? f.units_h[1] : f.units_h[2]);
    const int pw = p == 0 ? f.pw[0] : (p == 1 ? f.pw[1] : f.pw[2]), ph = p == 0 ? f.ph[0] : (p == 1 ? f.ph[1] : f.ph[2]);
    const int tile = svt_xcd_order(blockIdx.x, tiles_x * tiles_y);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x, x0 = tx * kFW, y0 = ty * kFH, tid = threadIdx.x;
    const int tw = min(kFW, pw - x0), th = min(kFH, ph - y0);
    // t[r][c] <-> plane (x0 - 8 + c, y0 - 7 + r).  Whole dwords when the plane allows it (base and stride dword-aligned, the dword inside the row), else sample by sample.
    const bool fast = ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0) && ((stride * (int)sizeof(PIX)) & 3) == 0;
    {
        for (int i = tid; i < kFRH * RD; i += 256) {
            const int r = i / RD, d = i - r * RD, y = y0 - kFHalo + r, x = x0 - kFPad + d * SPD;
            uint32_t v = 0;
            if (y >= 0 && y < ph) {
                const PIX* row = src + (size_t)y * stride;
                if (fast && x >= 0 && x + SPD <= pw) v = *(const uint32_t*)(row + x);
                else {
#pragma unroll
                    for (int k = 0; k < SPD; k++)
                        if (x + k >= 0 && x + k < pw) v |= (uint32_t)row[x + k] << (8 * (int)sizeof(PIX) * k);
                }
            }
            *(uint32_t*)(t + r * RS + d * SPD) = v;
        }
    }
    __syncthreads();
    // vertical edges at columns x0 + 4k: k = 0 .. 31 by the lanes of a 32-lane group, the rows by the eight groups; k = tw / 4 (the right neighbour's first edge, which
    // changes this tile's last columns) by one more sweep over the rows
    auto vertical = [&](int r, int k) {
        const int y = y0 - kFHalo + r, x = x0 + 4 * k;
        if (y < 0 || y >= ph || x >= pw || (x >> 2) >= units_w || (y >> 2) >= units_h) return;
        const uint32_t e = ev[(y >> 2) * units_w + (x >> 2)];
        const int len = e & 0xff, level = (int)(e >> 8);
        if (!len || !level) return;
        const int half = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
        PIX* s = t + r * RS + kFPad + 4 * k;
        int px[14];
#pragma unroll
        for (int q = 1; q <= 7; q++) { px[7 - q] = (q <= half) ? (int)s[-q] : 0; px[6 + q] = (q <= half) ? (int)s[q - 1] : 0; }
        const int changed = lpf_core<BD>(px, len, level, sharpness);
#pragma unroll
        for (int q = 1; q <= 6; q++)
            if (q <= changed) { s[-q] = (PIX)px[7 - q]; s[q - 1] = (PIX)px[6 + q]; }
    };
    {
        const int rows = th + 2 * kFHalo, k = tid & 31, ne = tw >> 2;
        if (k < ne)
            for (int r = tid >> 5; r < rows; r += 8) vertical(r, k);
        if (tid < rows) vertical(tid, ne);
    }
    __syncthreads();
    // horizontal edges at rows y0 + 4k, k = 0 .. th / 4: the tile's 128 columns by the lanes, even / odd k by the two halves of the workgroup
    {
        const int c = tid & 127, ne = (th >> 2) + 1;
        if (c < tw)
            for (int k = tid >> 7; k < ne; k += 2) {
                const int y = y0 + 4 * k, x = x0 + c;
                if (y >= ph || (y >> 2) >= units_h) continue;
                const uint32_t e = eh[(y >> 2) * units_w + (x >> 2)];
                const int len = e & 0xff, level = (int)(e >> 8);
                if (!len || !level) continue;
                const int half = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
                PIX* s = t + (kFHalo + 4 * k) * RS + kFPad + c;
                int px[14];
#pragma unroll
                for (int q = 1; q <= 7; q++) { px[7 - q] = (q <= half) ? (int)s[-q * RS] : 0; px[6 + q] = (q <= half) ? (int)s[(q - 1) * RS] : 0; }
                const int changed = lpf_core<BD>(px, len, level, sharpness);
#pragma unroll
                for (int q = 1; q <= 6; q++)
                    if (q <= changed) { s[-q * RS] = (PIX)px[7 - q]; s[(q - 1) * RS] = (PIX)px[6 + q]; }
            }
    }
    __syncthreads();
    {
        constexpr int TD = kFW / SPD;   // dwords per tile row: 32 / 64
        const int d = tid % TD, x = x0 + d * SPD;
        for (int r = tid / TD; r < th; r += 256 / TD) {
            const PIX* s = t + (kFHalo + r) * RS + kFPad + d * SPD;
            PIX* o = dst + (size_t)(y0 + r) * stride + x;
            if (fast && x + SPD <= pw) *(uint32_t*)o = *(const uint32_t*)s;
            else {
#pragma unroll
                for (int k = 0; k < SPD; k++)
                    if (x + k < pw) o[k] = s[k];
            }
        }
    }
}
template <typename PIX, int BD>
int launch_fused(hipStream_t st, DeblockFused& f, int sharp) {
    int n = 0;
    for (int p = 0; p < 3; p++) {
        f.tiles_x[p] = (f.pw[p] + kFW - 1) / kFW; f.tiles_y[p] = (f.ph[p] + kFH - 1) / kFH;
        if (f.src[p]) n = f.tiles_x[p] * f.tiles_y[p] > n ? f.tiles_x[p] * f.tiles_y[p] : n;
    }
    if (n <= 0) return 0;
    hipLaunchKernelGGL((deblock_fused_kernel<PIX, BD>), dim3(n, 1, 3), dim3(256), 0, st, f, sharp);
    return (int)hipGetLastError();
}

template <typename PIX, int BD>
int launch_both(hipStream_t st, PIX* plane, int stride, const uint16_t* ev, const uint16_t* eh, int uw, int uh, int sharp, int lv_v, int lv_h) {
    if (ev) hipLaunchKernelGGL((deblock_pass_kernel<PIX, BD, 0>), dim3((uw + 255) / 256, 4 * uh), dim3(256), 0, st, plane, stride, ev, uw, uh, sharp, lv_v);
    if (eh) hipLaunchKernelGGL((deblock_pass_kernel<PIX, BD, 1>), dim3((4 * uw + 255) / 256, uh), dim3(256), 0, st, plane, stride, eh, uw, uh, sharp, lv_h);
    return (int)hipGetLastError();
}

// svt_spatial_full_distortion_kernel_c / svt_full_distortion_kernel16_bits_c (Common/Codec/EbPictureOperators.c; called by
// picture_sse_calculations, EbDeblockingFilter.c:830-961): sum of squared differences of two planes.  A workgroup owns 8 rows
// of 1024 columns; lanes read 4 consecutive samples, u32 partials (<= 8 * 4 * 1023^2 < 2^32), one u64 atomic per wave.
template <typename PIX>
__global__ void __launch_bounds__(256)
plane_sse_kernel(const PIX* __restrict__ a, int a_stride, const PIX* __restrict__ b, int b_stride, int w, int h,
                 unsigned long long* __restrict__ out) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = blockIdx.y * 8;
    uint32_t acc = 0;
    if (x < w) {
        for (int y = y0; y < min(y0 + 8, h); y++) {
            const PIX* pa = a + (size_t)y * a_stride + x;
            const PIX* pb = b + (size_t)y * b_stride + x;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (x + k < w) { const int d = (int)pa[k] - (int)pb[k]; acc += (uint32_t)(d * d); }
        }
    }
    unsigned long long v = acc;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += ((unsigned long long)(uint32_t)__shfl_xor((int)(v >> 32), m, 64) << 32) | (uint32_t)__shfl_xor((int)v, m, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// set_lpf_parameters (Encoder/Codec/EbDeblockingFilter.c:168-319) for every 4x4 unit of a picture's planes, both directions, in one launch: the device form of
// svt_hip_dlf_build_edges_crop (svt_hip_host.cpp), which stays the statement the tests compare with.  One thread per unit; blockIdx.z = plane * 2 + direction.
// lvl[plane][dir] >= 0 stands for the level of EVERY record (frame-uniform levels: the grid is uploaded once per picture and serves the level search and the
// filter itself); < 0 reads the records' own.
struct EdgeBuild {
    const SvtHipDlfModeInfo* mi;
    int mi_cols, mi_rows, ss_x, ss_y;
    int pw[3], ph[3], fw[3], fh[3], uw[3], uh[3], lvl[3][2];
    uint16_t* out[3][2];
};
__global__ void __launch_bounds__(256) dlf_build_edges_kernel(const EdgeBuild a) {
    const int plane = blockIdx.z >> 1, dir = blockIdx.z & 1;
    uint16_t* out = a.out[plane][dir];
    const int uw = a.uw[plane], uh = a.uh[plane], idx = blockIdx.x * 256 + threadIdx.x;
    if (!out || idx >= uw * uh) return;
    const int uy = idx / uw, ux = idx - uy * uw;
    uint16_t v = 0;
    if (ux < a.fw[plane] && uy < a.fh[plane]) {   // outside: beyond the reference loops' range in a padded picture, never visited
        const int ss_x = plane ? a.ss_x : 0, ss_y = plane ? a.ss_y : 0, x = 4 * ux, y = 4 * uy;
        const int mr = min(ss_y | ((y << ss_y) >> 2), a.mi_rows - 1), mc = min(ss_x | ((x << ss_x) >> 2), a.mi_cols - 1);   // chroma: the bottom / right mi of the co-located 8x8 (:196-197)
        const SvtHipDlfModeInfo cur = a.mi[mr * a.mi_cols + mc];
        const int ts = plane == 0 ? (dir == 0 ? cur.tx_w_log2 : cur.tx_h_log2) : (dir == 0 ? cur.uv_tx_w_log2 : cur.uv_tx_h_log2);
        const int coord = dir == 0 ? x : y;
        const int pr = dir == 0 ? mr : mr - (1 << ss_y), pc = dir == 0 ? mc - (1 << ss_x) : mc;
        if (!(coord & ((1 << ts) - 1)) && coord && pr >= 0 && pc >= 0) {
            const SvtHipDlfModeInfo prv = a.mi[pr * a.mi_cols + pc];
            const int pts = plane == 0 ? (dir == 0 ? prv.tx_w_log2 : prv.tx_h_log2) : (dir == 0 ? prv.uv_tx_w_log2 : prv.uv_tx_h_log2);
            const int ov = a.lvl[plane][dir], cl = ov >= 0 ? ov : cur.level[plane][dir], pl = ov >= 0 ? ov : prv.level[plane][dir];
            const int bdim = max(dir == 0 ? cur.bw_log2 - ss_x : cur.bh_log2 - ss_y, 2);
            const bool pu_edge = !(coord & ((1 << bdim) - 1));
            if ((cl || pl) && (!prv.skip_inter || !cur.skip_inter || pu_edge)) {
                const int mts = min(ts, pts);
                const int len = mts <= 2 ? 4 : (mts == 3 ? (plane ? 6 : 8) : (plane ? 6 : 14));
                v = (uint16_t)(((cl ? cl : pl) << 8) | len);
            }
        }
    }
    out[idx] = v;
}

}  // namespace

extern "C" int svt_hip_launch_lpf_edge_list(hipStream_t st, void* plane, int pix_bytes, int stride, int bd, const void* jobs, int n) {
    if (n <= 0) return 0;
    dim3 grid((n + 15) / 16);
    if (pix_bytes == 1) hipLaunchKernelGGL((lpf_edge_list_kernel<uint8_t, 8>), grid, dim3(64), 0, st, (uint8_t*)plane, stride, (const SvtHipLpfEdge*)jobs, n);
    else if (bd == 8) hipLaunchKernelGGL((lpf_edge_list_kernel<uint16_t, 8>), grid, dim3(64), 0, st, (uint16_t*)plane, stride, (const SvtHipLpfEdge*)jobs, n);
    else hipLaunchKernelGGL((lpf_edge_list_kernel<uint16_t, 10>), grid, dim3(64), 0, st, (uint16_t*)plane, stride, (const SvtHipLpfEdge*)jobs, n);
    return (int)hipGetLastError();
}


extern "C" int svt_hip_launch_deblock_plane(hipStream_t st, void* plane, int pix_bytes, int stride, int bd, const uint16_t* edges_v,
                                            const uint16_t* edges_h, int units_w, int units_h, int sharpness, int level_v, int level_h) {
    if (units_w <= 0 || units_h <= 0) return 0;
    if (pix_bytes == 1) return launch_both<uint8_t, 8>(st, (uint8_t*)plane, stride, edges_v, edges_h, units_w, units_h, sharpness, level_v, level_h);
    if (bd == 8) return launch_both<uint16_t, 8>(st, (uint16_t*)plane, stride, edges_v, edges_h, units_w, units_h, sharpness, level_v, level_h);
    return launch_both<uint16_t, 10>(st, (uint16_t*)plane, stride, edges_v, edges_h, units_w, units_h, sharpness, level_v, level_h);
}

extern "C" int svt_hip_launch_deblock_frame(hipStream_t st, void* const plane[3], int pix_bytes, const int stride[3], int bd, const uint16_t* const ev[3],
                                            const uint16_t* const eh[3], const int units_w[3], const int units_h[3], int sharpness) {
    DeblockFrame f;
    for (int p = 0; p < 3; p++) { f.plane[p] = plane[p]; f.stride[p] = stride[p]; f.ev[p] = ev[p]; f.eh[p] = eh[p]; f.units_w[p] = units_w[p]; f.units_h[p] = units_h[p]; }
    if (pix_bytes == 1) return launch_frame<uint8_t, 8>(st, f, sharpness);
    if (bd == 8) return launch_frame<uint16_t, 8>(st, f, sharpness);
    return launch_frame<uint16_t, 10>(st, f, sharpness);
}

// *out must be zero before the launch (the caller enqueues the memset)
extern "C" int svt_hip_launch_plane_sse(hipStream_t st, int pix_bytes, const void* a, int a_stride, const void* b, int b_stride, int w, int h,
                                        uint64_t* out) {
    if (w <= 0 || h <= 0) return 0;
    dim3 grid((w + 1023) / 1024, (h + 7) / 8);
    if (pix_bytes == 1) hipLaunchKernelGGL((plane_sse_kernel<uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)a, a_stride, (const uint8_t*)b, b_stride, w, h, (unsigned long long*)out);
    else hipLaunchKernelGGL((plane_sse_kernel<uint16_t>), grid, dim3(256), 0, st, (const uint16_t*)a, a_stride, (const uint16_t*)b, b_stride, w, h, (unsigned long long*)out);
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_deblock_fused(hipStream_t st, const void* const src[3], void* const dst[3], int pix_bytes, const int stride[3], int bd, const int pw[3],
                                            const int ph[3], const uint16_t* const ev[3], const uint16_t* const eh[3], const int units_w[3], const int units_h[3], int sharpness) {
    DeblockFused f;
    for (int p = 0; p < 3; p++) {
        f.src[p] = src[p]; f.dst[p] = dst[p]; f.stride[p] = stride[p]; f.ev[p] = ev[p]; f.eh[p] = eh[p]; f.units_w[p] = units_w[p]; f.units_h[p] = units_h[p];
        f.pw[p] = pw[p]; f.ph[p] = ph[p];
    }
    if (pix_bytes == 1) return launch_fused<uint8_t, 8>(st, f, sharpness);
    if (bd == 8) return launch_fused<uint16_t, 8>(st, f, sharpness);
    return launch_fused<uint16_t, 10>(st, f, sharpness);
}

SVT_HIP_TU_PROBE(deblock)

extern "C" int svt_hip_launch_dlf_build_edges(hipStream_t st, const SvtHipDlfModeInfo* mi, int mi_cols, int mi_rows, int ss_x, int ss_y, const int pw[3], const int ph[3],
                                              const int fw[3], const int fh[3], const int level[3][2], uint16_t* const ev[3], uint16_t* const eh[3]) {
    EdgeBuild a;
    a.mi = mi; a.mi_cols = mi_cols; a.mi_rows = mi_rows; a.ss_x = ss_x; a.ss_y = ss_y;
    int n = 0;
    for (int p = 0; p < 3; p++) {
        a.pw[p] = pw[p]; a.ph[p] = ph[p]; a.fw[p] = fw[p]; a.fh[p] = fh[p]; a.uw[p] = (pw[p] + 3) >> 2; a.uh[p] = (ph[p] + 3) >> 2;
        a.out[p][0] = ev[p]; a.out[p][1] = eh[p];
        for (int d = 0; d < 2; d++) a.lvl[p][d] = level ? level[p][d] : -1;
        if ((ev[p] || eh[p]) && a.uw[p] * a.uh[p] > n) n = a.uw[p] * a.uh[p];
    }
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dlf_build_edges_kernel, dim3((n + 255) / 256, 1, 6), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
