// sgr_walk.hip — the per-unit search of the self-guided restoration filter entirely on the device; gfx950.
//
// Replaces, for every (restoration unit, parameter set) of a plane (file:line under /root/reference/Source/Lib/Encoder/Codec):
//   EbRestorationPick.c:448-538  svt_get_proj_subspace_c   the 2x2 solve in IEEE double on the exact integer sums (operation order of the reference)
//   EbRestorationPick.c:539-552  encode_xq
//   EbRestorationPick.c:353-446  finer_search_pixel_proj_error   (coordinate descent; every probe = get_pixel_proj_error :317 over the unit)
//   EbRestorationPick.c:583-671  search_selfguided_restoration    (best parameter set of the unit: first set with the smallest error)
//
// Data flow.  sgr_search8_kernel<.., STORE> (sgr.hip) has left, per plane, the five projection sums of every (unit, set) and three kinds of
// int16 planes: flt0 - u per r0-filter, flt1 - u per r1-filter (|.| <= 4084 at bit depth 8, < 2^14.1 at 10) and dat - src.  One workgroup owns one
// (unit, set): lane 0 solves and encodes the start point, then the workgroup alternates between
//   * REPLAY (lane 0): the reference's walk, decision by decision, on a cache of exactly evaluated points; at the first unknown point it
//     turns speculative and keeps walking on the quadratic model the five sums give (exact up to the per-pixel rounding), collecting the
//     points it visits (up to kMaxCand);
//   * EVALUATE (all lanes): one pass over the unit's three int16 planes gives the exact 64-bit error of all collected points.
// The result is the reference's by construction (a misprediction only costs another pass); no host round trip anywhere.  The last
// workgroup of a unit to finish picks the unit's best set.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"

namespace {

constexpr int kMaxCand = 10;     // points evaluated per pass
constexpr int kCache   = 256;    // >= the longest possible walk (tap ranges 128 / 128 at step 2, plus the step-1 probes)

__device__ __constant__ int kR[16][2] = {{2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {0, 1}, {0, 1}, {0, 1}, {0, 1}, {2, 0}, {2, 0}};
__device__ __constant__ int kTapMin[2] = {-96, -32}, kTapMax[2] = {31, 95};   // SGRPROJ_PRJ_MIN0 / MAX0, MIN1 / MAX1 (EbRestoration.h:100-103)

struct WalkLds {
    int       cx[kCache], cy[kCache];   // evaluated points
    long long ce[kCache];               // and their exact errors
    int       n_cache;
    int       wx[kMaxCand], wy[kMaxCand], n_want;
    int       xq0[kMaxCand], xq1[kMaxCand];
    long long part[4][kMaxCand];        // per-wave partial sums
    int       done, res_x, res_y;
    long long res_err;
    int       last;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// svt_get_proj_subspace_c's solve + encode_xq.  Every operation is an IEEE double operation in the reference's order (the library is built
// with -ffp-contract=off; AMDGPU's f64 division is correctly rounded, v_rndne_f64 is rint() in the default rounding mode).
__device__ void solve_and_encode(const long long* sums, int size, int ep, int xqd[2]) {
    double H00 = (double)sums[0], H01 = (double)sums[1], H11 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    const double dsize = (double)size;
    H00 /= dsize; H01 /= dsize; H11 /= dsize; C0 /= dsize; C1 /= dsize;
    const double H10 = H01;
    int xq[2] = {0, 0};
    if (kR[ep][0] == 0) { if (!(H11 < 1e-8)) xq[1] = (int)rint((C1 / H11) * 128.0); }
    else if (kR[ep][1] == 0) { if (!(H00 < 1e-8)) xq[0] = (int)rint((C0 / H00) * 128.0); }
    else {
        const double det = H00 * H11 - H01 * H10;
        if (!(det < 1e-8)) {
            const double x0 = (H11 * C0 - H01 * C1) / det, x1 = (H00 * C1 - H10 * C0) / det;
            xq[0] = (int)rint(x0 * 128.0); xq[1] = (int)rint(x1 * 128.0);
        }
    }
    if (kR[ep][0] == 0) { xqd[0] = 0; xqd[1] = clampi(128 - xq[1], kTapMin[1], kTapMax[1]); }
    else if (kR[ep][1] == 0) { xqd[0] = clampi(xq[0], kTapMin[0], kTapMax[0]); xqd[1] = clampi(128 - xqd[0], kTapMin[1], kTapMax[1]); }
    else { xqd[0] = clampi(xq[0], kTapMin[0], kTapMax[0]); xqd[1] = clampi(128 - xqd[0] - xq[1], kTapMin[1], kTapMax[1]); }
}

// finer_search_pixel_proj_error replayed on the cache.  Returns true when the walk finished on exact errors only.
__device__ bool replay(WalkLds& L, int ep, const int start[2], const long long* sums) {
    const bool   has0 = kR[ep][0] > 0, has1 = kR[ep][1] > 0;
    const double H00 = (double)sums[0], H01 = (double)sums[1], H11 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    auto model = [&](int x, int y) {
        const double a = has0 ? x : 0, b = !has1 ? 0 : (has0 ? 128 - x - y : 128 - y);   // svt_decode_xq
        return a * a * H00 + 2 * a * b * H01 + b * b * H11 - 256.0 * (a * C0 + b * C1);
    };
    auto lookup = [&](int x, int y, long long& e) {
        for (int i = 0; i < L.n_cache; i++)
            if (L.cx[i] == x && L.cy[i] == y) { e = L.ce[i]; return true; }
        return false;
    };
    bool spec = false;
    int  nw = 0;
    // value of a point: its exact error while everything so far was cached, the model afterwards (cur = the walk's current point, whose
    // value is switched to the model at that moment so that comparisons stay like with like)
    auto value = [&](int x, int y, const int cur[2], double& cur_err) {
        long long e;
        if (!spec && lookup(x, y, e)) return (double)e;
        if (!spec) { spec = true; cur_err = model(cur[0], cur[1]); }
        if (!lookup(x, y, e)) {
            bool dup = false;
            for (int i = 0; i < nw; i++) dup = dup || (L.wx[i] == x && L.wy[i] == y);
            if (!dup && nw < kMaxCand) { L.wx[nw] = x; L.wy[nw] = y; nw++; }
        }
        return model(x, y);
    };
    int    q[2] = {start[0], start[1]};
    double err = 0, err2;
    err = value(q[0], q[1], q, err);
    for (int s = 2; s >= 1 && nw < kMaxCand; s >>= 1) {
        for (int p = 0; p < 2 && nw < kMaxCand; p++) {
            if (kR[ep][p] == 0) continue;
            bool skip = false;
            for (;;) {
                if (q[p] - s >= kTapMin[p] && nw < kMaxCand) {
                    int c[2] = {q[0], q[1]}; c[p] -= s;
                    err2 = value(c[0], c[1], q, err);
                    if (!(err2 > err)) { q[p] -= s; err = err2; skip = true; if (s == 2) continue; }
                }
                break;
            }
            if (skip) break;   // EbRestorationPick.c:406-407: leaves the parameter loop of this step size
            for (;;) {
                if (q[p] + s <= kTapMax[p] && nw < kMaxCand) {
                    int c[2] = {q[0], q[1]}; c[p] += s;
                    err2 = value(c[0], c[1], q, err);
                    if (!(err2 > err)) { q[p] += s; err = err2; if (s == 2) continue; }
                }
                break;
            }
        }
    }
    L.n_want = nw;
    if (spec) return false;
    L.res_x = q[0]; L.res_y = q[1]; L.res_err = (long long)err;
    return true;
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid: (units, 16); block 256.  diff0 / diff1: [16] planes each (plane `ep`; sets 11 / 12 / 13 read the r1 plane of 2 / 5 / 8), sd: one plane.
template <int BD>
__global__ void __launch_bounds__(256)
sgr_walk_kernel(const int16_t* __restrict__ diff0, const int16_t* __restrict__ diff1, const int16_t* __restrict__ sd, int dstride, size_t dplane,
                const long long* __restrict__ sums, int pw, int ph, int unit_size, int units_x, int units_y, int voff, uint32_t ep_mask,
                int32_t* __restrict__ xqd_out, long long* __restrict__ err_out, uint32_t* __restrict__ counters, uint8_t* __restrict__ best_ep,
                int32_t* __restrict__ best_xqd) {
    __shared__ WalkLds L;
    const int unit = blockIdx.x, ep = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!((ep_mask >> ep) & 1)) return;
    // the unit's rectangle: foreach_rest_unit_in_tile (Common/Codec/EbRestoration.c:1369-1411)
    const int uj = unit % units_x, ui = unit / units_x;
    const int x0 = uj * unit_size, w = uj == units_x - 1 ? pw - x0 : unit_size;
    const int y0 = ui * unit_size, h = ui == units_y - 1 ? ph - y0 : unit_size;
    const int v0 = max(y0 - voff, 0), v1 = (y0 + h < ph) ? y0 + h - voff : y0 + h;
    const bool has0 = kR[ep][0] > 0, has1 = kR[ep][1] > 0;
    const int  ce = ep == 11 ? 2 : (ep == 12 ? 5 : (ep == 13 ? 8 : ep));
    const int16_t* __restrict__ D0 = diff0 + (size_t)ep * dplane;
    const int16_t* __restrict__ D1 = diff1 + (size_t)ce * dplane;
    const long long* S = sums + ((size_t)unit * 16 + ep) * 5;

    int start[2];
    if (tid == 0) {
        solve_and_encode(S, w * (v1 - v0), ep, start);
        L.n_cache = 0; L.done = 0; L.last = 0; L.n_want = 0;
        L.res_x = start[0]; L.res_y = start[1]; L.res_err = -1;
    }
    __syncthreads();
    const int cw = (w + 7) >> 3, nchunk = cw * (v1 - v0);
    for (int pass = 0; pass < 64; pass++) {
        if (tid == 0) {
            L.done = replay(L, ep, start, S) ? 1 : 0;
            for (int c = 0; c < L.n_want; c++) {   // svt_decode_xq (Common/Codec/EbRestoration.c:707-718)
                L.xq0[c] = has0 ? L.wx[c] : 0;
                L.xq1[c] = !has1 ? 0 : (has0 ? 128 - L.wx[c] - L.wy[c] : 128 - L.wy[c]);
            }
        }
        __syncthreads();
        if (L.done) break;
        const int nc = L.n_want;
        int xq0[kMaxCand], xq1[kMaxCand];
        long long acc[kMaxCand];
#pragma unroll
        for (int c = 0; c < kMaxCand; c++) { xq0[c] = c < nc ? L.xq0[c] : 0; xq1[c] = c < nc ? L.xq1[c] : 0; acc[c] = 0; }
        // ---- one pass over the unit: e = ((dat - src) << 11 | rounding) + xq0 (flt0 - u) + xq1 (flt1 - u)) >> 11   (svt_av1_{lowbd,highbd}_pixel_proj_error, :174-316)
        for (int k = tid; k < nchunk; k += 256) {
            const int row = k / cw, cx = k - row * cw;
            const size_t off = (size_t)(v0 + row) * dstride + x0 + 8 * cx;
            const int4 a = has0 ? *(const int4*)(D0 + off) : make_int4(0, 0, 0, 0);
            const int4 b = has1 ? *(const int4*)(D1 + off) : make_int4(0, 0, 0, 0);
            const int4 s = *(const int4*)(sd + off);
            const int  n = min(8, w - 8 * cx);
            int d0[8], d1[8], bs[8];
            const int aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, sw[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                d0[2 * i] = (int)(int16_t)aw[i]; d0[2 * i + 1] = aw[i] >> 16;
                d1[2 * i] = (int)(int16_t)bw[i]; d1[2 * i + 1] = bw[i] >> 16;
                bs[2 * i] = ((int)(int16_t)sw[i] << 11) + 1024; bs[2 * i + 1] = ((sw[i] >> 16) << 11) + 1024;
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i >= n) { d0[i] = 0; d1[i] = 0; bs[i] = 0; }   // columns past the unit: e = 0
#pragma unroll
            for (int c = 0; c < kMaxCand; c++) {
                if (c < nc) {   // workgroup-uniform
                    int p = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int e = (bs[i] + __mul24(xq0[c], d0[i]) + __mul24(xq1[c], d1[i])) >> 11;
                        p += __mul24(e, e);   // |e| < 2^13 at bit depth 10: eight squares stay below 2^31
                    }
                    acc[c] += p;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < kMaxCand; c++)
            if (c < nc) {
                const long long t = wave_sum_i64(acc[c]);
                if (lane == 0) L.part[wave][c] = t;
            }
        __syncthreads();
        if (tid == 0) {
            for (int c = 0; c < nc && L.n_cache < kCache; c++) {
                L.cx[L.n_cache] = L.wx[c]; L.cy[L.n_cache] = L.wy[c];
                L.ce[L.n_cache] = L.part[0][c] + L.part[1][c] + L.part[2][c] + L.part[3][c];
                L.n_cache++;
            }
        }
        __syncthreads();
    }
    // ---- results, and the unit's best set once all of its sets are in: search_selfguided_restoration :661-665 (first set with the smallest error)
    if (tid == 0) {
        xqd_out[((size_t)unit * 16 + ep) * 2] = L.res_x;
        xqd_out[((size_t)unit * 16 + ep) * 2 + 1] = L.res_y;
        err_out[(size_t)unit * 16 + ep] = L.done ? L.res_err : -1;   // -1: walk not finished within the pass budget (never observed; callers treat it as a failure)
        __threadfence();
        const uint32_t arrived = atomicAdd(&counters[unit], 1u) + 1u;
        if (arrived == (uint32_t)__popc(ep_mask)) {
            __threadfence();
            int be = -1; long long berr = -1;
            for (int e2 = 0; e2 < 16; e2++) {
                if (!((ep_mask >> e2) & 1)) continue;
                const long long v = ((volatile long long*)err_out)[(size_t)unit * 16 + e2];
                if (be < 0 || v < berr) { be = e2; berr = v; }
            }
            if (best_ep) best_ep[unit] = (uint8_t)be;
            if (best_xqd) {
                best_xqd[2 * unit] = ((volatile int32_t*)xqd_out)[((size_t)unit * 16 + be) * 2];
                best_xqd[2 * unit + 1] = ((volatile int32_t*)xqd_out)[((size_t)unit * 16 + be) * 2 + 1];
            }
        }
    }
}

}  // namespace

extern "C" int svt_hip_launch_sgr_walk(hipStream_t st, int bd, const int16_t* diff0, const int16_t* diff1, const int16_t* sd, int dstride, size_t dplane,
                                       const int64_t* sums, int pw, int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask, int32_t* xqd_out,
                                       int64_t* err_out, uint32_t* counters, uint8_t* best_ep, int32_t* best_xqd) {
    const int voff = 8 >> ss_y;
    dim3 grid(units_x * units_y, 16);
    if (bd == 8)
        hipLaunchKernelGGL((sgr_walk_kernel<8>), grid, dim3(256), 0, st, diff0, diff1, sd, dstride, dplane, (const long long*)sums, pw, ph, unit_size, units_x, units_y, voff,
                           ep_mask, xqd_out, (long long*)err_out, counters, best_ep, best_xqd);
    else
        hipLaunchKernelGGL((sgr_walk_kernel<10>), grid, dim3(256), 0, st, diff0, diff1, sd, dstride, dplane, (const long long*)sums, pw, ph, unit_size, units_x, units_y, voff,
                           ep_mask, xqd_out, (long long*)err_out, counters, best_ep, best_xqd);
    return (int)hipGetLastError();
}
