// sgr_walk.hip — the per-unit search of the self-guided restoration filter entirely on the device; gfx950.
//
// Replaces, for every (restoration unit, parameter set) of a plane (file:line under /root/reference/Source/Lib/Encoder/Codec):
//   EbRestorationPick.c:448-538  svt_get_proj_subspace_c   the 2x2 solve in IEEE double on the exact integer sums (operation order of the reference)
//   EbRestorationPick.c:539-552  encode_xq
//   EbRestorationPick.c:353-446  finer_search_pixel_proj_error   (coordinate descent; every probe = get_pixel_proj_error :317 over the unit)
//   EbRestorationPick.c:583-671  search_selfguided_restoration    (best parameter set of the unit: first set with the smallest error)
//
// Data flow.  sgr_search8_kernel<.., STORE> (sgr.hip) has left, per plane, the five projection sums of every (unit, set) and, per pixel, one 32-bit word per
// filter pair = (flt0 - u) | (flt1 - u) << 16 (int16 halves: |.| <= 4084 at bit depth 8, < 2^14.1 at 10) plus dat - src (int16).  One workgroup owns one
// (unit, set): wave 0 solves and encodes the start point, then the workgroup alternates between
//   * REPLAY (wave 0, all lanes with identical values): the reference's walk, decision by decision, on a cache of exactly evaluated points; at the
//     first unknown point it turns speculative and keeps walking on the quadratic model the five sums give (exact up to the per-pixel rounding),
//     collecting the points it visits (up to kMaxCand);
//   * EVALUATE (all lanes): one pass over the unit's planes gives the exact 64-bit error of all collected points — v_dot2_i32_i16 forms
//     xq0 (flt0 - u) + xq1 (flt1 - u) + ((dat - src) << 11 | rounding) of a pixel in one instruction (scaled by 32, so that the >> 11 of the reference is
//     "take the high half"), a second one squares and accumulates two pixels' errors.
// The result is the reference's by construction (a misprediction only costs another pass); no host round trip anywhere.  The last workgroup of a unit
// to finish picks the unit's best set.
// Measured alternatives (MI355X, 4K, 16 sets; profiles/r02/sgr_walk_notes.md): one wave per walk replaying all walks at once + separate evaluation
// launches, best-first hedging of close decisions with 16 / 32 points per pass, 1024-thread workgroups — all slower than this form, whose evaluation of
// one walk overlaps the serial replay of the three others on the same CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "svt_hip_internal.h"

namespace {

constexpr int kMaxCand = 10;     // points evaluated per pass
typedef short s16x2 __attribute__((ext_vector_type(2)));
constexpr int kCache   = 256;    // >= the longest possible walk (tap ranges 128 / 128 at step 2, plus the step-1 probes)

// eb_sgr_params (Common/Codec/EbRestoration.c:136-153): r0 > 0 for sets 0-9, 14, 15; r1 > 0 for sets 0-13.  Tap ranges: SGRPROJ_PRJ_MIN0 / MAX0 = -96 / 31,
// MIN1 / MAX1 = -32 / 95 (EbRestoration.h:100-103).  Both are spelled out as arithmetic where they are used: no table loads on the serial path.

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
    const bool has0 = ep < 10 || ep >= 14, has1 = ep < 14;   // eb_sgr_params r0 / r1 > 0
    if (!has0) { if (!(H11 < 1e-8)) xq[1] = (int)rint((C1 / H11) * 128.0); }
    else if (!has1) { if (!(H00 < 1e-8)) xq[0] = (int)rint((C0 / H00) * 128.0); }
    else {
        const double det = H00 * H11 - H01 * H10;
        if (!(det < 1e-8)) {
            const double x0 = (H11 * C0 - H01 * C1) / det, x1 = (H00 * C1 - H10 * C0) / det;
            xq[0] = (int)rint(x0 * 128.0); xq[1] = (int)rint(x1 * 128.0);
        }
    }
    if (!has0) { xqd[0] = 0; xqd[1] = clampi(128 - xq[1], -32, 95); }
    else if (!has1) { xqd[0] = clampi(xq[0], -96, 31); xqd[1] = clampi(128 - xqd[0], -32, 95); }
    else { xqd[0] = clampi(xq[0], -96, 31); xqd[1] = clampi(128 - xqd[0] - xq[1], -32, 95); }
}

// finer_search_pixel_proj_error replayed on the cache.  Returns true when the walk finished on exact errors only.
// Executed by ALL 64 lanes of wave 0 with identical values (uniform control flow): the scalar walk logic runs as before, but the two
// searches it performs over and over — "is this point in the cache?", "is it already wanted?" — compare 64 entries at a time across the
// lanes (one LDS read per lane + a ballot) instead of looping over them.  Parameter-set constants are arithmetic (no table loads).
__device__ bool replay(WalkLds& L, int ep, const int start[2], const long long* sums, int lane) {
    const bool   has0 = ep < 10 || ep >= 14, has1 = ep < 14;
    const double H00 = (double)sums[0], H01 = (double)sums[1], H11 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    auto model = [&](int x, int y) {
        const double a = has0 ? x : 0, b = !has1 ? 0 : (has0 ? 128 - x - y : 128 - y);   // svt_decode_xq
        return a * a * H00 + 2 * a * b * H01 + b * b * H11 - 256.0 * (a * C0 + b * C1);
    };
    const int n_cache = L.n_cache;
    auto lookup = [&](int x, int y, long long& e) {
        for (int base = 0; base < n_cache; base += 64) {
            const int  i = base + lane;
            const bool hit = i < n_cache && L.cx[i] == x && L.cy[i] == y;
            const unsigned long long m = __ballot(hit);
            if (m) { e = L.ce[base + __ffsll((long long)m) - 1]; return true; }
        }
        return false;
    };
    bool spec = false;
    int  nw = 0;
    // value of a point: its exact error while everything so far was cached, the model afterwards (cur = the walk's current point, whose
    // value is switched to the model at that moment so that comparisons stay like with like)
    auto value = [&](int x, int y, int curx, int cury, double& cur_err) {
        long long e;
        if (!spec && lookup(x, y, e)) return (double)e;
        if (!spec) { spec = true; cur_err = model(curx, cury); }
        if (!lookup(x, y, e)) {
            const bool dup = __ballot(lane < nw && ((volatile int*)L.wx)[lane] == x && ((volatile int*)L.wy)[lane] == y) != 0;   // kMaxCand <= 64; lane 0 wrote the list
            if (!dup && nw < kMaxCand) { if (lane == 0) { L.wx[nw] = x; L.wy[nw] = y; } nw++; }
        }
        return model(x, y);
    };
    int    q0 = start[0], q1 = start[1];
    double err = 0, err2;
    err = value(q0, q1, q0, q1, err);
    for (int s = 2; s >= 1 && nw < kMaxCand; s >>= 1) {
        for (int p = 0; p < 2 && nw < kMaxCand; p++) {
            if (p == 0 ? !has0 : !has1) continue;
            const int tmin = p == 0 ? -96 : -32, tmax = p == 0 ? 31 : 95;   // SGRPROJ_PRJ_MIN0 / MAX0, MIN1 / MAX1 (EbRestoration.h:100-103)
            bool skip = false;
            for (;;) {
                const int qp = p == 0 ? q0 : q1;
                if (qp - s >= tmin && nw < kMaxCand) {
                    const int c0 = p == 0 ? q0 - s : q0, c1 = p == 0 ? q1 : q1 - s;
                    err2 = value(c0, c1, q0, q1, err);
                    if (!(err2 > err)) { q0 = c0; q1 = c1; err = err2; skip = true; if (s == 2) continue; }
                }
                break;
            }
            if (skip) break;   // EbRestorationPick.c:406-407: leaves the parameter loop of this step size
            for (;;) {
                const int qp = p == 0 ? q0 : q1;
                if (qp + s <= tmax && nw < kMaxCand) {
                    const int c0 = p == 0 ? q0 + s : q0, c1 = p == 0 ? q1 : q1 + s;
                    err2 = value(c0, c1, q0, q1, err);
                    if (!(err2 > err)) { q0 = c0; q1 = c1; err = err2; if (s == 2) continue; }
                }
                break;
            }
        }
    }
    if (lane == 0) {
        L.n_want = nw;
        if (!spec) { L.res_x = q0; L.res_y = q1; L.res_err = (long long)err; }
    }
    return !spec;
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid: (units, 16); block 256.  pairs: [16] planes of (flt0 - u) | (flt1 - u) << 16 (plane `ep`; sets 11 / 12 / 13 read the plane of 2 / 5 / 8 — their
// xq0 is 0, so the r0 half does not matter), sd: one int16 plane of dat - src.
template <int BD>
__global__ void __launch_bounds__(256)
sgr_walk_kernel(const uint32_t* __restrict__ pairs, const int16_t* __restrict__ sd, int dstride, size_t dplane,
                const long long* __restrict__ sums, int pw, int ph, int unit_size, int units_x, int units_y, int voff, uint32_t ep_mask,
                int32_t* __restrict__ xqd_out, long long* __restrict__ err_out, uint32_t* __restrict__ counters, uint8_t* __restrict__ best_ep,
                int32_t* __restrict__ best_xqd, uint32_t* __restrict__ stats) {
    __shared__ WalkLds L;
    const int unit = blockIdx.x, ep = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!((ep_mask >> ep) & 1)) return;
    // the unit's rectangle: foreach_rest_unit_in_tile (Common/Codec/EbRestoration.c:1369-1411)
    const int uj = unit % units_x, ui = unit / units_x;
    const int x0 = uj * unit_size, w = uj == units_x - 1 ? pw - x0 : unit_size;
    const int y0 = ui * unit_size, h = ui == units_y - 1 ? ph - y0 : unit_size;
    const int v0 = max(y0 - voff, 0), v1 = (y0 + h < ph) ? y0 + h - voff : y0 + h;
    const bool has0 = ep < 10 || ep >= 14, has1 = ep < 14;
    const int  ce = ep == 11 ? 2 : (ep == 12 ? 5 : (ep == 13 ? 8 : ep));
    const uint32_t* __restrict__ PP = pairs + (size_t)ce * dplane;
    const long long* S = sums + ((size_t)unit * 16 + ep) * 5;

    int start[2] = {0, 0};
    if (wave == 0) {   // all lanes of wave 0 carry the same values
        solve_and_encode(S, w * (v1 - v0), ep, start);
        if (lane == 0) {
            L.n_cache = 0; L.done = 0; L.last = 0; L.n_want = 0;
            L.res_x = start[0]; L.res_y = start[1]; L.res_err = -1;
        }
    }
    __syncthreads();
    const int cw = (w + 7) >> 3, nchunk = cw * (v1 - v0);
    int n_pass = 0, n_eval = 0;
    for (int pass = 0; pass < 64; pass++) {
        if (wave == 0) {
            const bool fin = replay(L, ep, start, S, lane);
            if (lane == 0) L.done = fin ? 1 : 0;
            __builtin_amdgcn_wave_barrier();
            if (lane < kMaxCand) {   // svt_decode_xq (Common/Codec/EbRestoration.c:707-718); entries past n_want are not read
                const int x = L.wx[lane], y = L.wy[lane];
                L.xq0[lane] = has0 ? x : 0;
                L.xq1[lane] = !has1 ? 0 : (has0 ? 128 - x - y : 128 - y);
            }
        }
        __syncthreads();
        if (L.done) break;
        const int nc = L.n_want;
        n_pass++; n_eval += nc;
        int xq[kMaxCand];   // both taps scaled by 32 and packed for v_dot2_i32_i16 (|32 xq| <= 8192)
        long long acc[kMaxCand];
#pragma unroll
        for (int c = 0; c < kMaxCand; c++) {
            xq[c] = c < nc ? (int)(((uint32_t)(L.xq0[c] * 32) & 0xFFFFu) | ((uint32_t)(L.xq1[c] * 32) << 16)) : 0;
            acc[c] = 0;
        }
        // ---- one pass over the unit: e = ((dat - src) << 11 | rounding) + xq0 (flt0 - u) + xq1 (flt1 - u)) >> 11   (svt_av1_{lowbd,highbd}_pixel_proj_error, :174-316).
        // Everything is scaled by 32 so that the >> 11 becomes "take the high half": one v_dot2_i32_i16 forms a pixel's sum, v_perm_b32 packs the high halves
        // of two of them, a second v_dot2_i32_i16 squares and accumulates both errors.
        for (int k = tid; k < nchunk; k += 256) {
            const int row = k / cw, cx = k - row * cw;
            const size_t off = (size_t)(v0 + row) * dstride + x0 + 8 * cx;
            const int4 a0 = *(const int4*)(PP + off), a1 = *(const int4*)(PP + off + 4);
            const int4 s = *(const int4*)(sd + off);
            const int  n = min(8, w - 8 * cx);
            int pr[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bs[8];
            const int sw[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int i = 0; i < 4; i++) { bs[2 * i] = (sw[i] << 16) + 32768; bs[2 * i + 1] = (int)((uint32_t)sw[i] & 0xFFFF0000u) + 32768; }   // 32 x (((dat - src) << 11) + 2^10)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i >= n) { pr[i] = 0; bs[i] = 0; }   // columns past the unit: e = 0
#pragma unroll
            for (int c = 0; c < kMaxCand; c++) {
                if (c < nc) {   // workgroup-uniform
                    const s16x2 q = __builtin_bit_cast(s16x2, xq[c]);
                    int p = 0;
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        const int t0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pr[i]), q, bs[i], false);
                        const int t1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pr[i + 1]), q, bs[i + 1], false);
                        const int ee = (int)__builtin_amdgcn_perm((uint32_t)t1, (uint32_t)t0, 0x07060302u);   // (t0 >> 16) | (t1 & 0xffff0000): the two errors, int16 each
                        p = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, ee), __builtin_bit_cast(s16x2, ee), p, false);   // |e| < 2^13 at bit depth 10: eight squares stay below 2^31
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
        atomicAdd(&stats[0], (uint32_t)n_pass); atomicAdd(&stats[1], (uint32_t)n_eval); if (!L.done) atomicAdd(&stats[2], 1u);   // diagnostics
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

extern "C" size_t svt_hip_sgr_walk_state_bytes(int n_units) { return sizeof(uint32_t) * (size_t)n_units; }   // arrival counter per unit

extern "C" int svt_hip_launch_sgr_walk(hipStream_t st, int bd, const uint32_t* pairs, const int16_t* sd, int dstride, size_t dplane, const int64_t* sums,
                                       const int64_t* d2, void* states, int pw, int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask,
                                       int32_t* xqd_out, int64_t* err_out, uint8_t* best_ep, int32_t* best_xqd, uint32_t* stats) {
    (void)d2;
    const int voff = 8 >> ss_y;
    dim3 grid(units_x * units_y, 16);
    uint32_t* counters = (uint32_t*)states;
    (void)hipMemsetAsync(counters, 0, sizeof(uint32_t) * (size_t)units_x * units_y, st);
    // tuning knob (tools/hbd_time.py): unused dynamic LDS per workgroup limits how many (unit, set) walks are in flight
    static const int lds_pad = getenv("SVT_HIP_SGR_WALK_LDS_PAD") ? atoi(getenv("SVT_HIP_SGR_WALK_LDS_PAD")) * 1024 : 0;
    if (bd == 8)
        hipLaunchKernelGGL((sgr_walk_kernel<8>), grid, dim3(256), lds_pad, st, pairs, sd, dstride, dplane, (const long long*)sums, pw, ph, unit_size, units_x, units_y, voff,
                           ep_mask, xqd_out, (long long*)err_out, counters, best_ep, best_xqd, stats);
    else
        hipLaunchKernelGGL((sgr_walk_kernel<10>), grid, dim3(256), lds_pad, st, pairs, sd, dstride, dplane, (const long long*)sums, pw, ph, unit_size, units_x, units_y, voff,
                           ep_mask, xqd_out, (long long*)err_out, counters, best_ep, best_xqd, stats);
    return (int)hipGetLastError();
}
