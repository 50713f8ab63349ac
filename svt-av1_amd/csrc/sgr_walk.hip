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
#include <string.h>
#include "svt_hip_internal.h"

#ifdef SVT_SGR_NT
typedef int sgr_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 sgr_ld4_nt(const void* p) { const sgr_v4i v = __builtin_nontemporal_load((const sgr_v4i*)p); return make_int4(v.x, v.y, v.z, v.w); }
#define SGR_LD4(p) sgr_ld4_nt(p)
#else
#define SGR_LD4(p) (*(const int4*)(p))
#endif

namespace {

constexpr int kStreamCand = 10;  // the streamed form keeps one accumulator per candidate in registers
constexpr int kMaxCand = 16;     // capacity of the per-pass candidate list (the launch's `cap` <= this is the number actually used)
typedef short s16x2 __attribute__((ext_vector_type(2)));
struct WalkPlane {   // one plane's arguments of a walk launch
    const uint32_t* pairs; const int16_t* sd; const int64_t* sums; size_t dplane;
    int dstride, pw, ph, unit_size, units_x, units_y, voff; uint32_t ep_mask;
    int32_t* xqd_out; int64_t* err_out; uint32_t* counters; uint8_t* best_ep; int32_t* best_xqd; uint32_t* stats;
    const uint2* esc; const uint32_t* esc_cnt;   // packed form only (sgr_walk_packed_kernel): the samples whose differences do not fit the packed word, per (unit, set)
};
constexpr int kWalkMaxPlanes = 12;   // the planes of up to four pictures share a launch (grid.z)
struct WalkPic { WalkPlane p[kWalkMaxPlanes]; int cap, clocks, hist_w; };   // hist_w: largest |flt - u| the histogram evaluation takes (<= the instance's kHistW; tests narrow it to reach both paths)   // clocks: also accumulate the walks' phase clocks (diagnostics; SVT_HIP_SGR_WALK_CLOCKS=1 switches them on)
constexpr int kCache   = 256;    // evaluated points a walk can remember, see kThrottle
// The exact walk (finer_search_pixel_proj_error, EbRestorationPick.c:353-440) evaluates at most 1 + 2 x (1 + 63) + 4 = 133 points: per parameter at step 2 one rejected
// downward probe and then <= 63 upward ones (or <= 63 downward ones), at step 1 two probes per parameter.  Speculative requests (points the quadratic model walks
// to but the exact walk does not) share the cache; on content where the model predicts nothing (binary 0 / max pictures: rounding and clamping dominate) nearly every
// request beyond the first of a pass is wasted.  Once kThrottle points are cached a pass asks for ONE point -- the first unknown point, which is always on the exact
// path -- so the cache holds <= kThrottle + kMaxCand - 1 + 133 <= 244 points and a walk ends within kThrottle + 133 passes <= kPassBudget: every walk finishes.
// (Round 3 had 64 passes and no throttle: the extreme-content test of round 4 did not finish.)
constexpr int kThrottle = 96, kPassBudget = 256;

// eb_sgr_params (Common/Codec/EbRestoration.c:136-153): r0 > 0 for sets 0-9, 14, 15; r1 > 0 for sets 0-13.  Tap ranges: SGRPROJ_PRJ_MIN0 / MAX0 = -96 / 31,
// MIN1 / MAX1 = -32 / 95 (EbRestoration.h:100-103).  Both are spelled out as arithmetic where they are used: no table loads on the serial path.

struct WalkLds {
    int       cx[kCache], cy[kCache];   // evaluated points
    long long ce[kCache];               // and their exact errors
    int       n_cache;
    int       wx[kMaxCand], wy[kMaxCand], n_want;
    int       xq0[kMaxCand], xq1[kMaxCand];
    long long part[16][kMaxCand];       // per-wave partial sums
    int       done, res_x, res_y;
    long long res_err;
    int       last;
    int       ovf_n;                    // histogram evaluation: samples outside the histogram's window so far (the first kCache of them are listed in cx / cy)
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
// lanes (a ballot) instead of looping over them.  Parameter-set constants are arithmetic (no table loads).  The store of evaluated and wanted
// points is a policy: LdsStore keeps them in the workgroup's LDS (streamed form), RegStore in the registers of the walking wave (resident form).
struct LdsStore {
    WalkLds& L; int lane, n_cache, nw, cap;
    __device__ LdsStore(WalkLds& l, int ln, int cp) : L(l), lane(ln), n_cache(l.n_cache), nw(0), cap(cp) {}
    __device__ bool lookup(int x, int y, long long& e) const {
        for (int base = 0; base < n_cache; base += 64) {
            const int  i = base + lane;
            const bool hit = i < n_cache && L.cx[i] == x && L.cy[i] == y;
            const unsigned long long m = __ballot(hit);
            if (m) { e = L.ce[base + __ffsll((long long)m) - 1]; return true; }
        }
        return false;
    }
    __device__ void want(int x, int y) {
        const bool dup = __ballot(lane < nw && ((volatile int*)L.wx)[lane] == x && ((volatile int*)L.wy)[lane] == y) != 0;   // kMaxCand <= 64; lane 0 wrote the list
        if (!dup && nw < cap) { if (lane == 0) { L.wx[nw] = x; L.wy[nw] = y; } nw++; }
    }
    __device__ void finish(bool exact, int q0, int q1, double err) {
        if (lane == 0) {
            L.n_want = nw;
            if (exact) { L.res_x = q0; L.res_y = q1; L.res_err = (long long)err; }
        }
    }
};
// lane i of bank b holds evaluated point 64 b + i: a lookup is one compare + ballot + two v_readlane per bank in use, no memory on the serial path
constexpr int kBanks = kCache / 64;
__device__ __forceinline__ int point_key(int x, int y) { return (x + 128) | ((y + 128) << 8); }   // taps lie in [-96, 95]
struct RegStore {
    int key[kBanks]; long long err[kBanks];   // the cache, one entry per lane and bank
    int wkey;                                   // lane i: wanted point i
    int lane, n_cache, nw, cap;
    int res_x, res_y; long long res_err;
    __device__ bool lookup(int x, int y, long long& e) const {
        const int k = point_key(x, y);
        bool found = false;
#pragma unroll
        for (int b = 0; b < kBanks; b++) {   // no early exit: the bank index must stay a compile-time constant (registers, not scratch)
            if (!found && 64 * b < n_cache) {
                const unsigned long long m = __ballot(64 * b + lane < n_cache && key[b] == k);
                if (m) {
                    const int src = __ffsll((long long)m) - 1;
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)err[b], src);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)err[b] >> 32), src);
                    e = (long long)(((unsigned long long)hi << 32) | lo);
                    found = true;
                }
            }
        }
        return found;
    }
    __device__ void want(int x, int y) {
        const int k = point_key(x, y);
        const bool dup = __ballot(lane < nw && wkey == k) != 0;
        if (!dup && nw < cap) { if (lane == nw) wkey = k; nw++; }
    }
    __device__ void finish(bool exact, int q0, int q1, double e) {
        if (exact) { res_x = q0; res_y = q1; res_err = (long long)e; }
    }
};

// The walk as a state machine, so that a replay can RESUME where the previous one turned speculative instead of re-deciding the exact prefix:
//   s = step (2, then 1), p = parameter (0, 1; 2 = this step size is finished), up = 0 probing q - s / 1 probing q + s, (q0, q1) = current point,
//   err = its error, moved = "the downward probe of this parameter was accepted" (EbRestorationPick.c:406-407 then leaves the parameter loop).
struct WalkState { int s, p, up, q0, q1, moved, have_err; double err; };
__device__ __forceinline__ void walk_begin(WalkState& W, const int start[2]) { W.s = 2; W.p = 0; W.up = 0; W.q0 = start[0]; W.q1 = start[1]; W.moved = 0; W.have_err = 0; W.err = 0; }

// One replay: continues from W on exact errors; at the first unknown point it freezes W there (the next replay resumes from it) and keeps walking on
// the quadratic model, collecting up to K.cap unknown points.  Returns true when the walk finished on exact errors only (W.q0, W.q1, W.err = result).
struct ModelSums { double H00, H01, H11, C0, C1; };   // the five projection sums of the (unit, set), read from memory once per walk
__device__ __forceinline__ ModelSums load_model(const long long* sums) { return ModelSums{(double)sums[0], (double)sums[1], (double)sums[2], (double)sums[3], (double)sums[4]}; }
// HEDGE (opt-in instance): what the OTHER outcome of a speculative decision would probe next is asked for as well.  A pass costs ~44 k cycles whatever it evaluates
// (it streams the part of the unit that is not resident) and a point ~2 k (profiles/r03/sgr_walk_cand_sweep.txt), while half of the decisions near the optimum are
// decided by rounding noise the model cannot see (tools/sgr_walk_sim.c: 2.97 -> 2.20 passes per walk with 16 points per pass).  These two helpers restate the walk's
// state transitions for that look-ahead only; whatever they return changes which points are evaluated, never the result.
__device__ __forceinline__ void walk_apply(WalkState& A, bool accept, int c0, int c1) {
    bool again = false;
    if (accept) { A.q0 = c0; A.q1 = c1; if (!A.up) A.moved = 1; again = A.s == 2; }
    if (!again) {
        if (!A.up) { if (A.moved) A.p = 2; else A.up = 1; }
        else { A.p++; A.up = 0; A.moved = 0; }
    }
}
__device__ __forceinline__ bool walk_peek(WalkState A, bool has0, bool has1, int& c0, int& c1) {
    for (int guard = 0; guard < 12 && A.s >= 1; guard++) {
        if (A.p >= 2) { A.s >>= 1; A.p = 0; A.up = 0; A.moved = 0; continue; }
        if (A.p == 0 ? !has0 : !has1) { A.p++; A.up = 0; A.moved = 0; continue; }
        const int tmin = A.p == 0 ? -96 : -32, tmax = A.p == 0 ? 31 : 95;
        const int qp = A.p == 0 ? A.q0 : A.q1, d = A.up ? A.s : -A.s;
        if (A.up ? qp + A.s <= tmax : qp - A.s >= tmin) { c0 = A.p == 0 ? A.q0 + d : A.q0; c1 = A.p == 0 ? A.q1 : A.q1 + d; return true; }
        walk_apply(A, false, 0, 0);
    }
    return false;
}
template <class Store, bool HEDGE = false>
__device__ bool replay(Store& K, WalkState& W, int ep, const ModelSums& MS) {
    const bool   has0 = ep < 10 || ep >= 14, has1 = ep < 14;
    const double H00 = MS.H00, H01 = MS.H01, H11 = MS.H11, C0 = MS.C0, C1 = MS.C1;
    auto model = [&](int x, int y) {
        const double a = has0 ? x : 0, b = !has1 ? 0 : (has0 ? 128 - x - y : 128 - y);   // svt_decode_xq
        return a * a * H00 + 2 * a * b * H01 + b * b * H11 - 256.0 * (a * C0 + b * C1);
    };
    bool spec = false;
    K.nw = 0;
    WalkState T = W;   // the running state; W follows it while everything is exact
    // value of a point: its exact error while everything so far was cached, the model afterwards (the current point's
    // value is switched to the model at that moment so that comparisons stay like with like)
    auto value = [&](int x, int y) {   // one cache lookup per probe
        long long e;
        const bool hit = K.lookup(x, y, e);
        if (!spec) {
            if (hit) return (double)e;
            spec = true; T.err = model(T.q0, T.q1);
        }
        if (!hit) K.want(x, y);
        return model(x, y);
    };
    if (!T.have_err) {
        T.err = value(T.q0, T.q1);
        T.have_err = 1;
        if (!spec) W = T;
    }
    while (T.s >= 1 && K.nw < K.cap) {
        if (T.p >= 2) { T.s >>= 1; T.p = 0; T.up = 0; T.moved = 0; if (!spec) W = T; continue; }
        if (T.p == 0 ? !has0 : !has1) { T.p++; T.up = 0; T.moved = 0; if (!spec) W = T; continue; }
        const int tmin = T.p == 0 ? -96 : -32, tmax = T.p == 0 ? 31 : 95;   // SGRPROJ_PRJ_MIN0 / MAX0, MIN1 / MAX1 (EbRestoration.h:100-103)
        const int qp = T.p == 0 ? T.q0 : T.q1, d = T.up ? T.s : -T.s;
        bool again = false;
        if (T.up ? qp + T.s <= tmax : qp - T.s >= tmin) {
            const int c0 = T.p == 0 ? T.q0 + d : T.q0, c1 = T.p == 0 ? T.q1 : T.q1 + d;
            const double err2 = value(c0, c1);
            if (HEDGE && spec && K.nw < K.cap) {   // the other outcome's next probe
                WalkState A = T;
                walk_apply(A, err2 > T.err, c0, c1);
                int a0, a1;
                long long e;
                if (walk_peek(A, has0, has1, a0, a1) && !K.lookup(a0, a1, e)) K.want(a0, a1);
            }
            if (!(err2 > T.err)) { T.q0 = c0; T.q1 = c1; T.err = err2; if (!T.up) T.moved = 1; again = T.s == 2; }
        }
        if (!again) {
            if (!T.up) { if (T.moved) T.p = 2; else T.up = 1; }   // an accepted downward probe ends this step size (the reference's `if (skip) break`)
            else { T.p++; T.up = 0; T.moved = 0; }
        }
        if (!spec) W = T;   // still exact: this decision is final
    }
    K.finish(!spec, T.q0, T.q1, T.err);
    return !spec;
}

// sum of a non-negative value below 2^48 over the wave, without LDS traffic: two 24-bit limbs, each reduced with four DPP adds (within a row of 16
// lanes) and four v_readlane.  Uniform result.
__device__ __forceinline__ int row_sum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1 0 3 2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2 3 0 1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror
    return v;
}
__device__ __forceinline__ long long wave_sum_u48(long long v) {
    const int lo = row_sum_dpp((int)(v & 0xFFFFFF)), hi = row_sum_dpp((int)(v >> 24));
    const long long slo = (long long)__builtin_amdgcn_readlane(lo, 0) + __builtin_amdgcn_readlane(lo, 16) + __builtin_amdgcn_readlane(lo, 32) + __builtin_amdgcn_readlane(lo, 48);
    const long long shi = (long long)__builtin_amdgcn_readlane(hi, 0) + __builtin_amdgcn_readlane(hi, 16) + __builtin_amdgcn_readlane(hi, 32) + __builtin_amdgcn_readlane(hi, 48);
    return slo + (shi << 24);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid: (units, 16); block 256.  pairs: [16] planes of (flt0 - u) | (flt1 - u) << 16 (plane `ep`; sets 11 / 12 / 13 read the plane of 2 / 5 / 8 — their
// xq0 is 0, so the r0 half does not matter), sd: one int16 plane of dat - src.
// Publishing a walk's result to the workgroup that finishes the unit: write-through (agent-scope) stores, drained, then the arrival count; the reader uses
// agent-scope loads.  A __threadfence() pair here meant an L2 write-back + invalidate per walk (thousands per picture), which also cost the kernels of other
// frames running beside this one.
__device__ __forceinline__ void publish_walk(int32_t* xqd_out, long long* err_out, size_t ue, int rx, int ry, long long err) {
    __hip_atomic_store(&xqd_out[ue * 2], rx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&xqd_out[ue * 2 + 1], ry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&err_out[ue], err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void pick_unit_best(const int32_t* xqd_out, const long long* err_out, int unit, uint32_t ep_mask, uint8_t* best_ep, int32_t* best_xqd) {
    int be = -1; long long berr = -1;
    for (int e2 = 0; e2 < 16; e2++) {
        if (!((ep_mask >> e2) & 1)) continue;
        const long long v = __hip_atomic_load(&err_out[(size_t)unit * 16 + e2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (be < 0 || v < berr) { be = e2; berr = v; }
    }
    if (best_ep) best_ep[unit] = (uint8_t)be;
    if (best_xqd) {
        best_xqd[2 * unit] = __hip_atomic_load(&xqd_out[((size_t)unit * 16 + be) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        best_xqd[2 * unit + 1] = __hip_atomic_load(&xqd_out[((size_t)unit * 16 + be) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int BD>
__global__ void __launch_bounds__(256)
sgr_walk_kernel(const uint32_t* __restrict__ pairs, const int16_t* __restrict__ sd, int dstride, size_t dplane,
                const long long* __restrict__ sums, int pw, int ph, int unit_size, int units_x, int units_y, int voff, uint32_t ep_mask,
                int32_t* __restrict__ xqd_out, long long* __restrict__ err_out, uint32_t* __restrict__ counters, uint8_t* __restrict__ best_ep,
                int32_t* __restrict__ best_xqd, uint32_t* __restrict__ stats, int cap) {
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
    for (int pass = 0; pass < kPassBudget; pass++) {
        if (wave == 0) {
            LdsStore K(L, lane, ((volatile int&)L.n_cache) >= kThrottle ? 1 : cap);
            WalkState W0; walk_begin(W0, start);   // the streamed form re-decides the whole walk every pass
            const ModelSums MS0 = load_model(S);
            const bool fin = replay(K, W0, ep, MS0);
            if (lane == 0) L.done = fin ? 1 : 0;
            __builtin_amdgcn_wave_barrier();
            if (lane < kStreamCand) {   // svt_decode_xq (Common/Codec/EbRestoration.c:707-718); entries past n_want are not read
                const int x = L.wx[lane], y = L.wy[lane];
                L.xq0[lane] = has0 ? x : 0;
                L.xq1[lane] = !has1 ? 0 : (has0 ? 128 - x - y : 128 - y);
            }
        }
        __syncthreads();
        if (L.done) break;
        const int nc = L.n_want;
        n_pass++; n_eval += nc;
        int xq[kStreamCand];   // both taps scaled by 32 and packed for v_dot2_i32_i16 (|32 xq| <= 8192)
        long long acc[kStreamCand];
#pragma unroll
        for (int c = 0; c < kStreamCand; c++) {
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
        publish_walk(xqd_out, err_out, (size_t)unit * 16 + ep, L.res_x, L.res_y, L.done ? L.res_err : -1);   // -1: walk not finished within the pass budget (cannot happen, see kThrottle; callers treat it as a failure)
        atomicAdd(&stats[0], (uint32_t)n_pass); atomicAdd(&stats[1], (uint32_t)n_eval); if (!L.done) atomicAdd(&stats[2], 1u);   // diagnostics
        const uint32_t arrived = atomicAdd(&counters[unit], 1u) + 1u;
        if (arrived == (uint32_t)__popc(ep_mask)) pick_unit_best(xqd_out, err_out, unit, ep_mask, best_ep, best_xqd);
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// The same search with the unit RESIDENT on the compute unit: one 1024-thread workgroup per (unit, set).  Waves 1-15 load the unit's difference planes
// once — the (flt0 - u, flt1 - u) words into registers (9 chunks of 8 pixels per thread x 960 threads = 69 120 pixels, a whole 256 x 256 unit), dat - src
// into 135 KB of LDS — and every later pass of the walk is arithmetic on resident data; wave 0 holds no pixels and runs the solve and the replay (its
// registers are the replay's: the two roles are separate code paths that meet at the workgroup barriers, so neither spills).  The streamed form above
// re-reads 6 bytes per pixel and pass (3.3 passes per walk on coded content): 4 GB per 4K frame against 1.2 GB here.  Units larger than 69 120 pixels
// (the last row / column may be up to 1.5 x the unit size) keep the excess in memory and stream it per pass like the form above.  Per candidate and
// pixel pair: 2 v_dot2_i32_i16 (weighted sum, rounding constant as the accumulator), v_perm_b32 (high halves), v_pk_add_i16 (+ dat - src),
// v_dot2_i32_i16 (square-accumulate).
constexpr int kResT = 1024, kResJ = 9;   // full residency: threads, resident chunks per data thread
// The HYBRID instance (T = 512, J = 7, NA = 8): two workgroups share a compute unit, each keeps 7 x 448 chunks (38 % of a 256 x 256 unit) resident and
// streams the rest of the unit on every pass, chunk-outer with NA candidate accumulators — one workgroup's loads and replays overlap the other's
// evaluation (a compute unit streams only ~10 B per cycle from HBM, and with full residency nothing else can run beside the 1024 threads).
template <int T, int J>
struct ResLdsT {
    WalkLds W;
    int4    sd[J * (T - 64)];   // [chunk slot][data thread]: eight dat - src values
};

// Eight pixels of one candidate: 20 vector instructions, written out because the order matters on this pipeline — a dot product's result may
// not be read by a different opcode for three issue slots, so the eight weighted sums go first (the three-operand v_dot2_i32_i16 takes the rounding
// constant 2^15 from a scalar register; the compiler's own choice, the accumulate-in-place form, costs a v_mov per pixel), then the four
// v_perm_b32 that pack the high halves (= the rounded >> 11), the four v_pk_add_u16 that add dat - src, and four squaring dots that accumulate
// into p0 / p1 (a dot may feed the accumulator of the next dot back to back).  The caller waits three slots before reading p0 / p1 (dot_drain).
__device__ __forceinline__ void eval_chunk(const int4& a0, const int4& a1, const int4& s, int q, int rnd, int sel, int& p0, int& p1) {
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_dot2_i32_i16 %[t0], %[a0], %[q], %[r]\n\t"
        "v_dot2_i32_i16 %[t1], %[a1], %[q], %[r]\n\t"
        "v_dot2_i32_i16 %[t2], %[a2], %[q], %[r]\n\t"
        "v_dot2_i32_i16 %[t3], %[a3], %[q], %[r]\n\t"
        "v_dot2_i32_i16 %[t4], %[a4], %[q], %[r]\n\t"
        "v_dot2_i32_i16 %[t5], %[a5], %[q], %[r]\n\t"
        "v_dot2_i32_i16 %[t6], %[a6], %[q], %[r]\n\t"
        "v_dot2_i32_i16 %[t7], %[a7], %[q], %[r]\n\t"
        "v_perm_b32 %[t0], %[t1], %[t0], %[sel]\n\t"
        "v_perm_b32 %[t2], %[t3], %[t2], %[sel]\n\t"
        "v_perm_b32 %[t4], %[t5], %[t4], %[sel]\n\t"
        "v_perm_b32 %[t6], %[t7], %[t6], %[sel]\n\t"
        "v_pk_add_u16 %[t0], %[t0], %[s0]\n\t"
        "v_pk_add_u16 %[t2], %[t2], %[s1]\n\t"
        "v_pk_add_u16 %[t4], %[t4], %[s2]\n\t"
        "v_pk_add_u16 %[t6], %[t6], %[s3]\n\t"
        "v_dot2_i32_i16 %[p0], %[t0], %[t0], %[p0]\n\t"
        "v_dot2_i32_i16 %[p1], %[t2], %[t2], %[p1]\n\t"
        "v_dot2_i32_i16 %[p0], %[t4], %[t4], %[p0]\n\t"
        "v_dot2_i32_i16 %[p1], %[t6], %[t6], %[p1]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [p0] "+v"(p0), [p1] "+v"(p1)
        : [a0] "v"(a0.x), [a1] "v"(a0.y), [a2] "v"(a0.z), [a3] "v"(a0.w), [a4] "v"(a1.x), [a5] "v"(a1.y), [a6] "v"(a1.z), [a7] "v"(a1.w), [s0] "v"(s.x), [s1] "v"(s.y),
          [s2] "v"(s.z), [s3] "v"(s.w), [q] "v"(q), [r] "s"(rnd), [sel] "s"(sel));
}
// The same for a chunk that several candidates evaluate in one sweep: (dat - src) << 16 | 2^15 of every sample is formed once per sweep (expand_sd: one
// instruction per sample) and rides in the weighted sum's accumulator -- the upper half of the sum is then the sample's error with dat - src already in it, and
// the four packed adds per candidate go: 16 instructions per candidate and chunk instead of 20.
__device__ __forceinline__ void expand_sd(const int4& s, int (&c)[8]) {
    const int sw[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { c[2 * i] = (sw[i] << 16) | 0x8000; c[2 * i + 1] = (int)__builtin_amdgcn_perm(0x8000u, (uint32_t)sw[i], 0x03020504u); }   // v_lshl_or_b32; v_perm_b32: the upper half of sw over 0x8000 (an and + or pair otherwise: two literals do not fit one instruction)
}
__device__ __forceinline__ void eval_chunk_f(const int4& a0, const int4& a1, const int (&c)[8], int q, int sel, int& p0, int& p1) {
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_dot2_i32_i16 %[t0], %[a0], %[q], %[c0]\n\t"
        "v_dot2_i32_i16 %[t1], %[a1], %[q], %[c1]\n\t"
        "v_dot2_i32_i16 %[t2], %[a2], %[q], %[c2]\n\t"
        "v_dot2_i32_i16 %[t3], %[a3], %[q], %[c3]\n\t"
        "v_dot2_i32_i16 %[t4], %[a4], %[q], %[c4]\n\t"
        "v_dot2_i32_i16 %[t5], %[a5], %[q], %[c5]\n\t"
        "v_dot2_i32_i16 %[t6], %[a6], %[q], %[c6]\n\t"
        "v_dot2_i32_i16 %[t7], %[a7], %[q], %[c7]\n\t"
        "v_perm_b32 %[t0], %[t1], %[t0], %[sel]\n\t"
        "v_perm_b32 %[t2], %[t3], %[t2], %[sel]\n\t"
        "v_perm_b32 %[t4], %[t5], %[t4], %[sel]\n\t"
        "v_perm_b32 %[t6], %[t7], %[t6], %[sel]\n\t"
        "v_dot2_i32_i16 %[p0], %[t0], %[t0], %[p0]\n\t"
        "v_dot2_i32_i16 %[p1], %[t2], %[t2], %[p1]\n\t"
        "v_dot2_i32_i16 %[p0], %[t4], %[t4], %[p0]\n\t"
        "v_dot2_i32_i16 %[p1], %[t6], %[t6], %[p1]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [p0] "+v"(p0), [p1] "+v"(p1)
        : [a0] "v"(a0.x), [a1] "v"(a0.y), [a2] "v"(a0.z), [a3] "v"(a0.w), [a4] "v"(a1.x), [a5] "v"(a1.y), [a6] "v"(a1.z), [a7] "v"(a1.w), [c0] "v"(c[0]), [c1] "v"(c[1]),
          [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [q] "s"(q), [sel] "s"(sel));
}
// One accumulator per candidate (the 16-candidate instance: |e| < 2^10 at bit depth 8 and a thread sees < 400 samples of the largest unit, so 2^20 x 400 < 2^31):
// the four squaring dots chain through p (a dot may feed the next dot's accumulator back to back).
__device__ __forceinline__ void eval_chunk_f1(const int4& a0, const int4& a1, const int (&c)[8], int q, int sel, int& p) {
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_dot2_i32_i16 %[t0], %[a0], %[q], %[c0]\n\t"
        "v_dot2_i32_i16 %[t1], %[a1], %[q], %[c1]\n\t"
        "v_dot2_i32_i16 %[t2], %[a2], %[q], %[c2]\n\t"
        "v_dot2_i32_i16 %[t3], %[a3], %[q], %[c3]\n\t"
        "v_dot2_i32_i16 %[t4], %[a4], %[q], %[c4]\n\t"
        "v_dot2_i32_i16 %[t5], %[a5], %[q], %[c5]\n\t"
        "v_dot2_i32_i16 %[t6], %[a6], %[q], %[c6]\n\t"
        "v_dot2_i32_i16 %[t7], %[a7], %[q], %[c7]\n\t"
        "v_perm_b32 %[t0], %[t1], %[t0], %[sel]\n\t"
        "v_perm_b32 %[t2], %[t3], %[t2], %[sel]\n\t"
        "v_perm_b32 %[t4], %[t5], %[t4], %[sel]\n\t"
        "v_perm_b32 %[t6], %[t7], %[t6], %[sel]\n\t"
        "v_dot2_i32_i16 %[p], %[t0], %[t0], %[p]\n\t"
        "v_dot2_i32_i16 %[p], %[t2], %[t2], %[p]\n\t"
        "v_dot2_i32_i16 %[p], %[t4], %[t4], %[p]\n\t"
        "v_dot2_i32_i16 %[p], %[t6], %[t6], %[p]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [p] "+v"(p)
        : [a0] "v"(a0.x), [a1] "v"(a0.y), [a2] "v"(a0.z), [a3] "v"(a0.w), [a4] "v"(a1.x), [a5] "v"(a1.y), [a6] "v"(a1.z), [a7] "v"(a1.w), [c0] "v"(c[0]), [c1] "v"(c[1]),
          [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [q] "v"(q), [sel] "s"(sel));
}
// eval_chunk_f1 with the candidate's taps in a SCALAR register (each instruction still reads one scalar operand only) -- the packed walk's form: ONE int32 accumulator
// per candidate (|e| <= 1010 at bit depth 8 and a data thread of the 512-thread instances sees <= 42 chunks = 336 samples: 336 x 2^20 < 2^31), the four squaring dots
// chain through it (a dot may feed the next dot's accumulator back to back).  Against eval_chunk_f: sixteen vector registers fewer per pass.
__device__ __forceinline__ void eval_chunk_f1s(const int4& a0, const int4& a1, const int (&c)[8], int q, int sel, int& p) {
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_dot2_i32_i16 %[t0], %[a0], %[q], %[c0]\n\t"
        "v_dot2_i32_i16 %[t1], %[a1], %[q], %[c1]\n\t"
        "v_dot2_i32_i16 %[t2], %[a2], %[q], %[c2]\n\t"
        "v_dot2_i32_i16 %[t3], %[a3], %[q], %[c3]\n\t"
        "v_dot2_i32_i16 %[t4], %[a4], %[q], %[c4]\n\t"
        "v_dot2_i32_i16 %[t5], %[a5], %[q], %[c5]\n\t"
        "v_dot2_i32_i16 %[t6], %[a6], %[q], %[c6]\n\t"
        "v_dot2_i32_i16 %[t7], %[a7], %[q], %[c7]\n\t"
        "v_perm_b32 %[t0], %[t1], %[t0], %[sel]\n\t"
        "v_perm_b32 %[t2], %[t3], %[t2], %[sel]\n\t"
        "v_perm_b32 %[t4], %[t5], %[t4], %[sel]\n\t"
        "v_perm_b32 %[t6], %[t7], %[t6], %[sel]\n\t"
        "v_dot2_i32_i16 %[p], %[t0], %[t0], %[p]\n\t"
        "v_dot2_i32_i16 %[p], %[t2], %[t2], %[p]\n\t"
        "v_dot2_i32_i16 %[p], %[t4], %[t4], %[p]\n\t"
        "v_dot2_i32_i16 %[p], %[t6], %[t6], %[p]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [p] "+v"(p)
        : [a0] "v"(a0.x), [a1] "v"(a0.y), [a2] "v"(a0.z), [a3] "v"(a0.w), [a4] "v"(a1.x), [a5] "v"(a1.y), [a6] "v"(a1.z), [a7] "v"(a1.w), [c0] "v"(c[0]), [c1] "v"(c[1]),
          [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [q] "s"(q), [sel] "s"(sel));
}
__device__ __forceinline__ void dot_drain() { asm volatile("s_nop 2"); }
// zero the pixels of a chunk at or past column n (n < 8): their error is then 0 for every candidate
__device__ __forceinline__ void mask_chunk(int4& a0, int4& a1, int4& s, int n) {
    int pr[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    int sw[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (i >= n) { pr[i] = 0; sw[i >> 1] &= (i & 1) ? 0x0000FFFF : 0; }
    a0 = make_int4(pr[0], pr[1], pr[2], pr[3]); a1 = make_int4(pr[4], pr[5], pr[6], pr[7]); s = make_int4(sw[0], sw[1], sw[2], sw[3]);
}

// Accumulator ranges.  e = ((xq0 (flt0 - u) + xq1 (flt1 - u) + 2^10) >> 11) + dat - src (svt_av1_{lowbd,highbd}_pixel_proj_error, EbRestorationPick.c:174-330).
// |flt - u| <= D with D = 17 560 at bit depth 10 (flt <= 16 * 1023 * (1 + the one_by_x rounding) + 1, u = 16 dat; sgr.hip stores it as int16) and 4 390 at 8;
// svt_decode_xq (EbRestoration.c:707-718) gives xq0 in [-96, 31], xq1 = 128 - xqd0 - xqd1 in [2, 256], the extremes together at (-96, 256):
//   |e| <= ((96 + 256) D + 2^10) >> 11  +  (2^bd - 1)  =  3 019 + 1 023 = 4 042 at bit depth 10 (1 010 at 8),   e^2 <= 16 337 764 < 2^24 (< 2^20).
// A v_dot2_i32_i16 accumulator wraps modulo 2^32, so READ AS UNSIGNED it holds floor((2^32 - 1) / 16 337 764) = 262 squares at bit depth 10.  The largest restoration
// unit is 383 x 383 samples (1.5 x 256 rounds to two units; foreach_rest_unit_in_tile, EbRestoration.c:1369-1411) = 48 x 383 chunks of eight: a data thread of the
// 512-thread instances (448 data threads) sees <= 42 chunks = 336 squares, 168 per accumulator of the two-accumulator forms -> NO 64-bit drain inside a pass at either
// bit depth (round 3 emptied the 10-bit accumulators every third chunk on the looser |e| < 2^13).  One-accumulator forms (sixteen candidates) and the 256-thread
// instances (192 data threads: 96 chunks) exceed 262 squares at bit depth 10: they keep the periodic drain (DRAIN).  tests/test_sgr_gpu.py::test_search_units_largest_unit_extreme_content.
template <int BD, int kT, int kJ, int NA, int PF = 1, bool DRAIN = false>   // NA > 8 (opt-in instance): one accumulator per candidate, and the walker also asks for the other outcome's next probe; NA = 0: every candidate walks the resident chunks (and re-streams the excess of an over-sized unit) on its own; PF = streamed chunks in flight ahead of the one being evaluated
__global__ void __launch_bounds__(kT, (kT == 512 ? 4 : 1))   // the hybrid instances are built for two workgroups per compute unit: 128 registers
sgr_walk_resident_kernel(const WalkPic a) {
    constexpr int kResT = kT, kResJ = kJ, kResD = kT - 64;
    constexpr bool HEDGE = NA > 8;                        // the sixteen-candidate instance's walker hedges its requests
    constexpr bool ONE_ACC = NA > 8 || (DRAIN && PF != 2);   // one int32 accumulator per candidate: the sixteen-candidate instance (bit depth 8: a thread's squares fit), and the draining instances
    static_assert(NA <= 8 || BD == 8, "sixteen int32 accumulators hold a thread's squares at bit depth 8 only");
    static_assert(BD == 8 || DRAIN || kT >= 512, "bit depth 10 without drains: <= 42 chunks per data thread (see above)");
    __shared__ ResLdsT<kT, kJ> R;
    WalkLds& L = R.W;
    // the planes of a picture share one launch (grid.z): one tail instead of three.  Scalar copies of the plane's arguments (a reference into the
    // kernel-argument struct with a run-time index would force a private copy of the whole struct)
    const int z = blockIdx.z;
    const uint32_t* __restrict__ pairs = a.p[z].pairs; const int16_t* __restrict__ sd = a.p[z].sd; const long long* __restrict__ sums = (const long long*)a.p[z].sums;
    const int dstride = a.p[z].dstride, pw = a.p[z].pw, ph = a.p[z].ph, unit_size = a.p[z].unit_size, units_x = a.p[z].units_x, units_y = a.p[z].units_y, voff = a.p[z].voff;
    const size_t dplane = a.p[z].dplane;
    const uint32_t ep_mask = a.p[z].ep_mask;
    int32_t* __restrict__ xqd_out = a.p[z].xqd_out; long long* __restrict__ err_out = (long long*)a.p[z].err_out; uint32_t* __restrict__ counters = a.p[z].counters;
    uint8_t* __restrict__ best_ep = a.p[z].best_ep; int32_t* __restrict__ best_xqd = a.p[z].best_xqd; uint32_t* __restrict__ stats = a.p[z].stats;
    const int cap = a.cap;
    const int unit = blockIdx.x, ep = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (unit >= units_x * units_y || !((ep_mask >> ep) & 1)) return;
    const int uj = unit % units_x, ui = unit / units_x;
    const int x0 = uj * unit_size, w = uj == units_x - 1 ? pw - x0 : unit_size;
    const int y0 = ui * unit_size, h = ui == units_y - 1 ? ph - y0 : unit_size;
    const int v0 = max(y0 - voff, 0), v1 = (y0 + h < ph) ? y0 + h - voff : y0 + h;
    const bool has0 = ep < 10 || ep >= 14, has1 = ep < 14;
    const int  ce = ep == 11 ? 2 : (ep == 12 ? 5 : (ep == 13 ? 8 : ep));
    const uint32_t* __restrict__ PP = pairs + (size_t)ce * dplane;
    const long long* S = sums + ((size_t)unit * 16 + ep) * 5;
    // ---- HISTOGRAM EVALUATION of the one-filter sets (10 .. 13: r0 = 0, 14 / 15: r1 = 0).  With one tap the error of a sample is e = q(d) + (dat - src),
    // q(d) = (xq d + 2^10) >> 11, d = flt - u: samples with equal d share q, so  sum e^2 = sum_d [n_d q(d)^2 + 2 q(d) R_d] + sum (dat - src)^2  with n_d the number of
    // samples with that d and R_d the sum of their dat - src -- exactly, rounding included.  The unit is streamed ONCE into a histogram over d in LDS (one 64-bit
    // atomic per sample: n_d in the upper 24 bits, sum (dat - src + 1024) below; the storage of the resident dat - src values, which this form does not need), and a
    // candidate point then costs (2 W + 1) / threads bins per thread instead of a pass over the unit's samples: no resident chunks, no re-streaming, sixteen points per
    // pass.  |d| is a few hundred on coded pictures (the filter passes high-variance samples through), its bound is 17 560: a sample outside the window |d| <= W goes to
    // a list of kCache exact (d, dat - src) entries, and a unit with more of them than that is evaluated sample by sample like the two-filter sets (decided after the
    // streaming pass; SVT_HIP_SGR_WALK_HIST_W narrows W so that tests reach the list and the fall-back).
    constexpr int kHistWMax = (int)(sizeof(R.sd) / 16) - 1 < 3071 ? (int)(sizeof(R.sd) / 16) - 1 : 3071;   // bins d + W, 0 <= . <= 2 W, in the storage of R.sd
    const bool try_hist = NA > 0 && !(has0 && has1) && a.hist_w >= 0;   // workgroup-uniform
    const int  W = min(kHistWMax, a.hist_w);

    if (wave == 0) {
        // =================================== the walk: solve, replay, cache ===================================
        int start[2] = {0, 0};
        const unsigned long long c0 = __builtin_readcyclecounter();
        unsigned long long c_replay = 0, c_load = 0, c_eval = 0;
        __builtin_amdgcn_s_setprio(3);
        solve_and_encode(S, w * (v1 - v0), ep, start);
        __builtin_amdgcn_s_setprio(0);
        RegStore K;
#pragma unroll
        for (int b = 0; b < kBanks; b++) { K.key[b] = -1; K.err[b] = 0; }
        // a unit that is resident as a whole (every chroma unit of a 4:2:0 picture, small edge units) evaluates a candidate at ~1/8 of the cost of a streamed one
        // and needs no per-candidate accumulators: its passes take up to kMaxCand points, which lets most of its walks finish after ONE evaluation pass (two
        // replays instead of three: the serial replay is more than half of such a walk)
        const bool whole = NA > 0 && ((w + 7) >> 3) * (v1 - v0) <= kResJ * kResD;
        K.wkey = -1; K.lane = lane; K.n_cache = 0; K.nw = 0; K.cap = whole ? kMaxCand : cap; K.res_x = start[0]; K.res_y = start[1]; K.res_err = -1;
        bool hist = false;
        if (try_hist) { if (lane == 0) L.ovf_n = 0; __syncthreads(); }   // H: the data waves have cleared the histogram, the list is empty
        const unsigned long long c1 = __builtin_readcyclecounter();
        int n_pass = 0, n_eval = 0;
        bool fin = false;
        WalkState WS; walk_begin(WS, start);
        const ModelSums MS = load_model(S);
        int cap0 = K.cap;
        for (int pass = 0; pass < kPassBudget; pass++) {
            const unsigned long long r0 = __builtin_readcyclecounter();
            __builtin_amdgcn_s_setprio(3);   // the replay is the serial part of the walk: it goes ahead of the other workgroup's evaluation waves on this SIMD
            K.cap = K.n_cache >= kThrottle ? 1 : cap0;
            fin = replay<RegStore, HEDGE>(K, WS, ep, MS);
            __builtin_amdgcn_s_setprio(0);
            c_replay += __builtin_readcyclecounter() - r0;
            if (try_hist && pass == 0) {   // H2: the unit has been streamed (this first replay ran beside it); from the next pass on a histogram walk asks for sixteen points
                __syncthreads();
                hist = ((volatile int&)L.ovf_n) <= kCache;
                if (hist) cap0 = kMaxCand;
            }
            const int nc = K.nw;
            if (lane == 0) { L.done = fin ? 1 : 0; L.n_want = nc; }
            if (lane < nc) {   // svt_decode_xq (Common/Codec/EbRestoration.c:707-718)
                const int x = (K.wkey & 255) - 128, y = (K.wkey >> 8) - 128;
                L.xq0[lane] = has0 ? x : 0;
                L.xq1[lane] = !has1 ? 0 : (has0 ? 128 - x - y : 128 - y);
            }
            const unsigned long long a0 = __builtin_readcyclecounter();
            __syncthreads();   // A: the candidate list is published (or the walk is over)
            const unsigned long long a1 = __builtin_readcyclecounter();
            if (pass == 0) c_load = a1 - a0;
            if (fin) break;
            n_pass++; n_eval += nc;
            __syncthreads();   // B: every data wave has left its partial sums
            c_eval += __builtin_readcyclecounter() - a1;
            // the new exact points join the cache: point c goes to entry n_cache + c = lane (n_cache + c) & 63 of bank (n_cache + c) >> 6
#pragma unroll
            for (int b = 0; b < kBanks; b++) {
                const int c = 64 * b + lane - K.n_cache;
                const int k = __shfl(K.wkey, c & 63, 64);
                if (c >= 0 && c < nc) {
                    long long e = 0;
#pragma unroll
                    for (int v = 1; v < kResT / 64; v++) e += L.part[v][c];
                    K.key[b] = k; K.err[b] = e;
                }
            }
            K.n_cache = min(K.n_cache + nc, kCache);   // kCache is never reached (kThrottle)
        }
        // ---- results, and the unit's best set once all of its sets are in: search_selfguided_restoration :661-665 (first set with the smallest error)
        if (lane == 0) {
            publish_walk(xqd_out, err_out, (size_t)unit * 16 + ep, K.res_x, K.res_y, fin ? K.res_err : -1);   // -1: walk not finished within the pass budget (cannot happen, see kThrottle; callers treat it as a failure)
            atomicAdd(&stats[0], (uint32_t)n_pass); atomicAdd(&stats[1], (uint32_t)n_eval); if (!fin) atomicAdd(&stats[2], 1u);
            if (hist) atomicAdd(&stats[3], 1u);   // walks evaluated on the histogram
            // phase clocks of the walk, in units of 64 shader cycles (diagnostics: tools/hbd_time.py)
            if (a.clocks) {
                atomicAdd(&stats[24], (uint32_t)((c1 - c0) >> 6)); atomicAdd(&stats[25], (uint32_t)(c_replay >> 6)); atomicAdd(&stats[26], (uint32_t)(c_load >> 6));
                atomicAdd(&stats[27], (uint32_t)(c_eval >> 6)); atomicAdd(&stats[28], (uint32_t)((__builtin_readcyclecounter() - c0) >> 6));
            }
            const uint32_t arrived = atomicAdd(&counters[unit], 1u) + 1u;
            if (arrived == (uint32_t)__popc(ep_mask)) pick_unit_best(xqd_out, err_out, unit, ep_mask, best_ep, best_xqd);
        }
        return;
    }
    // =================================== the pixels: load once, evaluate every pass ===================================
    const int t = tid - 64;
    const int cw = (w + 7) >> 3, nchunk = cw * (v1 - v0);
    if (try_hist) {
        unsigned long long* H = (unsigned long long*)&R.sd[0];
        // (a private copy of the histogram per data wave was measured and bought nothing: streaming the unit, not the atomics, is what this phase costs)
        const int nb = 2 * W + 1;
        for (int b = t; b < nb; b += kResD) H[b] = 0ull;
        __syncthreads();   // H
        // ---- the unit, once: one atomic per sample (the next chunk's loads are issued before this chunk's atomics)
        unsigned long long r2 = 0;
        int k = t;
        int4 a0 = make_int4(0, 0, 0, 0), a1 = a0, s4 = a0;
        auto fetch = [&](int kk, int4& x0v, int4& x1v, int4& sv) {
            const int row = kk / cw, cx = kk - row * cw;
            const size_t off = (size_t)(v0 + row) * dstride + x0 + 8 * cx;
            x0v = SGR_LD4(PP + off); x1v = SGR_LD4(PP + off + 4); sv = SGR_LD4(sd + off);
        };
        if (k < nchunk) fetch(k, a0, a1, s4);
        while (k < nchunk) {
            const int kn = k + kResD;
            int4 b0 = make_int4(0, 0, 0, 0), b1 = b0, t4 = b0;
            if (kn < nchunk) fetch(kn, b0, b1, t4);
            const int cx = k % cw, n = min(8, w - 8 * cx);
            const int pr[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const int sw[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < n) {
                    const int d = has0 ? (int)(int16_t)(pr[i] & 0xFFFF) : (pr[i] >> 16);
                    const int r = (i & 1) ? (sw[i >> 1] >> 16) : (int)(int16_t)(sw[i >> 1] & 0xFFFF);
                    if (abs(d) <= W) atomicAdd(&H[d + W], (1ull << 40) | (unsigned long long)(uint32_t)(r + 1024));
                    else {   // outside the window: listed exactly (or, past kCache of them, the whole unit falls back)
                        const int at = atomicAdd(&L.ovf_n, 1);
                        if (at < kCache) { L.cx[at] = d; L.cy[at] = r; }
                    }
                    r2 += (unsigned long long)(uint32_t)(r * r);
                }
            a0 = b0; a1 = b1; s4 = t4; k = kn;
        }
        __syncthreads();   // H2: histogram and list complete
        const int n_ovf = ((volatile int&)L.ovf_n);
        if (n_ovf <= kCache) {
            const long long r2w = wave_sum_u48((long long)r2);   // this wave's share of sum (dat - src)^2 (< 2^48: 42 x 8 x 64 squares below 2^20)
            for (int pass = 0; pass < kPassBudget; pass++) {
                __syncthreads();   // A
                if (L.done) break;
                const int nc = L.n_want;
                for (int c = 0; c < nc; c++) {
                    const int xq = has0 ? L.xq0[c] : L.xq1[c];
                    long long acc = 0;
                    for (int b = t; b < nb; b += kResD) {
                        const unsigned long long h = H[b];
                        const int nd = (int)(h >> 40);
                        if (nd) {
                            const long long Rd = (long long)(h & ((1ull << 40) - 1)) - 1024ll * nd;
                            const int q = (xq * (b - W) + 1024) >> 11;   // |xq| <= 256, |d| <= 3071: 20 bits
                            acc += (long long)nd * (q * q) + 2ll * q * Rd;   // nd q^2 < 2^18 x 2^17.2, |q Rd| < 2^8.6 x 2^28: int64
                        }
                    }
                    for (int o = t; o < n_ovf; o += kResD) {   // a listed sample: n = 1, R = its dat - src (|xq d| < 2^8 x 2^14.1: 23 bits)
                        const int q = (xq * L.cx[o] + 1024) >> 11;
                        acc += (long long)q * q + 2ll * q * L.cy[o];
                    }
                    const long long sum = wave_sum_i64(acc);
                    if (lane == 0) L.part[wave][c] = sum + r2w;
                }
                __syncthreads();   // B
            }
            return;
        }
    }
    // The resident part of the unit: ALL of its loads are issued before the first one is consumed.  (Round 2 .. 4 loaded chunk j under `if (k < nchunk)` and wrote its
    // dat - src to LDS right away: a load under a condition is a branch, the LDS write behind it waits for it, and the kResJ chunks' loads ran one after the other —
    // kResJ memory round trips, the "40 k-cycle load of a unit" of DESIGN 4.5.)  Chunks past the end load the last chunk's address and are zeroed afterwards.
    int4 pa[kResJ], pb[kResJ], ps[kResJ];
#pragma unroll
    for (int j = 0; j < kResJ; j++) {
        const int k = min(t + j * kResD, max(nchunk - 1, 0));
        const int row = k / cw, cx = k - row * cw;
        const size_t off = (size_t)(v0 + row) * dstride + x0 + 8 * cx;
        pa[j] = SGR_LD4(PP + off); pb[j] = SGR_LD4(PP + off + 4);
        ps[j] = *(const int4*)(sd + off);
    }
#pragma unroll
    for (int j = 0; j < kResJ; j++) {
        const int k = t + j * kResD;
        const int kc = min(k, max(nchunk - 1, 0)), row = kc / cw, cx = kc - row * cw;
        const int n = k < nchunk ? w - 8 * cx : 0;
        if (n < 8) mask_chunk(pa[j], pb[j], ps[j], n);
        R.sd[j * kResD + t] = ps[j];   // read back by this thread only
    }
    int rnd, sel;
    asm volatile("s_mov_b32 %0, 0x8000\n\ts_mov_b32 %1, 0x07060302" : "=s"(rnd), "=s"(sel));   // opaque: kept in scalar registers
    for (int pass = 0; pass < kPassBudget; pass++) {
        __syncthreads();   // A
        if (L.done) break;
        const int nc = L.n_want;
        const int qv = lane < nc ? (int)(((uint32_t)(L.xq0[lane] * 32) & 0xFFFFu) | ((uint32_t)(L.xq1[lane] * 32) << 16)) : 0;   // lane c: both taps of candidate c, scaled by 32 (|32 xq| <= 8192)
        const unsigned long long e0 = __builtin_readcyclecounter();
        bool hybrid_done = false;
        if constexpr (NA > 0) if (nchunk > kResJ * kResD) {
            hybrid_done = true;
            // ---- hybrid: the streamed part of the unit first (its loads are in flight while the resident part is evaluated), one pass over it for all
            // candidates; per candidate two int32 accumulators (bit depth 8: |e| < 2^10, < 160 samples per thread) or one int32 accumulator emptied into a 64-bit sum every third chunk (bit depth 10)
            int pp0[NA], pp1[NA]; long long acc[NA];
            int qq[NA];
#pragma unroll
            for (int c = 0; c < NA; c++) { pp0[c] = pp1[c] = 0; acc[c] = 0; qq[c] = __builtin_amdgcn_readlane(qv, c); }
            int k = t + kResJ * kResD;
            int4 a0 = make_int4(0, 0, 0, 0), a1 = a0, s4 = a0;
            auto fetch = [&](int kk, int4& x0v, int4& x1v, int4& sv) {
                const int row = kk / cw, cx = kk - row * cw;
                const size_t off = (size_t)(v0 + row) * dstride + x0 + 8 * cx;
                x0v = SGR_LD4(PP + off); x1v = SGR_LD4(PP + off + 4); sv = SGR_LD4(sd + off);
                const int n = w - 8 * cx;
                if (n < 8) mask_chunk(x0v, x1v, sv, n);
            };
            if (PF == 2 && k < nchunk) fetch(k, a0, a1, s4);
            // bit depth 10: |e| < 2^13, eight squares per chunk: the candidate's int32 accumulator holds kDrain = 3 chunks (3 x 2^29 < 2^31) before it is emptied into
            // its 64-bit sum -- every third streamed chunk, once more before the resident slots, every third of those, and at the end
            constexpr int kDrain = 3;
            auto drain = [&]() {
                dot_drain();
#pragma unroll
                for (int c = 0; c < NA; c++)
                    if (c < nc) { acc[c] += (long long)(uint32_t)pp0[c] + (long long)(uint32_t)pp1[c]; pp0[c] = pp1[c] = 0; }
            };
            if constexpr (PF == 2) {   // two chunks in flight: a compute unit's streaming rate is set by the bytes it has outstanding
                int4 b0 = make_int4(0, 0, 0, 0), b1 = b0, t4 = b0;
                if (k + kResD < nchunk) fetch(k + kResD, b0, b1, t4);
                while (k < nchunk) {
                    const int kn2 = k + 2 * kResD;
                    int4 c0 = make_int4(0, 0, 0, 0), c1 = c0, u4 = c0;
                    if (kn2 < nchunk) fetch(kn2, c0, c1, u4);
                    int sx[8]; expand_sd(s4, sx);
#pragma unroll
                    for (int c = 0; c < NA; c++)
                        if (c < nc) {
                            if (ONE_ACC) eval_chunk_f1(a0, a1, sx, qq[c], sel, pp0[c]); else eval_chunk_f(a0, a1, sx, qq[c], sel, pp0[c], pp1[c]);
                            if (DRAIN) { dot_drain(); acc[c] += (long long)(uint32_t)pp0[c] + (long long)(uint32_t)pp1[c]; pp0[c] = pp1[c] = 0; }
                        }
                    a0 = b0; a1 = b1; s4 = t4; b0 = c0; b1 = c1; t4 = u4; k += kResD;
                }
            } else {
            // The streamed chunks of this thread are k, k + kResD, ...: (row, column chunk) and the byte offset advance by constants (one wrap test) instead of a
            // division and 64-bit address arithmetic per chunk; the loop is unrolled over two register sets, so a prefetched chunk is consumed where it landed.
            // (Round 5's loop spent ~80 instructions per chunk around the 17 per candidate: division 17, addresses 10, zeroing and copying the prefetch buffers 32,
            // forming the accumulators 12 -- profiles/r06/NOTES.md.)
            int since = 0;
            const int d_row = kResD / cw, d_cx = kResD - d_row * cw;                       // wave-uniform
            const uint32_t d_off = (uint32_t)(d_row * dstride + 8 * d_cx), d_wrap = (uint32_t)(dstride - 8 * cw);
            int cx = k % cw;
            uint32_t off = (uint32_t)((v0 + k / cw) * dstride + x0 + 8 * cx);             // samples from the plane's origin: < 2^24
            const char* __restrict__ PPb = (const char*)PP; const char* __restrict__ sdb = (const char*)sd;
            const bool ragged = (w & 7) != 0;   // only the last column chunk of a unit whose width is not a multiple of eight has samples to mask (wave-uniform)
            auto fetch2 = [&](int4& x0v, int4& x1v, int4& sv, int& n) {   // loads only: masking at fetch time made the compiler wait for the data right here
                x0v = SGR_LD4((const uint32_t*)(PPb + 4u * off)); x1v = SGR_LD4((const uint32_t*)(PPb + 4u * off + 16u)); sv = SGR_LD4((const int16_t*)(sdb + 2u * off));
                n = w - 8 * cx;
            };
            auto advance = [&]() { cx += d_cx; off += d_off; const bool wr = cx >= cw; cx -= wr ? cw : 0; off += wr ? d_wrap : 0u; };
            if (k < nchunk) {
                int na = 8;
                fetch2(a0, a1, s4, na);
                for (;;) {
                    // this chunk landed an iteration ago; its first use comes BEFORE the next chunk's loads are issued, so the wait for it does not cover them
                    if (ragged && na < 8) mask_chunk(a0, a1, s4, na);
                    int sx[8]; expand_sd(s4, sx);
                    k += kResD; advance();
                    const bool more = k < nchunk;
                    int4 b0, b1, t4; int nb = 8;
                    if (more) fetch2(b0, b1, t4, nb);   // next chunk's loads before this chunk's arithmetic
#pragma unroll
                    for (int c = 0; c < NA; c++)
                        if (c < nc) {
                            if (ONE_ACC) eval_chunk_f1(a0, a1, sx, qq[c], sel, pp0[c]); else eval_chunk_f(a0, a1, sx, qq[c], sel, pp0[c], pp1[c]);
                        }
                    if (DRAIN && ++since == kDrain) { drain(); since = 0; }
                    if (!more) break;
                    a0 = b0; a1 = b1; s4 = t4; na = nb;
                }
            }
            if (DRAIN && since) drain();
            }
            // ---- the resident chunks, slot by slot for all candidates (one LDS read of dat - src per slot)
#pragma unroll
            for (int j = 0; j < kResJ; j++) {
                if (j * kResD < nchunk) {
                    const int4 sc = R.sd[j * kResD + t];
                    int sx[8]; expand_sd(sc, sx);
#pragma unroll
                    for (int c = 0; c < NA; c++)
                        if (c < nc) {
                            if (ONE_ACC) eval_chunk_f1(pa[j], pb[j], sx, qq[c], sel, pp0[c]); else eval_chunk_f(pa[j], pb[j], sx, qq[c], sel, pp0[c], pp1[c]);
                            if (DRAIN && PF == 2) { dot_drain(); acc[c] += (long long)(uint32_t)pp0[c] + (long long)(uint32_t)pp1[c]; pp0[c] = pp1[c] = 0; }
                        }
                    if (DRAIN && PF != 2 && (j + 1) % kDrain == 0 && j + 1 < kResJ) drain();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            dot_drain();
#pragma unroll
            for (int c = 0; c < NA; c++)
                if (c < nc) {
                    const long long sum = wave_sum_u48(acc[c] + (long long)(uint32_t)pp0[c] + (long long)(uint32_t)pp1[c]);   // the accumulators are read as unsigned (see the ranges above)
                    if (lane == 0) L.part[wave][c] = sum;
                }
        }
        if (!hybrid_done)   // the unit is resident as a whole (or this is the fully resident instance): candidate by candidate over the resident chunks
        for (int c = 0; c < nc; c++) {
            const int q = __builtin_amdgcn_readlane(qv, c);
            long long acc = 0;
            int p0 = 0, p1 = 0;
            int4 sn = R.sd[t];
#pragma unroll
            for (int j = 0; j < kResJ; j++) {   // dat - src of the next chunk is fetched while this one is evaluated; the fence keeps the scheduler from
                if (j * kResD < nchunk) {       // hoisting all nine LDS reads (36 registers on top of the 72 resident ones); unused slots (small units) are skipped
                    const int4 sc = sn;
                    if (j + 1 < kResJ) sn = R.sd[(j + 1) * kResD + t];
                    eval_chunk(pa[j], pb[j], sc, q, rnd, sel, p0, p1);
                    if (DRAIN) { dot_drain(); acc += (long long)(uint32_t)p0 + (long long)(uint32_t)p1; p0 = p1 = 0; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!DRAIN) { dot_drain(); acc += (long long)(uint32_t)p0 + (long long)(uint32_t)p1; p0 = p1 = 0; }   // <= 9 resident chunks: 36 squares per accumulator
            for (int k = t + kResJ * kResD; k < nchunk; k += kResD) {   // the part of an over-sized unit that is not resident
                const int row = k / cw, cx = k - row * cw;
                const size_t off = (size_t)(v0 + row) * dstride + x0 + 8 * cx;
                int4 a0 = SGR_LD4(PP + off), a1 = SGR_LD4(PP + off + 4), s = SGR_LD4(sd + off);
                const int n = w - 8 * cx;
                if (n < 8) mask_chunk(a0, a1, s, n);
                eval_chunk(a0, a1, s, q, rnd, sel, p0, p1);
                dot_drain(); acc += (long long)(uint32_t)p0 + (long long)(uint32_t)p1; p0 = p1 = 0;
            }
            const long long sum = wave_sum_u48(acc);   // acc < 2^48
            if (lane == 0) L.part[wave][c] = sum;
        }
        if (a.clocks && tid == 64) { atomicAdd(&stats[29], (uint32_t)((__builtin_readcyclecounter() - e0) >> 6)); atomicAdd(&stats[30], (uint32_t)nc); }   // diagnostics: the candidate loop as wave 1 sees it
        __syncthreads();   // B
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same search on PACKED difference words (bit depth 8; sgr.hip, STORE == 2).  The walk above is bound by what a compute unit can stream: a pass re-reads
// the 62 % of the unit that is not resident at 6 bytes per sample (3 passes per walk: ~0.9 MB per walk, ~2 GB per 4K frame, ~60 % of the achievable memory rate while
// it runs).  Here a sample is ONE word [d1 : 11 | r_lo : 5 | d0 : 11 | r_hi : 5] -- both filter differences and dat - src -- so
//   * a chunk of eight samples is 32 bytes instead of 48 and there is no dat - src plane;
//   * the LDS that held the resident chunks' dat - src holds kL more chunks per data thread: (kJ + kL) x 448 x 8 = 43 008 samples = 66 % of a 256 x 256 unit are
//     resident (38 % before), and the streamed rest costs 4 bytes per sample: 88 KB per pass instead of 245 KB;
//   * the price is the unpacking, per sample and pass: `word & 0xFFE0FFE0` is the pair (32 d0, 32 d1) -- so the candidate's taps ride unscaled --, three more
//     operations rebuild (dat - src) << 16 | 2^15, the dot product's accumulator.
// Samples that do not fit the word are zero words in the plane (error 0 for every candidate) and sit in the (unit, set)'s escape list: every pass adds them one by one,
// whatever their number (a binary test picture escapes everywhere: slow, exact).  Wave 0 is the walker of the kernel above, unchanged.
template <int kT, int kL>
struct PackLdsT {
    WalkLds W;
    int4    pk[kL * 2 * (kT - 64)];   // [slot][half][data thread]: four packed words each (a wave's 64 lanes are contiguous: the direct global -> LDS load's layout); the one-filter sets' histogram lives here instead
};
__device__ __forceinline__ void unpack_chunk(const int4& w0, const int4& w1, int4& a0, int4& a1, int (&c)[8]) {
    const int w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    int pr[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        pr[i] = w[i] & (int)0xFFE0FFE0u;                                                      // (32 d0) | (32 d1) << 16
        const int lo = (w[i] & 0x001F0000) | 0x8000;                                          // r_lo << 16 | 2^15
        c[i] = (int)(((uint32_t)__builtin_amdgcn_sbfe(w[i], 0, 5) << 21) + (uint32_t)lo);     // + sext(r_hi) << 21
    }
    a0 = make_int4(pr[0], pr[1], pr[2], pr[3]); a1 = make_int4(pr[4], pr[5], pr[6], pr[7]);
}
__device__ __forceinline__ void mask_words(int4& w0, int4& w1, int n) {   // zero the words of a chunk at or past column n (n < 8)
    int w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int i = 0; i < 8; i++) if (i >= n) w[i] = 0;
    w0 = make_int4(w[0], w[1], w[2], w[3]); w1 = make_int4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ int packed_r(int w) { return (int)((uint32_t)__builtin_amdgcn_sbfe(w, 0, 5) << 5) | ((w >> 16) & 31); }

template <int kT, int kJ, int kL, int NA>
__global__ void __launch_bounds__(kT, 4)
sgr_walk_packed_kernel(const WalkPic a) {
    constexpr int kResD = kT - 64, kResC = (kJ + kL) * kResD;   // data threads, resident chunks
    __shared__ PackLdsT<kT, kL> R;
    WalkLds& L = R.W;
    const int z = blockIdx.z;
    const uint32_t* __restrict__ pairs = a.p[z].pairs; const long long* __restrict__ sums = (const long long*)a.p[z].sums;
    const int dstride = a.p[z].dstride, pw = a.p[z].pw, ph = a.p[z].ph, unit_size = a.p[z].unit_size, units_x = a.p[z].units_x, units_y = a.p[z].units_y, voff = a.p[z].voff;
    const size_t dplane = a.p[z].dplane;
    const uint32_t ep_mask = a.p[z].ep_mask;
    int32_t* __restrict__ xqd_out = a.p[z].xqd_out; long long* __restrict__ err_out = (long long*)a.p[z].err_out; uint32_t* __restrict__ counters = a.p[z].counters;
    uint8_t* __restrict__ best_ep = a.p[z].best_ep; int32_t* __restrict__ best_xqd = a.p[z].best_xqd; uint32_t* __restrict__ stats = a.p[z].stats;
    const int cap = a.cap;
    const int unit = blockIdx.x, ep = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (unit >= units_x * units_y || !((ep_mask >> ep) & 1)) return;
    const int uj = unit % units_x, ui = unit / units_x;
    const int x0 = uj * unit_size, w = uj == units_x - 1 ? pw - x0 : unit_size;
    const int y0 = ui * unit_size, h = ui == units_y - 1 ? ph - y0 : unit_size;
    const int v0 = max(y0 - voff, 0), v1 = (y0 + h < ph) ? y0 + h - voff : y0 + h;
    const bool has0 = ep < 10 || ep >= 14, has1 = ep < 14;
    const int  ce = ep == 11 ? 2 : (ep == 12 ? 5 : (ep == 13 ? 8 : ep));
    const uint32_t* __restrict__ PP = pairs + (size_t)ce * dplane;
    const long long* S = sums + ((size_t)unit * 16 + ep) * 5;
    const int cw = (w + 7) >> 3, nchunk = cw * (v1 - v0);
    // the one-filter sets are evaluated on a histogram over d as in the kernel above; |d| < 1024 here, so the window always covers the plane's samples and only the
    // escape list can send a unit down the sample-by-sample path (more than kCache listed samples)
    constexpr int kHistW = 1023;
    static_assert((2 * kHistW + 1) * 8 <= (int)sizeof(R.pk), "the histogram lives in the resident chunks' storage");
    const bool try_hist = !(has0 && has1) && a.hist_w >= 0;   // workgroup-uniform
    const int  W = min(kHistW, a.hist_w);

    if (wave == 0) {
        // =================================== the walk: solve, replay, cache (as in sgr_walk_resident_kernel) ===================================
        int start[2] = {0, 0};
        const unsigned long long c0 = __builtin_readcyclecounter();
        unsigned long long c_replay = 0, c_load = 0, c_eval = 0;
        __builtin_amdgcn_s_setprio(3);
        solve_and_encode(S, w * (v1 - v0), ep, start);
        __builtin_amdgcn_s_setprio(0);
        RegStore K;
#pragma unroll
        for (int b = 0; b < kBanks; b++) { K.key[b] = -1; K.err[b] = 0; }
        const bool whole = nchunk <= kResC;   // nothing to stream: a pass is arithmetic only and takes up to kMaxCand points (in groups of NA)
        K.wkey = -1; K.lane = lane; K.n_cache = 0; K.nw = 0; K.cap = whole ? kMaxCand : cap; K.res_x = start[0]; K.res_y = start[1]; K.res_err = -1;
        bool hist = false;
        if (try_hist) { if (lane == 0) L.ovf_n = 0; __syncthreads(); }   // H
        const unsigned long long c1 = __builtin_readcyclecounter();
        int n_pass = 0, n_eval = 0;
        bool fin = false;
        WalkState WS; walk_begin(WS, start);
        const ModelSums MS = load_model(S);
        int cap0 = K.cap;
        for (int pass = 0; pass < kPassBudget; pass++) {
            const unsigned long long r0 = __builtin_readcyclecounter();
            __builtin_amdgcn_s_setprio(3);
            K.cap = K.n_cache >= kThrottle ? 1 : cap0;
            fin = replay<RegStore, false>(K, WS, ep, MS);
            __builtin_amdgcn_s_setprio(0);
            c_replay += __builtin_readcyclecounter() - r0;
            if (try_hist && pass == 0) {   // H2
                __syncthreads();
                hist = ((volatile int&)L.ovf_n) <= kCache;
                if (hist) cap0 = kMaxCand;
            }
            const int nc = K.nw;
            if (lane == 0) { L.done = fin ? 1 : 0; L.n_want = nc; }
            if (lane < nc) {   // svt_decode_xq (Common/Codec/EbRestoration.c:707-718)
                const int x = (K.wkey & 255) - 128, y = (K.wkey >> 8) - 128;
                L.xq0[lane] = has0 ? x : 0;
                L.xq1[lane] = !has1 ? 0 : (has0 ? 128 - x - y : 128 - y);
            }
            const unsigned long long a0 = __builtin_readcyclecounter();
            __syncthreads();   // A
            const unsigned long long a1 = __builtin_readcyclecounter();
            if (pass == 0) c_load = a1 - a0;
            if (fin) break;
            n_pass++; n_eval += nc;
            __syncthreads();   // B
            c_eval += __builtin_readcyclecounter() - a1;
#pragma unroll
            for (int b = 0; b < kBanks; b++) {
                const int c = 64 * b + lane - K.n_cache;
                const int k = __shfl(K.wkey, c & 63, 64);
                if (c >= 0 && c < nc) {
                    long long e = 0;
#pragma unroll
                    for (int v = 1; v < kT / 64; v++) e += L.part[v][c];
                    K.key[b] = k; K.err[b] = e;
                }
            }
            K.n_cache = min(K.n_cache + nc, kCache);
        }
        if (lane == 0) {
            publish_walk(xqd_out, err_out, (size_t)unit * 16 + ep, K.res_x, K.res_y, fin ? K.res_err : -1);
            atomicAdd(&stats[0], (uint32_t)n_pass); atomicAdd(&stats[1], (uint32_t)n_eval); if (!fin) atomicAdd(&stats[2], 1u);
            if (hist) atomicAdd(&stats[3], 1u);
            if (a.clocks) {
                atomicAdd(&stats[24], (uint32_t)((c1 - c0) >> 6)); atomicAdd(&stats[25], (uint32_t)(c_replay >> 6)); atomicAdd(&stats[26], (uint32_t)(c_load >> 6));
                atomicAdd(&stats[27], (uint32_t)(c_eval >> 6)); atomicAdd(&stats[28], (uint32_t)((__builtin_readcyclecounter() - c0) >> 6));
            }
            const uint32_t arrived = atomicAdd(&counters[unit], 1u) + 1u;
            if (arrived == (uint32_t)__popc(ep_mask)) pick_unit_best(xqd_out, err_out, unit, ep_mask, best_ep, best_xqd);
        }
        return;
    }
    // =================================== the samples ===================================
    const int t = tid - 64;
    const int n_esc = (int)a.p[z].esc_cnt[(size_t)unit * 16 + ce];   // uniform
    const uint2* __restrict__ EL = a.p[z].esc + (size_t)(ce < 11 ? ce : ce - 3) * dplane + (size_t)v0 * dstride + (size_t)x0 * (size_t)(v1 - v0);
    auto chunk_off = [&](int k) { const int row = k / cw, cx = k - row * cw; return (size_t)(v0 + row) * dstride + x0 + 8 * cx; };
    if (try_hist) {
        unsigned long long* H = (unsigned long long*)&R.pk[0];
        const int nb = 2 * W + 1;
        for (int b = t; b < nb; b += kResD) H[b] = 0ull;
        __syncthreads();   // H
        unsigned long long r2 = 0;
        auto take = [&](int d, int r) {
            if (abs(d) <= W) atomicAdd(&H[d + W], (1ull << 40) | (unsigned long long)(uint32_t)(r + 1024));
            else { const int at = atomicAdd(&L.ovf_n, 1); if (at < kCache) { L.cx[at] = d; L.cy[at] = r; } }
            r2 += (unsigned long long)(uint32_t)(r * r);
        };
        int k = t;
        int4 w0 = make_int4(0, 0, 0, 0), w1 = w0;
        if (k < nchunk) { const size_t off = chunk_off(k); w0 = SGR_LD4(PP + off); w1 = SGR_LD4(PP + off + 4); }
        while (k < nchunk) {
            const int kn = k + kResD;
            int4 n0 = make_int4(0, 0, 0, 0), n1 = n0;
            if (kn < nchunk) { const size_t off = chunk_off(kn); n0 = SGR_LD4(PP + off); n1 = SGR_LD4(PP + off + 4); }
            const int cx = k % cw, n = min(8, w - 8 * cx);
            const int wd[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < n) take(has0 ? __builtin_amdgcn_sbfe(wd[i], 5, 11) : (wd[i] >> 21), packed_r(wd[i]));
            w0 = n0; w1 = n1; k = kn;
        }
        // the listed samples: their plane words are zero (d = 0, dat - src = 0: they only raised bin 0's count -- harmless, q(0) = 0); the exact values join now
        for (int o = t; o < n_esc; o += kResD) { const uint2 v = EL[o]; take(has0 ? (int)(int16_t)(v.x & 0xFFFFu) : ((int)v.x >> 16), (int)v.y); }
        __syncthreads();   // H2
        const int n_ovf = ((volatile int&)L.ovf_n);
        if (n_ovf <= kCache) {
            const long long r2w = wave_sum_u48((long long)r2);
            for (int pass = 0; pass < kPassBudget; pass++) {
                __syncthreads();   // A
                if (L.done) break;
                const int nc = L.n_want;
                for (int c = 0; c < nc; c++) {
                    const int xq = has0 ? L.xq0[c] : L.xq1[c];
                    long long acc = 0;
                    for (int b = t; b < nb; b += kResD) {
                        const unsigned long long hh = H[b];
                        const int nd = (int)(hh >> 40);
                        if (nd) {
                            const long long Rd = (long long)(hh & ((1ull << 40) - 1)) - 1024ll * nd;
                            const int q = (xq * (b - W) + 1024) >> 11;
                            acc += (long long)nd * (q * q) + 2ll * q * Rd;
                        }
                    }
                    for (int o = t; o < n_ovf; o += kResD) {
                        const int q = (xq * L.cx[o] + 1024) >> 11;
                        acc += (long long)q * q + 2ll * q * L.cy[o];
                    }
                    const long long sum = wave_sum_i64(acc);
                    if (lane == 0) L.part[wave][c] = sum + r2w;
                }
                __syncthreads();   // B
            }
            return;
        }
        // (a unit with more listed samples than that: nobody reads the histogram again, its storage takes resident chunks below)
    }
    // ---- the resident part: ALL of its loads are issued before the first one is consumed (unconditional, clamped addresses: a load under a condition is serialised).
    // The kL chunks that live in LDS go there directly (global_load_lds_dwordx4: a wave's 64 lanes x 16 bytes land contiguously at a wave-uniform base), so they cost no
    // registers while in flight; the kJ chunks of the registers follow.  A chunk past the unit's end is skipped by its thread in the sweeps (`mine`), a chunk that
    // straddles the unit's right edge (unit widths are multiples of eight except in the last column of a picture whose width is not) is patched after the loads landed.
    int4 pa[kJ], pb[kJ];
    {
        const int wbase = (wave - 1) * 64;
#pragma unroll
        for (int l = 0; l < kL; l++) {
            const uint32_t* g = PP + chunk_off(min(t + (kJ + l) * kResD, max(nchunk - 1, 0)));
            __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)&R.pk[(l * 2) * kResD + wbase], 16, 0, 0);
            __builtin_amdgcn_global_load_lds(g + 4, (__attribute__((address_space(3))) void*)&R.pk[(l * 2 + 1) * kResD + wbase], 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < kJ; j++) { const size_t off = chunk_off(min(t + j * kResD, max(nchunk - 1, 0))); pa[j] = SGR_LD4(PP + off); pb[j] = SGR_LD4(PP + off + 4); }
#pragma unroll
        for (int j = 0; j < kJ; j++) {
            const int k = t + j * kResD, kc = min(k, max(nchunk - 1, 0)), cx = kc % cw;
            const int n = k < nchunk ? w - 8 * cx : 0;
            if (n < 8) mask_words(pa[j], pb[j], n);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the direct loads have landed (this thread's slots are read back by this thread only)
        if ((w & 7) != 0) {
#pragma unroll
            for (int l = 0; l < kL; l++) {
                const int k = t + (kJ + l) * kResD;
                if (k < nchunk && w - 8 * (k % cw) < 8) {
                    int4 l0 = R.pk[(l * 2) * kResD + t], l1 = R.pk[(l * 2 + 1) * kResD + t];
                    mask_words(l0, l1, w - 8 * (k % cw));
                    R.pk[(l * 2) * kResD + t] = l0; R.pk[(l * 2 + 1) * kResD + t] = l1;
                }
            }
        }
    }
    int sel;
    asm volatile("s_mov_b32 %0, 0x07060302" : "=s"(sel));
    for (int pass = 0; pass < kPassBudget; pass++) {
        __syncthreads();   // A
        if (L.done) break;
        const int nc = L.n_want;
        const int qv = lane < nc ? (int)(((uint32_t)L.xq0[lane] & 0xFFFFu) | ((uint32_t)L.xq1[lane] << 16)) : 0;   // lane c: both taps of candidate c, UNSCALED (the pair carries the 32)
        const unsigned long long e0 = __builtin_readcyclecounter();
        for (int cb = 0; cb < nc; cb += NA) {   // a unit that is resident as a whole may ask for up to kMaxCand points: groups of NA
            const int ng = min(nc - cb, NA);
            int pp[NA], qq[NA];
#pragma unroll
            for (int c = 0; c < NA; c++) { pp[c] = 0; qq[c] = __builtin_amdgcn_readlane(qv, (cb + c) & 63); }   // qq: scalar registers
            auto sweep = [&](const int4& w0, const int4& w1) {
                int4 x0v, x1v; int sx[8];
                unpack_chunk(w0, w1, x0v, x1v, sx);
#pragma unroll
                for (int c = 0; c < NA; c++)
                    if (c < ng) eval_chunk_f1s(x0v, x1v, sx, qq[c], sel, pp[c]);
            };
            // ---- the streamed part first (next chunk's loads in flight while this one is evaluated)
            int k = t + kResC;
            int4 w0 = make_int4(0, 0, 0, 0), w1 = w0;
            auto fetch = [&](int kk, int4& f0, int4& f1) {
                const int row = kk / cw, cx = kk - row * cw;
                const size_t off = (size_t)(v0 + row) * dstride + x0 + 8 * cx;
                f0 = SGR_LD4(PP + off); f1 = SGR_LD4(PP + off + 4);
                const int n = w - 8 * cx;
                if (n < 8) mask_words(f0, f1, n);
            };
            if (k < nchunk) fetch(k, w0, w1);
            while (k < nchunk) {
                const int kn = k + kResD;
                int4 n0 = make_int4(0, 0, 0, 0), n1 = n0;
                if (kn < nchunk) fetch(kn, n0, n1);
                sweep(w0, w1);
                w0 = n0; w1 = n1; k = kn;
            }
            // ---- the chunks resident in LDS, then the ones in registers
#pragma unroll
            for (int l = 0; l < kL; l++) {
                if ((kJ + l) * kResD < nchunk) {   // uniform: the slot is in use
                    const bool mine = t + (kJ + l) * kResD < nchunk;
                    int4 l0 = R.pk[(l * 2) * kResD + t], l1 = R.pk[(l * 2 + 1) * kResD + t];
                    if (!mine) { l0 = make_int4(0, 0, 0, 0); l1 = l0; }   // a clamped load's copy of the last chunk: zero words add nothing
                    sweep(l0, l1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < kJ; j++) {
                if (j * kResD < nchunk) sweep(pa[j], pb[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            dot_drain();
            // ---- the listed samples, one by one (|e| <= 1010: the squares of a thread's share stay far below 2^63)
            long long le[NA];
#pragma unroll
            for (int c = 0; c < NA; c++) le[c] = 0;
            for (int o = t; o < n_esc; o += kResD) {
                const uint2 v = EL[o];
                const int d0 = (int)(int16_t)(v.x & 0xFFFFu), d1 = (int)v.x >> 16, rr = (int)v.y;
#pragma unroll
                for (int c = 0; c < NA; c++)
                    if (c < ng) { const int e = ((L.xq0[cb + c] * d0 + L.xq1[cb + c] * d1 + 1024) >> 11) + rr; le[c] += (long long)(e * e); }
            }
#pragma unroll
            for (int c = 0; c < NA; c++)
                if (c < ng) {
                    const long long sum = wave_sum_u48(le[c] + (long long)(uint32_t)pp[c]);
                    if (lane == 0) L.part[wave][cb + c] = sum;
                }
        }
        if (tid == 64) { if (a.clocks) { atomicAdd(&stats[29], (uint32_t)((__builtin_readcyclecounter() - e0) >> 6)); atomicAdd(&stats[30], (uint32_t)nc); } if (pass == 0 && n_esc) atomicAdd(&stats[4], (uint32_t)n_esc); }   // [4]: listed samples the sample-by-sample walks of the plane added (diagnostics, tests)
        __syncthreads();   // B
    }
}

}  // namespace

extern "C" size_t svt_hip_sgr_walk_state_bytes(int n_units) { return sizeof(uint32_t) * (size_t)n_units; }   // arrival counter per unit

// planes[i]: the arguments of svt_hip_launch_sgr_walk for plane i; one launch for all of them (resident / hybrid forms), one per plane (streamed form)
extern "C" int svt_hip_launch_sgr_walk_multi(hipStream_t st, int bd, int n_planes, const SvtHipSgrWalkPlane* planes) {
    if (n_planes < 1 || n_planes > kWalkMaxPlanes) return (int)hipErrorInvalidValue;
    // SVT_HIP_SGR_WALK = stream | resident | hybrid selects the form (A/B measurements, tools/hbd_time.py); default: hybrid
    static const char* form_env = getenv("SVT_HIP_SGR_WALK");
    const bool stream_form = form_env && !strcmp(form_env, "stream"), resident_form = form_env && !strcmp(form_env, "resident");
    static const int  cap_env = getenv("SVT_HIP_SGR_WALK_CAND") ? atoi(getenv("SVT_HIP_SGR_WALK_CAND")) : 0;
    constexpr int kHybT = 512, kHybJ = 7, kHybNA = 8;
    constexpr int kHybNA10 = 7;   // bit depth 10, DRAINING instances (256-thread forms, and SVT_HIP_SGR_WALK_NA10=7 for A/B runs: the round-3 default): an int32 accumulator + a 64-bit sum per candidate; 7 spill 20 registers outside the loops and still win (MI355X, configs[3] unit search: 5: 2.02-2.03, 6: 1.95, 7: 1.90 ms; 8 spill 51).  The default since round 4 is the bit-depth-8 form itself (eight candidates, two accumulators, no drains: the range argument above sgr_walk_resident_kernel)
    static const bool hyb16 = form_env && !strcmp(form_env, "hybrid16");   // opt-in: sixteen points per pass (one accumulator each) + hedged requests, bit depth 8
    static const int na10_env = getenv("SVT_HIP_SGR_WALK_NA10") ? atoi(getenv("SVT_HIP_SGR_WALK_NA10")) : 0;   // A/B: 7 = the draining seven-candidate instance of round 3
    static const bool small_form = form_env && (!strcmp(form_env, "hybrid256") || !strcmp(form_env, "hybrid256j"));
    const int na10 = (na10_env == kHybNA10 || small_form) ? kHybNA10 : kHybNA;
    const int cap_max = stream_form ? kStreamCand : (resident_form ? kMaxCand : (bd == 8 ? (hyb16 ? 16 : kHybNA) : na10));
    const int cap = cap_env >= 1 && cap_env <= cap_max ? cap_env : (stream_form ? kStreamCand : (resident_form ? 12 : cap_max));   // candidates per pass
    static const bool hist_off = getenv("SVT_HIP_SGR_WALK_HIST") && !atoi(getenv("SVT_HIP_SGR_WALK_HIST"));   // A/B: one-filter sets evaluated sample by sample like the others (round 3)
    WalkPic a = {};
    a.cap = cap;
    const char* hw_env = getenv("SVT_HIP_SGR_WALK_HIST_W");   // read per launch: tests/test_sgr_gpu.py narrows the window to send part of a plane's units down the sample-by-sample path
    a.hist_w = hist_off ? -1 : (hw_env ? atoi(hw_env) : 1 << 30);
    static const char* clk_env = getenv("SVT_HIP_SGR_WALK_CLOCKS");   // diagnostics, off by default since round 6 (the phase clocks cost 1.2 % of the stage: 0.923 -> 0.912 ms per 4K frame); =1: stats[24..30]
    a.clocks = clk_env && clk_env[0] == '1';
    int max_units = 0;
    for (int i = 0; i < n_planes; i++) {
        const SvtHipSgrWalkPlane& P = planes[i];
        const int nu = P.units_x * P.units_y;
        uint32_t* counters = (uint32_t*)P.states;
        // the arrival counters are zero: the caller clears the scratch up to the difference planes (svt_hip_api.cpp)
        a.p[i] = WalkPlane{P.pairs, P.sd, P.sums, P.dplane, P.dstride, P.pw, P.ph, P.unit_size, P.units_x, P.units_y, 8 >> P.ss_y, P.ep_mask,
                           P.xqd_out, P.err_out, counters, P.best_ep, P.best_xqd, P.stats, (const uint2*)P.esc, P.esc_cnt};
        if ((P.esc != nullptr) != (planes[0].esc != nullptr)) return (int)hipErrorInvalidValue;   // one form per launch
        max_units = nu > max_units ? nu : max_units;
    }
    if (planes[0].esc) {   // packed difference words (bit depth 8, sgr.hip STORE == 2): 512 threads, 7 chunks per data thread in registers + 5 in LDS, eight points per pass
        if (bd != 8) return (int)hipErrorInvalidValue;
        static const char* pk_env = getenv("SVT_HIP_SGR_WALK_PACKED");   // A/B: "j6l5", "j7l5": more resident chunks per data thread in registers (they spill)
        a.cap = cap_env >= 1 && cap_env <= kHybNA ? cap_env : kHybNA;
        dim3 grid(max_units, 16, n_planes);
        if (pk_env && !strcmp(pk_env, "j7l5")) hipLaunchKernelGGL((sgr_walk_packed_kernel<512, 7, 5, kHybNA>), grid, dim3(512), 0, st, a);
        else if (pk_env && !strcmp(pk_env, "j6l5")) hipLaunchKernelGGL((sgr_walk_packed_kernel<512, 6, 5, kHybNA>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((sgr_walk_packed_kernel<512, 5, 5, kHybNA>), grid, dim3(512), 0, st, a);   // five in registers: the form that does not spill (six: 28 registers, seven: 44)
        return (int)hipGetLastError();
    }
    if (stream_form) {
        for (int i = 0; i < n_planes; i++) {
            const WalkPlane& W = a.p[i];
            dim3 grid(W.units_x * W.units_y, 16);
#define WALK_ARGS W.pairs, W.sd, W.dstride, W.dplane, (const long long*)W.sums, W.pw, W.ph, W.unit_size, W.units_x, W.units_y, W.voff, W.ep_mask, W.xqd_out, (long long*)W.err_out, W.counters, W.best_ep, W.best_xqd, W.stats, cap
            if (bd == 8) hipLaunchKernelGGL((sgr_walk_kernel<8>), grid, dim3(256), 0, st, WALK_ARGS);
            else hipLaunchKernelGGL((sgr_walk_kernel<10>), grid, dim3(256), 0, st, WALK_ARGS);
#undef WALK_ARGS
        }
        return (int)hipGetLastError();
    }
    dim3 grid(max_units, 16, n_planes);
    static const bool hyb256 = form_env && !strcmp(form_env, "hybrid256"), hyb256j = form_env && !strcmp(form_env, "hybrid256j");
    static const bool hybpf5 = form_env && !strcmp(form_env, "hybridpf5");
    if (hyb16 && bd == 8) {   // experiment (MI355X, profiles/r03/sgr_walk_hybrid16_ab.txt: 2.03 instead of 2.79 passes, 15.2 instead of 10.9 points per walk; stage 1.012 -> 0.993 ms, four-frame step 8.61 -> 8.76 ms: not the default)
        hipLaunchKernelGGL((sgr_walk_resident_kernel<8, kHybT, kHybJ, 16>), grid, dim3(kHybT), 0, st, a);
        return (int)hipGetLastError();
    }
    if (hybpf5 && bd == 8) {   // experiment (MI355X: 1.190 vs 1.163 ms, no gain): five resident chunks instead of seven, two streamed chunks in flight
        hipLaunchKernelGGL((sgr_walk_resident_kernel<8, kHybT, 5, kHybNA, 2>), dim3(max_units, 16, n_planes), dim3(kHybT), 0, st, a);
        return (int)hipGetLastError();
    }
    if (hyb256 || hyb256j) {   // experiment: four smaller workgroups per compute unit (one walker + three data waves each)
        if (hyb256) {
            if (bd == 8) hipLaunchKernelGGL((sgr_walk_resident_kernel<8, 256, kHybJ, kHybNA>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((sgr_walk_resident_kernel<10, 256, kHybJ, kHybNA10, 1, true>), grid, dim3(256), 0, st, a);
        } else {
            if (bd == 8) hipLaunchKernelGGL((sgr_walk_resident_kernel<8, 256, 9, kHybNA>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((sgr_walk_resident_kernel<10, 256, 9, kHybNA10, 1, true>), grid, dim3(256), 0, st, a);
        }
        return (int)hipGetLastError();
    }
    if (resident_form) {
        if (bd == 8) hipLaunchKernelGGL((sgr_walk_resident_kernel<8, kResT, kResJ, 0>), grid, dim3(kResT), 0, st, a);
        else hipLaunchKernelGGL((sgr_walk_resident_kernel<10, kResT, kResJ, 0>), grid, dim3(kResT), 0, st, a);
    } else {
        if (bd == 8) hipLaunchKernelGGL((sgr_walk_resident_kernel<8, kHybT, kHybJ, kHybNA>), grid, dim3(kHybT), 0, st, a);
        else if (na10 == kHybNA10) hipLaunchKernelGGL((sgr_walk_resident_kernel<10, kHybT, kHybJ, kHybNA10, 1, true>), grid, dim3(kHybT), 0, st, a);
        else hipLaunchKernelGGL((sgr_walk_resident_kernel<10, kHybT, kHybJ, kHybNA>), grid, dim3(kHybT), 0, st, a);
    }
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_sgr_walk(hipStream_t st, int bd, const uint32_t* pairs, const int16_t* sd, int dstride, size_t dplane, const int64_t* sums,
                                       const int64_t* d2, void* states, int pw, int ph, int unit_size, int units_x, int units_y, int ss_y, uint32_t ep_mask,
                                       int32_t* xqd_out, int64_t* err_out, uint8_t* best_ep, int32_t* best_xqd, uint32_t* stats) {
    (void)d2;
    const SvtHipSgrWalkPlane P = {pairs, sd, sums, states, dplane, dstride, pw, ph, unit_size, units_x, units_y, ss_y, ep_mask, xqd_out, err_out, best_ep, best_xqd, stats, nullptr, nullptr};
    return svt_hip_launch_sgr_walk_multi(st, bd, 1, &P);
}

SVT_HIP_TU_PROBE(sgr_walk)
