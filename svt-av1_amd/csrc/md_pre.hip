// md_pre.hip — mode decision, picture-level precompute ("MD candidate batching").
//
// What the reference does, one block and one candidate at a time (EbProductCodingLoop.c:907 fast_loop_core, called by md_stage_0 :1461 for every candidate
// of every block of every superblock): predict the candidate (EbEncInterPrediction.c:6178 inter_pu_prediction_av1 -> :4040 av1_inter_prediction ->
// svt_inter_predictor) and measure its luma distortion against the source (svt_nxm_sad_kernel_sub_sampled, :953 — despite its name the plain SAD of all rows).
// In the first partitioning pass at presets above M4 the open-loop ME vectors go into stage 0 unrefined (EbEncDecProcess.c:3050-3093: md_sq / nsq / pme /
// sub-pel search levels 0), i.e. every ME candidate is a FULL-PEL, single-reference translation whose prediction is a copy of the reference block
// (svt_av1_convolve_2d_copy_sr).  None of that depends on a neighbouring block: the vectors are the open-loop ME's (known before the picture's mode decision
// starts), the reference pictures are complete, the source is the input.  So ONE launch per picture computes the distortion of every (superblock, square PU,
// reference picture) triple; the patched fast_loop_core reads it from a table (integration/svt_hip_md_bridge.c, hook "md_pre").
//
// Layout: one 256-thread workgroup per (superblock, reference picture); its four waves take the PUs of the list round-robin.  A wave covers a PU with one
// dword (four samples) per lane and iteration — lanes (w / 4) per row, 64 / (w / 4) rows per iteration — on v_sad_u8; neither plane has to be aligned
// (two aligned dwords + v_alignbyte).  The superblock's 4 KB of source and the <= 85 reference blocks around it stay in the L1 / L2 of the workgroup's CU.
// HBM-bound in principle (algorithmic bytes per (SB, reference): 4096 source + ~4 x 4096 reference samples + 85 x 8 table bytes), launch-bound in practice:
// a 1080p picture with 7 references is 3 570 workgroups of ~30 iterations.
#include "svt_hip_internal.h"
#include "lds_stage.h"
#include "interp_kernels.h"

namespace {

struct RefPlanes { SvtHipMdRefPlane r[SVT_HIP_MD_MAX_REFS]; };
struct PuList { uint8_t x[SVT_HIP_MD_MAX_PUS], y[SVT_HIP_MD_MAX_PUS], w[SVT_HIP_MD_MAX_PUS], h[SVT_HIP_MD_MAX_PUS]; };

// four consecutive samples starting at any byte address
__device__ __forceinline__ uint32_t load4_any(const uint8_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t lo = q[0], hi = q[1];
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(a & 3));
}

// four consecutive 16-bit samples starting at any even byte address: {lo = samples 0, 1; hi = samples 2, 3}
struct U16x4 { uint32_t lo, hi; };
__device__ __forceinline__ U16x4 load4_any16(const uint16_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    const uint32_t sh = (uint32_t)(a & 2);
    return U16x4{__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh)};
}
__device__ __forceinline__ uint32_t sad4(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
__device__ __forceinline__ uint32_t sad4(U16x4 a, U16x4 b, uint32_t acc) { return __builtin_amdgcn_sad_u16(a.hi, b.hi, __builtin_amdgcn_sad_u16(a.lo, b.lo, acc)); }
__device__ __forceinline__ uint32_t ld4(const uint8_t* p) { return load4_any(p); }
__device__ __forceinline__ U16x4 ld4(const uint16_t* p) { return load4_any16(p); }
__device__ __forceinline__ uint32_t avg2_round_up16(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7fff7fffu); }   // per 16-bit half (a + b + 1) >> 1

__device__ __forceinline__ uint32_t row_sum(uint32_t x) {      // every lane: the total of its 16-lane row
    int v = (int)x;
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);   // row_mirror
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t rows_total(uint32_t v) {   // the four row totals of a wave added up (wave-uniform)
    return (uint32_t)(__builtin_amdgcn_readlane((int)v, 0) + __builtin_amdgcn_readlane((int)v, 16) + __builtin_amdgcn_readlane((int)v, 32) + __builtin_amdgcn_readlane((int)v, 48));
}

// PIX = uint8_t: the 8-bit planes; uint16_t: the 16-bit planes a 10-bit encode's mode decision works on (sad_16b_kernel, Encoder/C_DEFAULT/EbComputeSAD_C.c:39; strides and the
// reference boxes in samples, d_plane = sample (0, 0) of the 16-bit plane)
template <typename PIX>
__global__ void __launch_bounds__(256)
md_fullpel_sad_kernel(const PIX* __restrict__ src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_pus, int n_refs, RefPlanes refs, PuList pus,
                      const uint32_t* __restrict__ mv, uint32_t* __restrict__ sad) {
    const int sb = blockIdx.x, r = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sb_x = (sb % sb_cols) * 64, sb_y = (sb / sb_cols) * 64;
    const SvtHipMdRefPlane ref = refs.r[r];
    for (int pu = wave; pu < n_pus; pu += 4) {
        const size_t slot = ((size_t)sb * n_pus + pu) * n_refs + r;
        const int w = pus.w[pu], h = pus.h[pu], x = sb_x + pus.x[pu], y = sb_y + pus.y[pu];
        const uint32_t m = mv[slot];
        const int mx = (int16_t)(m & 0xffff), my = (int16_t)(m >> 16);
        const int rx = x + mx, ry = y + my;
        // no vector, a PU that leaves the picture, a reference block that leaves the plane's allocation (the last four samples of a row pair are read as whole
        // dwords): the slot says "not computed"
        const bool ok = mx != SVT_HIP_MD_NO_MV && x + w <= pic_w && y + h <= pic_h && rx >= ref.x_min && ry >= ref.y_min && rx + w + 4 <= ref.x_max && ry + h <= ref.y_max;
        if (!ok) {   // wave-uniform
            if (lane == 0) sad[slot] = 0xffffffffu;
            continue;
        }
        const int n4 = w >> 2, rows = 64 / n4, row = lane / n4, c4 = (lane - row * n4) << 2;
        const PIX* ps = src + (ptrdiff_t)y * src_stride + x + c4;
        const PIX* pr = (const PIX*)ref.d_plane + (ptrdiff_t)ry * ref.stride + rx + c4;
        uint32_t s = 0;
        for (int r0 = 0; r0 < h; r0 += 4 * rows) {   // four row groups' loads in flight (unconditional, clamped rows: a load under a condition is a branch that waits for it)
            decltype(ld4(ps)) a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int yy = min(r0 + u * rows + row, h - 1);
                a[u] = ld4(ps + (ptrdiff_t)yy * src_stride); b[u] = ld4(pr + (ptrdiff_t)yy * ref.stride);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t t = sad4(a[u], b[u], s); s = r0 + u * rows + row < h ? t : s; }
        }
        s = rows_total(row_sum(s));
        if (lane == 0) sad[slot] = s;
    }
}

// ---- the same for the COMPOUND-AVERAGE candidates mode decision makes of two ME vectors (NEW_NEWMV from the open-loop ME's bi-directional candidates, EbModeDecision.c:3408-3540;
// MD_COMP_AVG: interinter_comp.type = COMPOUND_AVERAGE, compound_idx = 1): both predictions are full-pel copies in the compound domain (svt_av1_jnt_convolve_2d_copy:
// (sample << 4) + offset), the second call averages and rounds back: ((a << 4) + (b << 4)) >> 1 rounded by 4 bits = (a + b + 1) >> 1.  One workgroup per (superblock, pair of
// table columns); the pair's two vectors are the columns' own entries of the same PU.
struct PairList { uint8_t c0[SVT_HIP_MD_MAX_PAIRS], c1[SVT_HIP_MD_MAX_PAIRS]; };
__device__ __forceinline__ uint32_t avg_round_up(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }   // per byte (a + b + 1) >> 1
__device__ __forceinline__ U16x4 avg_round_up(U16x4 a, U16x4 b) { return U16x4{avg2_round_up16(a.lo, b.lo), avg2_round_up16(a.hi, b.hi)}; }   // (the high-bit-depth compound copy rounds the same way: svt_av1_highbd_jnt_convolve_2d_copy)
template <typename PIX>
__global__ void __launch_bounds__(256)
md_fullpel_avg_sad_kernel(const PIX* __restrict__ src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_pus, int n_refs, RefPlanes refs, PuList pus, int n_pairs,
                          PairList pairs, const uint32_t* __restrict__ mv, uint32_t* __restrict__ sad) {
    const int sb = blockIdx.x, pr_i = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sb_x = (sb % sb_cols) * 64, sb_y = (sb / sb_cols) * 64;
    const int c0 = pairs.c0[pr_i], c1 = pairs.c1[pr_i];
    const SvtHipMdRefPlane ref0 = refs.r[c0], ref1 = refs.r[c1];
    for (int pu = wave; pu < n_pus; pu += 4) {
        const size_t base = ((size_t)sb * n_pus + pu) * n_refs, slot = ((size_t)sb * n_pus + pu) * n_pairs + pr_i;
        const int w = pus.w[pu], h = pus.h[pu], x = sb_x + pus.x[pu], y = sb_y + pus.y[pu];
        const uint32_t m0 = mv[base + c0], m1 = mv[base + c1];
        const int mx0 = (int16_t)(m0 & 0xffff), my0 = (int16_t)(m0 >> 16), mx1 = (int16_t)(m1 & 0xffff), my1 = (int16_t)(m1 >> 16);
        const int rx0 = x + mx0, ry0 = y + my0, rx1 = x + mx1, ry1 = y + my1;
        const bool ok = mx0 != SVT_HIP_MD_NO_MV && mx1 != SVT_HIP_MD_NO_MV && x + w <= pic_w && y + h <= pic_h &&
                        rx0 >= ref0.x_min && ry0 >= ref0.y_min && rx0 + w + 4 <= ref0.x_max && ry0 + h <= ref0.y_max &&
                        rx1 >= ref1.x_min && ry1 >= ref1.y_min && rx1 + w + 4 <= ref1.x_max && ry1 + h <= ref1.y_max;
        if (!ok) {   // wave-uniform
            if (lane == 0) sad[slot] = 0xffffffffu;
            continue;
        }
        const int n4 = w >> 2, rows = 64 / n4, row = lane / n4, c4 = (lane - row * n4) << 2;
        const PIX* ps = src + (ptrdiff_t)y * src_stride + x + c4;
        const PIX* p0 = (const PIX*)ref0.d_plane + (ptrdiff_t)ry0 * ref0.stride + rx0 + c4;
        const PIX* p1 = (const PIX*)ref1.d_plane + (ptrdiff_t)ry1 * ref1.stride + rx1 + c4;
        uint32_t s = 0;
        for (int r0 = 0; r0 < h; r0 += 2 * rows) {
            decltype(ld4(ps)) a[2], b[2], c[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int yy = min(r0 + u * rows + row, h - 1);
                a[u] = ld4(ps + (ptrdiff_t)yy * src_stride); b[u] = ld4(p0 + (ptrdiff_t)yy * ref0.stride); c[u] = ld4(p1 + (ptrdiff_t)yy * ref1.stride);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) { const uint32_t t = sad4(a[u], avg_round_up(b[u], c[u]), s); s = r0 + u * rows + row < h ? t : s; }
        }
        s = rows_total(row_sum(s));
        if (lane == 0) sad[slot] = s;
    }
}

// ---- the sub-pel refinement's probes (md_subpel_search, EbProductCodingLoop.c:2063 -> svt_av1_find_best_sub_pixel_tree, mcomp.c:350): every probe is
// svt_upsampled_pref_error (:102) = svt_aom_upsampled_pred (C_DEFAULT/variance.c:212-269: an 8-tap horizontal pass and an 8-tap vertical pass, each rounded and clipped to
// 8 bits) + the block size's variance function against the source.  The tree starts at the block's full-pel vector and visits, in its half-pel and quarter-pel rounds,
// positions inside the 7 x 7 quarter-pel grid of +-6/8 sample around it: one workgroup per (superblock, PU, reference picture) computes (variance, sse) of ALL 49 grid
// positions — the probes of both rounds whichever way the comparisons go.  The passes are separable with an 8-bit intermediate, so the horizontal pass runs once per
// horizontal offset (7) over the PU's window and the vertical pass + statistics once per grid position (49), both as v_dot4_i32_i8 on four outputs per lane
// (samples biased by -128; the taps of the non-zero phases fit int8 and sum to 128: the bias is added back exactly).  LDS: the window, the seven intermediates
// (column-major: the vertical pass reads its eleven rows as three dwords) and the transposed source: 41 KB for a 64x64 PU.
__device__ __forceinline__ uint32_t bytes4(uint32_t d0, uint32_t d1, uint32_t d2, int sh) {   // bytes sh .. sh + 3 of the twelve bytes d0 d1 d2 (sh <= 8)
    return sh < 4 ? __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)sh) : (sh < 8 ? __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)(sh - 4)) : d2);
}
__device__ __forceinline__ uint32_t filt4(uint32_t d0, uint32_t d1, uint32_t d2, int o, int TA, int TB) {
    // four outputs of an 8-tap pass: output q reads bytes o + q .. o + q + 7 of (d0 d1 d2) ^ 0x80; result packed as four bytes
    uint32_t P = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int sum = 128 * 128 + 64;
        sum = __builtin_amdgcn_sdot4((int)bytes4(d0, d1, d2, o + q), TA, sum, false);
        sum = __builtin_amdgcn_sdot4((int)(o + q + 4 <= 8 ? bytes4(d0, d1, d2, o + q + 4) : 0u), TB, sum, false);
        int v = min(max(sum >> 7, 0), 255);
        // Without this the compiler (ROCm 7.2 clang, gfx950) fuses shift + clamp + pack of two outputs into v_ashr_pk_u8_i32, whose result on the MI355X had bit 7 set
        // in the bytes it produced (constant input 100 came out as 228; the other two outputs of the same lane, clamped with v_med3_i32, were right): the empty
        // asm keeps the clamp a v_med3_i32 for all four.  tests/test_md_pre_gpu.py::test_md_subpel_grid_picture is what noticed.
        asm volatile("" : "+v"(v));
        P |= (uint32_t)v << (8 * q);
    }
    return P;
}
__device__ __constant__ int8_t kGridOff7[7][2] = {{-1, 2}, {-1, 4}, {-1, 6}, {0, 0}, {0, 2}, {0, 4}, {0, 6}};   // grid offset -6 .. 6 (1/8 sample) = whole part, eighth-pel phase
__device__ __constant__ int8_t kGridOff3[3][2] = {{-1, 4}, {0, 0}, {0, 4}};                                         // the half-pel round alone: offsets -4, 0, 4
// kGrid = 7: the 7 x 7 quarter-pel grid (both rounds of the tree); kGrid = 3: the 3 x 3 half-pel grid (svt_first_level_check's eight neighbours + the centre)
// SMAX / NT: the largest PU a launch handles and its workgroup size — 8x8 and 16x16 PUs (80 of the 85 square PUs of a superblock) run as one wave with 3.5 KB of LDS,
// the larger ones as four waves; `sel` lists the PUs of the launch.
struct PuSel { uint8_t idx[SVT_HIP_MD_MAX_PUS]; };
template <int kGrid, int SMAX, int NT>
__global__ void __launch_bounds__(NT)
md_subpel_grid_kernel(const uint8_t* __restrict__ src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_pus, int n_refs, RefPlanes refs, PuList pus, PuSel sel,
                      const uint32_t* __restrict__ mv, int bank, uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t win[(SMAX + 8) * (SMAX + 8) + 16];   // [row][col], row pitch WIN
    constexpr int kGridN = kGrid * kGrid;
    const int8_t (*kGridOff)[2] = kGrid == 7 ? kGridOff7 : kGridOff3;
    __shared__ __attribute__((aligned(16))) uint8_t Hc[kGrid][SMAX * (SMAX + 8) + 16];  // [offset][col][row], column pitch WIN
    __shared__ __attribute__((aligned(16))) uint8_t Sc[SMAX * SMAX];       // source, [col][row]
    __shared__ uint32_t stat[kGridN][3];                                   // sum p, sum p^2, sum p s
    __shared__ uint32_t sstat[2];                                          // sum s, sum s^2
    const int sb = blockIdx.x, pu = sel.idx[blockIdx.y], r = blockIdx.z, tid = threadIdx.x;
    const size_t slot = ((size_t)sb * n_pus + pu) * n_refs + r;
    uint32_t* o = out + slot * (2 * kGridN);
    const int s = pus.w[pu], x = (sb % sb_cols) * 64 + pus.x[pu], y = (sb / sb_cols) * 64 + pus.y[pu], WIN = s + 8;
    const SvtHipMdRefPlane ref = refs.r[r];
    const uint32_t m = mv[slot];
    const int mx = (int16_t)(m & 0xffff), my = (int16_t)(m >> 16), wx = x + mx - 4, wy = y + my - 4;
    const bool ok = mx != SVT_HIP_MD_NO_MV && pus.h[pu] == s && s <= SMAX && (s == 8 || s == 16 || s == 32 || s == 64) && x + s <= pic_w && y + s <= pic_h && wx >= ref.x_min && wy >= ref.y_min &&
                    wx + WIN + 4 <= ref.x_max && wy + WIN <= ref.y_max;
    if (!ok) {   // workgroup-uniform
        for (int i = tid; i < 2 * kGridN; i += NT) o[i] = 0xffffffffu;
        return;
    }
    for (int i = tid; i < kGridN * 3 + 2; i += NT) { if (i < kGridN * 3) stat[i / 3][i % 3] = 0; else sstat[i - kGridN * 3] = 0; }
    // the packed taps of the three non-zero phases (2/8, 4/8, 6/8 sample), workgroup-uniform: scalar loads once instead of eight table reads per task
    int TAp[3], TBp[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int16_t* kp = kInterp[bank][4 * k + 4];
        TAp[k] = (kp[0] & 0xff) | (kp[1] & 0xff) << 8 | (kp[2] & 0xff) << 16 | (kp[3] & 0xff) << 24;
        TBp[k] = (kp[4] & 0xff) | (kp[5] & 0xff) << 8 | (kp[6] & 0xff) << 16 | (kp[7] & 0xff) << 24;
    }
    const int   sq = s >> 2, lq = 31 - __clz(sq);                 // s / 4 is a power of two
    const float r_hw = 1.0f / (float)(WIN * sq), r_per = 1.0f / (float)(s * sq);   // exact quotients for the task counts here (< 2^16, divisors <= 1152)
    // ---- stage the window (rows as dwords) and the source (transposed)
    const int   wdw = WIN >> 2;
    const float r_wdw = 1.0f / (float)wdw;
    batched_stage<4, uint32_t>(WIN * wdw, tid, NT,   // four window dwords in flight per lane
        [&](int i) { const int rr = (int)(((float)i + 0.5f) * r_wdw), cd = i - rr * wdw; return load4_any(ref.d_plane + (ptrdiff_t)(wy + rr) * ref.stride + wx + 4 * cd); },
        [&](int i, uint32_t v) { ((uint32_t*)win)[i] = v; });
    __syncthreads();   // (also orders the zeroing of the statistics before their first update)
    {
        uint32_t ss = 0, ss2 = 0;
        for (int i = tid; i < s * (s >> 2); i += NT) {
            const int rr = i / (s >> 2), c = (i - rr * (s >> 2)) << 2;
            const uint32_t v = load4_any(src + (ptrdiff_t)(y + rr) * src_stride + x + c);
#pragma unroll
            for (int q = 0; q < 4; q++) Sc[(c + q) * s + rr] = (uint8_t)(v >> (8 * q));
            ss += __builtin_amdgcn_udot4(v, 0x01010101u, 0u, false); ss2 += __builtin_amdgcn_udot4(v, v, 0u, false);
        }
        ss = rows_total(row_sum(ss)); ss2 = rows_total(row_sum(ss2));
        if ((tid & 63) == 0) { atomicAdd(&sstat[0], ss); atomicAdd(&sstat[1], ss2); }
    }
    // ---- horizontal pass, once per horizontal offset: Hc[a][col][row] over all WIN rows
    for (int t = tid; t < kGrid * WIN * (s >> 2); t += NT) {
        const int a = (int)(((float)t + 0.5f) * r_hw), rem = t - a * (WIN * sq), rr = rem >> lq, j = (rem - (rr << lq)) << 2;
        const int ix = kGridOff[a][0], fx = kGridOff[a][1];
        const int ob = j + ix + 1;                     // first byte of output 0's eight taps in the window row
        const uint32_t* wd = (const uint32_t*)(win + rr * WIN + (ob & ~3));
        const uint32_t d0 = wd[0], d1 = wd[1], d2 = wd[2];   // (the last dword of the last row may lie past the window: inside the array, unused bytes)
        uint32_t P;
        if (fx == 0) P = bytes4(d0, d1, d2, (ob & 3) + 3);
        else {
            const int TA = fx == 2 ? TAp[0] : (fx == 4 ? TAp[1] : TAp[2]), TB = fx == 2 ? TBp[0] : (fx == 4 ? TBp[1] : TBp[2]);
            P = filt4(d0 ^ 0x80808080u, d1 ^ 0x80808080u, d2 ^ 0x80808080u, ob & 3, TA, TB);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) Hc[a][(j + q) * WIN + rr] = (uint8_t)(P >> (8 * q));
    }
    __syncthreads();
    // ---- vertical pass + statistics, once per grid position: a task = four vertically adjacent outputs of one column
    const int per = s * (s >> 2);                       // tasks per grid position
    const int G = per < 64 ? per : 64;                  // lanes of a wave that share a grid position (16 for 8x8 PUs)
    for (int t0 = 0; t0 < kGridN * per; t0 += NT) {
        const int t = t0 + tid;
        uint32_t sp = 0, sp2 = 0, sps = 0;
        int c = 0;
        if (t < kGridN * per) {
            c = (int)(((float)t + 0.5f) * r_per);
            const int rem = t - c * per, j = rem >> lq, i = (rem - (j << lq)) << 2;
            const int b = c / kGrid, a = c - b * kGrid;   // grid position: row offset b, column offset a
            const int iy = kGridOff[b][0], fy = kGridOff[b][1];
            const int ob = i + iy + 1;
            const uint32_t* hd = (const uint32_t*)(&Hc[a][j * WIN + (ob & ~3)]);
            const uint32_t d0 = hd[0], d1 = hd[1], d2 = hd[2];
            uint32_t P;
            if (fy == 0) P = bytes4(d0, d1, d2, (ob & 3) + 3);
            else {
                const int TA = fy == 2 ? TAp[0] : (fy == 4 ? TAp[1] : TAp[2]), TB = fy == 2 ? TBp[0] : (fy == 4 ? TBp[1] : TBp[2]);
                P = filt4(d0 ^ 0x80808080u, d1 ^ 0x80808080u, d2 ^ 0x80808080u, ob & 3, TA, TB);
            }
            const uint32_t S = *(const uint32_t*)(&Sc[j * s + i]);
            sp = __builtin_amdgcn_udot4(P, 0x01010101u, 0u, false); sp2 = __builtin_amdgcn_udot4(P, P, 0u, false); sps = __builtin_amdgcn_udot4(P, S, 0u, false);
        }
        sp = row_sum(sp); sp2 = row_sum(sp2); sps = row_sum(sps);          // DPP inside the 16-lane rows (G = 16: a row is a grid position's lanes)
        if (G == 64) { sp = rows_total(sp); sp2 = rows_total(sp2); sps = rows_total(sps); }
        if (t < kGridN * per && (tid & (G - 1)) == 0) { atomicAdd(&stat[c][0], sp); atomicAdd(&stat[c][1], sp2); atomicAdd(&stat[c][2], sps); }
    }
    __syncthreads();
    if (tid < kGridN) {   // svt_aom_variance{W}x{H}: sse - (sum * sum) / (w h) in 32-bit unsigned arithmetic (Encoder/C_DEFAULT/variance.c)
        const long long sum = (long long)stat[tid][0] - (long long)sstat[0];
        const uint32_t sse = stat[tid][1] - 2u * stat[tid][2] + sstat[1];
        o[2 * tid] = sse - (uint32_t)((sum * sum) / (s * s));
        o[2 * tid + 1] = sse;
    }
}

}   // namespace

extern "C" int svt_hip_launch_md_fullpel_sad(hipStream_t st, int pix_bytes, const void* src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus,
                                             const SvtHipMdPu* pus, int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* mv, uint32_t* sad) {
    if (n_sb <= 0 || n_refs <= 0 || n_pus <= 0) return 0;
    RefPlanes rp;
    PuList    pl;
    for (int i = 0; i < SVT_HIP_MD_MAX_REFS; i++) rp.r[i] = refs[i < n_refs ? i : 0];
    for (int i = 0; i < SVT_HIP_MD_MAX_PUS; i++) {
        const SvtHipMdPu p = pus[i < n_pus ? i : 0];
        pl.x[i] = p.x; pl.y[i] = p.y; pl.w[i] = p.w; pl.h[i] = p.h;
    }
    if (pix_bytes == 2) hipLaunchKernelGGL(md_fullpel_sad_kernel<uint16_t>, dim3(n_sb, n_refs), dim3(256), 0, st, (const uint16_t*)src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, mv, sad);
    else hipLaunchKernelGGL(md_fullpel_sad_kernel<uint8_t>, dim3(n_sb, n_refs), dim3(256), 0, st, (const uint8_t*)src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, mv, sad);
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_md_fullpel_avg_sad(hipStream_t st, int pix_bytes, const void* src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                                 int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* mv, int n_pairs, const uint8_t (*pairs)[2], uint32_t* sad) {
    if (n_sb <= 0 || n_refs <= 0 || n_pus <= 0 || n_pairs <= 0) return 0;
    RefPlanes rp;
    PuList    pl;
    PairList  pp;
    for (int i = 0; i < SVT_HIP_MD_MAX_REFS; i++) rp.r[i] = refs[i < n_refs ? i : 0];
    for (int i = 0; i < SVT_HIP_MD_MAX_PUS; i++) {
        const SvtHipMdPu p = pus[i < n_pus ? i : 0];
        pl.x[i] = p.x; pl.y[i] = p.y; pl.w[i] = p.w; pl.h[i] = p.h;
    }
    for (int i = 0; i < SVT_HIP_MD_MAX_PAIRS; i++) { pp.c0[i] = pairs[i < n_pairs ? i : 0][0]; pp.c1[i] = pairs[i < n_pairs ? i : 0][1]; }
    if (pix_bytes == 2) hipLaunchKernelGGL(md_fullpel_avg_sad_kernel<uint16_t>, dim3(n_sb, n_pairs), dim3(256), 0, st, (const uint16_t*)src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, n_pairs, pp, mv, sad);
    else hipLaunchKernelGGL(md_fullpel_avg_sad_kernel<uint8_t>, dim3(n_sb, n_pairs), dim3(256), 0, st, (const uint8_t*)src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, n_pairs, pp, mv, sad);
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_md_subpel_grid(hipStream_t st, const uint8_t* src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                             int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* mv, int bank, int grid, uint32_t* out) {
    if (n_sb <= 0 || n_refs <= 0 || n_pus <= 0) return 0;
    RefPlanes rp;
    PuList    pl;
    for (int i = 0; i < SVT_HIP_MD_MAX_REFS; i++) rp.r[i] = refs[i < n_refs ? i : 0];
    for (int i = 0; i < SVT_HIP_MD_MAX_PUS; i++) {
        const SvtHipMdPu p = pus[i < n_pus ? i : 0];
        pl.x[i] = p.x; pl.y[i] = p.y; pl.w[i] = p.w; pl.h[i] = p.h;
    }
    PuSel small, large;   // PUs up to 16 wide: one wave each; the rest (and the ones the kernel declines: it writes their "not computed" pairs): four waves
    int   n_small = 0, n_large = 0;
    for (int i = 0; i < n_pus; i++) {
        if (pus[i].w <= 16 && pus[i].h <= 16) small.idx[n_small++] = (uint8_t)i;
        else large.idx[n_large++] = (uint8_t)i;
    }
    for (int i = n_small; i < SVT_HIP_MD_MAX_PUS; i++) small.idx[i] = 0;
    for (int i = n_large; i < SVT_HIP_MD_MAX_PUS; i++) large.idx[i] = 0;
    if (n_small) {
        if (grid == 3) hipLaunchKernelGGL((md_subpel_grid_kernel<3, 16, 64>), dim3(n_sb, n_small, n_refs), dim3(64), 0, st, src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, small, mv, bank, out);
        else hipLaunchKernelGGL((md_subpel_grid_kernel<7, 16, 64>), dim3(n_sb, n_small, n_refs), dim3(64), 0, st, src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, small, mv, bank, out);
    }
    if (n_large) {
        if (grid == 3) hipLaunchKernelGGL((md_subpel_grid_kernel<3, 64, 256>), dim3(n_sb, n_large, n_refs), dim3(256), 0, st, src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, large, mv, bank, out);
        else hipLaunchKernelGGL((md_subpel_grid_kernel<7, 64, 256>), dim3(n_sb, n_large, n_refs), dim3(256), 0, st, src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, large, mv, bank, out);
    }
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(md_pre)
