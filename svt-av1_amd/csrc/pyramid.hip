// pyramid.hip — picture-analysis kernels that feed open-loop ME, and the HME search kernel; gfx950.
//
// Replaces (file:line under /root/reference/Source/Lib):
//   Encoder/Codec/EbPictureAnalysisProcess.c:193,223    decimation_2d / downsample_2d
//   Encoder/Codec/EbPictureAnalysisProcess.c:1005-2575  compute_block_mean_compute_variance
//        (+ svt_compute_interm_var_four8x8 :352, svt_compute_sub_mean_8x8 :310, svt_compute_mean_squared_values :287)
//   Encoder/C_DEFAULT/EbComputeSAD_C.c:58               svt_sad_loop_kernel_c  (hme_level_0/1/2, EbMotionEstimation.c:852,1028,1177)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"
#include "lds_stage.h"

namespace {

__global__ void __launch_bounds__(256)
downsample_kernel(const uint8_t* __restrict__ in, int in_stride, int w, int h, uint8_t* __restrict__ out, int out_stride, int step, int filtered) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    const int ow = w / step;
    if (ox >= ow) return;
    if (!filtered) {
        out[(size_t)oy * out_stride + ox] = in[(size_t)(oy * step) * in_stride + ox * step];
    } else {
        const int half = step >> 1, y = oy * step + half, x = ox * step + half;
        const uint32_t s = in[(size_t)(y - 1) * in_stride + x - 1] + in[(size_t)(y - 1) * in_stride + x] + in[(size_t)y * in_stride + x - 1] +
                           in[(size_t)y * in_stride + x];
        out[(size_t)oy * out_stride + ox] = (uint8_t)((s + 2) >> 2);
    }
}

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
    return ((uint64_t)(uint32_t)__shfl_xor((int)(v >> 32), m, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64);
}

// one wave per 64x64 SB; lane = 8x8 block in raster order; outputs [sb][85]: 64x64, 32x32 x4, 16x16 x16, 8x8 x64 (raster per level)
__global__ void __launch_bounds__(64)
variance_pyramid_kernel(const uint8_t* __restrict__ plane, int stride, int sb_cols, int full_precision, uint8_t* __restrict__ mean_out,
                        uint16_t* __restrict__ var_out) {
    const int sb = blockIdx.x, lane = threadIdx.x;
    const int sx = (sb % sb_cols) * 64, sy = (sb / sb_cols) * 64;
    const uint8_t* p = plane + (size_t)(sy + (lane >> 3) * 8) * stride + sx + (lane & 7) * 8;
    uint32_t s = 0, s2 = 0;
    for (int y = 0; y < 8; y += full_precision ? 1 : 2) {
        const uint2 v = *(const uint2*)(p + (size_t)y * stride);   // SB-aligned + 8-aligned column: needs plane/stride 8-byte aligned
        s = __builtin_amdgcn_udot4(v.x, 0x01010101u, s, false); s = __builtin_amdgcn_udot4(v.y, 0x01010101u, s, false);
        s2 = __builtin_amdgcn_udot4(v.x, v.x, s2, false); s2 = __builtin_amdgcn_udot4(v.y, v.y, s2, false);
    }
    uint64_t m8 = full_precision ? ((uint64_t)s << 8) / 64 : (uint64_t)s << 3;
    uint64_t q8 = full_precision ? ((uint64_t)s2 << 16) / 64 : (uint64_t)s2 << 11;
    // raster lanes: 16x16 = lanes {l, l^1, l^8, l^9}; 32x32 adds {^2, ^16}; 64x64 adds {^4, ^32}
    uint64_t t = m8 + shfl_xor64(m8, 1); const uint64_t m16 = (t + shfl_xor64(t, 8)) >> 2;
    t = q8 + shfl_xor64(q8, 1);          const uint64_t q16 = (t + shfl_xor64(t, 8)) >> 2;
    t = m16 + shfl_xor64(m16, 2);        const uint64_t m32 = (t + shfl_xor64(t, 16)) >> 2;
    t = q16 + shfl_xor64(q16, 2);        const uint64_t q32 = (t + shfl_xor64(t, 16)) >> 2;
    t = m32 + shfl_xor64(m32, 4);        const uint64_t m64 = (t + shfl_xor64(t, 32)) >> 2;
    t = q32 + shfl_xor64(q32, 4);        const uint64_t q64 = (t + shfl_xor64(t, 32)) >> 2;
    uint8_t* mo = mean_out + (size_t)sb * 85;
    uint16_t* vo = var_out + (size_t)sb * 85;
#define PUT(i, m, q) do { mo[i] = (uint8_t)((m) >> 8); vo[i] = (uint16_t)(((q) - (m) * (m)) >> 16); } while (0)
    PUT(21 + lane, m8, q8);
    const int bx = lane & 7, by = lane >> 3;
    if (!(bx & 1) && !(by & 1)) PUT(5 + (by >> 1) * 4 + (bx >> 1), m16, q16);
    if (!(bx & 3) && !(by & 3)) PUT(1 + (by >> 2) * 2 + (bx >> 2), m32, q32);
    if (lane == 0) PUT(0, m64, q64);
#undef PUT
}

// svt_sad_loop_kernel for a list of searches; one workgroup per search.
// Fast path (block width 16 / 32 / 64, plane strides multiples of 4): the same v_qsad_pk_u16_u8 scheme as the
// integer ME kernel — source block and reference window staged in LDS with the byte misalignment removed, a
// lane owns 8 adjacent candidates of one candidate row, packed u16 partial SADs are flushed into 32-bit sums
// every 256/BW rows (before they can overflow), running (sad << 32 | raster index) key per lane.
// Other widths (partial SBs at the picture edge) take the generic byte-wise path.
constexpr int kHmeRefStrideDw = 48, kHmeRefRowDw = 36, kHmeSrcStrideDw = 16;

template <int BW>
__device__ __forceinline__ void sad_loop_fast(const SvtHipSadLoop& d, const uint8_t* __restrict__ src, int src_stride,
                                              const uint8_t* __restrict__ ref, int ref_stride, uint32_t* lds_src, uint32_t* lds_ref, int tid,
                                              unsigned long long& best) {
    constexpr int SEGS = BW / 8, FLUSH = 256 / BW;
    const int rstep = d.row_step, rows = d.bh / rstep;
    // HME searches are small (16x16 candidates = 32 units): the block's rows are dealt out to P adjacent lanes of a unit (lane p takes
    // rows p, p+P, ...) so that all 256 lanes work; the P partial SAD vectors are added with lane shuffles.  The P lanes of a unit read
    // P consecutive LDS rows at once, so the row strides are chosen per shape: 48 / 16 dwords put 4 consecutive rows of a 16-dword
    // read on disjoint bank quarters (large windows, P = 1); 40 / 24 put 8 consecutive rows of a <= 4-dword read on disjoint 8-bank slots.
    const int ng0 = (min(64, (int)d.sa_w) + 7) >> 3, th0 = min(64, (int)d.sa_h);
    int P = 1;
    while (P < 8 && th0 * ng0 * (2 * P) <= 256 && (rows % (2 * P)) == 0) P *= 2;
    const int rs = P > 1 ? 40 : kHmeRefStrideDw, ss = P > 1 ? 24 : kHmeSrcStrideDw;
    stage_rows(lds_src, ss, src + (size_t)d.src_y * src_stride + d.src_x, src_stride * rstep, rows, BW / 4, BW, tid, 256);
    for (int ty = 0; ty < d.sa_h; ty += 64) {
        const int th = min(64, d.sa_h - ty);
        for (int tx = 0; tx < d.sa_w; tx += 64) {
            const int tw = min(64, d.sa_w - tx), ng = (tw + 7) >> 3;
            __syncthreads();
            stage_rows(lds_ref, rs, ref + (size_t)(d.ref_y + ty) * ref_stride + d.ref_x + tx, ref_stride, th + (rows - 1) * rstep,
                       kHmeRefRowDw, tw + BW - 1, tid, 256);
            __syncthreads();
            const int rows_p = rows / P;
            for (int w = tid; w < th * ng * P; w += 256) {
                const int u = w / P, part = w & (P - 1);
                const int y = u / ng, g = u - y * ng;
                uint32_t acc32[8];
#pragma unroll
                for (int c = 0; c < 8; c++) acc32[c] = 0;
                for (int n0 = 0; n0 < rows_p; n0 += FLUSH) {
                    uint64_t a0 = 0, a1 = 0;
                    const int n1 = min(rows_p, n0 + FLUSH);
                    for (int n = n0; n < n1; n++) {
                        const int r = part + n * P;
                        const uint32_t* rp = lds_ref + (y + r * rstep) * rs + 2 * g;
                        const uint32_t* sp = lds_src + r * ss;
                        uint64_t ev[SEGS + 1], od[SEGS];
#pragma unroll
                        for (int k = 0; k <= SEGS; k++) ev[k] = *(const uint64_t*)(rp + 2 * k);
#pragma unroll
                        for (int k = 0; k < SEGS; k++) { Dw2 t = *(const Dw2*)(rp + 2 * k + 1); od[k] = pack64(t.x, t.y); }
#pragma unroll
                        for (int k = 0; k < SEGS; k++) {
                            const uint32_t s0 = sp[2 * k], s1 = sp[2 * k + 1];
                            a0 = __builtin_amdgcn_qsad_pk_u16_u8(ev[k], s0, a0);
                            a0 = __builtin_amdgcn_qsad_pk_u16_u8(od[k], s1, a0);
                            a1 = __builtin_amdgcn_qsad_pk_u16_u8(od[k], s0, a1);
                            a1 = __builtin_amdgcn_qsad_pk_u16_u8(ev[k + 1], s1, a1);
                        }
                    }
                    acc32[0] += (uint32_t)a0 & 0xFFFFu; acc32[1] += ((uint32_t)a0) >> 16; acc32[2] += (uint32_t)(a0 >> 32) & 0xFFFFu; acc32[3] += (uint32_t)(a0 >> 48);
                    acc32[4] += (uint32_t)a1 & 0xFFFFu; acc32[5] += ((uint32_t)a1) >> 16; acc32[6] += (uint32_t)(a1 >> 32) & 0xFFFFu; acc32[7] += (uint32_t)(a1 >> 48);
                }
                for (int m = 1; m < P; m <<= 1)   // P adjacent lanes hold the parts of one unit (groups never straddle the loop bound)
#pragma unroll
                    for (int c = 0; c < 8; c++) acc32[c] += (uint32_t)__shfl_xor((int)acc32[c], m, 64);
                const uint32_t idx0 = (uint32_t)((ty + y) * d.sa_w + tx + 8 * g);
                if (part == 0) {
#pragma unroll
                    for (int c = 0; c < 8; c++)
                        if (8 * g + c < tw) {
                            const unsigned long long key = ((unsigned long long)acc32[c] << 32) | (idx0 + c);
                            best = key < best ? key : best;
                        }
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256)
sad_loop_kernel(const uint8_t* __restrict__ src, int src_stride, const uint8_t* __restrict__ ref, int ref_stride,
                const SvtHipSadLoop* __restrict__ searches, uint32_t* __restrict__ best_sad, int16_t* __restrict__ best_xy) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_src[64 * 24];   // row stride 16 or 24 dwords (see sad_loop_fast)
    __shared__ __attribute__((aligned(16))) uint32_t lds_ref[127 * kHmeRefStrideDw];
    __shared__ unsigned long long s_best[4];
    const SvtHipSadLoop d = searches[blockIdx.x];
    const int tid = threadIdx.x;
    unsigned long long best = ((unsigned long long)0xffffffu << 32) | 0xffffffffu;  // initial best_sad 0xffffff (EbComputeSAD_C.c:73)
    const int ncand = d.sa_w * d.sa_h;
    const bool aligned = !((src_stride | ref_stride) & 3) && d.bh <= 64 && (d.bh % d.row_step) == 0;
    if (aligned && d.bw == 16) sad_loop_fast<16>(d, src, src_stride, ref, ref_stride, lds_src, lds_ref, tid, best);
    else if (aligned && d.bw == 32) sad_loop_fast<32>(d, src, src_stride, ref, ref_stride, lds_src, lds_ref, tid, best);
    else if (aligned && d.bw == 64) sad_loop_fast<64>(d, src, src_stride, ref, ref_stride, lds_src, lds_ref, tid, best);
    else {
        uint8_t* s_blk = (uint8_t*)lds_ref;  // generic path: source block in LDS, reference read through the caches
        const int rstep = d.row_step, rows = d.bh / rstep;
        for (int i = tid; i < rows * d.bw; i += 256) {
            const int y = i / d.bw, x = i - y * d.bw;
            s_blk[i] = src[(size_t)(d.src_y + y * rstep) * src_stride + d.src_x + x];
        }
        __syncthreads();
        for (int c = tid; c < ncand; c += 256) {
            const int cy = c / d.sa_w, cx = c - cy * d.sa_w;
            const uint8_t* r = ref + (size_t)(d.ref_y + cy) * ref_stride + d.ref_x + cx;
            uint32_t sad = 0;
            for (int y = 0; y < rows; y++)
                for (int x = 0; x < d.bw; x++) sad += (uint32_t)abs((int)s_blk[y * d.bw + x] - (int)r[(size_t)(y * rstep) * ref_stride + x]);
            const unsigned long long key = ((unsigned long long)sad << 32) | (uint32_t)c;
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const unsigned long long o = shfl_xor64(best, m); best = o < best ? o : best; }
    if ((tid & 63) == 0) s_best[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) best = s_best[w] < best ? s_best[w] : best;
        const uint32_t sad = (uint32_t)(best >> 32), c = (uint32_t)best;
        best_sad[blockIdx.x] = sad;
        if (sad < 0xffffffu && c < (uint32_t)ncand) {   // a candidate beat the initial value: centers are written
            best_xy[2 * blockIdx.x] = (int16_t)(c % d.sa_w);
            best_xy[2 * blockIdx.x + 1] = (int16_t)(c / d.sa_w);
        }
    }
}


// 16-bit twin of the search above for the high-bit-depth path: sad_16b_kernel (Encoder/C_DEFAULT/EbComputeSAD_C.c:39) over a window, with
// svt_sad_loop_kernel's candidate order and update rule (first minimum in raster order, initial best 0xffffff).  One workgroup per search:
// the source block sits in LDS, a lane owns candidates c, c + 256, ...; two samples per v_sad_u16.
__device__ void sad_loop16_generic(uint16_t* __restrict__ s_blk, unsigned long long* __restrict__ s_best, const uint16_t* __restrict__ src, int src_stride,
                                   const uint16_t* __restrict__ ref, int ref_stride, const SvtHipSadLoop d, uint32_t* __restrict__ best_sad, int16_t* __restrict__ best_xy) {
    const int tid = threadIdx.x;
    unsigned long long best = ((unsigned long long)0xffffffu << 32) | 0xffffffffu;
    const int ncand = d.sa_w * d.sa_h, rstep = d.row_step, rows = d.bh / rstep;
    for (int i = tid; i < rows * d.bw; i += 256) {
        const int y = i / d.bw, x = i - y * d.bw;
        s_blk[i] = src[(size_t)(d.src_y + y * rstep) * src_stride + d.src_x + x];
    }
    __syncthreads();
    for (int c = tid; c < ncand; c += 256) {
        const int cy = c / d.sa_w, cx = c - cy * d.sa_w;
        const uint16_t* r = ref + (size_t)(d.ref_y + cy) * ref_stride + d.ref_x + cx;
        uint32_t sad = 0;
        for (int y = 0; y < rows; y++) {
            const uint16_t* rr = r + (size_t)(y * rstep) * ref_stride;
            const uint16_t* ss = s_blk + y * d.bw;
            for (int x = 0; x < d.bw; x++) sad += (uint32_t)abs((int)ss[x] - (int)rr[x]);
        }
        const unsigned long long key = ((unsigned long long)sad << 32) | (uint32_t)c;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const unsigned long long o = shfl_xor64(best, m); best = o < best ? o : best; }
    if ((tid & 63) == 0) s_best[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) best = s_best[w] < best ? s_best[w] : best;
        const uint32_t sad = (uint32_t)(best >> 32), c = (uint32_t)best;
        best_sad[blockIdx.x] = sad;
        if (sad < 0xffffffu && c < (uint32_t)ncand) {
            best_xy[2 * blockIdx.x] = (int16_t)(c % d.sa_w);
            best_xy[2 * blockIdx.x + 1] = (int16_t)(c / d.sa_w);
        }
    }
}

// The same search for the shapes configs[3] of BASELINE.json asks for (block 16..64 wide in multiples of 16, search area a multiple of 8 wide, every row of
// the block): the reference window goes through LDS once as aligned sample pairs; a lane owns 8 horizontally adjacent candidates of one candidate row and feeds
// v_sad_u16 (two samples per instruction).  Per 8 source pairs: three 128-bit LDS reads of the window, eleven v_alignbit_b32 that form the pairs of the four
// candidates at odd columns from neighbouring dwords, and 64 SAD instructions; the source pairs are wave-uniform and come through scalar loads when the source
// block is dword-aligned (else from an LDS copy).  Round 6: one window copy instead of two (69 -> 34.5 KB: three workgroups per compute unit instead of two) and
// three LDS reads per 64 SADs instead of eight, no source copy in LDS (34.5 KB: four workgroups per compute unit instead of two) -- and the source pairs of a whole block row come through scalar loads issued a row ahead: 4K 10-bit 64x64 / 64x64 search 0.737 -> 0.60 ms (profiles/r06/NOTES.md).  Keys sad << 32 | candidate index keep the
// reference's first minimum in raster order.  16.8 M absolute differences per 64x64 / 64x64 search = 131 k wave instructions.
constexpr int kS16 = 68;   // row stride of the window in dwords (272 B: 16-byte aligned rows, consecutive rows 4 banks apart)
typedef const __attribute__((address_space(4))) uint32_t* sad16_const_u32_ptr;
// one group of 8 source pairs (16 samples) against the 8 candidates of the lane: w0 = 12 window dwords from the lane's first candidate on
__device__ __forceinline__ void sad16_group(const uint4* __restrict__ q, const uint32_t (&sv)[8], uint32_t (&acc)[8]) {
    const uint4 a0 = q[0], a1 = q[1], a2 = q[2];
    const uint32_t w0[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
    uint32_t w1[11];   // the pairs one sample further: (w0[k] >> 16) | (w0[k + 1] << 16)
#pragma unroll
    for (int k = 0; k < 11; k++) w1[k] = __builtin_amdgcn_alignbit(w0[k + 1], w0[k], 16);
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int j = 0; j < 8; j++)
            acc[j] = __builtin_amdgcn_sad_u16((j & 1) ? w1[p + (j >> 1)] : w0[p + (j >> 1)], sv[p], acc[j]);
}
// source pairs through scalar loads, a whole block row (NG groups) at a time and the NEXT row's loads issued before this row's arithmetic (a scalar load per
// group, waited for on the spot, left the SAD pipe idle half of the time)
template <int NG>
__device__ __forceinline__ void sad16_rows_scalar(const uint32_t* __restrict__ s_w, const uint16_t* __restrict__ src_blk, int src_stride, int bh, int cx, int cy, uint32_t (&acc)[8]) {
    uint32_t cur[8 * NG], nxt[8 * NG];
    {
        sad16_const_u32_ptr sp = (sad16_const_u32_ptr)src_blk;
#pragma unroll
        for (int k = 0; k < 8 * NG; k++) cur[k] = sp[k];
    }
    for (int y = 0; y < bh; y++) {
        sad16_const_u32_ptr sn = (sad16_const_u32_ptr)(src_blk + (size_t)min(y + 1, bh - 1) * src_stride);   // wave-uniform (the last row is loaded twice: no branch)
#pragma unroll
        for (int k = 0; k < 8 * NG; k++) nxt[k] = sn[k];
        const uint4* __restrict__ q = (const uint4*)(s_w + (cy + y) * kS16 + (cx >> 1));
#pragma unroll
        for (int c = 0; c < NG; c++) {
            const uint32_t sv[8] = {cur[8 * c], cur[8 * c + 1], cur[8 * c + 2], cur[8 * c + 3], cur[8 * c + 4], cur[8 * c + 5], cur[8 * c + 6], cur[8 * c + 7]};
            sad16_group(q + 2 * c, sv, acc);
        }
#pragma unroll
        for (int k = 0; k < 8 * NG; k++) cur[k] = nxt[k];
    }
}
// a source block that does not start on a dword (odd column, odd stride): its pairs are formed from 16-bit loads (wave-uniform addresses; the rare path)
__device__ __forceinline__ void sad16_rows_unaligned(const uint32_t* __restrict__ s_w, const uint16_t* __restrict__ src_blk, int src_stride, int bw, int bh, int cx, int cy,
                                                     uint32_t (&acc)[8]) {
    for (int y = 0; y < bh; y++) {
        const uint4* __restrict__ q = (const uint4*)(s_w + (cy + y) * kS16 + (cx >> 1));
        const uint16_t* __restrict__ sr = src_blk + (size_t)y * src_stride;
        for (int c = 0; c < (bw >> 4); c++) {   // 8 source pairs (16 samples) at a time
            uint32_t sv[8];
#pragma unroll
            for (int p = 0; p < 8; p++) sv[p] = (uint32_t)sr[16 * c + 2 * p] | ((uint32_t)sr[16 * c + 2 * p + 1] << 16);
            sad16_group(q + 2 * c, sv, acc);
        }
    }
}
__global__ void __launch_bounds__(256)
sad_loop16_lds_kernel(const uint16_t* __restrict__ src, int src_stride, const uint16_t* __restrict__ ref, int ref_stride,
                      const SvtHipSadLoop* __restrict__ searches, uint32_t* __restrict__ best_sad, int16_t* __restrict__ best_xy) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    __shared__ unsigned long long s_best[4];
    const SvtHipSadLoop d = searches[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool lds_form = d.row_step == 1 && d.bw >= 16 && d.bw <= 64 && (d.bw & 15) == 0 && d.bh >= 1 && d.bh <= 64 && d.sa_w >= 8 && d.sa_w <= 64 && (d.sa_w & 7) == 0 &&
                          d.sa_h >= 1 && d.sa_h <= 64;
    if (!lds_form) {   // any other shape (sub-sampled rows, odd widths, large areas): a lane per candidate straight from memory (workgroup-uniform branch)
        sad_loop16_generic((uint16_t*)s_dyn, s_best, src, src_stride, ref, ref_stride, d, best_sad, best_xy);
        return;
    }
    const int bw = d.bw, bh = d.bh, sa_w = d.sa_w, sa_h = d.sa_h, wr = bh + sa_h - 1, ww = bw + sa_w - 1;
    uint32_t* __restrict__ s_w = s_dyn;                        // [wr][kS16]
    const uint16_t* __restrict__ src_blk = src + (size_t)d.src_y * src_stride + d.src_x;
    const bool src_scalar = (((uintptr_t)src_blk | (uintptr_t)(2 * src_stride)) & 3) == 0;   // every source row starts on a dword: its pairs are scalar loads
    // dword i of a window row: samples 2i, 2i + 1; zero past the window.  Eight dwords = sixteen 16-bit loads in flight per thread (one dword per iteration is 32
    // dependent memory round trips per workgroup before the first SAD); the loads are unconditional on clamped columns, the zeroing is a select
    batched_stage<8, uint32_t>(wr * 64, tid, 256,
        [&](int i) {
            const int y = i >> 6, x = i & 63;
            const uint16_t* p = ref + (size_t)(d.ref_y + y) * ref_stride + d.ref_x;
            const uint32_t a = p[min(2 * x, ww - 1)], b = p[min(2 * x + 1, ww - 1)];
            return (2 * x < ww ? a : 0u) | ((2 * x + 1 < ww ? b : 0u) << 16);
        },
        [&](int i, uint32_t v) { s_w[(i >> 6) * kS16 + (i & 63)] = v; });
    __syncthreads();
    unsigned long long best = ((unsigned long long)0xffffffu << 32) | 0xffffffffu;
    const int g = lane & 7, r = lane >> 3, cx = 8 * g;
    for (int cy0 = 0; cy0 < sa_h; cy0 += 32) {
        const int cy = cy0 + 8 * wave + r;
        if (cx < sa_w && cy < sa_h) {   // sa_w is a multiple of 8: a lane's eight candidates are all inside or all outside
            uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (!src_scalar) sad16_rows_unaligned(s_w, src_blk, src_stride, bw, bh, cx, cy, acc);
            else if (bw == 64) sad16_rows_scalar<4>(s_w, src_blk, src_stride, bh, cx, cy, acc);
            else if (bw == 48) sad16_rows_scalar<3>(s_w, src_blk, src_stride, bh, cx, cy, acc);
            else if (bw == 32) sad16_rows_scalar<2>(s_w, src_blk, src_stride, bh, cx, cy, acc);
            else sad16_rows_scalar<1>(s_w, src_blk, src_stride, bh, cx, cy, acc);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned long long key = ((unsigned long long)acc[j] << 32) | (uint32_t)(cy * sa_w + cx + j);
                best = key < best ? key : best;
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const unsigned long long o = shfl_xor64(best, m); best = o < best ? o : best; }
    if (lane == 0) s_best[wave] = best;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) best = s_best[w] < best ? s_best[w] : best;
        const uint32_t sad = (uint32_t)(best >> 32), c = (uint32_t)best;
        best_sad[blockIdx.x] = sad;
        if (sad < 0xffffffu && c < (uint32_t)(sa_w * sa_h)) {
            best_xy[2 * blockIdx.x] = (int16_t)(c % sa_w);
            best_xy[2 * blockIdx.x + 1] = (int16_t)(c / sa_w);
        }
    }
}
}  // namespace

extern "C" int svt_hip_launch_downsample(hipStream_t st, const uint8_t* in, int in_stride, int w, int h, uint8_t* out, int out_stride, int step, int filtered) {
    hipLaunchKernelGGL(downsample_kernel, dim3((w / step + 255) / 256, h / step), dim3(256), 0, st, in, in_stride, w, h, out, out_stride, step, filtered);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_variance_pyramid(hipStream_t st, const uint8_t* plane, int stride, int sb_cols, int n_sb, int full_precision,
                                               uint8_t* mean_out, uint16_t* var_out) {
    if (n_sb <= 0) return 0;
    hipLaunchKernelGGL(variance_pyramid_kernel, dim3(n_sb), dim3(64), 0, st, plane, stride, sb_cols, full_precision, mean_out, var_out);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_sad_loop(hipStream_t st, const uint8_t* src, int src_stride, const uint8_t* ref, int ref_stride,
                                       const SvtHipSadLoop* searches, int n, uint32_t* best_sad, int16_t* best_xy) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sad_loop_kernel, dim3(n), dim3(256), 0, st, src, src_stride, ref, ref_stride, searches, best_sad, best_xy);
    return (int)hipGetLastError();
}
extern "C" int svt_hip_launch_sad_loop16(hipStream_t st, const uint16_t* src, int src_stride, const uint16_t* ref, int ref_stride, const SvtHipSadLoop* searches, int n,
                                             uint32_t* best_sad, int16_t* best_xy) {
    if (n <= 0) return 0;
    constexpr size_t lds = sizeof(uint32_t) * (127 * kS16);   // one window copy, 34.5 KB: four workgroups per compute unit (the generic form stages a source block of <= 8 KB there)
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)sad_loop16_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
    hipLaunchKernelGGL(sad_loop16_lds_kernel, dim3(n), dim3(256), lds, st, src, src_stride, ref, ref_stride, searches, best_sad, best_xy);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(pyramid)
