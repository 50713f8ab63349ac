// me_fullpel.hip — open-loop integer motion search, all 85 square PUs of a 64x64 superblock,
// hand-written for gfx950 (CDNA4, wave64).
//
// Replaces the per-SB inner loops of the reference (file:line under /root/reference/Source/Lib):
//   Encoder/Codec/EbMotionEstimation.c:814  open_loop_me_fullpel_search_sblock
//   Encoder/Codec/EbMotionEstimation.c:362  svt_ext_all_sad_calculation_8x8_16x16_c
//   Encoder/Codec/EbMotionEstimation.c:396  svt_ext_eight_sad_calculation_32x32_64x64_c
//   Encoder/Codec/EbMotionEstimation.c:122/:191 (single-candidate tails)
// One workgroup = one (superblock, reference) pair; one launch = every SB of the frame.
//
// Mapping (why it looks nothing like the x86 kernels):
//  * The SAD primitive is v_qsad_pk_u16_u8: one instruction = SADs of 4 source pixels against 4
//    horizontally adjacent candidates, accumulated into 4 packed u16 — the ref bytes are consumed
//    *unaligned* out of a 64-bit register pair, so no per-candidate byte realignment is needed.
//    Measured on MI355X (tools/ubench/sad_rate.hip): 16 cyc/wave-instr (=16 abs-diff/lane), the
//    same abs-diff rate as v_sad_u8 (4 cyc, 4 abs-diff) but with 1/4 of the LDS operand traffic.
//  * A lane owns a "unit" = 8 adjacent candidates (two qsad quads) on one candidate row and walks
//    all 64 8x8 blocks of the SB for them; 8x8 SADs live packed u16x4 in registers, 16x16 are packed
//    adds, 32x32/64x64 are unpacked u32 sums.
//  * Every PU keeps a running (sad<<16 | raster_candidate_index) key per lane, updated with
//    v_min3_u32; the reference's "strict <, raster order" tie-break is exactly the minimum of that
//    key.  32x32/64x64 SADs need 20 bits, so those 5 PUs use 64-bit keys.
//  * The source SB (4 KB) and the reference window ((64+63) rows) are staged in LDS once per
//    64x64-candidate tile with the byte misalignment removed at staging time (v_alignbyte).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"
#include "lds_stage.h"

#ifndef ME_R_UNROLL
#define ME_R_UNROLL 1
#endif
#ifndef ME_WAVES_PER_EU
#define ME_WAVES_PER_EU 2
#endif

namespace {

constexpr int kTile   = 64;            // candidates per tile edge
constexpr int kRefRows = kTile + 63;   // window rows per tile
constexpr int kRefRowDw = 36;          // dwords staged per window row (64+63 px + 8 over-read = 135 B -> 144 B)
constexpr int kRefStrideDw = 48;       // LDS row stride in dwords (192 B): rows y..y+3 of a 32-lane group hit
                                       // disjoint 16-bank quarters for ds_read_b64 (banks = dword % 64)

// Wave-wide unsigned minimum, result uniform.  4 DPP steps inside each 16-lane row, then the four
// row results are combined through SGPRs.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
    uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

// index of the 8x8 block (bx,by) in the reference's 85-PU layout (EbMotionEstimationContext.h:51-137;
// z-order of 16x16 blocks from the `offsets` table, EbMotionEstimation.c:367)
__host__ __device__ constexpr int z16_index(int X, int Y) { return (((Y >> 1) * 2 + (X >> 1)) << 2) | ((Y & 1) << 1) | (X & 1); }
__host__ __device__ constexpr int pu8_index(int bx, int by) { return 21 + z16_index(bx >> 1, by >> 1) * 4 + ((by & 1) << 1) + (bx & 1); }

// Fold the 8 candidates of a unit (packed u16 SADs a = cands 0..3, b = cands 4..7) into a key.
// key = sad << 16 | candidate raster index.
__device__ __forceinline__ uint32_t fold_key16(uint32_t key, uint32_t a_lo, uint32_t a_hi, uint32_t b_lo, uint32_t b_hi,
                                               uint32_t idx0) {
    uint32_t k0 = (a_lo << 16) | idx0;
    uint32_t k1 = (a_lo & 0xFFFF0000u) | (idx0 + 1);
    uint32_t k2 = (a_hi << 16) | (idx0 + 2);
    uint32_t k3 = (a_hi & 0xFFFF0000u) | (idx0 + 3);
    uint32_t k4 = (b_lo << 16) | (idx0 + 4);
    uint32_t k5 = (b_lo & 0xFFFF0000u) | (idx0 + 5);
    uint32_t k6 = (b_hi << 16) | (idx0 + 6);
    uint32_t k7 = (b_hi & 0xFFFF0000u) | (idx0 + 7);
    key = min(min(key, k0), k1);
    key = min(min(key, k2), k3);
    key = min(min(key, k4), k5);
    key = min(min(key, k6), k7);
    return key;
}

// The folds below only feed the next unit / the final reduction, so LLVM's machine-sink pass moves
// all 80 of them behind the last SAD loop and keeps every accumulator of the unit alive (448
// registers, one wave per SIMD).  An empty volatile asm that "modifies" the key pins each fold to
// the place it is written, which brings the kernel under 256 registers = two waves per SIMD.
// Read-only for the whole launch and wave-uniform: the constant address space makes the backend pick
// scalar (SMEM) loads even after barriers.
typedef const __attribute__((address_space(4))) uint32_t* const_u32_ptr;

#define PIN(x) asm volatile("" : "+v"(x))

struct Keys {
    uint32_t k8[64];   // indexed by 8x8 PU number - 21
    uint32_t k16[16];  // indexed by 16x16 PU number - 5
    uint64_t k32[4];
    uint64_t k64;
};

// All 85 PUs for the 8 candidates (row y, columns 8g..8g+7 of the tile) owned by this lane.
template <bool SUB>
__device__ __forceinline__ void search_unit(const_u32_ptr src_dw, int src_pitch_dw, uint32_t src_shift,
                                            const uint32_t* __restrict__ lds_ref, int y, int g, uint32_t idx0, Keys& K) {
    uint32_t h16[4][4];       // packed 16-wide partial sums of the even block row: [X][quad*2 + half]
    uint32_t s32[2][8];       // 32x32 running sums per candidate: [X>>1][cand]
    uint32_t s64[8];
#pragma unroll
    for (int c = 0; c < 8; c++) s64[c] = 0;

#pragma unroll
    for (int by = 0; by < 8; by++) {
        uint64_t acc[8][2];
#pragma unroll
        for (int bx = 0; bx < 8; bx++) { acc[bx][0] = 0; acc[bx][1] = 0; }

        const uint32_t* rrow = lds_ref + (y + 8 * by) * kRefStrideDw + 2 * g;
        const_u32_ptr srow = src_dw + (size_t)(8 * by) * src_pitch_dw;
#pragma unroll ME_R_UNROLL
        for (int r = 0; r < 8; r += (SUB ? 2 : 1)) {
            const uint32_t* rp = rrow + r * kRefStrideDw;
            const_u32_ptr sp = srow + (size_t)r * src_pitch_dw;   // wave-uniform: scalar loads, S[] lives in SGPRs
            uint64_t ev[9], od[8];
#pragma unroll
            for (int k = 0; k < 9; k++) ev[k] = *(const uint64_t*)(rp + 2 * k);           // 8-byte aligned
#pragma unroll
            for (int k = 0; k < 8; k++) { Dw2 t = *(const Dw2*)(rp + 2 * k + 1); od[k] = pack64(t.x, t.y); }
            uint32_t Wd[17], S[16];
#pragma unroll
            for (int k = 0; k < 16; k++) Wd[k] = sp[k];
            Wd[16] = src_shift ? sp[16] : 0u;   // the 17th dword only holds source bytes when the row is misaligned
#pragma unroll
            for (int k = 0; k < 16; k++) S[k] = (uint32_t)(pack64(Wd[k], Wd[k + 1]) >> src_shift);
#pragma unroll
            for (int bx = 0; bx < 8; bx++) {
                // quad 0: candidates 0..3 -> ref dwords (2bx, 2bx+1) for the left 4 px, (2bx+1, 2bx+2) for the right 4 px
                acc[bx][0] = __builtin_amdgcn_qsad_pk_u16_u8(ev[bx], S[2 * bx], acc[bx][0]);
                acc[bx][0] = __builtin_amdgcn_qsad_pk_u16_u8(od[bx], S[2 * bx + 1], acc[bx][0]);
                // quad 1: candidates 4..7 -> one dword further
                acc[bx][1] = __builtin_amdgcn_qsad_pk_u16_u8(od[bx], S[2 * bx], acc[bx][1]);
                acc[bx][1] = __builtin_amdgcn_qsad_pk_u16_u8(ev[bx + 1], S[2 * bx + 1], acc[bx][1]);
            }
        }

        // ---- 8x8 PUs of this block row
        uint32_t p[8][4];
#pragma unroll
        for (int bx = 0; bx < 8; bx++) {
            p[bx][0] = (uint32_t)acc[bx][0]; p[bx][1] = (uint32_t)(acc[bx][0] >> 32);
            p[bx][2] = (uint32_t)acc[bx][1]; p[bx][3] = (uint32_t)(acc[bx][1] >> 32);
            if (SUB) {  // rows 0,2,4,6 only, SAD doubled (EbMotionEstimation.c:243-301); 4*8*255*2 < 65536
#pragma unroll
                for (int k = 0; k < 4; k++) p[bx][k] <<= 1;
            }
            uint32_t& key = K.k8[pu8_index(bx, by) - 21];
            key = fold_key16(key, p[bx][0], p[bx][1], p[bx][2], p[bx][3], idx0);
            PIN(key);
        }
        // ---- 16x16: packed u16 adds never carry between halves (max 256*255 = 65280)
        if ((by & 1) == 0) {
#pragma unroll
            for (int X = 0; X < 4; X++)
#pragma unroll
                for (int k = 0; k < 4; k++) h16[X][k] = p[2 * X][k] + p[2 * X + 1][k];
        } else {
#pragma unroll
            for (int X = 0; X < 4; X++) {
                uint32_t q[4];
#pragma unroll
                for (int k = 0; k < 4; k++) q[k] = h16[X][k] + p[2 * X][k] + p[2 * X + 1][k];
                uint32_t& key = K.k16[z16_index(X, by >> 1)];
                key = fold_key16(key, q[0], q[1], q[2], q[3], idx0);
                PIN(key);
                // ---- 32x32 running sums (unpacked)
                const int xh = X >> 1;
                if ((X & 1) == 0 && (by & 3) == 1) {
#pragma unroll
                    for (int k = 0; k < 4; k++) { s32[xh][2 * k] = q[k] & 0xFFFFu; s32[xh][2 * k + 1] = q[k] >> 16; }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) { s32[xh][2 * k] += q[k] & 0xFFFFu; s32[xh][2 * k + 1] += q[k] >> 16; }
                }
            }
            if ((by & 3) == 3) {
#pragma unroll
                for (int xh = 0; xh < 2; xh++) {
                    uint64_t key = K.k32[(by >> 2) * 2 + xh];
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        uint64_t k = ((uint64_t)s32[xh][c] << 32) | (idx0 + c);
                        key = (k < key) ? k : key;
                        s64[c] += s32[xh][c];
                    }
                    PIN(key);
                    K.k32[(by >> 2) * 2 + xh] = key;
                }
            }
        }
    }
    {
        uint64_t key = K.k64;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            uint64_t k = ((uint64_t)s64[c] << 32) | (idx0 + c);
            key = (k < key) ? k : key;
        }
        K.k64 = key;
    }
}

// BIG = false: search areas of up to 65 536 candidates (the packed keys carry a 16-bit raster index); larger ones return at once and are
// taken by the BIG = true instance of the same code, which walks the window in strips of whole candidate rows (<= 65 536 candidates each,
// top to bottom) and merges a strip's winners into the result with the reference's strict '<', i.e. still "first minimum in raster order"
// (the reference configures search areas up to 750 x 750, EbMotionEstimationProcess.c:124-137).
template <int WAVES, bool SUB, bool BIG>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(ME_WAVES_PER_EU, ME_WAVES_PER_EU)))
me_fullpel_85pu_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ ref, int stride, int org_x, int org_y,
                       const SvtHipSbSearch* __restrict__ sbs, uint32_t* __restrict__ best_sad,
                       uint32_t* __restrict__ best_mv) {
    constexpr int NT = 64 * WAVES;
    __shared__ __attribute__((aligned(16))) uint32_t lds_ref[kRefRows * kRefStrideDw];
    __shared__ uint32_t lds_red[WAVES][2][8];   // per-wave reduced (sad, index) of the five 64-bit-key PUs
    __shared__ uint32_t lds_part[NT], lds_fin[88];

    const int sb  = svt_xcd_order(blockIdx.x, gridDim.x);   // neighbouring superblocks' search windows overlap: an XCD takes a band of superblock rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const SvtHipSbSearch d = sbs[sb];
    const int saw = d.width, sah_all = d.height;
    if (saw & 7) return;  // widths 1..7 (window clamped at a picture edge) go to me_fullpel_narrow_kernel
    if (((saw * sah_all) > 65536) != BIG) return;
    const int strip_rows = BIG ? max(1, 65536 / max(saw, 1)) : sah_all;
    int sy0 = 0;
    do {
    const int sah = BIG ? min(strip_rows, sah_all - sy0) : sah_all;   // candidate rows of this strip

    Keys K;
#pragma unroll
    for (int i = 0; i < 64; i++) K.k8[i] = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 16; i++) K.k16[i] = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 4; i++) K.k32[i] = ~0ull;
    K.k64 = ~0ull;

    // The source SB is the same for every lane, so it is read with scalar loads straight into SGPRs
    // (v_qsad's src1 may be an SGPR): no LDS traffic, no VGPRs.  Rows are fetched as aligned dwords
    // and the byte misalignment of the SB origin is shifted out on the scalar ALU.
    const uint8_t*  src_base  = src + (size_t)(org_y + d.sb_y) * stride + (org_x + d.sb_x);
    const uint32_t  src_shift = 8u * (uint32_t)((uintptr_t)src_base & 3);
    const_u32_ptr   src_dw    = (const_u32_ptr)((uintptr_t)src_base & ~(uintptr_t)3);
    const int       src_pitch_dw = stride >> 2;

    for (int ty = 0; ty < sah; ty += kTile) {
        const int th = min(kTile, sah - ty);
        for (int tx = 0; tx < saw; tx += kTile) {
            const int tw = min(kTile, saw - tx);
            const int ng = tw >> 3;  // 8-candidate groups per candidate row (tw is a multiple of 8 here)
            __syncthreads();               // previous tile fully consumed
            const uint8_t* ref_base = ref + (size_t)(org_y + d.sb_y + d.y_origin + sy0 + ty) * stride +
                                      (org_x + d.sb_x + d.x_origin + tx);
            stage_rows<4>(lds_ref, kRefStrideDw, ref_base, stride, th + 63, kRefRowDw, tw + 63, tid, NT);   // 4: the 90 key registers stay live here
            __syncthreads();
            const int units = th * ng;
            for (int u = tid; u < units; u += NT) {
                const int y = u / ng, g = u - y * ng;
                const uint32_t idx0 = (uint32_t)((ty + y) * saw + tx + 8 * g);
                search_unit<SUB>(src_dw, src_pitch_dw, src_shift, lds_ref, y, g, idx0, K);
            }
        }
    }

    // ---- reduce the per-lane keys over the workgroup.
    // The 80 32-bit keys (16x16 and 8x8 PUs) go through LDS transposed: every lane stores its keys
    // as lds_keys[pu][lane], then one thread per (PU, slice) takes the minimum of a slice and a last
    // step merges the slices -- ~200 instructions instead of 80 DPP/readlane wave reductions.
    constexpr int PPH   = 4096 / NT;   // PUs handled per pass (16 KB of the idle window buffer)
    constexpr int PARTS = NT / PPH;    // slices per PU, PPH entries each
    uint32_t* lds_keys = lds_ref;
    const int rp = tid % PPH, rq = tid / PPH;
#pragma unroll
    for (int ph = 0; ph * PPH < 80; ph++) {
        const int cnt = (80 - ph * PPH) < PPH ? (80 - ph * PPH) : PPH;
        __syncthreads();   // window buffer / previous pass no longer read
#pragma unroll
        for (int j = 0; j < cnt; j++) {
            const int i = ph * PPH + j;   // PU number - 5
            lds_keys[j * NT + tid] = (i < 16) ? K.k16[i] : K.k8[i - 16];
        }
        __syncthreads();
        if (rp < cnt) {
            uint32_t m = 0xFFFFFFFFu;
#pragma unroll 8
            for (int j = 0; j < PPH; j++) m = min(m, lds_keys[rp * NT + rq * PPH + ((j + rp) & (PPH - 1))]);  // rotated: conflict-free
            lds_part[rq * PPH + rp] = m;
        }
        __syncthreads();
        if (tid < cnt) {
            uint32_t m = lds_part[tid];
#pragma unroll
            for (int q = 1; q < PARTS; q++) m = min(m, lds_part[q * PPH + tid]);
            lds_fin[5 + ph * PPH + tid] = m;
        }
    }
    // The five 32x32 / 64x64 PUs carry 64-bit keys: two DPP wave reductions each.
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint64_t k = (i < 4) ? K.k32[i] : K.k64;
        const uint32_t s = (uint32_t)(k >> 32), ix = (uint32_t)k;
        const uint32_t ms = wave_min_u32(s);
        const uint32_t mi = wave_min_u32(s == ms ? ix : 0xFFFFFFFFu);
        const int pu = (i < 4) ? 1 + i : 0;
        if (lane == 0) { lds_red[wave][0][pu] = ms; lds_red[wave][1][pu] = mi; }
    }
    __syncthreads();
    for (int pu = tid; pu < SVT_HIP_SQUARE_PU_COUNT; pu += NT) {
        uint32_t bs, bi;
        if (pu >= 5) {
            bs = lds_fin[pu] >> 16; bi = lds_fin[pu] & 0xFFFFu;
        } else {
            bs = lds_red[0][0][pu]; bi = lds_red[0][1][pu];
#pragma unroll
            for (int w = 1; w < WAVES; w++) {
                const uint32_t s = lds_red[w][0][pu], ix = lds_red[w][1][pu];
                if (s < bs || (s == bs && ix < bi)) { bs = s; bi = ix; }
            }
        }
        uint32_t out_sad = SVT_HIP_MAX_SAD_VALUE, out_mv = 0;
        if (saw > 0 && sah > 0 && bi < (uint32_t)(saw * sah)) {
            const int cy = (int)bi / saw, cx = (int)bi - cy * saw;
            // MV word of a candidate: EbMotionEstimation.c:476-478 / :253-255 (quarter-pel int16 halves)
            const uint32_t ymv = (uint32_t)((d.y_origin + sy0 + cy) * 4) & 0xFFFFu;
            const uint32_t xmv = (uint32_t)((d.x_origin + cx) * 4) & 0xFFFFu;
            out_sad = bs;
            out_mv  = (ymv << 16) | xmv;
        }
        // later strips only replace a strictly smaller SAD (this thread wrote the previous strips' value of this PU itself)
        if (!BIG || sy0 == 0 || out_sad < best_sad[(size_t)sb * SVT_HIP_SQUARE_PU_COUNT + pu]) {
            best_sad[(size_t)sb * SVT_HIP_SQUARE_PU_COUNT + pu] = out_sad;
            best_mv[(size_t)sb * SVT_HIP_SQUARE_PU_COUNT + pu]  = out_mv;
        }
    }
    if (BIG) __syncthreads();   // lds_red / lds_fin are rewritten by the next strip
    } while (BIG && (sy0 += strip_rows) < sah_all);
}

// Search areas narrower than 8 candidates (EbMotionEstimation.c:2007-2008 keeps widths 1..7 when
// the window was cropped at a picture edge; the reference then runs the single-candidate kernels
// :122/:191).  Rare and tiny, so: one wave per SB, lane = 8x8 block in PU order, candidates visited
// in raster order with the reference's strict '<' update, sums combined through LDS.
template <bool SUB>
__global__ void __launch_bounds__(64)
me_fullpel_narrow_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ ref, int stride, int org_x, int org_y,
                         const SvtHipSbSearch* __restrict__ sbs, uint32_t* __restrict__ best_sad,
                         uint32_t* __restrict__ best_mv) {
    __shared__ uint32_t s8[64], s16[16], s32[4];
    const int sb = blockIdx.x, lane = threadIdx.x;
    const SvtHipSbSearch d = sbs[sb];
    const int saw = d.width, sah = d.height;
    if (!(saw & 7)) return;
    // lane -> 8x8 block position: PU order is z-order of 16x16 blocks, raster inside a 16x16
    const int z = lane >> 2, q = lane & 3;
    const int X = ((z >> 2) & 1) * 2 + (z & 1), Y = (z >> 3) * 2 + ((z >> 1) & 1);
    const int bx = 2 * X + (q & 1), by = 2 * Y + (q >> 1);
    const uint8_t* s = src + (size_t)(org_y + d.sb_y + 8 * by) * stride + (org_x + d.sb_x + 8 * bx);
    const uint8_t* r0 = ref + (size_t)(org_y + d.sb_y + d.y_origin + 8 * by) * stride + (org_x + d.sb_x + d.x_origin + 8 * bx);
    uint32_t b8 = SVT_HIP_MAX_SAD_VALUE, b16 = SVT_HIP_MAX_SAD_VALUE, b32 = SVT_HIP_MAX_SAD_VALUE, b64 = SVT_HIP_MAX_SAD_VALUE;
    uint32_t m8 = 0, m16 = 0, m32 = 0, m64 = 0;
    for (int cy = 0; cy < sah; cy++)
        for (int cx = 0; cx < saw; cx++) {
            const uint8_t* r = r0 + (size_t)cy * stride + cx;
            uint32_t sad = 0;
            for (int y = 0; y < 8; y += (SUB ? 2 : 1))
                for (int x = 0; x < 8; x++) {
                    const int a = s[y * stride + x], b = r[y * stride + x];
                    sad += (uint32_t)(a > b ? a - b : b - a);
                }
            if (SUB) sad <<= 1;
            const uint32_t mv = ((((uint32_t)((d.y_origin + cy) * 4)) & 0xFFFFu) << 16) | (((uint32_t)((d.x_origin + cx) * 4)) & 0xFFFFu);
            __syncthreads();
            s8[lane] = sad;
            __syncthreads();
            if (lane < 16) s16[lane] = s8[4 * lane] + s8[4 * lane + 1] + s8[4 * lane + 2] + s8[4 * lane + 3];
            __syncthreads();
            if (lane < 4) s32[lane] = s16[4 * lane] + s16[4 * lane + 1] + s16[4 * lane + 2] + s16[4 * lane + 3];
            __syncthreads();
            if (sad < b8) { b8 = sad; m8 = mv; }
            if (lane < 16 && s16[lane] < b16) { b16 = s16[lane]; m16 = mv; }
            if (lane < 4 && s32[lane] < b32) { b32 = s32[lane]; m32 = mv; }
            const uint32_t t = s32[0] + s32[1] + s32[2] + s32[3];
            if (lane == 0 && t < b64) { b64 = t; m64 = mv; }
        }
    uint32_t* os = best_sad + (size_t)sb * SVT_HIP_SQUARE_PU_COUNT;
    uint32_t* om = best_mv + (size_t)sb * SVT_HIP_SQUARE_PU_COUNT;
    os[21 + lane] = b8; om[21 + lane] = m8;
    if (lane < 16) { os[5 + lane] = b16; om[5 + lane] = m16; }
    if (lane < 4) { os[1 + lane] = b32; om[1 + lane] = m32; }
    if (lane == 0) { os[0] = b64; om[0] = m64; }
}

}  // namespace

extern "C" int svt_hip_launch_me_fullpel(hipStream_t stream, const uint8_t* d_src, const uint8_t* d_ref, int stride,
                                         int org_x, int org_y, const SvtHipSbSearch* d_sbs, int n_sb, int sub_sad,
                                         uint32_t* d_best_sad, uint32_t* d_best_mv, int waves_per_sb, int big_windows) {
    if (n_sb <= 0) return 0;
    dim3 grid(n_sb);
    // waves_per_sb >= 16: experimental "one workgroup per CU" mode -- (waves_per_sb >> 4) KiB of unused dynamic LDS keep a second ME
    // workgroup off the CU, leaving half of every SIMD's registers to kernels that run concurrently on other streams
    const int lds_pad = (waves_per_sb >> 4) * 1024;
    waves_per_sb &= 15;
#define LAUNCH(W, S) hipLaunchKernelGGL((me_fullpel_85pu_kernel<W, S, false>), grid, dim3(64 * W), lds_pad, stream, d_src, d_ref, stride, \
                                        org_x, org_y, d_sbs, d_best_sad, d_best_mv)
    if (waves_per_sb == 1) { if (sub_sad) LAUNCH(1, true); else LAUNCH(1, false); }
    else if (waves_per_sb == 2) { if (sub_sad) LAUNCH(2, true); else LAUNCH(2, false); }
    else { if (sub_sad) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
    if (big_windows) {   // search areas above 65 536 candidates: same kernel, strip by strip (SBs with smaller windows return at once)
        if (sub_sad) hipLaunchKernelGGL((me_fullpel_85pu_kernel<4, true, true>), grid, dim3(256), 0, stream, d_src, d_ref, stride, org_x, org_y, d_sbs, d_best_sad, d_best_mv);
        else         hipLaunchKernelGGL((me_fullpel_85pu_kernel<4, false, true>), grid, dim3(256), 0, stream, d_src, d_ref, stride, org_x, org_y, d_sbs, d_best_sad, d_best_mv);
    }
    if (sub_sad) hipLaunchKernelGGL((me_fullpel_narrow_kernel<true>), grid, dim3(64), 0, stream, d_src, d_ref, stride, org_x, org_y, d_sbs, d_best_sad, d_best_mv);
    else         hipLaunchKernelGGL((me_fullpel_narrow_kernel<false>), grid, dim3(64), 0, stream, d_src, d_ref, stride, org_x, org_y, d_sbs, d_best_sad, d_best_mv);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(me_fullpel)
