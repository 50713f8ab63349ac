/*
 * me_oracle.c — CPU restatement of SVT-AV1's open-loop motion-estimation SAD kernels.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 */
#include "svt_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline uint32_t absdiff(int a, int b) { return (uint32_t)(a > b ? a - b : b - a); }

/* Encoder/C_DEFAULT/EbComputeSAD_C.c:20-37 */
uint32_t orc_nxm_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                     uint32_t height, uint32_t width) {
    uint32_t sad = 0;
    for (uint32_t y = 0; y < height; y++, src += src_stride, ref += ref_stride)
        for (uint32_t x = 0; x < width; x++) sad += absdiff(src[x], ref[x]);
    return sad;
}

/* Encoder/C_DEFAULT/EbComputeSAD_C.c:39-56 */
uint32_t orc_sad_16b(const uint16_t *src, uint32_t src_stride, const uint16_t *ref, uint32_t ref_stride,
                     uint32_t height, uint32_t width) {
    uint32_t sad = 0;
    for (uint32_t y = 0; y < height; y++, src += src_stride, ref += ref_stride)
        for (uint32_t x = 0; x < width; x++) sad += absdiff(src[x], ref[x]);
    return sad;
}

/* Encoder/C_DEFAULT/EbComputeSAD_C.c:58-96: exhaustive search, strict '<', raster candidate order,
 * initial best 0xffffff; the ref pointer advances by src_stride_raw per candidate row while the SAD
 * rows advance by ref_stride (this is how the callers implement row sub-sampling). */
void orc_sad_loop(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                  uint32_t block_height, uint32_t block_width, uint64_t *best_sad, int16_t *x_center,
                  int16_t *y_center, uint32_t src_stride_raw, int16_t sa_width, int16_t sa_height) {
    *best_sad = 0xffffff;
    for (int16_t cy = 0; cy < sa_height; cy++) {
        const uint8_t *row = ref + (size_t)cy * src_stride_raw;
        for (int16_t cx = 0; cx < sa_width; cx++) {
            uint32_t sad = 0;
            for (uint32_t y = 0; y < block_height; y++)
                for (uint32_t x = 0; x < block_width; x++)
                    sad += absdiff(src[y * src_stride + x], row[cx + y * ref_stride + x]);
            if (sad < *best_sad) {
                *best_sad = sad;
                *x_center = cx;
                *y_center = cy;
            }
        }
    }
}

/* 8x8 SAD with optional row sub-sampling: EbMotionEstimation.c:66-120 (compute8x4/8x8_sad_kernel_c)
 * and the "<< 1" of the sub_sad branches (:243-301). */
static uint32_t sad8x8(const uint8_t *s, uint32_t ss, const uint8_t *r, uint32_t rs, int sub_sad) {
    uint32_t sad = 0;
    if (sub_sad) {
        for (int y = 0; y < 8; y += 2)
            for (int x = 0; x < 8; x++) sad += absdiff(s[y * ss + x], r[y * rs + x]);
        return sad << 1;
    }
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) sad += absdiff(s[y * ss + x], r[y * rs + x]);
    return sad;
}

/* The MV word a candidate stores: EbMotionEstimation.c:251-255 — x/y of `mv` are quarter-pel
 * int16 halves (EbDefinitions.h:2346-2347); candidate k of an 8-group adds 4*k to x. */
static inline uint32_t mv_plus_x(uint32_t mv, int k) {
    int16_t x = (int16_t)(mv & 0xffff) + (int16_t)(4 * k);
    int16_t y = (int16_t)(mv >> 16);
    return ((uint32_t)(uint16_t)y << 16) | (uint16_t)x;
}

/* z-order index of the 16x16 block (X,Y) in the 4x4 grid: the `offsets` table of
 * EbMotionEstimation.c:367 (== tab16x16, EbMotionEstimation.h:108). */
static inline int z16(int X, int Y) { return (((Y >> 1) * 2 + (X >> 1)) << 2) | ((Y & 1) << 1) | (X & 1); }

/* EbMotionEstimation.c:362-391 with the inlined per-16x16 helper :230-359. */
void orc_ext_all_sad_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                               uint32_t ref_stride, uint32_t mv, uint32_t *best_sad8, uint32_t *best_sad16,
                               uint32_t *best_mv8, uint32_t *best_mv16, uint32_t eight_sad16[16][8],
                               uint32_t eight_sad8[64][8], int sub_sad) {
    for (int Y = 0; Y < 4; Y++)
        for (int X = 0; X < 4; X++) {
            const int      p16 = z16(X, Y), p8 = 4 * p16;
            const uint8_t *s   = src + 16 * Y * src_stride + 16 * X;
            const uint8_t *r   = ref + 16 * Y * ref_stride + 16 * X;
            for (int k = 0; k < 8; k++) {
                uint32_t sum = 0;
                for (int q = 0; q < 4; q++) { /* q: TL, TR, BL, BR */
                    const uint32_t off_s = (q >> 1) * 8 * src_stride + (q & 1) * 8;
                    const uint32_t off_r = (q >> 1) * 8 * ref_stride + (q & 1) * 8;
                    const uint32_t sad   = sad8x8(s + off_s, src_stride, r + off_r + k, ref_stride, sub_sad);
                    eight_sad8[p8 + q][k] = sad;
                    if (sad < best_sad8[p8 + q]) {
                        best_sad8[p8 + q] = sad;
                        best_mv8[p8 + q]  = mv_plus_x(mv, k);
                    }
                    sum += sad;
                }
                eight_sad16[p16][k] = sum;
                if (sum < best_sad16[p16]) {
                    best_sad16[p16] = sum;
                    best_mv16[p16]  = mv_plus_x(mv, k);
                }
            }
        }
}

/* EbMotionEstimation.c:396-459 */
void orc_ext_eight_sad_32x32_64x64(uint32_t sad16[16][8], uint32_t *best_sad32, uint32_t *best_sad64,
                                   uint32_t *best_mv32, uint32_t *best_mv64, uint32_t mv,
                                   uint32_t sad32[4][8]) {
    for (int k = 0; k < 8; k++) {
        uint32_t total = 0;
        for (int q = 0; q < 4; q++) {
            const uint32_t s = sad16[4 * q][k] + sad16[4 * q + 1][k] + sad16[4 * q + 2][k] + sad16[4 * q + 3][k];
            sad32[q][k]      = s;
            if (s < best_sad32[q]) {
                best_sad32[q] = s;
                best_mv32[q]  = mv_plus_x(mv, k);
            }
            total += s;
        }
        if (total < best_sad64[0]) {
            best_sad64[0] = total;
            best_mv64[0]  = mv_plus_x(mv, k);
        }
    }
}

/* EbMotionEstimation.c:122-186 (single candidate; stores `mv` unchanged) */
void orc_ext_sad_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                           uint32_t *best_sad8, uint32_t *best_sad16, uint32_t *best_mv8,
                           uint32_t *best_mv16, uint32_t mv, uint32_t *sad16, uint32_t *sad8, int sub_sad) {
    uint32_t sum = 0;
    for (int q = 0; q < 4; q++) {
        sad8[q] = sad8x8(src + (q >> 1) * 8 * src_stride + (q & 1) * 8, src_stride,
                         ref + (q >> 1) * 8 * ref_stride + (q & 1) * 8, ref_stride, sub_sad);
        if (sad8[q] < best_sad8[q]) {
            best_sad8[q] = sad8[q];
            best_mv8[q]  = mv;
        }
        sum += sad8[q];
    }
    if (sum < best_sad16[0]) {
        best_sad16[0] = sum;
        best_mv16[0]  = mv;
    }
    *sad16 = sum;
}

/* EbMotionEstimation.c:191-225 */
void orc_ext_sad_32x32_64x64(const uint32_t *sad16, uint32_t *best_sad32, uint32_t *best_sad64,
                             uint32_t *best_mv32, uint32_t *best_mv64, uint32_t mv, uint32_t *sad32) {
    uint32_t total = 0;
    for (int q = 0; q < 4; q++) {
        sad32[q] = sad16[4 * q] + sad16[4 * q + 1] + sad16[4 * q + 2] + sad16[4 * q + 3];
        if (sad32[q] < best_sad32[q]) {
            best_sad32[q] = sad32[q];
            best_mv32[q]  = mv;
        }
        total += sad32[q];
    }
    if (total < best_sad64[0]) {
        best_sad64[0] = total;
        best_mv64[0]  = mv;
    }
}

/* EbMotionEstimation.c:814-850 (candidate loops: groups of 8 through the "eight" kernels, the
 * width%8 tail through the single-candidate kernels :508-812), :462-506 (MV word of a candidate:
 * ((uint16)y << 18) | (uint16)(x << 2)), :2086 (best SAD init to MAX_SAD_VALUE; MVs untouched -> we
 * define them as 0 so that outputs are deterministic when sa is empty). */
void orc_me_fullpel_sb(const uint8_t *src, uint32_t src_stride, const uint8_t *ref_tl, uint32_t ref_stride,
                       int x_sa_origin, int y_sa_origin, uint32_t sa_width, uint32_t sa_height, int sub_sad,
                       uint32_t best_sad[ORC_SQUARE_PU_COUNT], uint32_t best_mv[ORC_SQUARE_PU_COUNT]) {
    uint32_t eight16[16][8], eight8[64][8], eight32[4][8];
    uint32_t one16[16], one8[64], one32[4];
    for (int i = 0; i < ORC_SQUARE_PU_COUNT; i++) {
        best_sad[i] = ORC_MAX_SAD_VALUE;
        best_mv[i]  = 0;
    }
    uint32_t *bs64 = best_sad, *bs32 = best_sad + 1, *bs16 = best_sad + 5, *bs8 = best_sad + 21;
    uint32_t *bm64 = best_mv, *bm32 = best_mv + 1, *bm16 = best_mv + 5, *bm8 = best_mv + 21;
    const uint32_t w8 = sa_width & ~7u;
    for (uint32_t cy = 0; cy < sa_height; cy++) {
        const uint8_t *row = ref_tl + (size_t)cy * ref_stride;
        const int      ys  = (int)cy + y_sa_origin;
        for (uint32_t cx = 0; cx < sa_width; cx += (cx < w8 ? 8 : 1)) {
            const int      xs = (int)cx + x_sa_origin;
            const uint32_t mv = ((uint32_t)(uint16_t)ys << 18) | (uint16_t)((uint16_t)xs << 2);
            if (cx < w8) {
                orc_ext_all_sad_8x8_16x16(src, src_stride, row + cx, ref_stride, mv, bs8, bs16, bm8, bm16,
                                          eight16, eight8, sub_sad);
                orc_ext_eight_sad_32x32_64x64(eight16, bs32, bs64, bm32, bm64, mv, eight32);
            } else {
                for (int Y = 0; Y < 4; Y++)
                    for (int X = 0; X < 4; X++) {
                        const int p16 = z16(X, Y);
                        orc_ext_sad_8x8_16x16(src + 16 * Y * src_stride + 16 * X, src_stride,
                                              row + cx + 16 * Y * ref_stride + 16 * X, ref_stride,
                                              bs8 + 4 * p16, bs16 + p16, bm8 + 4 * p16, bm16 + p16, mv,
                                              one16 + p16, one8 + 4 * p16, sub_sad);
                    }
                orc_ext_sad_32x32_64x64(one16, bs32, bs64, bm32, bm64, mv, one32);
            }
        }
    }
}

/* EbMotionEstimation.c:1945-2066, unrestricted_motion_vector branch.  All quantities are int16 in
 * the reference; the intermediate sums there are int (integer promotion) and then truncated on
 * assignment — reproduced with explicit casts. */
OrcSearchWindow orc_me_search_window(int sb_origin_x, int sb_origin_y, int x_center, int y_center,
                                     int sa_width, int sa_height, int pic_width, int pic_height) {
    const int16_t pad = 63; /* BLOCK_SIZE_64 - 1, :1886-1887 */
    int16_t ox = (int16_t)sb_origin_x, oy = (int16_t)sb_origin_y;
    int16_t w = (int16_t)sa_width, h = (int16_t)sa_height;
    int16_t pw = (int16_t)pic_width, ph = (int16_t)pic_height;
    int16_t xo = (int16_t)(x_center - (w >> 1)); /* :1945 */
    int16_t yo = (int16_t)(y_center - (h >> 1));
    /* left edge (:1981-1988).  The reference first clamps the origin and only then evaluates the
     * width expression with the *already clamped* origin, so the width branch never fires; the
     * statements are restated in the same order to keep that behaviour. */
    xo = (int16_t)((ox + xo < -pad) ? -pad - ox : xo);
    w  = (int16_t)((ox + xo < -pad) ? w - (-pad - (ox + xo)) : w);
    /* right edge (:1991-2003) */
    xo = (int16_t)((ox + xo > pw - 1) ? xo - ((ox + xo) - (pw - 1)) : xo);
    if (ox + xo + w > pw) {
        int v = w - ((ox + xo + w) - pw);
        w     = (int16_t)(v > 1 ? v : 1);
    }
    w = (int16_t)((w < 8) ? w : (w & ~0x07)); /* :2007-2008 */
    /* top edge (:2044-2051), same ordering remark as for the left edge */
    yo = (int16_t)((oy + yo < -pad) ? -pad - oy : yo);
    h  = (int16_t)((oy + yo < -pad) ? h - (-pad - (oy + yo)) : h);
    /* bottom edge (:2054-2065) */
    yo = (int16_t)((oy + yo > ph - 1) ? yo - ((oy + yo) - (ph - 1)) : yo);
    if (oy + yo + h > ph) {
        int v = h - ((oy + yo + h) - ph);
        h     = (int16_t)(v > 1 ? v : 1);
    }
    OrcSearchWindow out = {xo, yo, w, h};
    return out;
}

void orc_me_fullpel_frame(const uint8_t *src, const uint8_t *ref, int stride, int org_x, int org_y,
                          const OrcSbSearch *sbs, int n_sb, int sub_sad, uint32_t *best_sad,
                          uint32_t *best_mv, int sb_begin, int sb_end) {
    (void)n_sb;
    for (int i = sb_begin; i < sb_end; i++) {
        const OrcSbSearch *d = &sbs[i];
        const uint8_t *s = src + (size_t)(org_y + d->sb_y) * stride + org_x + d->sb_x;
        const uint8_t *r = ref + (size_t)(org_y + d->sb_y + d->y_origin) * stride + org_x + d->sb_x + d->x_origin;
        orc_me_fullpel_sb(s, stride, r, stride, d->x_origin, d->y_origin, (uint32_t)d->width, (uint32_t)d->height,
                          sub_sad, best_sad + (size_t)i * ORC_SQUARE_PU_COUNT, best_mv + (size_t)i * ORC_SQUARE_PU_COUNT);
    }
}
