/*
 * conv_oracle.c — CPU restatement of SVT-AV1's single-reference sub-pel convolve (normative AV1
 * interpolation, *_sr), the sub-pel-search predictor (svt_aom_upsampled_pred: two convolve8 passes
 * with an 8-bit intermediate) and block variance / SAD helpers.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 */
#include "svt_oracle.h"
#include <string.h>
#include <stdlib.h>

/* AV1 interpolation kernels (normative constants): Common/Codec/EbInterPrediction.c:258-291, :1181-1249.
 * bank 0 EIGHTTAP_REGULAR, 1 EIGHTTAP_SMOOTH, 2 MULTITAP_SHARP, 3 BILINEAR, 4 4-tap regular (w <= 4),
 * 5 4-tap smooth (w <= 4) */
const int16_t orc_interp_kernels[6][16][8] = {
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 2, -6, 126, 8, -2, 0, 0}, {0, 2, -10, 122, 18, -4, 0, 0}, {0, 2, -12, 116, 28, -8, 2, 0},
     {0, 2, -14, 110, 38, -10, 2, 0}, {0, 2, -14, 102, 48, -12, 2, 0}, {0, 2, -16, 94, 58, -12, 2, 0}, {0, 2, -14, 84, 66, -12, 2, 0},
     {0, 2, -14, 76, 76, -14, 2, 0}, {0, 2, -12, 66, 84, -14, 2, 0}, {0, 2, -12, 58, 94, -16, 2, 0}, {0, 2, -12, 48, 102, -14, 2, 0},
     {0, 2, -10, 38, 110, -14, 2, 0}, {0, 2, -8, 28, 116, -12, 2, 0}, {0, 0, -4, 18, 122, -10, 2, 0}, {0, 0, -2, 8, 126, -6, 2, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 2, 28, 62, 34, 2, 0, 0}, {0, 0, 26, 62, 36, 4, 0, 0}, {0, 0, 22, 62, 40, 4, 0, 0},
     {0, 0, 20, 60, 42, 6, 0, 0}, {0, 0, 18, 58, 44, 8, 0, 0}, {0, 0, 16, 56, 46, 10, 0, 0}, {0, -2, 16, 54, 48, 12, 0, 0},
     {0, -2, 14, 52, 52, 14, -2, 0}, {0, 0, 12, 48, 54, 16, -2, 0}, {0, 0, 10, 46, 56, 16, 0, 0}, {0, 0, 8, 44, 58, 18, 0, 0},
     {0, 0, 6, 42, 60, 20, 0, 0}, {0, 0, 4, 40, 62, 22, 0, 0}, {0, 0, 4, 36, 62, 26, 0, 0}, {0, 0, 2, 34, 62, 28, 2, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {-2, 2, -6, 126, 8, -2, 2, 0}, {-2, 6, -12, 124, 16, -6, 4, -2}, {-2, 8, -18, 120, 26, -10, 6, -2},
     {-4, 10, -22, 116, 38, -14, 6, -2}, {-4, 10, -22, 108, 48, -18, 8, -2}, {-4, 10, -24, 100, 60, -20, 8, -2}, {-4, 10, -24, 90, 70, -22, 10, -2},
     {-4, 12, -24, 80, 80, -24, 12, -4}, {-2, 10, -22, 70, 90, -24, 10, -4}, {-2, 8, -20, 60, 100, -24, 10, -4}, {-2, 8, -18, 48, 108, -22, 10, -4},
     {-2, 6, -14, 38, 116, -22, 10, -4}, {-2, 6, -10, 26, 120, -18, 8, -2}, {-2, 4, -6, 16, 124, -12, 6, -2}, {0, 2, -2, 8, 126, -6, 2, -2}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, 0, 120, 8, 0, 0, 0}, {0, 0, 0, 112, 16, 0, 0, 0}, {0, 0, 0, 104, 24, 0, 0, 0},
     {0, 0, 0, 96, 32, 0, 0, 0}, {0, 0, 0, 88, 40, 0, 0, 0}, {0, 0, 0, 80, 48, 0, 0, 0}, {0, 0, 0, 72, 56, 0, 0, 0},
     {0, 0, 0, 64, 64, 0, 0, 0}, {0, 0, 0, 56, 72, 0, 0, 0}, {0, 0, 0, 48, 80, 0, 0, 0}, {0, 0, 0, 40, 88, 0, 0, 0},
     {0, 0, 0, 32, 96, 0, 0, 0}, {0, 0, 0, 24, 104, 0, 0, 0}, {0, 0, 0, 16, 112, 0, 0, 0}, {0, 0, 0, 8, 120, 0, 0, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, -4, 126, 8, -2, 0, 0}, {0, 0, -8, 122, 18, -4, 0, 0}, {0, 0, -10, 116, 28, -6, 0, 0},
     {0, 0, -12, 110, 38, -8, 0, 0}, {0, 0, -12, 102, 48, -10, 0, 0}, {0, 0, -14, 94, 58, -10, 0, 0}, {0, 0, -12, 84, 66, -10, 0, 0},
     {0, 0, -12, 76, 76, -12, 0, 0}, {0, 0, -10, 66, 84, -12, 0, 0}, {0, 0, -10, 58, 94, -14, 0, 0}, {0, 0, -10, 48, 102, -12, 0, 0},
     {0, 0, -8, 38, 110, -12, 0, 0}, {0, 0, -6, 28, 116, -10, 0, 0}, {0, 0, -4, 18, 122, -8, 0, 0}, {0, 0, -2, 8, 126, -4, 0, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, 30, 62, 34, 2, 0, 0}, {0, 0, 26, 62, 36, 4, 0, 0}, {0, 0, 22, 62, 40, 4, 0, 0},
     {0, 0, 20, 60, 42, 6, 0, 0}, {0, 0, 18, 58, 44, 8, 0, 0}, {0, 0, 16, 56, 46, 10, 0, 0}, {0, 0, 14, 54, 48, 12, 0, 0},
     {0, 0, 12, 52, 52, 12, 0, 0}, {0, 0, 12, 48, 54, 14, 0, 0}, {0, 0, 10, 46, 56, 16, 0, 0}, {0, 0, 8, 44, 58, 18, 0, 0},
     {0, 0, 6, 42, 60, 20, 0, 0}, {0, 0, 4, 40, 62, 22, 0, 0}, {0, 0, 4, 36, 62, 26, 0, 0}, {0, 0, 2, 34, 62, 30, 0, 0}}};

static inline int rp2(int v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); }
static inline int clipbd(int v, int bd) { const int m = (1 << bd) - 1; return v < 0 ? 0 : (v > m ? m : v); }
static inline int rdp(const void *b, int pb, ptrdiff_t i) { return pb == 1 ? ((const uint8_t *)b)[i] : ((const uint16_t *)b)[i]; }
static inline void wrp(void *b, int pb, ptrdiff_t i, int v) { if (pb == 1) ((uint8_t *)b)[i] = (uint8_t)v; else ((uint16_t *)b)[i] = (uint16_t)v; }

/* svt_av1_[highbd_]convolve_{2d_copy,x,y,2d}_sr_c, selected like convolve[sx != 0][sy != 0][0]
 * (Common/Codec/EbInterPrediction.c:349-469, :744-866, :1161-1174), round_0 = 3, round_1 = 11
 * (non-compound, bd <= 10).  src points at the block's top-left integer sample. */
void orc_convolve_sr(const void *src, int src_stride, void *dst, int dst_stride, int pix_bytes, int w, int h, int bank_x, int bank_y,
                     int subpel_x_q4, int subpel_y_q4, int bd) {
    const int16_t *xf = orc_interp_kernels[bank_x][subpel_x_q4 & 15], *yf = orc_interp_kernels[bank_y][subpel_y_q4 & 15];
    const int r0 = 3, r1 = 11, fo = 3;
    if (!(subpel_x_q4 & 15) && !(subpel_y_q4 & 15)) {
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) wrp(dst, pix_bytes, (ptrdiff_t)y * dst_stride + x, rdp(src, pix_bytes, (ptrdiff_t)y * src_stride + x));
    } else if (!(subpel_y_q4 & 15)) {
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            int res = 0;
            for (int k = 0; k < 8; k++) res += xf[k] * rdp(src, pix_bytes, (ptrdiff_t)y * src_stride + x - fo + k);
            res = rp2(res, r0);
            wrp(dst, pix_bytes, (ptrdiff_t)y * dst_stride + x, clipbd(rp2(res, 7 - r0), bd));
        }
    } else if (!(subpel_x_q4 & 15)) {
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            int res = 0;
            for (int k = 0; k < 8; k++) res += yf[k] * rdp(src, pix_bytes, (ptrdiff_t)(y - fo + k) * src_stride + x);
            wrp(dst, pix_bytes, (ptrdiff_t)y * dst_stride + x, clipbd(rp2(res, 7), bd));
        }
    } else {
        int16_t im[(128 + 7) * 128];
        const int im_h = h + 7, offset_bits = bd + 14 - r0, bits = 14 - r0 - r1;
        for (int y = 0; y < im_h; y++) for (int x = 0; x < w; x++) {
            int sum = 1 << (bd + 6);
            for (int k = 0; k < 8; k++) sum += xf[k] * rdp(src, pix_bytes, (ptrdiff_t)(y - fo) * src_stride + x - fo + k);
            im[y * w + x] = (int16_t)rp2(sum, r0);
        }
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            int sum = 1 << offset_bits;
            for (int k = 0; k < 8; k++) sum += yf[k] * im[(y + k) * w + x];
            int res = rp2(sum, r1) - ((1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1)));
            if (pix_bytes == 1) res = (int16_t)res;
            wrp(dst, pix_bytes, (ptrdiff_t)y * dst_stride + x, clipbd(rp2(res, bits), bd));
        }
    }
}

/* svt_aom_upsampled_pred_c (Encoder/C_DEFAULT/variance.c:212-269) over svt_aom_convolve8_{horiz,vert}_c
 * (Common/Codec/convolve.c:249-307): 1/8-pel phases, 8-bit intermediate, output packed with stride `width`.
 * bank: 3 (USE_2_TAPS), 4 (USE_4_TAPS) or 0 (USE_8_TAPS) — av1_get_filter, variance.c:198-209. */
void orc_upsampled_pred(const uint8_t *ref, int ref_stride, uint8_t *pred, int width, int height, int subpel_x_q3, int subpel_y_q3, int bank) {
    const int16_t *kx = orc_interp_kernels[bank][(subpel_x_q3 << 1) & 15], *ky = orc_interp_kernels[bank][(subpel_y_q3 << 1) & 15];
    if (!subpel_x_q3 && !subpel_y_q3) {
        for (int i = 0; i < height; i++) memcpy(pred + i * width, ref + (ptrdiff_t)i * ref_stride, width);
    } else if (!subpel_y_q3) {
        for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) {
            int s = 0; for (int k = 0; k < 8; k++) s += ref[(ptrdiff_t)y * ref_stride + x - 3 + k] * kx[k];
            pred[y * width + x] = (uint8_t)clipbd(rp2(s, 7), 8);
        }
    } else if (!subpel_x_q3) {
        for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) {
            int s = 0; for (int k = 0; k < 8; k++) s += ref[(ptrdiff_t)(y - 3 + k) * ref_stride + x] * ky[k];
            pred[y * width + x] = (uint8_t)clipbd(rp2(s, 7), 8);
        }
    } else {
        static uint8_t temp[(128 * 2 + 32) * 128];
        const int ih = (((height - 1) * 8 + subpel_y_q3) >> 3) + 8;
        uint8_t *t = (uint8_t *)__builtin_alloca((size_t)ih * 128);
        (void)temp;
        for (int y = 0; y < ih; y++) for (int x = 0; x < width; x++) {
            int s = 0; for (int k = 0; k < 8; k++) s += ref[(ptrdiff_t)(y - 3) * ref_stride + x - 3 + k] * kx[k];
            t[y * 128 + x] = (uint8_t)clipbd(rp2(s, 7), 8);
        }
        for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) {
            int s = 0; for (int k = 0; k < 8; k++) s += t[(y + k) * 128 + x] * ky[k];
            pred[y * width + x] = (uint8_t)clipbd(rp2(s, 7), 8);
        }
    }
}

/* svt_aom_variance{W}x{H}_c (Encoder/C_DEFAULT/EbComputeVariance_C.c:14-77) */
uint32_t orc_variance(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse) {
    int sum = 0; uint32_t s2 = 0;
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) { const int d = a[(ptrdiff_t)i * a_stride + j] - b[(ptrdiff_t)i * b_stride + j]; sum += d; s2 += (uint32_t)(d * d); }
    *sse = s2;
    return s2 - (uint32_t)(((int64_t)sum * sum) / (w * h));
}
/* svt_aom_highbd_10_variance{W}x{H}_c (Encoder/Codec/EbPsnr.c:170-233) */
uint32_t orc_variance_hbd10(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h, uint32_t *sse) {
    int64_t tsum = 0; uint64_t tsse = 0;
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) { const int d = a[(ptrdiff_t)i * a_stride + j] - b[(ptrdiff_t)i * b_stride + j]; tsum += d; tsse += (uint32_t)(d * d); }
    *sse = (uint32_t)((tsse + 8) >> 4);
    const int sum = (int)((tsum + 2) >> 2);
    const int64_t var = (int64_t)(*sse) - (((int64_t)sum * sum) / (w * h));
    return var >= 0 ? (uint32_t)var : 0;
}

/* The prediction batch svt_hip_subpel_predict_batch_dev receives (include/svt_hip.h, SvtHipConvBlk), blocks [begin, end), walked
 * with the single-block restatements above (mode 0: convolve_*_sr, mode 1: upsampled_pred).  Tests / bench.py CPU baseline. */
typedef struct {
    int32_t src_x, src_y, dst_x, dst_y;
    uint8_t w, h, bank_x, bank_y, subpel_x, subpel_y, mode, reserved;
} OrcConvBlk;
void orc_subpel_predict_batch(int pix_bytes, int bd, const void *ref, int ref_stride, void *dst, int dst_stride, const void *blks_, int begin, int end) {
    const OrcConvBlk *blks = (const OrcConvBlk *)blks_;
    for (int i = begin; i < end; i++) {
        const OrcConvBlk *b = &blks[i];
        const uint8_t *s = (const uint8_t *)ref + ((ptrdiff_t)b->src_y * ref_stride + b->src_x) * pix_bytes;
        uint8_t *d = (uint8_t *)dst + ((size_t)b->dst_y * dst_stride + b->dst_x) * pix_bytes;
        if (b->mode == 0) {
            orc_convolve_sr(s, ref_stride, d, dst_stride, pix_bytes, b->w, b->h, b->bank_x, b->bank_y, b->subpel_x, b->subpel_y, bd);
        } else {
            uint8_t tmp[128 * 128];
            orc_upsampled_pred(s, ref_stride, tmp, b->w, b->h, b->subpel_x >> 1, b->subpel_y >> 1, b->bank_x);
            for (int y = 0; y < b->h; y++) memcpy(d + (size_t)y * dst_stride, tmp + y * b->w, b->w);
        }
    }
}

/* ---------------------------------------------------------------------------------------------------------------------------------
 * Compound inter prediction.
 * svt_av1_[highbd_]jnt_convolve_{2d_copy,x,y,2d}_c with do_average == 0 (Common/Codec/EbInterPrediction.c:552-741, :944-1143): the
 * 16-bit intermediate ("ConvBufType") prediction of one reference; round_0 = 3 (5 for 12-bit), round_1 = COMPOUND_ROUND1_BITS = 7. */
void orc_jnt_convolve_d16(const void *src, int src_stride, int pix_bytes, int w, int h, int bank_x, int bank_y, int subpel_x_q4, int subpel_y_q4, int bd,
                          uint16_t *out, int out_stride) {
    const int16_t *xf = orc_interp_kernels[bank_x][subpel_x_q4 & 15], *yf = orc_interp_kernels[bank_y][subpel_y_q4 & 15];
    const int r0 = bd == 12 ? 5 : 3, r1 = 7, fo = 3;
    const int offset_bits = bd + 14 - r0, round_offset = (1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1));
    const int sx = subpel_x_q4 & 15, sy = subpel_y_q4 & 15;
    if (!sx && !sy) {                                   /* 2d_copy :704-741 */
        const int bits = 14 - r1 - r0;
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
            out[y * out_stride + x] = (uint16_t)((rdp(src, pix_bytes, (ptrdiff_t)y * src_stride + x) << bits) + round_offset);
    } else if (!sy) {                                   /* x :658-702 */
        const int bits = 7 - r1;
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            int res = 0;
            for (int k = 0; k < 8; k++) res += xf[k] * rdp(src, pix_bytes, (ptrdiff_t)y * src_stride + x - fo + k);
            res = (1 << bits) * rp2(res, r0) + round_offset;
            out[y * out_stride + x] = (uint16_t)res;
        }
    } else if (!sx) {                                   /* y :612-656 */
        const int bits = 7 - r0;
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            int res = 0;
            for (int k = 0; k < 8; k++) res += yf[k] * rdp(src, pix_bytes, (ptrdiff_t)(y - fo + k) * src_stride + x);
            res *= (1 << bits);
            res = rp2(res, r1) + round_offset;
            out[y * out_stride + x] = (uint16_t)res;
        }
    } else {                                            /* 2d :552-610 */
        int16_t im[(128 + 7) * 128];
        for (int y = 0; y < h + 7; y++) for (int x = 0; x < w; x++) {
            int sum = 1 << (bd + 6);
            for (int k = 0; k < 8; k++) sum += xf[k] * rdp(src, pix_bytes, (ptrdiff_t)(y - fo) * src_stride + x - fo + k);
            im[y * w + x] = (int16_t)rp2(sum, r0);
        }
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            int sum = 1 << offset_bits;
            for (int k = 0; k < 8; k++) sum += yf[k] * im[(y + k) * w + x];
            out[y * out_stride + x] = (uint16_t)rp2(sum, r1);
        }
    }
}

/* The blocks svt_hip_compound_predict_batch_dev receives (include/svt_hip.h, SvtHipCompBlk) */
typedef struct {
    int32_t src0_x, src0_y, src1_x, src1_y, dst_x, dst_y;
    uint8_t w, h, bank_x, bank_y;
    uint8_t subpel0_x, subpel0_y, subpel1_x, subpel1_y;
    uint8_t type, fwd_offset, bck_offset, mask_type;
    uint8_t mask_sub, reserved[3];
    int32_t mask_off, mask_stride;
} OrcCompBlk;

/* Both references of a compound block and their combination:
 *   type 0 COMPOUND_AVERAGE / 1 COMPOUND_DISTANCE: the do_average branch of the jnt_convolve functions (:593-606), use_jnt_comp_avg = type;
 *   type 2 COMPOUND_DIFFWTD: svt_av1_build_compound_diffwtd_mask_d16_c (Common/C_DEFAULT/EbInterPrediction_c.c:15-43) written to
 *          masks + mask_off (stride w; skipped when mask_off < 0), then the blend;
 *   type 3 mask supplied (wedge, or a previously built seg_mask; mask_sub = 1: the mask is at luma resolution for a 4:2:0 chroma block)
 *   blend = svt_aom_{lowbd,highbd}_blend_a64_d16_mask_c (Common/Codec/EbBlend_a64_mask.c:34-110 / :112-230) as build_masked_compound_no_round
 *   calls it (Encoder/Codec/EbEncInterPrediction.c:60-120). */
void orc_compound_predict_batch(int pix_bytes, int bd, const void *ref0, int ref0_stride, const void *ref1, int ref1_stride, void *dst, int dst_stride,
                                uint8_t *masks, const void *blks_, int begin, int end) {
    const OrcCompBlk *blks = (const OrcCompBlk *)blks_;
    static _Thread_local uint16_t a[128 * 128], b[128 * 128];
    static _Thread_local uint8_t seg[128 * 128];
    const int r0 = bd == 12 ? 5 : 3, r1 = 7;
    const int offset_bits = bd + 14 - r0, round_offset = (1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1)), round_bits = 14 - r0 - r1;
    for (int i = begin; i < end; i++) {
        const OrcCompBlk *c = &blks[i];
        const int w = c->w, h = c->h;
        orc_jnt_convolve_d16((const uint8_t *)ref0 + ((ptrdiff_t)c->src0_y * ref0_stride + c->src0_x) * pix_bytes, ref0_stride, pix_bytes, w, h, c->bank_x, c->bank_y,
                             c->subpel0_x, c->subpel0_y, bd, a, w);
        orc_jnt_convolve_d16((const uint8_t *)ref1 + ((ptrdiff_t)c->src1_y * ref1_stride + c->src1_x) * pix_bytes, ref1_stride, pix_bytes, w, h, c->bank_x, c->bank_y,
                             c->subpel1_x, c->subpel1_y, bd, b, w);
        const uint8_t *m = NULL; int ms = 0;
        if (c->type == 2) {
            const int round = 14 - r0 - r1 + (bd - 8);
            for (int k = 0; k < w * h; k++) {
                int diff = abs((int)a[k] - (int)b[k]);
                diff = rp2(diff, round);
                int mm = 38 + diff / 16;                                       /* DIFF_FACTOR = 16 */
                mm = mm < 0 ? 0 : (mm > 64 ? 64 : mm);
                seg[k] = (uint8_t)(c->mask_type ? 64 - mm : mm);
            }
            if (c->mask_off >= 0) for (int y = 0; y < h; y++) memcpy(masks + c->mask_off + (size_t)y * w, seg + y * w, (size_t)w);
            m = seg; ms = w;
        } else if (c->type == 3) { m = masks + c->mask_off; ms = c->mask_stride; }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int k = y * w + x;
                int tmp;
                if (c->type <= 1) {
                    tmp = c->type ? ((int)a[k] * c->fwd_offset + (int)b[k] * c->bck_offset) >> 4 : ((int)a[k] + (int)b[k]) >> 1;
                } else {
                    int mm;
                    if (c->type == 3 && c->mask_sub)
                        mm = rp2(m[(2 * y) * ms + 2 * x] + m[(2 * y + 1) * ms + 2 * x] + m[(2 * y) * ms + 2 * x + 1] + m[(2 * y + 1) * ms + 2 * x + 1], 2);
                    else mm = m[y * ms + x];
                    tmp = (mm * (int)a[k] + (64 - mm) * (int)b[k]) >> 6;
                }
                tmp -= round_offset;
                wrp(dst, pix_bytes, (ptrdiff_t)(c->dst_y + y) * dst_stride + c->dst_x + x, clipbd(rp2(tmp, round_bits), bd));
            }
    }
}

/* ---------------------------------------------------------------------------------------------------------------------------------
 * OBMC motion-search costs: svt_aom_obmc_sad{W}x{H}_c (Encoder/C_DEFAULT/sad_av1.c:18-38),
 * svt_aom_obmc_variance{W}x{H}_c and svt_aom_obmc_sub_pixel_variance{W}x{H}_c (Encoder/C_DEFAULT/variance.c:270-318; the 2-tap bilinear
 * passes :32-75, bilinear_filters_2t[k] = {128 - 16 k, 16 k}).  wsrc / mask: int32, stride w.  out = {sad (unfiltered pre), sse, variance
 * (pre filtered at (xoffset, yoffset) eighths; offset 0 is the identity, i.e. the plain variance function)}. */
void orc_obmc_block(const uint8_t *pre, int pre_stride, const int32_t *wsrc, const int32_t *mask, int w, int h, int xoffset, int yoffset, uint32_t out[3]) {
    uint32_t sad = 0, sse = 0;
    int sum = 0;
    static _Thread_local uint16_t f1[129 * 128];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int d = wsrc[y * w + x] - pre[(ptrdiff_t)y * pre_stride + x] * mask[y * w + x];
            sad += (uint32_t)rp2(abs(d), 12);
        }
    for (int y = 0; y < h + 1; y++)
        for (int x = 0; x < w; x++)
            f1[y * w + x] = (uint16_t)rp2(pre[(ptrdiff_t)y * pre_stride + x] * (128 - 16 * xoffset) + pre[(ptrdiff_t)y * pre_stride + x + 1] * (16 * xoffset), 7);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int p = (uint8_t)rp2(f1[y * w + x] * (128 - 16 * yoffset) + f1[(y + 1) * w + x] * (16 * yoffset), 7);
            const int v = wsrc[y * w + x] - p * mask[y * w + x];
            const int d = v < 0 ? -rp2(-v, 12) : rp2(v, 12);
            sum += d; sse += (uint32_t)(d * d);
        }
    out[0] = sad; out[1] = sse; out[2] = sse - (uint32_t)(((int64_t)sum * sum) / (w * h));
}
typedef struct { int32_t pre_x, pre_y; uint8_t w, h, xoffset, yoffset; int32_t wm_off; } OrcObmcBlk;
void orc_obmc_batch(const uint8_t *pre, int pre_stride, const int32_t *wsrc, const int32_t *mask, const void *blks_, int n, uint32_t *out) {
    const OrcObmcBlk *b = (const OrcObmcBlk *)blks_;
    for (int i = 0; i < n; i++)
        orc_obmc_block(pre + (ptrdiff_t)b[i].pre_y * pre_stride + b[i].pre_x, pre_stride, wsrc + b[i].wm_off, mask + b[i].wm_off, b[i].w, b[i].h, b[i].xoffset,
                       b[i].yoffset, out + 3 * i);
}

/* ---------------------------------------------------------------------------------------------------------------------------------
 * Pixel-domain mask blends: svt_aom_[highbd_]blend_a64_mask_c (Common/Codec/EbBlend_a64_mask.c:214-330; 2-D mask, the four
 * sub-sampling combinations), svt_aom_[highbd_]blend_a64_vmask (:332-382; one mask value per row) and _hmask (:384-434; per column) — what OBMC
 * (av1_build_obmc_inter_prediction), inter-intra and pixel-domain masked compound use.  dst may alias src0. */
typedef struct {
    int32_t src0_x, src0_y, src1_x, src1_y, dst_x, dst_y;
    uint8_t w, h, mode, subw, subh, reserved[3];      /* mode 0: 2-D mask (mask_stride, subw, subh); 1: hmask; 2: vmask */
    int32_t mask_off, mask_stride;
} OrcBlendBlk;
void orc_blend_a64_batch(int pix_bytes, const void *src0, int src0_stride, const void *src1, int src1_stride, void *dst, int dst_stride, const uint8_t *masks,
                         const void *blks_, int n) {
    const OrcBlendBlk *blks = (const OrcBlendBlk *)blks_;
    for (int i = 0; i < n; i++) {
        const OrcBlendBlk *b = &blks[i];
        const uint8_t *mk = masks + b->mask_off;
        const int ms = b->mask_stride;
        for (int y = 0; y < b->h; y++)
            for (int x = 0; x < b->w; x++) {
                int m;
                if (b->mode == 1) m = mk[x];
                else if (b->mode == 2) m = mk[y];
                else if (!b->subw && !b->subh) m = mk[y * ms + x];
                else if (b->subw && b->subh) m = rp2(mk[(2 * y) * ms + 2 * x] + mk[(2 * y + 1) * ms + 2 * x] + mk[(2 * y) * ms + 2 * x + 1] + mk[(2 * y + 1) * ms + 2 * x + 1], 2);
                else if (b->subw) m = rp2(mk[y * ms + 2 * x] + mk[y * ms + 2 * x + 1], 1);
                else m = rp2(mk[(2 * y) * ms + x] + mk[(2 * y + 1) * ms + x], 1);
                const int v0 = rdp(src0, pix_bytes, (ptrdiff_t)(b->src0_y + y) * src0_stride + b->src0_x + x);
                const int v1 = rdp(src1, pix_bytes, (ptrdiff_t)(b->src1_y + y) * src1_stride + b->src1_x + x);
                wrp(dst, pix_bytes, (ptrdiff_t)(b->dst_y + y) * dst_stride + b->dst_x + x, rp2(m * v0 + (64 - m) * v1, 6));
            }
    }
}
