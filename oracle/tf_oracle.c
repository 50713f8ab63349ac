/*
 * tf_oracle.c — CPU restatement of SVT-AV1's alt-ref temporal filter, plane-wise strategy
 * (SURVEY 8(f) rank 3).  TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 *
 * Pinned by tests/test_oracle_vs_ref.py against svt_av1_apply_temporal_filter_planewise_c /
 * svt_av1_apply_temporal_filter_planewise_hbd_c / estimate_noise / estimate_noise_highbd of
 * oracle/_ref/libsvtav1_ref.so.  Citations are file:line under /root/reference/Source/Lib.
 *
 * The filter weight goes through libm's expf exactly like the reference does
 * (Encoder/Codec/EbTemporalFiltering.c:740); tools/expf_pin.c shows that on the whole domain
 * the weight can see ([-7, 0], every float) glibc 2.35's expf equals the table + cubic
 * algorithm the HIP kernel evaluates in double precision.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "svt_oracle.h"

#define TF_WINDOW 5            /* TF_PLANEWISE_FILTER_WINDOW_LENGTH, Encoder/Codec/EbTemporalFiltering.h:37 */
#define TF_BALANCE 5           /* TF_WINDOW_BLOCK_BALANCE_WEIGHT, :49 */
#define TF_SCALE 1000          /* TF_WEIGHT_SCALE / TF_PLANEWISE_FILTER_WEIGHT_SCALE, :40,:45 */
#define TF_DIST_THRESHOLD 0.1  /* TF_SEARCH_DISTANCE_THRESHOLD, :74 */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline unsigned px(const void *p, int pix_bytes, size_t i) {
    return pix_bytes == 1 ? ((const uint8_t *)p)[i] : ((const uint16_t *)p)[i];
}

/* EbTemporalFiltering.c:736-741 (8-bit) / :926-931 (hbd): weight of one pixel from its normalised error */
static int tf_weight(double window_error, double block_error, double d_factor, double n_decay) {
    const double combined_error = (TF_BALANCE * window_error + block_error) / (TF_BALANCE + 1);
    double scaled_diff = combined_error * d_factor / (2 * n_decay * n_decay) / 1 / 1;
    if (!(scaled_diff < 7)) scaled_diff = 7;   /* AOMMIN(x, 7) */
    return (int)(expf((float)(-scaled_diff)) * TF_SCALE);
}

/* EbTemporalFiltering.c:643-813 (svt_av1_apply_temporal_filter_planewise_c) and :829-1003 (_hbd_c): one 32x32
 * (block_width x block_height) block of one reference frame.  `blk` = the 64x64 block's MeContext TF fields,
 * (block_row, block_col) = tf_block_row / tf_block_col.  accum / count are indexed with the predictor strides. */
void orc_tf_planewise(const OrcTfBlk64 *blk, int block_row, int block_col, int tf_chroma, int min_frame_size, int pix_bytes, int bd,
                      const void *y_src, int y_src_stride, const void *y_pre, int y_pre_stride, const void *u_src, const void *v_src,
                      int uv_src_stride, const void *u_pre, const void *v_pre, int uv_pre_stride, unsigned block_width,
                      unsigned block_height, int ss_x, int ss_y, const double *noise_levels, int decay_control, uint32_t *y_accum,
                      uint16_t *y_count, uint32_t *u_accum, uint16_t *u_count, uint32_t *v_accum, uint16_t *v_count) {
    const unsigned uvw = block_width >> ss_x, uvh = block_height >> ss_y;
    const int hbd = pix_bytes == 2;
    const int sh = hbd ? (bd - 8) * 2 : 0;                      /* :880 sum_square_diff >>= (bd - 8) * 2 */
    uint32_t *yd = calloc(4096, 4), *ud = calloc(4096, 4), *vd = calloc(4096, 4);
    /* :525-555 calculate_squared_errors(_highbd); the 8-bit buffer is uint16_t, 255^2 fits */
    for (unsigned i = 0; i < block_height; i++)
        for (unsigned j = 0; j < block_width; j++) {
            const int d = (int)px(y_src, pix_bytes, (size_t)i * y_src_stride + j) - (int)px(y_pre, pix_bytes, (size_t)i * y_pre_stride + j);
            yd[i * block_width + j] = (uint32_t)(d * d);
        }
    if (tf_chroma)
        for (unsigned i = 0; i < uvh; i++)
            for (unsigned j = 0; j < uvw; j++) {
                const int du = (int)px(u_src, pix_bytes, (size_t)i * uv_src_stride + j) - (int)px(u_pre, pix_bytes, (size_t)i * uv_pre_stride + j);
                const int dv = (int)px(v_src, pix_bytes, (size_t)i * uv_src_stride + j) - (int)px(v_pre, pix_bytes, (size_t)i * uv_pre_stride + j);
                ud[i * uvw + j] = (uint32_t)(du * du); vd[i * uvw + j] = (uint32_t)(dv * dv);
            }
    const int half = TF_WINDOW >> 1;
    const int idx32 = block_col + block_row * 2;
    for (unsigned i = 0; i < block_height; i++)
        for (unsigned j = 0; j < block_width; j++) {
            const int pixel_value = (int)px(y_pre, pix_bytes, (size_t)i * y_pre_stride + j);
            int num = 0;
            uint64_t sum = 0;
            for (int dy = -half; dy <= half; dy++)
                for (int dx = -half; dx <= half; dx++) {
                    sum += yd[clampi((int)i + dy, 0, (int)block_height - 1) * (int)block_width + clampi((int)j + dx, 0, (int)block_width - 1)];
                    num++;
                }
            sum >>= sh;
            double window_error = (double)sum / num;
            const int sub = (i >= block_height / 2) * 2 + (j >= block_width / 2);
            double block_error;
            int16_t mv_col, mv_row;
            if (blk->split[idx32]) {        /* :710-717, :721-730 ; hbd: error >> 4 (:890-895) */
                block_error = (double)(hbd ? blk->err16[idx32 * 4 + sub] >> 4 : blk->err16[idx32 * 4 + sub]) / 256;
                mv_col = blk->mv16_x[idx32 * 4 + sub]; mv_row = blk->mv16_y[idx32 * 4 + sub];
            } else {
                block_error = (double)(hbd ? blk->err32[idx32] >> 4 : blk->err32[idx32]) / 1024;
                mv_col = blk->mv32_x[idx32]; mv_row = blk->mv32_y[idx32];
            }
            const float distance = sqrtf(powf(mv_row, 2) + powf(mv_col, 2));                         /* :731 */
            const double thr = min_frame_size * TF_DIST_THRESHOLD;
            const double distance_threshold = (double)(thr > 1 ? thr : 1);                            /* :732-733 */
            const double dd = distance / distance_threshold;
            const double d_factor = dd > 1 ? dd : 1;                                                  /* :734 */
            double n_decay = (double)decay_control * (0.7 + log1p(noise_levels[0]));                  /* :706 */
            int w = tf_weight(window_error, block_error, d_factor, n_decay);
            size_t k = (size_t)i * y_pre_stride + j;
            y_count[k] += w; y_accum[k] += w * pixel_value;
            if (tf_chroma && !(i & ss_y) && !(j & ss_x)) {                                           /* :746-812 */
                const int uv_r = i >> ss_y, uv_c = j >> ss_x;
                const int upix = (int)px(u_pre, pix_bytes, (size_t)uv_r * uv_pre_stride + uv_c);
                const int vpix = (int)px(v_pre, pix_bytes, (size_t)uv_r * uv_pre_stride + uv_c);
                num = 0;
                uint64_t ys = 0;
                for (int dy = 0; dy < (1 << ss_y); dy++)
                    for (int dx = 0; dx < (1 << ss_x); dx++) { ys += yd[((int)i + dy) * (int)block_width + (int)j + dx]; num++; }
                uint64_t us = ys, vs = ys;
                for (int dy = -half; dy <= half; dy++)
                    for (int dx = -half; dx <= half; dx++) {
                        const int o = clampi(uv_r + dy, 0, (int)uvh - 1) * (int)uvw + clampi(uv_c + dx, 0, (int)uvw - 1);
                        us += ud[o]; vs += vd[o]; num++;
                    }
                us >>= sh; vs >>= sh;
                const size_t m = (size_t)uv_r * uv_pre_stride + uv_c;
                n_decay = (double)decay_control * (0.7 + log1p(noise_levels[1]));
                w = tf_weight((double)us / num, block_error, d_factor, n_decay);
                u_count[m] += w; u_accum[m] += w * upix;
                n_decay = (double)decay_control * (0.7 + log1p(noise_levels[2]));
                w = tf_weight((double)vs / num, block_error, d_factor, n_decay);
                v_count[m] += w; v_accum[m] += w * vpix;
            }
        }
    free(yd); free(ud); free(vd);
}

/* Frame driver = the pixel side of produce_temporally_filtered_pic (EbTemporalFiltering.c:2136-2412) once motion search has produced
 * a predictor picture and the per-64x64 TF fields for every reference frame:  for every 64x64 block, reset accum / count (:2142-2143),
 * add the central picture with weight 1000 (apply_filtering_central :557-590), add every other frame 32x32 by 32x32 (:2358-2381), then
 * normalise (get_final_filtered_pixels :1943-1990; OD_DIVU == plain division, pinned in tests) and sum the squared change.
 * refs[f].blocks == NULL marks the central picture.  w / h are the multiple-of-64 extents the reference walks (:2076-2079). */
void orc_tf_filter_frame(int pix_bytes, int bd, const void *const src[3], const int src_stride[3], void *const dst[3], const int dst_stride[3],
                         int w, int h, int ss_x, int ss_y, int tf_chroma, const OrcTfRef *refs, int n_refs, const double *noise_levels,
                         int decay_control, int min_frame_size, uint64_t sse[2]) {
    const int bc = w / 64, br = h / 64;
    const int cw = 64 >> ss_x, ch = 64 >> ss_y;
    uint32_t *accum = malloc(3 * 4096 * 4);
    uint16_t *count = malloc(3 * 4096 * 2);
    sse[0] = sse[1] = 0;
    for (int by = 0; by < br; by++)
        for (int bx = 0; bx < bc; bx++) {
            memset(accum, 0, 3 * 4096 * 4); memset(count, 0, 3 * 4096 * 2);
            uint32_t *acc[3] = {accum, accum + 4096, accum + 8192};
            uint16_t *cnt[3] = {count, count + 4096, count + 8192};
            const int pstride[3] = {64, cw, cw};                                                      /* stride_pred :2083 */
            for (int f = 0; f < n_refs; f++) {
                if (!refs[f].blocks) {
                    for (int p = 0; p < (tf_chroma ? 3 : 1); p++) {
                        const int pw = p ? cw : 64, ph = p ? ch : 64;
                        const size_t o = (size_t)by * ph * src_stride[p] + (size_t)bx * pw;
                        for (int i = 0, k = 0; i < ph; i++)
                            for (int j = 0; j < pw; j++, k++) {
                                acc[p][k] += TF_SCALE * px(src[p], pix_bytes, o + (size_t)i * src_stride[p] + j);
                                cnt[p][k] += TF_SCALE;
                            }
                    }
                    continue;
                }
                const OrcTfBlk64 *blk = refs[f].blocks + (size_t)by * bc + bx;
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 2; c++) {
                        const void *s[3], *q[3];
                        for (int p = 0; p < 3; p++) {
                            const int pw = p ? cw : 64, ph = p ? ch : 64;
                            s[p] = (const char *)src[p] + ((size_t)(by * ph + r * (ph >> 1)) * src_stride[p] + (size_t)bx * pw + c * (pw >> 1)) * pix_bytes;
                            q[p] = (const char *)refs[f].pred[p] + ((size_t)(by * ph + r * (ph >> 1)) * refs[f].pred_stride[p] + (size_t)bx * pw + c * (pw >> 1)) * pix_bytes;
                        }
                        /* the reference's predictor is a 64-wide block buffer; ours is a picture, so accumulate into a 32x32 scratch with the
                         * picture's predictor stride semantics kept (accum index = row * stride + col) by using a private stride-64 view */
                        uint32_t ya[32 * 64] = {0}, ua[32 * 64] = {0}, va[32 * 64] = {0};
                        uint16_t yc[32 * 64] = {0}, uc[32 * 64] = {0}, vc[32 * 64] = {0};
                        /* run the block function on copies of the predictor with stride 64 / cw (its accum index uses the predictor stride) */
                        uint16_t yp16[32 * 64], up16[32 * 64], vp16[32 * 64];
                        uint8_t yp8[32 * 64], up8[32 * 64], vp8[32 * 64];
                        for (int p = 0; p < 3; p++) {
                            const int pw = (p ? cw : 64) >> 1, ph = (p ? ch : 64) >> 1;
                            for (int i = 0; i < ph; i++)
                                for (int j = 0; j < pw; j++) {
                                    const unsigned v = px(q[p], pix_bytes, (size_t)i * refs[f].pred_stride[p] + j);
                                    if (pix_bytes == 1) (p == 0 ? yp8 : p == 1 ? up8 : vp8)[i * pstride[p] + j] = (uint8_t)v;
                                    else (p == 0 ? yp16 : p == 1 ? up16 : vp16)[i * pstride[p] + j] = (uint16_t)v;
                                }
                        }
                        orc_tf_planewise(blk, r, c, tf_chroma, min_frame_size, pix_bytes, bd, s[0], src_stride[0],
                                         pix_bytes == 1 ? (void *)yp8 : (void *)yp16, pstride[0], s[1], s[2], src_stride[1],
                                         pix_bytes == 1 ? (void *)up8 : (void *)up16, pix_bytes == 1 ? (void *)vp8 : (void *)vp16, pstride[1], 32, 32,
                                         ss_x, ss_y, noise_levels, decay_control, ya, yc, ua, uc, va, vc);
                        for (int p = 0; p < (tf_chroma ? 3 : 1); p++) {
                            const int pw = (p ? cw : 64) >> 1, ph = (p ? ch : 64) >> 1;
                            const uint32_t *a = p == 0 ? ya : p == 1 ? ua : va;
                            const uint16_t *n = p == 0 ? yc : p == 1 ? uc : vc;
                            for (int i = 0; i < ph; i++)
                                for (int j = 0; j < pw; j++) {
                                    const int k = (r * ph + i) * pstride[p] + c * pw + j;
                                    acc[p][k] += a[i * pstride[p] + j];
                                    cnt[p][k] = (uint16_t)(cnt[p][k] + n[i * pstride[p] + j]);
                                }
                        }
                    }
            }
            for (int p = 0; p < (tf_chroma ? 3 : 1); p++) {
                const int pw = p ? cw : 64, ph = p ? ch : 64;
                for (int i = 0, k = 0; i < ph; i++)
                    for (int j = 0; j < pw; j++, k++) {
                        const size_t so = (size_t)(by * ph + i) * src_stride[p] + (size_t)bx * pw + j;
                        const size_t dofs = (size_t)(by * ph + i) * dst_stride[p] + (size_t)bx * pw + j;
                        const int32_t v = (int32_t)((acc[p][k] + (cnt[p][k] >> 1)) / cnt[p][k]);
                        const int32_t d = (int32_t)px(src[p], pix_bytes, so) - v;
                        sse[p ? 1 : 0] += (uint64_t)(d * d);
                        if (pix_bytes == 1) ((uint8_t *)dst[p])[dofs] = (uint8_t)v; else ((uint16_t *)dst[p])[dofs] = (uint16_t)v;
                    }
            }
        }
    free(accum); free(count);
}

/* EbTemporalFiltering.c:2414-2448 (estimate_noise) and :2451-2486 (estimate_noise_highbd): returns the Laplacian sum and the number of
 * smooth pixels; sigma = sum / (6 * num) * SQRT_PI_BY_2, or -1 when num < SMOOTH_THRESHOLD (16). */
double orc_tf_estimate_noise(const void *src, int pix_bytes, int bd, int width, int height, int stride, int64_t out[2]) {
    int64_t sum = 0, num = 0;
    const int sh = pix_bytes == 2 ? bd - 8 : 0;
    for (int i = 1; i < height - 1; i++)
        for (int j = 1; j < width - 1; j++) {
#define P(dy, dx) ((int)px(src, pix_bytes, (size_t)(i + (dy)) * stride + j + (dx)))
            const int gx = (P(-1, -1) - P(-1, 1)) + (P(1, -1) - P(1, 1)) + 2 * (P(0, -1) - P(0, 1));
            const int gy = (P(-1, -1) - P(1, -1)) + (P(-1, 1) - P(1, 1)) + 2 * (P(-1, 0) - P(1, 0));
            int ga = abs(gx) + abs(gy);
            if (sh) ga = (ga + ((1 << sh) >> 1)) >> sh;                                               /* ROUND_POWER_OF_TWO, :2463 */
            if (ga < 50) {                                                                            /* EDGE_THRESHOLD */
                const int v = 4 * P(0, 0) - 2 * (P(0, -1) + P(0, 1) + P(-1, 0) + P(1, 0)) + (P(-1, -1) + P(-1, 1) + P(1, -1) + P(1, 1));
                int a = abs(v);
                if (sh) a = (a + ((1 << sh) >> 1)) >> sh;                                             /* :2472 */
                sum += a; num++;
            }
#undef P
        }
    if (out) { out[0] = sum; out[1] = num; }
    if (num < 16) return -1.0;
    return (double)sum / (6 * num) * 1.25331413732;
}

/* ---------------------------------------------------------------------------------------------------------------------------------
 * The sub-pel stage of the temporal filter: tf_32x32_sub_pel_search (Encoder/Codec/EbTemporalFiltering.c:1469), tf_16x16_sub_pel_search
 * (:1133), derive_tf_32x32_block_split_flag (:284), tf_inter_prediction (:1768) for a list of (64x64 block, frame) pairs of one reference
 * picture.  A candidate = av1_inter_prediction's single-reference unscaled path (Encoder/Codec/EbEncInterPrediction.c:4040 ->
 * enc_make_inter_predictor :3663 -> compute_subpel_params :3593 with clamp_mv_to_umv_border_sb :24) + svt_aom_variance{W}x{H}_c /
 * variance_highbd_c (Encoder/C_DEFAULT/EbComputeVariance_C.c:54 / :34).  Pinned by the end-to-end encodes (tests/test_encode_e2e.py, hook
 * "tf_subpel" on the CPU test double against the unpatched reference encoder): the reference's functions are static and need a whole
 * PictureParentControlSet. */
typedef struct {   /* SvtHipTfSubpelBlk of include/svt_hip.h */
    int32_t  x, y, dst_x, dst_y, blk_index;
    uint32_t mv32[4], mv16[16];
} OrcTfSubpelBlk;
typedef struct {
    int pix_bytes, bd, mi_cols, mi_rows, tf_hp, tf_chroma;
    const void *const *src; const int *src_stride; const void *const *ref; const int *ref_stride; void *const *pred; const int *pred_stride;
} TfSp;

/* clamp_mv_to_umv_border_sb + the position arithmetic of compute_subpel_params' unscaled branch; mv in 1/8 luma pel, (px, py) the block's luma
 * position, bs its luma size, (bw, bh) its size in the plane with subsampling ss, (pre_x, pre_y) its position there */
static void tf_sp_params(const TfSp *t, int mvx, int mvy, int px, int py, int bs, int bw, int bh, int ss, int pre_x, int pre_y, int *pos_x, int *pos_y, int *sx, int *sy) {
    const int mirow = py >> 2, micol = px >> 2, mi = bs >> 2;
    const int to_top = -((mirow * 4) * 8), to_bottom = ((t->mi_rows - mi - mirow) * 4) * 8, to_left = -((micol * 4) * 8), to_right = ((t->mi_cols - mi - micol) * 4) * 8;
    const int spel_left = (4 + bw) << 4, spel_right = spel_left - 16, spel_top = (4 + bh) << 4, spel_bottom = spel_top - 16;
    int16_t row = (int16_t)(mvy * (1 << (1 - ss))), col = (int16_t)(mvx * (1 << (1 - ss)));
    col = (int16_t)clampi(col, to_left * (1 << (1 - ss)) - spel_left, to_right * (1 << (1 - ss)) + spel_right);
    row = (int16_t)clampi(row, to_top * (1 << (1 - ss)) - spel_top, to_bottom * (1 << (1 - ss)) + spel_bottom);
    *sx = col & 15; *sy = row & 15;
    *pos_x = ((pre_x << 4) + col) >> 4; *pos_y = ((pre_y << 4) + row) >> 4;
}
static void tf_sp_predict(const TfSp *t, int plane, int pos_x, int pos_y, int sx, int sy, int bank, int bw, int bh, void *dst, int dst_stride) {
    const uint8_t *r = (const uint8_t *)t->ref[plane] + ((ptrdiff_t)pos_y * t->ref_stride[plane] + pos_x) * t->pix_bytes;
    orc_convolve_sr(r, t->ref_stride[plane], dst, dst_stride, t->pix_bytes, bw, bh, bank, bank, sx, sy, t->bd);
}
static uint64_t tf_sp_distortion(const TfSp *t, const void *pred, int bs, int lx, int ly) {
    const int n = bs * bs;
    if (t->pix_bytes == 1) {
        const uint8_t *s = (const uint8_t *)t->src[0] + (ptrdiff_t)ly * t->src_stride[0] + lx, *p = (const uint8_t *)pred;
        int sum = 0; uint32_t sse = 0;
        for (int y = 0; y < bs; y++) for (int x = 0; x < bs; x++) { const int d = p[y * bs + x] - s[(ptrdiff_t)y * t->src_stride[0] + x]; sum += d; sse += (uint32_t)(d * d); }
        return sse - (uint32_t)(((int64_t)sum * sum) / n);
    }
    const uint16_t *s = (const uint16_t *)t->src[0] + (ptrdiff_t)ly * t->src_stride[0] + lx, *p = (const uint16_t *)pred;
    int sad = 0; uint32_t sse = 0;
    for (int y = 0; y < bs; y++) for (int x = 0; x < bs; x++) { const int d = p[y * bs + x] - s[(ptrdiff_t)y * t->src_stride[0] + x]; sad += d; sse += (uint32_t)(d * d); }
    return (uint32_t)(sse - (uint32_t)((int)((unsigned)sad * (unsigned)sad) / n));   /* `sad * sad` in int: wraps like the compiled C */
}
static void tf_sp_search(const TfSp *t, int bs, int px, int py, int lx, int ly, uint32_t word, int16_t *out_x, int16_t *out_y, uint64_t *out_err) {
    uint16_t pred[32 * 32];
    int16_t  mv_x = (int16_t)((int16_t)(word & 0xffff) * 2), mv_y = (int16_t)((int16_t)(word >> 16) * 2), best_x = mv_x, best_y = mv_y;   /* the reference shifts the (possibly negative) vector left by one */
    uint64_t best = 0x7fffffff;   /* INT_MAX */
    for (int round = 0; round < (t->tf_hp ? 3 : 2); round++) {
        const int step = 4 >> round;
        for (int i = -step; i <= step; i += step)
            for (int j = -step; j <= step; j += step) {
                const int16_t cx = (int16_t)(mv_x + i), cy = (int16_t)(mv_y + j);
                int pos_x, pos_y, sx, sy;
                tf_sp_params(t, cx, cy, px, py, bs, bs, bs, 0, px, py, &pos_x, &pos_y, &sx, &sy);
                tf_sp_predict(t, 0, pos_x, pos_y, sx, sy, 0, bs, bs, pred, bs);   /* EIGHTTAP_REGULAR */
                const uint64_t d = tf_sp_distortion(t, pred, bs, lx, ly);
                if (d < best) { best = d; best_x = cx; best_y = cy; }
            }
        mv_x = best_x; mv_y = best_y;
    }
    *out_x = best_x; *out_y = best_y; *out_err = best;
}
static void tf_sp_final(const TfSp *t, int bs, int px, int py, int lx, int ly, int mvx, int mvy) {
    int pos_x, pos_y, sx, sy;
    tf_sp_params(t, mvx, mvy, px, py, bs, bs, bs, 0, px, py, &pos_x, &pos_y, &sx, &sy);
    tf_sp_predict(t, 0, pos_x, pos_y, sx, sy, 2, bs, bs, (uint8_t *)t->pred[0] + ((ptrdiff_t)ly * t->pred_stride[0] + lx) * t->pix_bytes, t->pred_stride[0]);   /* MULTITAP_SHARP */
    if (!t->tf_chroma) return;
    const int cb = bs >> 1, cpx = ((px >> 3) << 3) / 2, cpy = ((py >> 3) << 3) / 2, clx = ((lx >> 3) << 3) / 2, cly = ((ly >> 3) << 3) / 2;
    tf_sp_params(t, mvx, mvy, px, py, bs, cb, cb, 1, cpx, cpy, &pos_x, &pos_y, &sx, &sy);
    for (int p = 1; p < 3; p++)
        tf_sp_predict(t, p, pos_x, pos_y, sx, sy, 2, cb, cb, (uint8_t *)t->pred[p] + ((ptrdiff_t)cly * t->pred_stride[p] + clx) * t->pix_bytes, t->pred_stride[p]);
}
void orc_tf_subpel_frame(int pix_bytes, int bd, const void *const src[3], const int src_stride[3], const void *const ref[3], const int ref_stride[3],
                         void *const pred[3], const int pred_stride[3], int mi_cols, int mi_rows, uint64_t th16, int tf_hp, int tf_chroma, const void *jobs_,
                         int n_jobs, OrcTfBlk64 *blocks) {
    const OrcTfSubpelBlk *jobs = (const OrcTfSubpelBlk *)jobs_;
    const TfSp t = {pix_bytes, bd, mi_cols, mi_rows, tf_hp, tf_chroma, src, src_stride, ref, ref_stride, pred, pred_stride};
    for (int n = 0; n < n_jobs; n++) {
        const OrcTfSubpelBlk *J = &jobs[n];
        OrcTfBlk64           *B = &blocks[J->blk_index];
        int                   search_do[4];
        for (int q = 0; q < 4; q++)   /* tf_32x32_sub_pel_search */
            tf_sp_search(&t, 32, J->x + 32 * (q & 1), J->y + 32 * (q >> 1), J->dst_x + 32 * (q & 1), J->dst_y + 32 * (q >> 1), J->mv32[q], &B->mv32_x[q], &B->mv32_y[q],
                         &B->err32[q]);
        for (int q = 0; q < 4; q++) {   /* tf_16x16_sub_pel_search; 16x16 k of quadrant q sits at raster (2 (q >> 1) + (k >> 1), 2 (q & 1) + (k & 1)) */
            search_do[q] = B->err32[q] < th16 ? 0 : 1;
            for (int k = 0; k < 4; k++) {
                const int ox = 32 * (q & 1) + 16 * (k & 1), oy = 32 * (q >> 1) + 16 * (k >> 1);
                if (search_do[q]) tf_sp_search(&t, 16, J->x + ox, J->y + oy, J->dst_x + ox, J->dst_y + oy, J->mv16[4 * q + k], &B->mv16_x[4 * q + k], &B->mv16_y[4 * q + k], &B->err16[4 * q + k]);
                else { B->mv16_x[4 * q + k] = B->mv16_y[4 * q + k] = 0; B->err16[4 * q + k] = 0; }   /* never read: split = 0 */
            }
        }
        for (int q = 0; q < 4; q++) {   /* derive_tf_32x32_block_split_flag */
            if (!search_do[q]) { B->split[q] = 0; continue; }
            const int block_error = (int)B->err32[q];
            int       mn = 0x7fffffff, mx = (int)0x80000000, sum = 0;
            for (int k = 0; k < 4; k++) { const int e = (int)B->err16[4 * q + k]; sum = (int)((unsigned)sum + (unsigned)e); mn = e < mn ? e : mn; mx = e > mx ? e : mx; }
            const int b15 = (int)((unsigned)block_error * 15u), b14 = (int)((unsigned)block_error * 14u), s16 = (int)((unsigned)sum * 16u);
            B->split[q] = ((b15 < s16 && mx - mn < 12000) || (b14 < s16 && mx - mn < 6000)) ? 0 : 1;
        }
        for (int q = 0; q < 4; q++) {   /* tf_inter_prediction */
            if (B->split[q])
                for (int k = 0; k < 4; k++) {
                    const int ox = 32 * (q & 1) + 16 * (k & 1), oy = 32 * (q >> 1) + 16 * (k >> 1);
                    tf_sp_final(&t, 16, J->x + ox, J->y + oy, J->dst_x + ox, J->dst_y + oy, B->mv16_x[4 * q + k], B->mv16_y[4 * q + k]);
                }
            else tf_sp_final(&t, 32, J->x + 32 * (q & 1), J->y + 32 * (q >> 1), J->dst_x + 32 * (q & 1), J->dst_y + 32 * (q >> 1), B->mv32_x[q], B->mv32_y[q]);
        }
    }
}
