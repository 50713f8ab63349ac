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
