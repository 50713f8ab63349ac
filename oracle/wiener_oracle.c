/*
 * wiener_oracle.c — CPU restatement of the Wiener restoration path.  TEST INFRASTRUCTURE ONLY (see
 * svt_oracle.h): pinned to the reference's own functions by tests/test_oracle_vs_ref.py.
 *
 *   svt_av1_compute_stats_c / _highbd_c          Encoder/Codec/EbRestorationPick.c:704-790  (find_average: EbRestorationPick.h:24-42)
 *   svt_av1_[highbd_]wiener_convolve_add_src_c   Common/Codec/convolve.c:60-241
 *   wiener_filter_stripe[_highbd]                Common/Codec/EbRestoration.c:1040-1085 / :1110-1132
 */
#include "svt_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int rdp(const void *b, int pb, ptrdiff_t i) { return pb == 1 ? ((const uint8_t *)b)[i] : ((const uint16_t *)b)[i]; }

/* M[win2], H[win2 * win2] of one restoration unit; feature index = (dx + half) * win + (dy + half) (column offset outer, as the
 * reference's loops run); 10-bit sums are divided by 4 (12-bit: 16) at the end, towards zero like C's int64 division. */
void orc_wiener_compute_stats(int win, const void *dgd, const void *src, int pix_bytes, int bd, int h_start, int h_end, int v_start, int v_end,
                              int dgd_stride, int src_stride, int64_t *M, int64_t *H) {
    const int win2 = win * win, half = win >> 1;
    uint64_t sum = 0;
    for (int i = v_start; i < v_end; i++)
        for (int j = h_start; j < h_end; j++) sum += (uint64_t)rdp(dgd, pix_bytes, (ptrdiff_t)i * dgd_stride + j);
    const int avg = (int)(sum / (uint64_t)((v_end - v_start) * (h_end - h_start)));   /* find_average: truncating */
    memset(M, 0, sizeof(int64_t) * win2);
    memset(H, 0, sizeof(int64_t) * win2 * win2);
    int32_t y[49];
    for (int i = v_start; i < v_end; i++)
        for (int j = h_start; j < h_end; j++) {
            const int32_t x = rdp(src, pix_bytes, (ptrdiff_t)i * src_stride + j) - avg;
            int idx = 0;
            for (int k = -half; k <= half; k++)
                for (int l = -half; l <= half; l++) y[idx++] = rdp(dgd, pix_bytes, (ptrdiff_t)(i + l) * dgd_stride + (j + k)) - avg;
            for (int k = 0; k < win2; k++) {
                M[k] += (int64_t)y[k] * x;
                for (int l = k; l < win2; l++) H[k * win2 + l] += (int64_t)y[k] * y[l];
            }
        }
    const int64_t div = bd == 12 ? 16 : (bd == 10 ? 4 : 1);
    for (int k = 0; k < win2; k++) {
        if (pix_bytes == 2) { M[k] /= div; H[k * win2 + k] /= div; }
        for (int l = k + 1; l < win2; l++) {
            if (pix_bytes == 2) H[k * win2 + l] /= div;
            H[l * win2 + k] = H[k * win2 + l];
        }
    }
}

/* 7-tap separable Wiener filter with the "add source" identity tap, w x h block, 3 rows / columns of context around it.
 * filter_x / filter_y: 8 int16 (tap 7 = 0), round_0 = 3, round_1 = 11 for 8 / 10 bit (get_conv_params_wiener, convolve.h:78-95). */
void orc_wiener_convolve_add_src(const void *src, int src_stride, void *dst, int dst_stride, int pix_bytes, const int16_t *filter_x,
                                 const int16_t *filter_y, int w, int h, int bd) {
    const int r0 = 3, r1 = 11, ih = h + 6;   /* tap 7 is 0 (the reference still reads row h + 3 / column w + 3 for it, its pictures have wide
                                               * borders; here nothing outside the 3-sample context is touched) */
    const int lim = (1 << (bd + 1 + 7 - r0)) - 1;   /* WIENER_CLAMP_LIMIT - 1 */
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * ih);
    for (int y = 0; y < ih; y++)           /* rows -3 .. h+2 */
        for (int x = 0; x < w; x++) {
            int32_t sum = 0;
            for (int k = 0; k < 7; k++) sum += rdp(src, pix_bytes, (ptrdiff_t)(y - 3) * src_stride + x - 3 + k) * filter_x[k];
            sum += (rdp(src, pix_bytes, (ptrdiff_t)(y - 3) * src_stride + x) << 7) + (1 << (bd + 7 - 1));
            int32_t v = (sum + (1 << (r0 - 1))) >> r0;
            tmp[y * w + x] = (uint16_t)(v < 0 ? 0 : (v > lim ? lim : v));
        }
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++) {
            int32_t sum = 0;
            for (int k = 0; k < 7; k++) sum += tmp[(y + k) * w + x] * filter_y[k];
            sum += ((int32_t)tmp[(y + 3) * w + x] << 7) - (1 << (bd + r1 - 1));
            int32_t v = (sum + (1 << (r1 - 1))) >> r1;
            const int mx = (1 << bd) - 1;
            v = v < 0 ? 0 : (v > mx ? mx : v);
            if (pix_bytes == 1) ((uint8_t *)dst)[(size_t)y * dst_stride + x] = (uint8_t)v;
            else ((uint16_t *)dst)[(size_t)y * dst_stride + x] = (uint16_t)v;
        }
    free(tmp);
}

/* per-unit statistics of a whole plane with the reference's unit geometry (search_wiener_seg, EbRestorationPick.c:1347):
 * M[unit][win2], H[unit][win2 * win2] */
void orc_wiener_stats_plane(int win, const void *dgd, int dgd_stride, const void *src, int src_stride, int pix_bytes, int bd, int pw, int ph, int ss_y,
                            int unit_size, int64_t *M, int64_t *H) {
    const int nu = orc_rest_units(pw, unit_size) * orc_rest_units(ph, unit_size), win2 = win * win;
    int32_t *lim = (int32_t *)malloc(sizeof(int32_t) * 4 * nu);
    orc_rest_unit_limits(pw, ph, ss_y, unit_size, lim);
    for (int u = 0; u < nu; u++)
        orc_wiener_compute_stats(win, dgd, src, pix_bytes, bd, lim[4 * u], lim[4 * u + 1], lim[4 * u + 2], lim[4 * u + 3], dgd_stride, src_stride,
                                 M + (size_t)u * win2, H + (size_t)u * win2 * win2);
    free(lim);
}
