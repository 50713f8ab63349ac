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

/* ---- initial Wiener filter of one restoration unit from its statistics: search_wiener_seg between svt_av1_compute_stats and the tap refinement
 * (Encoder/Codec/EbRestorationPick.c:1388-1407): wiener_decompose_sep_sym (:946-979: four rounds of update_a_sep_sym :841 / update_b_sep_sym :895, each a small
 * linear system solved by linsolve_wiener :800), finalize_sym_filter (:1022-1052), compute_score (:980-1020).  All of it is 64-bit INTEGER arithmetic with
 * truncating divisions, written here the way the device kernel organises it: the two tap vectors are symmetric, so the normal equations are folded onto the
 * first `half + 1` taps (fold(i) below = the reference's wrap_index, :793). */
#define WN_SCALE ((int64_t)1 << 16)   /* WIENER_TAP_SCALE_FACTOR, :42 */
#define WN_STEP 128                   /* WIENER_FILT_STEP, Common/Codec/EbRestoration.h:125 */
static int wn_fold(int i, int win) { return i > (win >> 1) ? win - 1 - i : i; }

/* Gaussian elimination with the reference's neighbour-swap pivoting and its truncations (:800-838); A is n x n with row stride `st`; 0 = singular */
static int wn_solve(int n, int64_t *A, int st, int64_t *b, int32_t *x) {
    for (int k = 0; k + 1 < n; k++) {
        for (int i = n - 1; i > k; i--) {
            const int64_t lo = A[(i - 1) * st + k], hi = A[i * st + k];
            if ((lo < 0 ? -lo : lo) >= (hi < 0 ? -hi : hi)) continue;
            for (int j = 0; j < n; j++) { const int64_t t = A[i * st + j]; A[i * st + j] = A[(i - 1) * st + j]; A[(i - 1) * st + j] = t; }
            const int64_t t = b[i]; b[i] = b[i - 1]; b[i - 1] = t;
        }
        for (int i = k + 1; i < n; i++) {
            const int64_t piv = A[k * st + k], c = A[i * st + k];
            if (piv == 0) return 0;
            for (int j = 0; j < n; j++) A[i * st + j] -= c / 256 * A[k * st + j] / piv * 256;
            b[i] -= c * b[k] / piv;
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        if (A[i * st + i] == 0) return 0;
        int64_t c = 0;
        for (int j = i + 1; j < n; j++) c += A[i * st + j] * x[j] / WN_SCALE;
        x[i] = (int32_t)(WN_SCALE * (b[i] - c) / A[i * st + i]);
    }
    return 1;
}

/* One half-round: `fixed` stays, `upd` is solved for.  which = 0: update_a_sep_sym (the vertical taps a from the horizontal b), 1: update_b_sep_sym.
 * H is read as the four-index array H[(i * win + k) * win2 + j * win + l] (hc[i * win + j][k * win2 + l], :962-968). */
static void wn_half_round(int win, const int64_t *M, const int64_t *H, int which, const int32_t *fixed, int32_t *upd) {
    const int win2 = win * win, h1 = (win >> 1) + 1;
    int64_t A[4] = {0, 0, 0, 0}, B[16];
    for (int i = 0; i < 16; i++) B[i] = 0;
    for (int i = 0; i < win; i++)
        for (int j = 0; j < win; j++) {
            if (which == 0) A[wn_fold(j, win)] += M[i * win + j] * fixed[i] / WN_SCALE;
            else A[wn_fold(i, win)] += M[i * win + j] * fixed[j] / WN_SCALE;
        }
    for (int i = 0; i < win; i++)
        for (int j = 0; j < win; j++)
            for (int k = 0; k < win; k++)
                for (int l = 0; l < win; l++) {
                    if (which == 0)   /* :856-867: hc[j * win + i][k * win2 + l] * b[i] / S * b[j] / S into B[fold(l)][fold(k)] */
                        B[wn_fold(l, win) * h1 + wn_fold(k, win)] += H[(j * win + k) * win2 + i * win + l] * fixed[i] / WN_SCALE * fixed[j] / WN_SCALE;
                    else              /* :909-921: hc[i * win + j][k * win2 + l] * a[k] / S * a[l] / S into B[fold(j)][fold(i)] */
                        B[wn_fold(j, win) * h1 + wn_fold(i, win)] += H[(i * win + k) * win2 + j * win + l] * fixed[k] / WN_SCALE * fixed[l] / WN_SCALE;
                }
    /* the taps sum to one: the centre tap is eliminated from the system (:868-882 / :923-937) */
    const int64_t a_c = A[h1 - 1], b_cc = B[(h1 - 1) * h1 + h1 - 1];
    for (int i = 0; i < h1 - 1; i++) A[i] -= a_c * 2 + B[i * h1 + h1 - 1] - 2 * b_cc;
    for (int i = 0; i < h1 - 1; i++)
        for (int j = 0; j < h1 - 1; j++) B[i * h1 + j] -= 2 * (B[i * h1 + h1 - 1] + B[(h1 - 1) * h1 + j] - 2 * b_cc);
    int32_t S[7];
    if (!wn_solve(h1 - 1, B, h1, A, S)) return;   /* singular: the vector keeps its value (:883, :938) */
    S[h1 - 1] = (int32_t)WN_SCALE;
    for (int i = h1; i < win; i++) { S[i] = S[win - 1 - i]; S[h1 - 1] -= 2 * S[i]; }
    for (int i = 0; i < win; i++) upd[i] = S[i];
}

/* finalize_sym_filter (:1022-1052): round to WIENER_FILT_STEP units, clamp to the coded ranges, centre tap implicit.  fi[8] is zeroed first (the reference's 3-tap
 * branch reads fi[1] before anything wrote it; the hook hands it a zeroed WienerInfo, integration/patch_reference.py svt_hip_wiener_unit_init). */
static void wn_finalize(int win, const int32_t *f, int16_t fi[8]) {
    for (int i = 0; i < 8; i++) fi[i] = 0;
    for (int i = 0; i < (win >> 1); i++) {
        const int64_t v = (int64_t)f[i] * WN_STEP;
        fi[i] = (int16_t)((v < 0 ? v - WN_SCALE / 2 : v + WN_SCALE / 2) / WN_SCALE);
    }
    /* WIENER_FILT_TAPn_{MIN,MAX}V (EbRestoration.h:130-149): tap 0 in [-5, 10], tap 1 in [-23, 8], tap 2 in [-17, 46] */
    if (win == 7) {
        fi[0] = (int16_t)(fi[0] < -5 ? -5 : fi[0] > 10 ? 10 : fi[0]);
        fi[1] = (int16_t)(fi[1] < -23 ? -23 : fi[1] > 8 ? 8 : fi[1]);
        fi[2] = (int16_t)(fi[2] < -17 ? -17 : fi[2] > 46 ? 46 : fi[2]);
    } else {   /* the narrower windows sit in the inner taps: what was rounded into fi[0], fi[1] moves one place in */
        fi[2] = (int16_t)(fi[1] < -17 ? -17 : fi[1] > 46 ? 46 : fi[1]);
        fi[1] = (int16_t)(fi[0] < -23 ? -23 : fi[0] > 8 ? 8 : fi[0]);
        fi[0] = 0;
    }
    fi[6] = fi[0]; fi[5] = fi[1]; fi[4] = fi[2];
    fi[3] = (int16_t)(-2 * (fi[0] + fi[1] + fi[2]));
}

/* compute_score (:980-1020): the filter's modelled error minus the identity filter's, from the statistics alone */
static int64_t wn_score(int win, const int64_t *M, const int64_t *H, const int16_t v[8], const int16_t h[8]) {
    const int win2 = win * win, off = (7 - win) >> 1;
    int16_t a[7], b[7];
    a[3] = b[3] = WN_STEP;
    for (int i = 0; i < 3; i++) { a[i] = a[6 - i] = v[i]; b[i] = b[6 - i] = h[i]; a[3] -= 2 * a[i]; b[3] -= 2 * b[i]; }
    int32_t ab[49];
    for (int k = 0; k < win; k++)
        for (int l = 0; l < win; l++) ab[k * win + l] = a[l + off] * b[k + off];
    int64_t P = 0, Q = 0;
    for (int k = 0; k < win2; k++) {
        P += ab[k] * M[k] / WN_STEP / WN_STEP;
        for (int l = 0; l < win2; l++) Q += ab[k] * H[k * win2 + l] * ab[l] / WN_STEP / WN_STEP / WN_STEP / WN_STEP;
    }
    const int64_t ident = H[(win2 >> 1) * win2 + (win2 >> 1)] - 2 * M[win2 >> 1];
    return Q - 2 * P - ident;
}

/* -> 1: vfilter / hfilter (8 x int16 each, InterpKernel layout) hold the unit's initial filter and it beats the identity filter by the model (the refinement follows);
 *    2: it does not (the unit gets no Wiener filter).  win = 7 / 5 / 3. */
int orc_wiener_unit_init(int win, const int64_t *M, const int64_t *H, int16_t vfilter[8], int16_t hfilter[8]) {
    static const int32_t mid[7] = {3, -7, 15, 128 - 2 * (3 - 7 + 15), 15, -7, 3};   /* WIENER_FILT_TAPn_MIDV */
    int32_t a[7], b[7];
    for (int i = 0; i < win; i++) a[i] = b[i] = (int32_t)(WN_SCALE / WN_STEP) * mid[i + ((7 - win) >> 1)];
    for (int round = 1; round < 5; round++) {   /* NUM_WIENER_ITERS = 5 (:40): four rounds */
        wn_half_round(win, M, H, 0, b, a);
        wn_half_round(win, M, H, 1, a, b);
    }
    wn_finalize(win, a, vfilter);
    wn_finalize(win, b, hfilter);
    return wn_score(win, M, H, vfilter, hfilter) > 0 ? 2 : 1;
}
