/*
 * sgr_oracle.c — CPU restatement of SVT-AV1's self-guided restoration (SGRPROJ): the two box filters,
 * the projection sums / solve / error used by the search, and the final application.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 *
 * The reference computes the (2r+1)^2 box sums with running-sum helpers over a 3-pixel-extended copy
 * of the processing unit (Common/Codec/EbRestoration.c:541-705); only full windows are ever consumed,
 * so here every A/B entry is simply summed over its window.
 */
#include "svt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* eb_sgr_params (EbRestoration.c:136-153): {r0, r1, s0, s1} */
const int32_t orc_sgr_params[16][4] = {{2, 1, 140, 3236}, {2, 1, 112, 2158}, {2, 1, 93, 1618}, {2, 1, 80, 1438}, {2, 1, 70, 1295}, {2, 1, 58, 1177},
                                       {2, 1, 47, 1079},  {2, 1, 37, 996},   {2, 1, 30, 925},  {2, 1, 25, 863},  {0, 1, -1, 2589}, {0, 1, -1, 1618},
                                       {0, 1, -1, 1177},  {0, 1, -1, 925},   {2, 0, 56, -1},   {2, 0, 22, -1}};
/* eb_x_by_xplus1 (EbRestoration.c:720-736): round(256 z / (z + 1)) with the two documented exceptions
 * [0] = 1 and [255] = 256; eb_one_by_x (:738-741): round(4096 / n). */
int32_t orc_x_by_xplus1(int z) { return z == 0 ? 1 : (z == 255 ? 256 : (256 * z + (z + 1) / 2) / (z + 1)); }
int32_t orc_one_by_x(int n) { return (4096 + n / 2) / n; }

static inline int rdp(const void *b, int pb, ptrdiff_t i) { return pb == 1 ? ((const uint8_t *)b)[i] : ((const uint16_t *)b)[i]; }
static inline uint32_t rp2u(uint32_t v, int n) { return n == 0 ? v : ((v + (1u << (n - 1))) >> n); }

/* A'/B' at position (i, j) of the unit: EbRestoration.c:787-858 (r = 2) / :926-985 (r = 1) */
static void ab_at(const void *dgd, int pb, int stride, int i, int j, int r, uint32_t s, int bd, int32_t *A, int32_t *B) {
    uint32_t sum = 0, sq = 0;
    for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) { const uint32_t v = (uint32_t)rdp(dgd, pb, (ptrdiff_t)(i + dy) * stride + j + dx); sum += v; sq += v * v; }
    const uint32_t n = (uint32_t)((2 * r + 1) * (2 * r + 1));
    const uint32_t a = rp2u(sq, 2 * (bd - 8)), b = rp2u(sum, bd - 8);
    const uint32_t p = (a * n < b * b) ? 0 : a * n - b * b;
    const uint32_t z = rp2u(p * s, 20);
    const int32_t Ak = orc_x_by_xplus1(z < 255 ? (int)z : 255);
    *A = Ak;
    *B = (int32_t)rp2u((uint32_t)(256 - Ak) * sum * (uint32_t)orc_one_by_x((int)n), 12);
}

/* svt_av1_selfguided_restoration_c (EbRestoration.c:1012-1045) on one processing unit (w, h <= 64;
 * 3 valid samples around it in `dgd`).  flt0 / flt1 untouched when the corresponding radius is 0. */
void orc_sgr_filter(const void *dgd, int pix_bytes, int w, int h, int stride, int32_t *flt0, int32_t *flt1, int flt_stride, int ep, int bd) {
    const int32_t *prm = orc_sgr_params[ep];
    int32_t *A = (int32_t *)malloc(sizeof(int32_t) * 2 * 66 * 66), *B = A + 66 * 66;
#define AT(M, i, j) M[((i) + 1) * 66 + (j) + 1]
    if (prm[0] > 0) { /* selfguided_restoration_fast_internal, r = 2: A/B on rows -1, 1, 3, ... */
        for (int i = -1; i < h + 1; i += 2)
            for (int j = -1; j < w + 1; j++) ab_at(dgd, pix_bytes, stride, i, j, 2, (uint32_t)prm[2], bd, &AT(A, i, j), &AT(B, i, j));
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                int32_t a, b, nb;
                if (!(i & 1)) {
                    nb = 5;
                    a = (AT(A, i - 1, j) + AT(A, i + 1, j)) * 6 + (AT(A, i - 1, j - 1) + AT(A, i + 1, j - 1) + AT(A, i - 1, j + 1) + AT(A, i + 1, j + 1)) * 5;
                    b = (AT(B, i - 1, j) + AT(B, i + 1, j)) * 6 + (AT(B, i - 1, j - 1) + AT(B, i + 1, j - 1) + AT(B, i - 1, j + 1) + AT(B, i + 1, j + 1)) * 5;
                } else {
                    nb = 4;
                    a = AT(A, i, j) * 6 + (AT(A, i, j - 1) + AT(A, i, j + 1)) * 5;
                    b = AT(B, i, j) * 6 + (AT(B, i, j - 1) + AT(B, i, j + 1)) * 5;
                }
                const int32_t v = a * rdp(dgd, pix_bytes, (ptrdiff_t)i * stride + j) + b;
                flt0[i * flt_stride + j] = (v + (1 << (8 + nb - 4 - 1))) >> (8 + nb - 4);
            }
    }
    if (prm[1] > 0) { /* selfguided_restoration_internal, r = 1 */
        for (int i = -1; i < h + 1; i++)
            for (int j = -1; j < w + 1; j++) ab_at(dgd, pix_bytes, stride, i, j, 1, (uint32_t)prm[3], bd, &AT(A, i, j), &AT(B, i, j));
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                const int32_t a = (AT(A, i, j) + AT(A, i, j - 1) + AT(A, i, j + 1) + AT(A, i - 1, j) + AT(A, i + 1, j)) * 4 +
                                  (AT(A, i - 1, j - 1) + AT(A, i + 1, j - 1) + AT(A, i - 1, j + 1) + AT(A, i + 1, j + 1)) * 3;
                const int32_t b = (AT(B, i, j) + AT(B, i, j - 1) + AT(B, i, j + 1) + AT(B, i - 1, j) + AT(B, i + 1, j)) * 4 +
                                  (AT(B, i - 1, j - 1) + AT(B, i + 1, j - 1) + AT(B, i - 1, j + 1) + AT(B, i + 1, j + 1)) * 3;
                const int32_t v = a * rdp(dgd, pix_bytes, (ptrdiff_t)i * stride + j) + b;
                flt1[i * flt_stride + j] = (v + (1 << (8 + 5 - 4 - 1))) >> (8 + 5 - 4);
            }
    }
#undef AT
    free(A);
}

/* svt_decode_xq (EbRestoration.c:707-718) */
void orc_sgr_decode_xq(const int32_t *xqd, int32_t *xq, int ep) {
    const int32_t *prm = orc_sgr_params[ep];
    if (prm[0] == 0) { xq[0] = 0; xq[1] = 128 - xqd[1]; }
    else if (prm[1] == 0) { xq[0] = xqd[0]; xq[1] = 0; }
    else { xq[0] = xqd[0]; xq[1] = 128 - xq[0] - xqd[1]; }
}

/* svt_apply_selfguided_restoration_c (EbRestoration.c:1047-1084) on a unit of any size, walked in
 * 64x64 processing units like sgrproj_filter_stripe (:1086-1132) does. */
void orc_sgr_apply(const void *dat, int pix_bytes, int w, int h, int stride, int ep, const int32_t *xqd, void *dst, int dst_stride, int bd) {
    const int32_t *prm = orc_sgr_params[ep];
    int32_t xq[2];
    orc_sgr_decode_xq(xqd, xq, ep);
    int32_t *f0 = (int32_t *)malloc(sizeof(int32_t) * 2 * 64 * 64), *f1 = f0 + 64 * 64;
    for (int y0 = 0; y0 < h; y0 += 64)
        for (int x0 = 0; x0 < w; x0 += 64) {
            const int pw = w - x0 < 64 ? w - x0 : 64, ph = h - y0 < 64 ? h - y0 : 64;
            const uint8_t *d = (const uint8_t *)dat + ((size_t)y0 * stride + x0) * pix_bytes;
            orc_sgr_filter(d, pix_bytes, pw, ph, stride, f0, f1, 64, ep, bd);
            for (int i = 0; i < ph; i++)
                for (int j = 0; j < pw; j++) {
                    const int32_t u = rdp(d, pix_bytes, (ptrdiff_t)i * stride + j) << 4;
                    int32_t v = u << 7;
                    if (prm[0] > 0) v += xq[0] * (f0[i * 64 + j] - u);
                    if (prm[1] > 0) v += xq[1] * (f1[i * 64 + j] - u);
                    const int16_t wv = (int16_t)((v + (1 << 10)) >> 11);
                    const int mx = (1 << bd) - 1, o = wv < 0 ? 0 : (wv > mx ? mx : wv);
                    if (pix_bytes == 1) ((uint8_t *)dst)[(size_t)(y0 + i) * dst_stride + x0 + j] = (uint8_t)o;
                    else ((uint16_t *)dst)[(size_t)(y0 + i) * dst_stride + x0 + j] = (uint16_t)o;
                }
        }
    free(f0);
}

/* The five sums svt_get_proj_subspace_c accumulates (Encoder/Codec/EbRestorationPick.c:448-496) as exact
 * integers: sums[0..4] = H00, H01, H11, C0, C1 (before the division by the pixel count). */
void orc_sgr_proj_sums(const void *src, int src_stride, const void *dat, int dat_stride, int pix_bytes, int w, int h, const int32_t *flt0,
                       int f0_stride, const int32_t *flt1, int f1_stride, int ep, int64_t sums[5]) {
    const int32_t *prm = orc_sgr_params[ep];
    memset(sums, 0, sizeof(int64_t) * 5);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const int64_t u = rdp(dat, pix_bytes, (ptrdiff_t)i * dat_stride + j) << 4;
            const int64_t s = (rdp(src, pix_bytes, (ptrdiff_t)i * src_stride + j) << 4) - u;
            const int64_t f1 = prm[0] > 0 ? flt0[i * f0_stride + j] - u : 0, f2 = prm[1] > 0 ? flt1[i * f1_stride + j] - u : 0;
            sums[0] += f1 * f1; sums[1] += f1 * f2; sums[2] += f2 * f2; sums[3] += f1 * s; sums[4] += f2 * s;
        }
}
/* the FP64 solve of svt_get_proj_subspace_c (:497-538) from the integer sums */
void orc_sgr_solve(const int64_t sums[5], int size, int ep, int32_t xq[2]) {
    const int32_t *prm = orc_sgr_params[ep];
    double H00 = (double)sums[0], H01 = (double)sums[1], H11 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    H00 /= size; H01 /= size; H11 /= size; C0 /= size; C1 /= size;
    const double H10 = H01;
    xq[0] = xq[1] = 0;
    if (prm[0] == 0) {
        if (H11 < 1e-8) return;
        xq[1] = (int32_t)rint((C1 / H11) * 128);
    } else if (prm[1] == 0) {
        if (H00 < 1e-8) return;
        xq[0] = (int32_t)rint((C0 / H00) * 128);
    } else {
        const double det = H00 * H11 - H01 * H10;
        if (det < 1e-8) return;
        const double x0 = (H11 * C0 - H01 * C1) / det, x1 = (H00 * C1 - H10 * C0) / det;
        xq[0] = (int32_t)rint(x0 * 128);
        xq[1] = (int32_t)rint(x1 * 128);
    }
}
/* svt_av1_{lowbd,highbd}_pixel_proj_error_c (EbRestorationPick.c:174-316) */
int64_t orc_sgr_proj_error(const void *src, int src_stride, const void *dat, int dat_stride, int pix_bytes, int w, int h, const int32_t *flt0,
                           int f0_stride, const int32_t *flt1, int f1_stride, const int32_t xq[2], int ep) {
    const int32_t *prm = orc_sgr_params[ep];
    int64_t err = 0;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const int32_t d = rdp(dat, pix_bytes, (ptrdiff_t)i * dat_stride + j), s = rdp(src, pix_bytes, (ptrdiff_t)i * src_stride + j);
            int32_t e;
            if (prm[0] <= 0 && prm[1] <= 0) e = d - s;
            else {
                const int32_t u = d << 4;
                int32_t v = u << 7;
                if (prm[0] > 0) v += xq[0] * (flt0[i * f0_stride + j] - u);
                if (prm[1] > 0) v += xq[1] * (flt1[i * f1_stride + j] - u);
                e = ((v + (1 << 10)) >> 11) - s;
            }
            err += (int64_t)e * e;
        }
    return err;
}

/* ---------------------------------------------------------------- frame level -- */
/* count_units_in_tile (EbRestoration.c:172-180) */
int orc_rest_units(int size, int unit_size) { const int n = (size + (unit_size >> 1)) / unit_size; return n > 1 ? n : 1; }

/* foreach_rest_unit_in_tile (EbRestoration.c:1369-1411) for the single-tile frame: limits[u][4] =
 * {h_start, h_end, v_start, v_end}; unit rows are shifted up by RESTORATION_UNIT_OFFSET >> ss_y,
 * the last row / column of units absorbs a remainder smaller than half a unit. */
int orc_rest_unit_limits(int pw, int ph, int ss_y, int unit_size, int32_t *limits) {
    const int ext = unit_size * 3 / 2, voff = 8 >> ss_y;
    const int hunits = orc_rest_units(pw, unit_size);
    int y0 = 0, i = 0, n = 0;
    while (y0 < ph) {
        const int rem_h = ph - y0, h = rem_h < ext ? rem_h : unit_size;
        int v_start = y0, v_end = y0 + h;
        v_start = v_start - voff > 0 ? v_start - voff : 0;
        if (v_end < ph) v_end -= voff;
        int x0 = 0, j = 0;
        while (x0 < pw) {
            const int rem_w = pw - x0, w = rem_w < ext ? rem_w : unit_size;
            int32_t *o = limits + 4 * (i * hunits + j);
            o[0] = x0; o[1] = x0 + w; o[2] = v_start; o[3] = v_end;
            x0 += w; j++; n++;
        }
        y0 += h; i++;
    }
    return n;
}

/* The integer sums of search_selfguided_restoration (Encoder/Codec/EbRestorationPick.c:583-671) for every unit of a plane and every
 * parameter set in ep_mask: apply_sgr (:554-581) walks the unit in (64 >> ss) processing units from the unit's own origin, then
 * svt_get_proj_subspace accumulates over the whole unit.  dgd = pixel (0,0) of the 3-px extended picture.  sums[unit][16][5]. */
void orc_sgr_search_plane(const void *dgd, int pix_bytes, int stride, const void *src, int src_stride, int pw, int ph, int ss_x, int ss_y,
                          int unit_size, int bd, uint32_t ep_mask, int64_t *sums) {
    const int nu = orc_rest_units(pw, unit_size) * orc_rest_units(ph, unit_size);
    int32_t *lim = (int32_t *)malloc(sizeof(int32_t) * 4 * nu);
    orc_rest_unit_limits(pw, ph, ss_y, unit_size, lim);
    const int puw = 64 >> ss_x, puh = 64 >> ss_y;
    for (int u = 0; u < nu; u++) {
        const int x0 = lim[4 * u], x1 = lim[4 * u + 1], y0 = lim[4 * u + 2], y1 = lim[4 * u + 3], w = x1 - x0, h = y1 - y0;
        const int fs = ((w + 7) & ~7) + 8;
        int32_t *f0 = (int32_t *)malloc(sizeof(int32_t) * 2 * fs * h), *f1 = f0 + (size_t)fs * h;
        const uint8_t *d = (const uint8_t *)dgd + ((size_t)y0 * stride + x0) * pix_bytes;
        const uint8_t *s = (const uint8_t *)src + ((size_t)y0 * src_stride + x0) * pix_bytes;
        for (int ep = 0; ep < 16; ep++) {
            int64_t *o = sums + ((size_t)u * 16 + ep) * 5;
            if (!((ep_mask >> ep) & 1)) continue;
            for (int i = 0; i < h; i += puh)
                for (int j = 0; j < w; j += puw)
                    orc_sgr_filter(d + ((size_t)i * stride + j) * pix_bytes, pix_bytes, w - j < puw ? w - j : puw, h - i < puh ? h - i : puh, stride,
                                   f0 + (size_t)i * fs + j, f1 + (size_t)i * fs + j, fs, ep, bd);
            orc_sgr_proj_sums(s, src_stride, d, stride, pix_bytes, w, h, f0, fs, f1, fs, ep, o);
        }
        free(f0);
    }
    free(lim);
}

/* encode_xq (EbRestorationPick.c:539-552); SGRPROJ_PRJ_MIN0/MAX0 = -96/31, MIN1/MAX1 = -32/95 (EbRestoration.h:100-103) */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
void orc_sgr_encode_xq(const int32_t *xq, int32_t *xqd, int ep) {
    const int32_t *prm = orc_sgr_params[ep];
    if (prm[0] == 0) { xqd[0] = 0; xqd[1] = clampi(128 - xq[1], -32, 95); }
    else if (prm[1] == 0) { xqd[0] = clampi(xq[0], -96, 31); xqd[1] = clampi(128 - xqd[0], -32, 95); }
    else { xqd[0] = clampi(xq[0], -96, 31); xqd[1] = clampi(128 - xqd[0] - xq[1], -32, 95); }
}

/* get_pixel_proj_error (EbRestorationPick.c:317-351): decode the xqd pair, then the projected SSE */
static int64_t xqd_error(const void *src, int src_stride, const void *dat, int dat_stride, int pix_bytes, int w, int h, const int32_t *flt0,
                         int f0_stride, const int32_t *flt1, int f1_stride, const int32_t *xqd, int ep) {
    int32_t xq[2];
    orc_sgr_decode_xq(xqd, xq, ep);
    return orc_sgr_proj_error(src, src_stride, dat, dat_stride, pix_bytes, w, h, flt0, f0_stride, flt1, f1_stride, xq, ep);
}

/* finer_search_pixel_proj_error (EbRestorationPick.c:353-446): coordinate descent on the xqd pair with steps start_step, .., 1.  For each
 * step and each active coefficient: try -step (at the largest step keep walking while the error does not increase); only when that
 * did not help try +step the same way.  "Not worse" (err2 <= err) counts as an improvement. */
int64_t orc_sgr_finer_search(const void *src, int src_stride, const void *dat, int dat_stride, int pix_bytes, int w, int h, const int32_t *flt0,
                             int f0_stride, const int32_t *flt1, int f1_stride, int start_step, int32_t *xqd, int ep) {
    const int32_t *prm = orc_sgr_params[ep];
    const int tap_min[2] = {-96, -32}, tap_max[2] = {31, 95};
    int64_t err = xqd_error(src, src_stride, dat, dat_stride, pix_bytes, w, h, flt0, f0_stride, flt1, f1_stride, xqd, ep);
    for (int s = start_step; s >= 1; s >>= 1) {
        for (int p = 0; p < 2; p++) {
            if ((prm[0] == 0 && p == 0) || (prm[1] == 0 && p == 1)) continue;
            int skip = 0;
            for (;;) {
                if (xqd[p] - s >= tap_min[p]) {
                    xqd[p] -= s;
                    const int64_t err2 = xqd_error(src, src_stride, dat, dat_stride, pix_bytes, w, h, flt0, f0_stride, flt1, f1_stride, xqd, ep);
                    if (err2 > err) xqd[p] += s;
                    else { err = err2; skip = 1; if (s == start_step) continue; }
                }
                break;
            }
            if (skip) break;   /* as in the reference: leaves the loop over p for this step */
            for (;;) {
                if (xqd[p] + s <= tap_max[p]) {
                    xqd[p] += s;
                    const int64_t err2 = xqd_error(src, src_stride, dat, dat_stride, pix_bytes, w, h, flt0, f0_stride, flt1, f1_stride, xqd, ep);
                    if (err2 > err) xqd[p] -= s;
                    else { err = err2; if (s == start_step) continue; }
                }
                break;
            }
        }
    }
    return err;
}

/* search_selfguided_restoration (EbRestorationPick.c:583-671) for every restoration unit of a plane and every parameter set in ep_mask:
 * box filters (apply_sgr :554), svt_get_proj_subspace, encode_xq, finer search with start step 2.  xqd_out[unit][16][2], err_out[unit][16]
 * (untouched for sets outside the mask); best_ep[unit] = the first set of the mask with the smallest error (strict <, :659).  The
 * reference's window [start_ep, end_ep) around the reference frames' sets (:596-607) is the caller's ep_mask. */
void orc_sgr_search_units_plane(const void *dgd, int pix_bytes, int stride, const void *src, int src_stride, int pw, int ph, int ss_x, int ss_y,
                                int unit_size, int bd, uint32_t ep_mask, int32_t *xqd_out, int64_t *err_out, uint8_t *best_ep) {
    const int nu = orc_rest_units(pw, unit_size) * orc_rest_units(ph, unit_size);
    int32_t *lim = (int32_t *)malloc(sizeof(int32_t) * 4 * nu);
    orc_rest_unit_limits(pw, ph, ss_y, unit_size, lim);
    const int puw = 64 >> ss_x, puh = 64 >> ss_y;
    for (int u = 0; u < nu; u++) {
        const int x0 = lim[4 * u], x1 = lim[4 * u + 1], y0 = lim[4 * u + 2], y1 = lim[4 * u + 3], w = x1 - x0, h = y1 - y0;
        const int fs = ((w + 7) & ~7) + 8;
        int32_t *f0 = (int32_t *)malloc(sizeof(int32_t) * 2 * fs * h), *f1 = f0 + (size_t)fs * h;
        const uint8_t *d = (const uint8_t *)dgd + ((size_t)y0 * stride + x0) * pix_bytes;
        const uint8_t *s = (const uint8_t *)src + ((size_t)y0 * src_stride + x0) * pix_bytes;
        int64_t besterr = -1;
        for (int ep = 0; ep < 16; ep++) {
            if (!((ep_mask >> ep) & 1)) continue;
            for (int i = 0; i < h; i += puh)
                for (int j = 0; j < w; j += puw)
                    orc_sgr_filter(d + ((size_t)i * stride + j) * pix_bytes, pix_bytes, w - j < puw ? w - j : puw, h - i < puh ? h - i : puh, stride,
                                   f0 + (size_t)i * fs + j, f1 + (size_t)i * fs + j, fs, ep, bd);
            int64_t sums[5];
            int32_t xq[2], *xqd = xqd_out + ((size_t)u * 16 + ep) * 2;
            orc_sgr_proj_sums(s, src_stride, d, stride, pix_bytes, w, h, f0, fs, f1, fs, ep, sums);
            orc_sgr_solve(sums, w * h, ep, xq);
            orc_sgr_encode_xq(xq, xqd, ep);
            const int64_t err = orc_sgr_finer_search(s, src_stride, d, stride, pix_bytes, w, h, f0, fs, f1, fs, 2, xqd, ep);
            err_out[(size_t)u * 16 + ep] = err;
            if (besterr == -1 || err < besterr) { besterr = err; if (best_ep) best_ep[u] = (uint8_t)ep; }
        }
        free(f0);
    }
    free(lim);
}

/* svt_av1_loop_restoration_filter_frame for one plane (EbRestoration.c:1293-1366) = for every unit svt_av1_loop_restoration_filter_unit
 * (:1162-1249): the unit is filtered stripe by stripe (64 >> ss_y rows, the first 8 >> ss_y shorter); the 3 rows above / below a
 * stripe are replaced by the DEBLOCKED picture's rows (2 saved rows stretched to 3: get_stripe_boundary_info :321,
 * setup_processing_stripe_boundary :353-453, saved by save_deblock_boundary_lines :1645-1697 with edge replication) unless the stripe
 * touches the top / bottom of the frame, where the CDEF picture's own 3-px extension stays.
 * dbl = deblocked plane, cdef = CDEF output plane with a valid 3-px extension (modified and restored like the reference does);
 * unit_ep[u] > 15 = RESTORE_NONE (copy). */
void orc_sgr_apply_plane(const void *dbl, int dbl_stride, void *cdef, int stride, int pix_bytes, int pw, int ph, int ss_x, int ss_y,
                         int unit_size, int bd, const uint8_t *unit_ep, const int32_t *unit_xqd, void *dst, int dst_stride) {
    orc_lr_apply_plane(dbl, dbl_stride, cdef, stride, pix_bytes, pw, ph, ss_x, ss_y, unit_size, bd, unit_ep, unit_xqd, NULL, dst, dst_stride);
}
/* ... with Wiener units as well: unit_ep[u] == 254 -> wiener_filter_stripe[_highbd] (EbRestoration.c:1040-1085, :1110-1132) with the taps
 * unit_wiener[u][0][8] (vertical) / [u][1][8] (horizontal).  The reference rounds a stripe's last column block up to 16 columns and lets
 * the neighbouring unit (or the padding) take the surplus; the restatement writes the unit's own columns only -- same picture. */
void orc_lr_apply_plane(const void *dbl, int dbl_stride, void *cdef, int stride, int pix_bytes, int pw, int ph, int ss_x, int ss_y, int unit_size,
                        int bd, const uint8_t *unit_ep, const int32_t *unit_xqd, const int16_t *unit_wiener, void *dst, int dst_stride) {
    const int nu = orc_rest_units(pw, unit_size) * orc_rest_units(ph, unit_size);
    int32_t *lim = (int32_t *)malloc(sizeof(int32_t) * 4 * nu);
    orc_rest_unit_limits(pw, ph, ss_y, unit_size, lim);
    const int full = 64 >> ss_y, off = 8 >> ss_y, puw = 64 >> ss_x;
    uint8_t *save = (uint8_t *)malloc((size_t)6 * (pw + 6) * pix_bytes);
    for (int u = 0; u < nu; u++) {
        const int x0 = lim[4 * u], x1 = lim[4 * u + 1], v0 = lim[4 * u + 2], v1 = lim[4 * u + 3], uw = x1 - x0;
        const int is_wiener = unit_ep[u] == 254 && unit_wiener;
        if (unit_ep[u] > 15 && !is_wiener) {   /* copy_tile */
            for (int y = v0; y < v1; y++)
                memcpy((uint8_t *)dst + ((size_t)y * dst_stride + x0) * pix_bytes, (const uint8_t *)cdef + ((size_t)y * stride + x0) * pix_bytes, (size_t)uw * pix_bytes);
            continue;
        }
        int i = 0;
        while (i < v1 - v0) {
            const int v = v0 + i;
            const int first = v == 0, this_h = full - (first ? off : 0), last = v + this_h >= ph;
            const int tile_stripe = (v + off) / full;
            const int nominal = full - (tile_stripe == 0 ? off : 0), h = nominal < v1 - v ? nominal : v1 - v;
            const int lx0 = x0 - 3, lw = uw + 6;   /* columns the filter reads (the reference swaps 4 extra px each side) */
            uint8_t *sv = save;
            for (int pass = 0; pass < 2; pass++) {            /* 0: above, 1: below */
                if (pass == 0 ? first : last) continue;
                for (int k = 0; k < 3; k++) {
                    const int row = pass == 0 ? v - 3 + k : v + h + k;
                    int srow = pass == 0 ? (v - 2) + (k - 1 > 0 ? k - 1 : 0) : v + h + (k < 1 ? k : 1);
                    if (srow > ph - 1) srow = ph - 1;   /* lines_to_save == 1: the single line is duplicated */
                    uint8_t *drow = (uint8_t *)cdef + ((ptrdiff_t)row * stride + lx0) * pix_bytes;
                    memcpy(sv, drow, (size_t)lw * pix_bytes); sv += (size_t)lw * pix_bytes;
                    for (int x = 0; x < lw; x++) {
                        int sx = lx0 + x; sx = sx < 0 ? 0 : (sx > pw - 1 ? pw - 1 : sx);
                        if (pix_bytes == 1) drow[x] = ((const uint8_t *)dbl)[(size_t)srow * dbl_stride + sx];
                        else ((uint16_t *)drow)[x] = ((const uint16_t *)dbl)[(size_t)srow * dbl_stride + sx];
                    }
                }
            }
            for (int j = 0; j < uw; j += puw)   /* wiener_filter_stripe / sgrproj_filter_stripe[_highbd] (:1040-1160) */
                if (is_wiener)
                    orc_wiener_convolve_add_src((const uint8_t *)cdef + ((size_t)v * stride + x0 + j) * pix_bytes, stride,
                                                (uint8_t *)dst + ((size_t)v * dst_stride + x0 + j) * pix_bytes, dst_stride, pix_bytes, unit_wiener + 16 * u + 8,
                                                unit_wiener + 16 * u, uw - j < puw ? uw - j : puw, h, bd);
                else
                orc_sgr_apply((const uint8_t *)cdef + ((size_t)v * stride + x0 + j) * pix_bytes, pix_bytes, uw - j < puw ? uw - j : puw, h, stride, unit_ep[u],
                              unit_xqd + 2 * u, (uint8_t *)dst + ((size_t)v * dst_stride + x0 + j) * pix_bytes, dst_stride, bd);
            sv = save;
            for (int pass = 0; pass < 2; pass++) {            /* restore_processing_stripe_boundary */
                if (pass == 0 ? first : last) continue;
                for (int k = 0; k < 3; k++) {
                    const int row = pass == 0 ? v - 3 + k : v + h + k;
                    memcpy((uint8_t *)cdef + ((ptrdiff_t)row * stride + lx0) * pix_bytes, sv, (size_t)lw * pix_bytes); sv += (size_t)lw * pix_bytes;
                }
            }
            i += h;
        }
    }
    free(save); free(lim);
}
