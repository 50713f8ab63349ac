/*
 * dlf_oracle.c — CPU restatement of SVT-AV1's deblocking edge filters (4/6/8/14 taps, 8-bit and
 * high bit-depth) and of a whole-plane two-pass application driven by per-4x4 edge descriptors.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 *
 * The reference has separate 8-bit (Common/Codec/EbDeblockingCommon.c:148-393, :810-921) and 16-bit
 * (:396-582, :607-805) code; both are the same arithmetic once the sample offset 0x80 << (bd-8),
 * the clamp range and the threshold scaling << (bd-8) are parameters, which is how it is written here.
 */
#include "svt_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* px[0..13] = p6 p5 p4 p3 p2 p1 p0 q0 q1 q2 q3 q4 q5 q6 (only the taps the length needs are read).
 * len in {4,6,8,14}; blimit/limit/thresh are the 8-bit table values (scaled by bd inside). */
void orc_lpf_core(int *px, int len, int blimit, int limit, int thresh, int bd) {
    const int sh = bd - 8, t80 = 0x80 << sh, lo = -t80, hi = t80 - 1;
    const int lim = limit << sh, blim = blimit << sh, thr = thresh << sh, one = 1 << sh;
    int *P = px + 6, *Q = px + 7; /* P[-i] = p_i, Q[i] = q_i */
#define p(i) P[-(i)]
#define q(i) Q[(i)]
    int mask; /* 1 = filter */
    {
        int m = 0;
        if (len == 4) {
            m |= iabs(p(1) - p(0)) > lim; m |= iabs(q(1) - q(0)) > lim;
        } else if (len == 6) {
            m |= iabs(p(2) - p(1)) > lim; m |= iabs(p(1) - p(0)) > lim; m |= iabs(q(1) - q(0)) > lim; m |= iabs(q(2) - q(1)) > lim;
        } else {
            m |= iabs(p(3) - p(2)) > lim; m |= iabs(p(2) - p(1)) > lim; m |= iabs(p(1) - p(0)) > lim;
            m |= iabs(q(1) - q(0)) > lim; m |= iabs(q(2) - q(1)) > lim; m |= iabs(q(3) - q(2)) > lim;
        }
        m |= (iabs(p(0) - q(0)) * 2 + iabs(p(1) - q(1)) / 2) > blim;
        mask = !m;
    }
    int flat = 0, flat2 = 0;
    if (len == 6)
        flat = !(iabs(p(1) - p(0)) > one || iabs(q(1) - q(0)) > one || iabs(p(2) - p(0)) > one || iabs(q(2) - q(0)) > one);
    if (len >= 8)
        flat = !(iabs(p(1) - p(0)) > one || iabs(q(1) - q(0)) > one || iabs(p(2) - p(0)) > one || iabs(q(2) - q(0)) > one ||
                 iabs(p(3) - p(0)) > one || iabs(q(3) - q(0)) > one);
    if (len == 14)
        flat2 = !(iabs(p(4) - p(0)) > one || iabs(q(4) - q(0)) > one || iabs(p(5) - p(0)) > one || iabs(q(5) - q(0)) > one ||
                  iabs(p(6) - p(0)) > one || iabs(q(6) - q(0)) > one);
#define RP2(v, n) (((v) + (1 << ((n)-1))) >> (n))
    if (len == 14 && flat2 && flat && mask) { /* filter14, EbDeblockingCommon.c:810-843 */
        const int p6 = p(6), p5 = p(5), p4 = p(4), p3 = p(3), p2 = p(2), p1 = p(1), p0 = p(0);
        const int q0 = q(0), q1 = q(1), q2 = q(2), q3 = q(3), q4 = q(4), q5 = q(5), q6 = q(6);
        p(5) = RP2(p6 * 7 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0, 4);
        p(4) = RP2(p6 * 5 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1, 4);
        p(3) = RP2(p6 * 4 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2, 4);
        p(2) = RP2(p6 * 3 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3, 4);
        p(1) = RP2(p6 * 2 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4, 4);
        p(0) = RP2(p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5, 4);
        q(0) = RP2(p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6, 4);
        q(1) = RP2(p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 * 2, 4);
        q(2) = RP2(p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 * 3, 4);
        q(3) = RP2(p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 * 4, 4);
        q(4) = RP2(p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 * 5, 4);
        q(5) = RP2(p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 7, 4);
        return;
    }
    if (len >= 8 && flat && mask) { /* filter8, :294-314 */
        const int p3 = p(3), p2 = p(2), p1 = p(1), p0 = p(0), q0 = q(0), q1 = q(1), q2 = q(2), q3 = q(3);
        p(2) = RP2(p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0, 3);
        p(1) = RP2(p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1, 3);
        p(0) = RP2(p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2, 3);
        q(0) = RP2(p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3, 3);
        q(1) = RP2(p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3, 3);
        q(2) = RP2(p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3, 3);
        return;
    }
    if (len == 6 && flat && mask) { /* filter6, :278-292 */
        const int p2 = p(2), p1 = p(1), p0 = p(0), q0 = q(0), q1 = q(1), q2 = q(2);
        p(1) = RP2(p2 * 3 + p1 * 2 + p0 * 2 + q0, 3);
        p(0) = RP2(p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1, 3);
        q(0) = RP2(p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2, 3);
        q(1) = RP2(p0 + q0 * 2 + q1 * 2 + q2 * 3, 3);
        return;
    }
    { /* filter4, :218-249 / highbd_filter4 :449-481 */
        const int ps1 = p(1) - t80, ps0 = p(0) - t80, qs0 = q(0) - t80, qs1 = q(1) - t80;
        const int hev = (iabs(p(1) - p(0)) > thr) || (iabs(q(1) - q(0)) > thr);
        int f = hev ? clampi(ps1 - qs1, lo, hi) : 0;
        f = mask ? clampi(f + 3 * (qs0 - ps0), lo, hi) : 0;
        const int f1 = clampi(f + 4, lo, hi) >> 3, f2 = clampi(f + 3, lo, hi) >> 3;
        q(0) = clampi(qs0 - f1, lo, hi) + t80;
        p(0) = clampi(ps0 + f2, lo, hi) + t80;
        const int f3 = hev ? 0 : ((f1 + 1) >> 1);
        q(1) = clampi(qs1 - f3, lo, hi) + t80;
        p(1) = clampi(ps1 + f3, lo, hi) + t80;
    }
#undef p
#undef q
#undef RP2
}

static int rd(const void *b, int pix_bytes, ptrdiff_t i) { return pix_bytes == 1 ? ((const uint8_t *)b)[i] : ((const uint16_t *)b)[i]; }
static void wr(void *b, int pix_bytes, ptrdiff_t i, int v) { if (pix_bytes == 1) ((uint8_t *)b)[i] = (uint8_t)v; else ((uint16_t *)b)[i] = (uint16_t)v; }

/* One 4-sample edge segment, like svt_aom_[highbd_]lpf_{vertical,horizontal}_{4,6,8,14}_c
 * (EbDeblockingCommon.c:251-393, :483-582, :698-921).  `s` points at q0 of the first sample;
 * vertical edge (dir 0): taps along x, samples step by `pitch`; horizontal edge (dir 1): taps along
 * y (step `pitch`), samples step by 1. */
void orc_lpf_edge(void *s, int pix_bytes, int pitch, int dir, int len, int blimit, int limit, int thresh, int bd) {
    const ptrdiff_t tap = dir == 0 ? 1 : pitch, step = dir == 0 ? pitch : 1;
    const int half = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
    for (int i = 0; i < 4; i++) {
        int px[14] = {0};
        for (int k = 1; k <= half; k++) { px[7 - k] = rd(s, pix_bytes, i * step - k * tap); px[6 + k] = rd(s, pix_bytes, i * step + (k - 1) * tap); }
        orc_lpf_core(px, len, blimit, limit, thresh, bd);
        for (int k = 1; k <= half; k++) { wr(s, pix_bytes, i * step - k * tap, px[7 - k]); wr(s, pix_bytes, i * step + (k - 1) * tap, px[6 + k]); }
    }
}

/* limits of a filter level: update_sharpness (EbDeblockingCommon.c:587-606) + hev_thr = lvl >> 4
 * (Encoder/Codec/EbDeblockingFilter.c:36-38) */
void orc_lf_limits(int level, int sharpness, int *lim, int *mblim, int *hev_thr) {
    int inside = level >> ((sharpness > 0) + (sharpness > 4));
    if (sharpness > 0 && inside > 9 - sharpness) inside = 9 - sharpness;
    if (inside < 1) inside = 1;
    *lim = inside;
    *mblim = 2 * (level + 2) + inside;
    *hev_thr = level >> 4;
}

/* Whole plane, normative order: every vertical edge, then every horizontal edge
 * (svt_av1_loop_filter_frame / loop_filter_sb, Encoder/Codec/EbDeblockingFilter.c:614-751, whose
 * per-SB interleaving is order-equivalent).  edges_v / edges_h: [units_h][units_w] uint16 =
 * (level << 8) | filter_length for the edge on the left / top side of each 4x4 unit (0 = none). */
void orc_deblock_plane(void *plane, int pix_bytes, int stride, int bd, const uint16_t *edges_v, const uint16_t *edges_h,
                       int units_w, int units_h, int sharpness) {
    for (int dir = 0; dir < 2; dir++) {
        const uint16_t *e = dir == 0 ? edges_v : edges_h;
        for (int uy = 0; uy < units_h; uy++)
            for (int ux = 0; ux < units_w; ux++) {
                const int len = e[uy * units_w + ux] & 0xff, level = e[uy * units_w + ux] >> 8;
                if (!len) continue;
                int lim, mblim, hev;
                orc_lf_limits(level, sharpness, &lim, &mblim, &hev);
                uint8_t *s = (uint8_t *)plane + ((size_t)(4 * uy) * stride + 4 * ux) * pix_bytes;
                orc_lpf_edge(s, pix_bytes, stride, dir, len, mblim, lim, hev, bd);
            }
    }
}

/* svt_spatial_full_distortion_kernel_c / svt_full_distortion_kernel16_bits_c (Common/Codec/EbPictureOperators.c:182-208
 * and the 8-bit twin), as used by picture_sse_calculations (Encoder/Codec/EbDeblockingFilter.c:830-961). */
uint64_t orc_plane_sse(int pix_bytes, const void *a, int a_stride, const void *b, int b_stride, int w, int h) {
    uint64_t sse = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int64_t d = pix_bytes == 1 ? (int64_t)((const uint8_t *)a)[(size_t)y * a_stride + x] - ((const uint8_t *)b)[(size_t)y * b_stride + x]
                                             : (int64_t)((const uint16_t *)a)[(size_t)y * a_stride + x] - ((const uint16_t *)b)[(size_t)y * b_stride + x];
            sse += (uint64_t)(d * d);
        }
    return sse;
}

/* try_filter_frame (EbDeblockingFilter.c:966-1024) with a frame-uniform level: copy, deblock, SSE vs source. */
static int64_t orc_try_level(const void *recon, void *tmp, int pix_bytes, int stride, int bd, int w, int h, const void *src, int src_stride,
                             const uint16_t *ev, const uint16_t *eh, int uw, int uh, int sharpness, int lv_v, int lv_h) {
    uint16_t *tv = (uint16_t *)malloc((size_t)uw * uh * 2), *th = (uint16_t *)malloc((size_t)uw * uh * 2);
    for (int i = 0; i < uw * uh; i++) {   /* the level of every edge becomes the probed one; level 0 = edge not filtered (:245) */
        tv[i] = (ev[i] & 0xff) && lv_v ? (uint16_t)((lv_v << 8) | (ev[i] & 0xff)) : 0;
        th[i] = (eh[i] & 0xff) && lv_h ? (uint16_t)((lv_h << 8) | (eh[i] & 0xff)) : 0;
    }
    for (int y = 0; y < h; y++) memcpy((uint8_t *)tmp + (size_t)y * stride * pix_bytes, (const uint8_t *)recon + (size_t)y * stride * pix_bytes, (size_t)w * pix_bytes);
    orc_deblock_plane(tmp, pix_bytes, stride, bd, tv, th, uw, uh, sharpness);
    free(tv); free(th);
    return (int64_t)orc_plane_sse(pix_bytes, src, src_stride, tmp, stride, w, h);
}

/* search_filter_level (EbDeblockingFilter.c:1026-1187).  Returns filt_best; *best_err = ss_err[filt_best]; probes[] (64 entries,
 * optional) receives ss_err[] (-1 = level never probed) so a test can compare the probe sequence as well. */
int orc_dlf_search_level(const void *recon, void *tmp, int pix_bytes, int stride, int bd, int w, int h, const void *src, int src_stride,
                         const uint16_t *ev, const uint16_t *eh, int uw, int uh, int sharpness, int plane, int dir, int other_level,
                         int start_level, int loop_filter_mode, int tx_mode_only_4x4, int64_t *best_err_out, int64_t *probes) {
    int64_t ss_err[64];
    memset(ss_err, 0xFF, sizeof(ss_err));
    int filt_direction = 0;
    int filt_mid = start_level < 0 ? 0 : (start_level > 63 ? 63 : start_level);
    int filter_step = filt_mid < 16 ? 4 : filt_mid / 4;
#define TRY(l) orc_try_level(recon, tmp, pix_bytes, stride, bd, w, h, src, src_stride, ev, eh, uw, uh, sharpness, \
                             (plane == 0 && dir == 1) ? other_level : (l), (plane == 0 && dir == 0) ? other_level : (l))
    int64_t best_err = TRY(filt_mid);
    int filt_best = filt_mid;
    ss_err[filt_mid] = best_err;
    if (loop_filter_mode <= 2) {
        filter_step = 2;
        const int filt_high = filt_mid + filter_step > 63 ? 63 : filt_mid + filter_step;
        const int filt_low = filt_mid - filter_step < 0 ? 0 : filt_mid - filter_step;
        int64_t bias = (best_err >> (15 - (filt_mid / 8))) * filter_step;
        if (!tx_mode_only_4x4) bias >>= 1;
        if (filt_direction <= 0 && filt_low != filt_mid) {
            if (ss_err[filt_low] < 0) ss_err[filt_low] = TRY(filt_low);
            if (ss_err[filt_low] < (best_err + bias)) {
                if (ss_err[filt_low] < best_err) best_err = ss_err[filt_low];
                filt_best = filt_low;
            }
        }
        if (filt_direction >= 0 && filt_high != filt_mid) {
            if (ss_err[filt_high] < 0) ss_err[filt_high] = TRY(filt_high);
            if (ss_err[filt_high] < (best_err - bias)) filt_best = filt_high;
        }
    } else {
        while (filter_step > 0) {
            const int filt_high = filt_mid + filter_step > 63 ? 63 : filt_mid + filter_step;
            const int filt_low = filt_mid - filter_step < 0 ? 0 : filt_mid - filter_step;
            int64_t bias = (best_err >> (15 - (filt_mid / 8))) * filter_step;
            if (!tx_mode_only_4x4) bias >>= 1;
            if (filt_direction <= 0 && filt_low != filt_mid) {
                if (ss_err[filt_low] < 0) ss_err[filt_low] = TRY(filt_low);
                if (ss_err[filt_low] < (best_err + bias)) {
                    if (ss_err[filt_low] < best_err) best_err = ss_err[filt_low];
                    filt_best = filt_low;
                }
            }
            if (filt_direction >= 0 && filt_high != filt_mid) {
                if (ss_err[filt_high] < 0) ss_err[filt_high] = TRY(filt_high);
                if (ss_err[filt_high] < (best_err - bias)) {
                    best_err = ss_err[filt_high];
                    filt_best = filt_high;
                }
            }
            if (filt_best == filt_mid) {
                filter_step /= 2;
                filt_direction = 0;
            } else {
                filt_direction = (filt_best < filt_mid) ? -1 : 1;
                filt_mid = filt_best;
            }
        }
    }
#undef TRY
    if (best_err_out) *best_err_out = ss_err[filt_best];
    if (probes) memcpy(probes, ss_err, sizeof(ss_err));
    return filt_best;
}
