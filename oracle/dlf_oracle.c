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

/* =====================================================================================================================================================
 * E2: which edges the deblocking filter touches, how long the filter is and at which level — set_lpf_parameters (Encoder/Codec/EbDeblockingFilter.c:168-319)
 * with get_transform_size (:134-166), the level table of svt_av1_loop_filter_frame_init (Common/Codec/EbDeblockingCommon.c:71-146) and the unit ranges of
 * svt_av1_filter_block_plane_vert / _horz (:322-367, :463-508).  Written from those functions; pinned to svt_av1_loop_filter_frame itself (all three planes of synthetic
 * pictures, every block size / transform depth / skip / reference / mode combination) by tests/test_oracle_vs_ref.py::test_deblocking_edges_of_a_frame through
 * oracle/ref_shim.c, which records what the reference's frame loop hands to the sixteen edge filters.
 * ===================================================================================================================================================== */

/* AV1 block sizes in the order of the BlockSize enum (Common/Codec/EbDefinitions.h): luma width / height */
static const uint8_t k_bs_w[22] = {4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 128, 4, 16, 8, 32, 16, 64};
static const uint8_t k_bs_h[22] = {4, 8, 4, 8, 16, 8, 16, 32, 16, 32, 64, 32, 64, 128, 64, 128, 16, 4, 32, 8, 64, 16};
int orc_block_width(int bsize) { return bsize >= 0 && bsize < 22 ? k_bs_w[bsize] : 0; }
int orc_block_height(int bsize) { return bsize >= 0 && bsize < 22 ? k_bs_h[bsize] : 0; }

/* tx_depth_to_tx_size[depth][bsize] (Common/Codec/EbDefinitions.h:636-657) as transform dimensions.  Depth 0 is the largest transform of the block (64 at most);
 * the deeper entries follow the table's own pattern, quirks included: a block with a 128 side stays 64x64, blocks with a 4 side that is not 4:1 stay whole,
 * 8x8 goes 8x8 -> 4x4 -> 8x8. */
void orc_tx_dims_for_depth(int bsize, int depth, int *tw, int *th) {
    const int w = k_bs_w[bsize], h = k_bs_h[bsize], lo = w < h ? w : h, hi = w < h ? h : w;
    if (hi == 128) { *tw = *th = 64; return; }
    if (depth == 0 || (lo == 4 && hi <= 8)) { *tw = w; *th = h; return; }
    if (hi == lo) {                               /* square */
        const int s = depth == 1 ? lo / 2 : (lo == 8 ? 8 : lo / 4);
        *tw = *th = s;
    } else if (hi == 2 * lo) {                    /* 2:1 -> the square of the short side, then half of it */
        *tw = *th = depth == 1 ? lo : lo / 2;
    } else {                                      /* 4:1 -> 2:1, then the square of the short side */
        if (depth == 1) { *tw = w > h ? w / 2 : w; *th = h > w ? h / 2 : h; }
        else *tw = *th = lo;
    }
}
/* av1_get_max_uv_txsize(bsize, 1, 1) (4:2:0): the chroma block is half the luma block with a floor of 4 samples, its transform is capped at 32
 * (av1_get_adjusted_tx_size, EbDefinitions.h:677-686) */
void orc_uv_tx_dims(int bsize, int *tw, int *th) {
    int w = k_bs_w[bsize] / 2, h = k_bs_h[bsize] / 2;
    if (w < 4) w = 4;
    if (h < 4) h = 4;
    *tw = w > 32 ? 32 : w; *th = h > 32 ? 32 : h;
}
static int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

/* svt_av1_loop_filter_frame_init without segmentation: lvl[plane][dir][ref 0..7][mode delta 0..1].  lf = {filter_level[0], filter_level[1], filter_level_u, filter_level_v,
 * sharpness (unused here), mode_ref_delta_enabled, ref_deltas[8], mode_deltas[2]}.  Planes the init skips (zero frame level) keep zeros: the frame loop skips them too. */
void orc_dlf_level_table(const int32_t *lf, uint8_t lvl[3][2][8][2]) {
    memset(lvl, 0, 3 * 2 * 8 * 2);
    const int filt[3] = {lf[0], lf[2], lf[3]}, filt_r[3] = {lf[1], lf[2], lf[3]};
    for (int plane = 0; plane < 3; plane++) {
        if (plane == 0 && !filt[0] && !filt_r[0]) break;
        if (plane > 0 && !filt[plane]) continue;
        for (int dir = 0; dir < 2; dir++) {
            const int seg = dir == 0 ? filt[plane] : filt_r[plane];
            if (!lf[5]) { memset(lvl[plane][dir], seg, 16); continue; }
            const int scale = 1 << (seg >> 5);
            lvl[plane][dir][0][0] = (uint8_t)clampi(seg + lf[6] * scale, 0, 63);   /* INTRA_FRAME has one entry; [0][1] is never read */
            for (int ref = 1; ref < 8; ref++)
                for (int m = 0; m < 2; m++) lvl[plane][dir][ref][m] = (uint8_t)clampi(seg + lf[6 + ref] * scale + lf[14 + m] * scale, 0, 63);
        }
    }
}

/* The per-4x4 summary set_lpf_parameters works from (the product's SvtHipDlfModeInfo layout, 13 bytes per unit: tx_w_log2, tx_h_log2, uv_tx_w_log2, uv_tx_h_log2,
 * bw_log2, bh_log2, skip_inter, level[3][2]) out of the reference's mode-info fields per unit: sb_type, tx_depth, ref_frame[0], skip, prediction mode.
 * get_transform_size: inter blocks use the largest transform unless they carry coefficients, intra blocks the depth's; mode_lf_lut: 0 for intra modes, GLOBALMV
 * and GLOBAL_GLOBALMV, 1 for the other inter modes (Common/Codec/EbDeblockingCommon.h:66-70). */
void orc_dlf_mode_info_summary(int n_units, const uint8_t *sb_type, const uint8_t *tx_depth, const uint8_t *ref_frame0, const uint8_t *skip, const uint8_t *mode,
                               const uint8_t lvl[3][2][8][2], uint8_t *out) {
    for (int i = 0; i < n_units; i++) {
        const int bs = sb_type[i], inter = ref_frame0[i] > 0;   /* is_inter_block_no_intrabc: ref_frame[0] > INTRA_FRAME */
        int tw, th, uw, uh;
        orc_tx_dims_for_depth(bs, (inter && skip[i]) ? 0 : tx_depth[i], &tw, &th);
        orc_uv_tx_dims(bs, &uw, &uh);
        const int m = mode[i] == 25 ? 0 : mode[i];   /* INTRA_MODE_4x4 counts as DC_PRED */
        const int delta = m < 13 ? 0 : (m == 15 || m == 23 ? 0 : 1);
        uint8_t *o = out + 13 * (size_t)i;
        o[0] = (uint8_t)ilog2(tw); o[1] = (uint8_t)ilog2(th); o[2] = (uint8_t)ilog2(uw); o[3] = (uint8_t)ilog2(uh);
        o[4] = (uint8_t)ilog2(k_bs_w[bs]); o[5] = (uint8_t)ilog2(k_bs_h[bs]); o[6] = (uint8_t)(skip[i] && inter);
        for (int p = 0; p < 3; p++)
            for (int d = 0; d < 2; d++) o[7 + 2 * p + d] = lvl[p][d][ref_frame0[i]][delta];
    }
}

/* units of a plane the frame loop visits along one axis (svt_av1_filter_block_plane_vert :340-366): whole superblocks, except that the last superblock row / column of
 * a coded size that is not a multiple of the superblock size ends at the unpadded extent */
int orc_dlf_filtered_units(int coded_luma, int pad, int sb_size, int ss) {
    const int full = ((coded_luma >> ss) + 3) >> 2, n_sb = (coded_luma + sb_size - 1) / sb_size;
    int units = 0;
    for (int s = 0; s < n_sb; s++) {
        int range = (sb_size >> 2) >> ss;                                                   /* MAX_MIB_SIZE / SB64_MIB_SIZE >> scale */
        if (s * sb_size == coded_luma / sb_size * sb_size) range = ((((coded_luma - pad) % sb_size) >> ss) + 3) >> 2;
        units = ((s * sb_size) >> ss >> 2) + range;                                          /* where this superblock's loop ends */
    }
    return units < full ? units : full;
}

/* set_lpf_parameters for every 4x4 unit of one plane, both directions: edges_v / edges_h [ceil(ph / 4)][ceil(pw / 4)] = level << 8 | filter length for the edge on the
 * left / top of the unit; units at or beyond filt_units_w / filt_units_h are not visited (0).  mi = the 13-byte summaries, [mi_rows][mi_cols] luma units. */
void orc_dlf_build_edges(const uint8_t *mi, int mi_cols, int mi_rows, int plane, int ss_x, int ss_y, int pw, int ph, int filt_units_w, int filt_units_h,
                         uint16_t *edges_v, uint16_t *edges_h) {
    const int uw = (pw + 3) >> 2, uh = (ph + 3) >> 2;
    for (int dir = 0; dir < 2; dir++)
        for (int uy = 0; uy < uh; uy++)
            for (int ux = 0; ux < uw; ux++) {
                uint16_t *out = (dir ? edges_h : edges_v) + (size_t)uy * uw + ux;
                *out = 0;
                const int x = 4 * ux, y = 4 * uy;
                if (ux >= filt_units_w || uy >= filt_units_h || x >= pw || y >= ph) continue;   /* (:185-189) outside the plane: TX_4X4, no filter */
                const int mi_row = ss_y | ((y << ss_y) >> 2), mi_col = ss_x | ((x << ss_x) >> 2);   /* chroma: the bottom / right unit of the 8x8 (:196-197) */
                if (mi_row >= mi_rows || mi_col >= mi_cols) continue;   /* never the case for coded sizes that are multiples of 8 */
                const uint8_t *cur = mi + 13 * ((size_t)mi_row * mi_cols + mi_col);
                const int ts = cur[(plane ? 2 : 0) + dir];   /* log2 of the transform's extent across the edge */
                const int coord = dir ? y : x;
                if (coord & ((1 << ts) - 1)) continue;       /* not a transform edge (:214-219) */
                if (!coord) continue;                        /* picture border: nothing to filter against (:245) */
                const uint8_t *prv = mi + 13 * ((size_t)(dir ? mi_row - (1 << ss_y) : mi_row) * mi_cols + (dir ? mi_col : mi_col - (1 << ss_x)));
                const int pts = prv[(plane ? 2 : 0) + dir];
                const int cl = cur[7 + 2 * plane + dir], pl = prv[7 + 2 * plane + dir];
                /* get_plane_block_size: the prediction block in the plane's samples, never below 4 */
                int bdim = cur[4 + dir] - (plane ? (dir ? ss_y : ss_x) : 0);
                if (bdim < 2) bdim = 2;
                const int pu_edge = !(coord & ((1 << bdim) - 1));
                if (!(cl || pl) || (prv[6] && cur[6] && !pu_edge)) continue;   /* both sides skipped inter blocks: only prediction edges (:281-283) */
                const int mts = ts < pts ? ts : pts;
                const int len = mts <= 2 ? 4 : (mts == 3 ? (plane ? 6 : 8) : (plane ? 6 : 14));
                *out = (uint16_t)(((cl ? cl : pl) << 8) | len);
            }
}
