// svt_hip_host.cpp — the HOST-side logic of libsvtav1_hip.so that involves no device work: mode-info -> edge descriptors
// (set_lpf_parameters), the ME search-window clamp, the control flow of the deblocking filter-level search.  No HIP header is
// included here on purpose: the same translation unit is compiled into the CPU test double of the library (test
// infrastructure) so that end-to-end encodes on a box without a GPU exercise exactly this code against the reference encoder.
#include <stdint.h>
#include "../../include/svt_hip.h"
#include "svt_hip_host.h"

extern "C" {

/* ------------------------------------------------------------------------------------------- ME */
// Host-side restatement of the search-window clamp (EbMotionEstimation.c:1945-2066, unrestricted-MV
// branch; int16 arithmetic with int intermediates, statements in the reference's order).
SvtHipSbSearch svt_hip_me_search_window(int sb_origin_x, int sb_origin_y, int x_center, int y_center, int sa_width,
                                        int sa_height, int pic_width, int pic_height) {
    const int16_t pad = 63;
    const int16_t ox = (int16_t)sb_origin_x, oy = (int16_t)sb_origin_y, pw = (int16_t)pic_width, ph = (int16_t)pic_height;
    int16_t w = (int16_t)sa_width, h = (int16_t)sa_height;
    int16_t xo = (int16_t)(x_center - (w >> 1)), yo = (int16_t)(y_center - (h >> 1));
    xo = (int16_t)((ox + xo < -pad) ? -pad - ox : xo);
    w  = (int16_t)((ox + xo < -pad) ? w - (-pad - (ox + xo)) : w);
    xo = (int16_t)((ox + xo > pw - 1) ? xo - ((ox + xo) - (pw - 1)) : xo);
    if (ox + xo + w > pw) { const int v = w - ((ox + xo + w) - pw); w = (int16_t)(v > 1 ? v : 1); }
    w  = (int16_t)((w < 8) ? w : (w & ~0x07));
    yo = (int16_t)((oy + yo < -pad) ? -pad - oy : yo);
    h  = (int16_t)((oy + yo < -pad) ? h - (-pad - (oy + yo)) : h);
    yo = (int16_t)((oy + yo > ph - 1) ? yo - ((oy + yo) - (ph - 1)) : yo);
    if (oy + yo + h > ph) { const int v = h - ((oy + yo + h) - ph); h = (int16_t)(v > 1 ? v : 1); }
    SvtHipSbSearch s;
    s.sb_x = sb_origin_x; s.sb_y = sb_origin_y; s.x_origin = xo; s.y_origin = yo; s.width = w; s.height = h;
    return s;
}

/* ------------------------------------------------------------------------------- deblocking */
// Restatement of set_lpf_parameters (Encoder/Codec/EbDeblockingFilter.c:168-319) over a plain grid.
// svt_av1_filter_block_plane_vert / _horz (Encoder/Codec/EbDeblockingFilter.c:338-367, :479-508): the 4x4 units of a plane the reference's loops
// visit along one axis.  Every superblock covers sb_size >> ss samples, except that the LAST superblock row / column of a coded size that is
// not a multiple of the superblock size stops at the unpadded source extent (rounded up to 4 samples).  A coded size that IS a multiple of the
// superblock size is filtered completely even when it contains padding (the reference's `mi_row ==` test then never fires).
int svt_hip_dlf_filtered_units(int coded_luma, int pad, int sb_size, int ss) {
    if (coded_luma <= 0 || pad < 0 || pad >= coded_luma || (sb_size != 64 && sb_size != 128) || ss < 0 || ss > 1) return -1;
    const int full = ((coded_luma >> ss) + 3) >> 2;
    if (coded_luma % sb_size == 0) return full;
    const int last = coded_luma / sb_size * sb_size, rem = (coded_luma - pad) % sb_size;
    const int units = ((last >> ss) >> 2) + (((rem >> ss) + 3) >> 2);
    return units < full ? units : full;
}

int svt_hip_dlf_build_edges(const SvtHipDlfModeInfo* mi, int mi_cols, int mi_rows, int plane, int ss_x, int ss_y, int plane_w,
                            int plane_h, uint16_t* edges_v, uint16_t* edges_h) {
    return svt_hip_dlf_build_edges_crop(mi, mi_cols, mi_rows, plane, ss_x, ss_y, plane_w, plane_h, (plane_w + 3) >> 2, (plane_h + 3) >> 2, edges_v, edges_h);
}

int svt_hip_dlf_build_edges_crop(const SvtHipDlfModeInfo* mi, int mi_cols, int mi_rows, int plane, int ss_x, int ss_y, int plane_w,
                                 int plane_h, int filt_units_w, int filt_units_h, uint16_t* edges_v, uint16_t* edges_h) {
    if (!mi || mi_cols <= 0 || mi_rows <= 0 || plane < 0 || plane > 2 || !edges_v || !edges_h || filt_units_w < 0 || filt_units_h < 0) return SVT_HIP_ERR_BAD_ARG;
    const int uw = (plane_w + 3) >> 2, uh = (plane_h + 3) >> 2;
    for (int dir = 0; dir < 2; dir++) {
        uint16_t* out = dir == 0 ? edges_v : edges_h;
        for (int uy = 0; uy < uh; uy++)
            for (int ux = 0; ux < uw; ux++) {
                uint16_t v = 0;
                if (ux >= filt_units_w || uy >= filt_units_h) { out[uy * uw + ux] = 0; continue; }   // outside the loops' range: never visited
                const int x = 4 * ux, y = 4 * uy;
                // chroma maps to the bottom/right mi of the co-located 8x8 (:196-197)
                int mr = ss_y | ((y << ss_y) >> 2), mc = ss_x | ((x << ss_x) >> 2);
                if (mr >= mi_rows) mr = mi_rows - 1;
                if (mc >= mi_cols) mc = mi_cols - 1;
                const SvtHipDlfModeInfo& cur = mi[mr * mi_cols + mc];
                const int ts = plane == 0 ? (dir == 0 ? cur.tx_w_log2 : cur.tx_h_log2) : (dir == 0 ? cur.uv_tx_w_log2 : cur.uv_tx_h_log2);
                const int coord = dir == 0 ? x : y;
                if (!(coord & ((1 << ts) - 1)) && coord) {
                    const int pr = dir == 0 ? mr : mr - (1 << ss_y), pc = dir == 0 ? mc - (1 << ss_x) : mc;
                    if (pr >= 0 && pc >= 0) {
                        const SvtHipDlfModeInfo& prv = mi[pr * mi_cols + pc];
                        const int pts = plane == 0 ? (dir == 0 ? prv.tx_w_log2 : prv.tx_h_log2) : (dir == 0 ? prv.uv_tx_w_log2 : prv.uv_tx_h_log2);
                        const int cl = cur.level[plane][dir], pl = prv.level[plane][dir];
                        int bdim = dir == 0 ? cur.bw_log2 - (plane ? ss_x : 0) : cur.bh_log2 - (plane ? ss_y : 0);
                        if (bdim < 2) bdim = 2;
                        const bool pu_edge = !(coord & ((1 << bdim) - 1));
                        if ((cl || pl) && (!prv.skip_inter || !cur.skip_inter || pu_edge)) {
                            const int mts = ts < pts ? ts : pts;
                            const int len = mts <= 2 ? 4 : (mts == 3 ? (plane ? 6 : 8) : (plane ? 6 : 14));
                            v = (uint16_t)(((cl ? cl : pl) << 8) | len);
                        }
                    }
                }
                out[uy * uw + ux] = v;
            }
    }
    return SVT_HIP_OK;
}


// search_filter_level (Encoder/Codec/EbDeblockingFilter.c:1026-1187): the probe sequence, the integer bias rule and the
// mode <= 2 single refinement, as a PLAN over the errors known so far: the walk is replayed on ss_err[] (-1 = not measured) until it needs a level that has not been
// measured; the one or two levels that iteration needs (filt_low and filt_high are both measured before either is compared when the direction is open) come back in
// need[], the caller measures them (try_filter_frame, :966-1024), stores them in ss_err[] and asks again.  Return value: how many levels are needed; 0 = the walk is
// over, *best_level / *best_err_out are its result.  Measuring the two levels of an iteration together halves the round trips of a device search, and the three planes'
// searches of a picture can be advanced in lockstep (svt_hip_dlf_search_levels_picture_dev).
int svt_hip_dlf_search_plan(const SvtHipDlfSearch* p, const int64_t* ss_err, int need[2], int* best_level, int64_t* best_err_out) {
    if (!p || !ss_err || !need || !best_level) return SVT_HIP_ERR_BAD_ARG;
    const int kMaxLoopFilter = 63;   // MAX_LOOP_FILTER
    int filt_direction = 0;
    int filt_mid = p->start_level < 0 ? 0 : (p->start_level > kMaxLoopFilter ? kMaxLoopFilter : p->start_level);
    int filter_step = filt_mid < 16 ? 4 : filt_mid / 4;
    if (ss_err[filt_mid] < 0) { need[0] = filt_mid; return 1; }
    int64_t best_err = ss_err[filt_mid];
    int filt_best = filt_mid;
    const bool single = p->loop_filter_mode <= 2;
    if (single) filter_step = 2;
    while (filter_step > 0) {
        const int filt_high = filt_mid + filter_step > kMaxLoopFilter ? kMaxLoopFilter : filt_mid + filter_step;
        const int filt_low = filt_mid - filter_step < 0 ? 0 : filt_mid - filter_step;
        int64_t bias = (best_err >> (15 - (filt_mid / 8))) * filter_step;   // bias against raising the level
        if (!p->tx_mode_only_4x4) bias >>= 1;
        const bool want_low = filt_direction <= 0 && filt_low != filt_mid, want_high = filt_direction >= 0 && filt_high != filt_mid;
        int n = 0;
        if (want_low && ss_err[filt_low] < 0) need[n++] = filt_low;
        if (want_high && ss_err[filt_high] < 0 && !(n && need[0] == filt_high)) need[n++] = filt_high;
        if (n) return n;
        if (want_low && ss_err[filt_low] < best_err + bias) {
            if (ss_err[filt_low] < best_err) best_err = ss_err[filt_low];
            filt_best = filt_low;
        }
        if (want_high && ss_err[filt_high] < best_err - bias) {
            if (!single) best_err = ss_err[filt_high];   // the mode <= 2 branch does not update best_err (:1121-1122)
            filt_best = filt_high;
        }
        if (single) break;
        if (filt_best == filt_mid) {
            filter_step /= 2;
            filt_direction = 0;
        } else {
            filt_direction = filt_best < filt_mid ? -1 : 1;
            filt_mid = filt_best;
        }
    }
    *best_level = filt_best;
    if (best_err_out) *best_err_out = ss_err[filt_best];
    return 0;
}
// the levels a probe of `lvl` filters with: plane 0 with dir 2 (svt_av1_pick_filter_level :1281) and chroma use the probed level in both directions
void svt_hip_dlf_search_probe_levels(const SvtHipDlfSearch* p, int lvl, int* lv_v, int* lv_h) {
    *lv_v = lvl; *lv_h = lvl;
    if (p->plane == 0 && p->dir == 0) *lv_h = p->other_level;
    if (p->plane == 0 && p->dir == 1) *lv_v = p->other_level;
}
// the whole walk with a callback per probe (one probe at a time: svt_hip_dlf_search_level_dev, the CPU test double)
int svt_hip_dlf_search_levels_host(const SvtHipDlfSearch* p, SvtHipTryLevelFn try_fn, void* user, int* best_level, int64_t* best_err_out) {
    if (!p || !try_fn || !best_level) return SVT_HIP_ERR_BAD_ARG;
    int64_t ss_err[64];
    for (int i = 0; i < 64; i++) ss_err[i] = -1;
    for (;;) {
        int need[2];
        const int n = svt_hip_dlf_search_plan(p, ss_err, need, best_level, best_err_out);
        if (n <= 0) return n < 0 ? n : SVT_HIP_OK;
        for (int i = 0; i < n; i++) {
            int lv_v, lv_h;
            svt_hip_dlf_search_probe_levels(p, need[i], &lv_v, &lv_h);
            const int64_t e = try_fn(user, lv_v, lv_h);
            if (e < 0) return SVT_HIP_ERR_RUNTIME;
            ss_err[need[i]] = e;
        }
    }
}

/* ------------------------------------------------------------------------------- temporal filter */
double svt_hip_tf_noise_sigma(int64_t sum, int64_t num) {   // EbTemporalFiltering.c:2442-2447
    if (num < 16) return -1.0;
    return (double)sum / (6 * num) * 1.25331413732;
}


}  // extern "C"
