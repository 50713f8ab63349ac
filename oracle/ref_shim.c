/*
 * ref_shim.c — tiny accessor file compiled INTO oracle/_ref/libsvtav1_ref.so (never into the
 * product, never on the GPU box): exposes constant tables that the reference keeps `static` in its
 * headers (scan orders) or inside encoder objects (quantizer tables), so that tests and
 * tests/golden/make_golden.py can feed the oracle / HIP path with the reference's real tables.
 * It includes the reference headers from /root/reference at build time only.
 */
#include <stdint.h>
#include <string.h>
#include "EbDefinitions.h"
#include "EbCoefficients.h"
#include "EbPictureControlSet.h"
#include "EbFullLoop.h"
#include "EbInvTransforms.h"

void svt_av1_build_quantizer(AomBitDepth bit_depth, int32_t y_dc_delta_q, int32_t u_dc_delta_q, int32_t u_ac_delta_q,
                             int32_t v_dc_delta_q, int32_t v_ac_delta_q, Quants *const quants, Dequants *const deq);

int ref_shim_scan(int tx_size, int tx_type, const int16_t **scan, const int16_t **iscan) {
    *scan  = av1_scan_orders[tx_size][tx_type].scan;
    *iscan = av1_scan_orders[tx_size][tx_type].iscan;
    return av1_get_max_eob((TxSize)tx_size);
}
int ref_shim_tx_scale(int tx_size) { return av1_get_tx_scale_tab[tx_size]; }
void ref_shim_flip(int tx_type, int *ud, int *lr) { get_flip_cfg((TxType)tx_type, ud, lr); }

/* out[k][0..1] = {dc, ac} of: 0 zbin, 1 round, 2 quant, 3 quant_shift, 4 dequant, 5 round_fp, 6 quant_fp
 * for plane 0 (Y), 1 (U), 2 (V); tables as built at encoder init with zero delta-q
 * (Encoder/Codec/EbModeDecisionConfigurationProcess.c:205-287). */
void ref_shim_qparams(int bd, int qindex, int plane, int16_t out[7][2]) {
    static Quants   q[2];
    static Dequants d[2];
    static int      ready[2] = {0, 0};
    const int       b        = bd == 8 ? 0 : 1;
    if (!ready[b]) {
        svt_av1_build_quantizer(bd == 8 ? AOM_BITS_8 : AOM_BITS_10, 0, 0, 0, 0, 0, &q[b], &d[b]);
        ready[b] = 1;
    }
    const int16_t *src[7];
    if (plane == 0) {
        src[0] = q[b].y_zbin[qindex]; src[1] = q[b].y_round[qindex]; src[2] = q[b].y_quant[qindex];
        src[3] = q[b].y_quant_shift[qindex]; src[4] = d[b].y_dequant_qtx[qindex];
        src[5] = q[b].y_round_fp[qindex]; src[6] = q[b].y_quant_fp[qindex];
    } else if (plane == 1) {
        src[0] = q[b].u_zbin[qindex]; src[1] = q[b].u_round[qindex]; src[2] = q[b].u_quant[qindex];
        src[3] = q[b].u_quant_shift[qindex]; src[4] = d[b].u_dequant_qtx[qindex];
        src[5] = q[b].u_round_fp[qindex]; src[6] = q[b].u_quant_fp[qindex];
    } else {
        src[0] = q[b].v_zbin[qindex]; src[1] = q[b].v_round[qindex]; src[2] = q[b].v_quant[qindex];
        src[3] = q[b].v_quant_shift[qindex]; src[4] = d[b].v_dequant_qtx[qindex];
        src[5] = q[b].v_round_fp[qindex]; src[6] = q[b].v_quant_fp[qindex];
    }
    for (int k = 0; k < 7; k++) { out[k][0] = src[k][0]; out[k][1] = src[k][1]; }
}

/* ------------------------------------------------------------------------------------------------
 * Loop-restoration frame level: drive the reference's own unit iterator, stripe
 * boundary save / setup / restore and stripe filters for ONE plane with plain arguments, so that the
 * oracle's restatement (oracle/sgr_oracle.c: orc_rest_unit_limits, orc_sgr_apply_plane) can be pinned.
 * Nothing here restates an algorithm: it only builds the structs the reference functions take. */
#include <stdlib.h>
#include "Av1Common.h"
#include "EbRestoration.h"
#include "common_dsp_rtcd.h"

void save_tile_row_boundary_lines(uint8_t *src, int32_t src_stride, int32_t src_width, int32_t src_height, int32_t use_highbd,
                                  int32_t plane, Av1Common *cm, int32_t after_cdef, RestorationStripeBoundaries *boundaries);
void av1_foreach_rest_unit_in_frame(Av1Common *cm, int32_t plane, RestTileStartVisitor on_tile, RestUnitVisitor on_rest_unit, void *priv);
void svt_av1_loop_restoration_filter_unit(uint8_t need_bounadaries, const RestorationTileLimits *limits, const RestorationUnitInfo *rui,
                                          const RestorationStripeBoundaries *rsb, RestorationLineBuffers *rlbs,
                                          const Av1PixelRect *tile_rect, int32_t tile_stripe0, int32_t ss_x, int32_t ss_y,
                                          int32_t highbd, int32_t bit_depth, uint8_t *data8, int32_t stride, uint8_t *dst8,
                                          int32_t dst_stride, int32_t *tmpbuf, int32_t optimized_lr);

int ref_shim_rtcd_ready = 0;   /* also set by refb_setup (ref_bench.c), which installs the host's SIMD kernels instead of the C ones */
static void shim_rtcd(void) {
    if (!ref_shim_rtcd_ready) { setup_common_rtcd_internal(0); ref_shim_rtcd_ready = 1; }
}
static int shim_units(int unit_size, int size) { int n = (size + (unit_size >> 1)) / unit_size; return n > 1 ? n : 1; }  /* count_units_in_tile */

typedef struct {
    int32_t *limits;   /* [unit][4] = h_start, h_end, v_start, v_end */
    int      n;
} ShimLimitsCtx;
static void shim_limits_visitor(const RestorationTileLimits *limits, const Av1PixelRect *tile_rect, int32_t unit_idx, void *priv) {
    (void)tile_rect;
    ShimLimitsCtx *c = (ShimLimitsCtx *)priv;
    int32_t *o = c->limits + 4 * unit_idx;
    o[0] = limits->h_start; o[1] = limits->h_end; o[2] = limits->v_start; o[3] = limits->v_end;
    if (unit_idx + 1 > c->n) c->n = unit_idx + 1;
}
static Av1Common *shim_cm(int frame_w, int frame_h, int bd, int highbd, int plane, int unit_size) {
    Av1Common *cm = (Av1Common *)calloc(1, sizeof(Av1Common));
    cm->frm_size.frame_width = (uint16_t)frame_w; cm->frm_size.frame_height = (uint16_t)frame_h;
    cm->frm_size.superres_upscaled_width = (uint16_t)frame_w; cm->frm_size.superres_upscaled_height = (uint16_t)frame_h;
    cm->frm_size.superres_denominator = 8;
    cm->subsampling_x = 1; cm->subsampling_y = 1; cm->bit_depth = bd; cm->use_highbitdepth = highbd;
    cm->mi_rows = (frame_h + 3) >> 2; cm->mi_cols = (frame_w + 3) >> 2;
    const int ss = plane > 0;
    const int pw = (frame_w + ss) >> ss, ph = (frame_h + ss) >> ss;
    RestorationInfo *rsi = &cm->rst_info[plane];
    rsi->restoration_unit_size = unit_size;
    rsi->frame_restoration_type = RESTORE_SGRPROJ;
    rsi->horz_units_per_tile = shim_units(unit_size, pw);
    rsi->vert_units_per_tile = shim_units(unit_size, ph);
    rsi->units_per_tile = rsi->horz_units_per_tile * rsi->vert_units_per_tile;
    return cm;
}
/* limits of every restoration unit of a plane as av1_foreach_rest_unit_in_frame hands them out; returns the unit count */
int ref_shim_rest_unit_limits(int frame_w, int frame_h, int plane, int unit_size, int32_t *limits) {
    Av1Common *cm = shim_cm(frame_w, frame_h, 8, 0, plane, unit_size);
    ShimLimitsCtx c = {limits, 0};
    av1_foreach_rest_unit_in_frame(cm, plane, NULL, shim_limits_visitor, &c);
    free(cm);
    return c.n;
}

typedef struct {
    Av1Common *cm; RestorationInfo *rsi; RestorationLineBuffers *rlbs; int plane, highbd, bd;
    uint8_t *data8, *dst8; int stride, dst_stride; int32_t *tmpbuf;
} ShimFilterCtx;
static void shim_filter_visitor(const RestorationTileLimits *limits, const Av1PixelRect *tile_rect, int32_t unit_idx, void *priv) {
    ShimFilterCtx *c = (ShimFilterCtx *)priv;   /* filter_frame_on_unit, EbRestoration.c:1269-1291 */
    svt_av1_loop_restoration_filter_unit(1, limits, &c->rsi->unit_info[unit_idx], &c->rsi->boundaries, c->rlbs, tile_rect, 0, c->plane > 0,
                                         c->plane > 0, c->highbd, c->bd, c->data8, c->stride, c->dst8, c->dst_stride, c->tmpbuf, 0);
}
/* svt_av1_loop_restoration_save_boundary_lines (deblocked frame, then CDEF frame) + svt_av1_loop_restoration_filter_frame for one
 * plane.  dbl / cdef / dst point at pixel (0,0); cdef needs a 3-pixel writable border (svt_extend_frame fills it);
 * strides in pixels; unit_ep[u] = parameter set or 255 for RESTORE_NONE; unit_xqd[u][2]. */
int ref_shim_lr_apply_plane_ex(int plane, int bd, int highbd, int frame_w, int frame_h, void *dbl, int dbl_stride, void *cdef, int stride,
                               void *dst, int dst_stride, int unit_size, const uint8_t *unit_ep, const int32_t *unit_xqd, const int16_t *unit_wiener);
int ref_shim_lr_apply_plane(int plane, int bd, int highbd, int frame_w, int frame_h, void *dbl, int dbl_stride, void *cdef, int stride,
                            void *dst, int dst_stride, int unit_size, const uint8_t *unit_ep, const int32_t *unit_xqd) {
    return ref_shim_lr_apply_plane_ex(plane, bd, highbd, frame_w, frame_h, dbl, dbl_stride, cdef, stride, dst, dst_stride, unit_size, unit_ep, unit_xqd, NULL);
}
/* unit_ep[u] == 254: RESTORE_WIENER with taps unit_wiener[u][0][8] (vertical) / [1][8] (horizontal) */
int ref_shim_lr_apply_plane_ex(int plane, int bd, int highbd, int frame_w, int frame_h, void *dbl, int dbl_stride, void *cdef, int stride,
                               void *dst, int dst_stride, int unit_size, const uint8_t *unit_ep, const int32_t *unit_xqd, const int16_t *unit_wiener) {
    shim_rtcd();
    Av1Common *cm = shim_cm(frame_w, frame_h, bd, highbd, plane, unit_size);
    const int ss = plane > 0;
    const int pw = (frame_w + ss) >> ss, ph = (frame_h + ss) >> ss;
    RestorationInfo *rsi = &cm->rst_info[plane];
    rsi->unit_info = (RestorationUnitInfo *)calloc(rsi->units_per_tile, sizeof(RestorationUnitInfo));
    for (int u = 0; u < rsi->units_per_tile; u++) {
        rsi->unit_info[u].restoration_type = unit_ep[u] == 254 && unit_wiener ? RESTORE_WIENER : (unit_ep[u] > 15 ? RESTORE_NONE : RESTORE_SGRPROJ);
        if (unit_ep[u] == 254 && unit_wiener) {
            memcpy(rsi->unit_info[u].wiener_info.vfilter, unit_wiener + 16 * u, 16);
            memcpy(rsi->unit_info[u].wiener_info.hfilter, unit_wiener + 16 * u + 8, 16);
        }
        rsi->unit_info[u].sgrproj_info.ep = unit_ep[u] > 15 ? 0 : unit_ep[u];
        rsi->unit_info[u].sgrproj_info.xqd[0] = unit_xqd[2 * u]; rsi->unit_info[u].sgrproj_info.xqd[1] = unit_xqd[2 * u + 1];
    }
    /* svt_av1_alloc_restoration_buffers (EbRestoration.c:1872-1930): stripe count from the luma height, 32-aligned line stride */
    const int num_stripes = (RESTORATION_UNIT_OFFSET + (cm->mi_rows << 2) + 63) / 64;
    const int bstride = (pw + 2 * RESTORATION_EXTRA_HORZ + 31) & ~31;
    const size_t bsize = (size_t)num_stripes * bstride * RESTORATION_CTX_VERT << highbd;
    rsi->boundaries.stripe_boundary_above = (uint8_t *)calloc(bsize + 64, 1);
    rsi->boundaries.stripe_boundary_below = (uint8_t *)calloc(bsize + 64, 1);
    rsi->boundaries.stripe_boundary_stride = bstride;
    rsi->boundaries.stripe_boundary_size = (int32_t)bsize;
    uint8_t *dbl8 = highbd ? CONVERT_TO_BYTEPTR(dbl) : (uint8_t *)dbl;
    uint8_t *cdef8 = highbd ? CONVERT_TO_BYTEPTR(cdef) : (uint8_t *)cdef;
    uint8_t *dst8 = highbd ? CONVERT_TO_BYTEPTR(dst) : (uint8_t *)dst;
    save_tile_row_boundary_lines(highbd ? (uint8_t *)dbl : dbl8, dbl_stride, pw, ph, highbd, plane, cm, 0, &rsi->boundaries);
    save_tile_row_boundary_lines(highbd ? (uint8_t *)cdef : cdef8, stride, pw, ph, highbd, plane, cm, 1, &rsi->boundaries);
    svt_extend_frame(cdef8, pw, ph, stride, RESTORATION_BORDER, RESTORATION_BORDER, highbd);
    RestorationLineBuffers *rlbs = (RestorationLineBuffers *)calloc(1, sizeof(RestorationLineBuffers));
    int32_t *tmpbuf = NULL;
    if (posix_memalign((void **)&tmpbuf, 32, RESTORATION_TMPBUF_SIZE)) return -1;
    ShimFilterCtx c = {cm, rsi, rlbs, plane, highbd, bd, cdef8, dst8, stride, dst_stride, tmpbuf};
    av1_foreach_rest_unit_in_frame(cm, plane, NULL, shim_filter_visitor, &c);
    free(tmpbuf); free(rlbs); free(rsi->boundaries.stripe_boundary_above); free(rsi->boundaries.stripe_boundary_below);
    free(rsi->unit_info); free(cm);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Temporal filtering: drive the reference's plane-wise filter with a MeContext built from plain arrays. */
#include "EbMotionEstimationContext.h"
#include "EbTemporalFiltering.h"
#include "EbBitstreamUnit.h"
void ref_shim_tf_planewise(const int16_t *mv16_x, const int16_t *mv16_y, const uint64_t *err16, const int16_t *mv32_x, const int16_t *mv32_y,
                           const uint64_t *err32, const int32_t *split, int block_row, int block_col, int tf_chroma, int min_frame_size,
                           int highbd, int bd, const void *y_src, int y_src_stride, const void *y_pre, int y_pre_stride, const void *u_src,
                           const void *v_src, int uv_src_stride, const void *u_pre, const void *v_pre, int uv_pre_stride, unsigned bw,
                           unsigned bh, int ss_x, int ss_y, const double *noise_levels, int decay_control, uint32_t *y_accum,
                           uint16_t *y_count, uint32_t *u_accum, uint16_t *u_count, uint32_t *v_accum, uint16_t *v_count) {
    MeContext *c = (MeContext *)calloc(1, sizeof(MeContext));
    for (int i = 0; i < 16; i++) { c->tf_16x16_mv_x[i] = mv16_x[i]; c->tf_16x16_mv_y[i] = mv16_y[i]; c->tf_16x16_block_error[i] = err16[i]; }
    for (int i = 0; i < 4; i++) {
        c->tf_32x32_mv_x[i] = mv32_x[i]; c->tf_32x32_mv_y[i] = mv32_y[i]; c->tf_32x32_block_error[i] = err32[i];
        c->tf_32x32_block_split_flag[i] = split[i];
    }
    c->tf_block_row = block_row; c->tf_block_col = block_col; c->tf_chroma = (uint8_t)tf_chroma; c->min_frame_size = (uint16_t)min_frame_size;
    if (!highbd)
        svt_av1_apply_temporal_filter_planewise_c(c, y_src, y_src_stride, y_pre, y_pre_stride, u_src, v_src, uv_src_stride, u_pre, v_pre,
                                                  uv_pre_stride, bw, bh, ss_x, ss_y, noise_levels, decay_control, y_accum, y_count, u_accum,
                                                  u_count, v_accum, v_count);
    else
        svt_av1_apply_temporal_filter_planewise_hbd_c(c, y_src, y_src_stride, y_pre, y_pre_stride, u_src, v_src, uv_src_stride, u_pre, v_pre,
                                                      uv_pre_stride, bw, bh, ss_x, ss_y, noise_levels, decay_control, y_accum, y_count,
                                                      u_accum, u_count, v_accum, v_count, (uint32_t)bd);
    free(c);
}
/* OD_DIVU as get_final_filtered_pixels uses it (EbTemporalFiltering.c:1956-1961) */
uint32_t ref_shim_od_divu(uint32_t x, uint32_t d) { return OD_DIVU(x, d); }
double estimate_noise(const uint8_t *src, uint16_t width, uint16_t height, uint16_t stride_y);
double estimate_noise_highbd(const uint16_t *src, int width, int height, int stride, int bd);
double ref_shim_estimate_noise(const void *src, int highbd, int bd, int width, int height, int stride) {
    return highbd ? estimate_noise_highbd((const uint16_t *)src, width, height, stride, bd)
                  : estimate_noise((const uint8_t *)src, (uint16_t)width, (uint16_t)height, (uint16_t)stride);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Deblocking, frame level: svt_av1_loop_filter_frame (Encoder/Codec/EbDeblockingFilter.c:711) driven on a synthetic picture whose mode-info grid
 * is given per 4x4 unit, with the sixteen edge-filter pointers replaced by recorders — what comes out is, for every 4x4 unit of every plane and both directions, whether
 * the reference's frame loop filters the edge on its left / top, with which filter length and at which level: the outcome of set_lpf_parameters (:168, static),
 * get_transform_size (:134, static), svt_av1_loop_filter_frame_init and the unit ranges of svt_av1_filter_block_plane_vert / _horz.  Builds structs and records calls;
 * restates nothing. */
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbDeblockingFilter.h"
static struct {
    uint8_t *base[3]; int stride[3], uw[3], uh[3];
    const LoopFilterThresh *thr;
    uint16_t *ev[3], *eh[3];
    int bad;
} g_dlf_rec;
static void dlf_record(uint8_t *s, int dir, int len, const uint8_t *limit) {
    for (int p = 0; p < 3; p++) {
        const ptrdiff_t off = s - g_dlf_rec.base[p];
        if (off < 0 || off >= (ptrdiff_t)g_dlf_rec.stride[p] * g_dlf_rec.uh[p] * 4) continue;
        const int y = (int)(off / g_dlf_rec.stride[p]), x = (int)(off % g_dlf_rec.stride[p]);
        const ptrdiff_t lv = (limit - (const uint8_t *)g_dlf_rec.thr) / (ptrdiff_t)sizeof(LoopFilterThresh);   /* limit points into lfthr[level] */
        if ((x & 3) || (y & 3) || x >= 4 * g_dlf_rec.uw[p] || lv < 0 || lv > MAX_LOOP_FILTER) { g_dlf_rec.bad++; return; }
        uint16_t *o = (dir ? g_dlf_rec.eh[p] : g_dlf_rec.ev[p]) + (size_t)(y >> 2) * g_dlf_rec.uw[p] + (x >> 2);
        if (*o) g_dlf_rec.bad++;   /* an edge filtered twice */
        *o = (uint16_t)((lv << 8) | len);
        return;
    }
    g_dlf_rec.bad++;
}
#define DLF_REC(name, dir, len) static void name(uint8_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh) { (void)pitch; (void)blimit; (void)thresh; dlf_record(s, dir, len, limit); }
DLF_REC(rec_v4, 0, 4) DLF_REC(rec_v6, 0, 6) DLF_REC(rec_v8, 0, 8) DLF_REC(rec_v14, 0, 14)
DLF_REC(rec_h4, 1, 4) DLF_REC(rec_h6, 1, 6) DLF_REC(rec_h8, 1, 8) DLF_REC(rec_h14, 1, 14)

/* w x h = coded luma size (multiples of 8), pad_right / pad_bottom = how much of it is padding, sb_size 64 or 128; sb_type / tx_depth / ref_frame0 / skip / mode:
 * [h / 4][w / 4] per luma 4x4 unit; lf = {filter_level[0], filter_level[1], filter_level_u, filter_level_v, sharpness, mode_ref_delta_enabled, ref_deltas[8],
 * mode_deltas[2]}; ev / eh[plane]: [ceil(ph / 4)][ceil(pw / 4)] = level << 8 | length (zeroed here); lvl_out (may be NULL): lf_info.lvl[plane][0][dir][ref][mode] after the
 * reference's init.  Returns the number of inconsistencies the recorders saw (0 = fine), -1 on allocation failure. */
int ref_shim_dlf_frame_edges(int w, int h, int pad_right, int pad_bottom, int sb_size, const uint8_t *sb_type, const uint8_t *tx_depth, const uint8_t *ref_frame0,
                             const uint8_t *skip, const uint8_t *mode, const int32_t *lf, uint16_t *const ev[3], uint16_t *const eh[3], uint8_t *lvl_out) {
    shim_rtcd();
    const int mi_cols = w >> 2, mi_rows = h >> 2;
    PictureControlSet *pcs = (PictureControlSet *)calloc(1, sizeof(PictureControlSet));
    PictureParentControlSet *ppcs = (PictureParentControlSet *)calloc(1, sizeof(PictureParentControlSet));
    SequenceControlSet *scs = (SequenceControlSet *)calloc(1, sizeof(SequenceControlSet));
    EbObjectWrapper *wr = (EbObjectWrapper *)calloc(1, sizeof(EbObjectWrapper));
    ModeInfo *mi = (ModeInfo *)calloc((size_t)mi_cols * mi_rows, sizeof(ModeInfo));
    ModeInfo **grid = (ModeInfo **)calloc((size_t)mi_cols * mi_rows, sizeof(ModeInfo *));
    EbPictureBufferDesc *pic = (EbPictureBufferDesc *)calloc(1, sizeof(EbPictureBufferDesc));
    if (!pcs || !ppcs || !scs || !wr || !mi || !grid || !pic) return -1;
    for (int i = 0; i < mi_cols * mi_rows; i++) {
        grid[i] = &mi[i];
        mi[i].mbmi.block_mi.sb_type = (BlockSize)sb_type[i]; mi[i].mbmi.tx_depth = tx_depth[i]; mi[i].mbmi.block_mi.ref_frame[0] = (MvReferenceFrame)ref_frame0[i];
        mi[i].mbmi.block_mi.skip = skip[i]; mi[i].mbmi.block_mi.mode = (PredictionMode)mode[i];
    }
    wr->object_ptr = scs;
    pcs->parent_pcs_ptr = ppcs; pcs->mi_grid_base = grid; pcs->mi_stride = mi_cols;
    ppcs->scs_wrapper_ptr = wr; ppcs->scs_ptr = scs; ppcs->aligned_width = (uint16_t)w; ppcs->aligned_height = (uint16_t)h;
    scs->sb_size_pix = (uint8_t)sb_size; scs->seq_header.sb_size = sb_size == 128 ? BLOCK_128X128 : BLOCK_64X64;
    scs->max_input_luma_width = (uint16_t)w; scs->max_input_luma_height = (uint16_t)h; scs->max_input_pad_right = (uint16_t)pad_right; scs->max_input_pad_bottom = (uint16_t)pad_bottom;
    scs->static_config.encoder_bit_depth = 8; scs->static_config.is_16bit_pipeline = 0;
    struct LoopFilter *lp = &ppcs->frm_hdr.loop_filter_params;
    lp->filter_level[0] = lf[0]; lp->filter_level[1] = lf[1]; lp->filter_level_u = lf[2]; lp->filter_level_v = lf[3]; lp->sharpness_level = lf[4];
    lp->mode_ref_delta_enabled = (uint8_t)lf[5]; lp->combine_vert_horz_lf = 1;
    for (int i = 0; i < 8; i++) lp->ref_deltas[i] = (int8_t)lf[6 + i];
    for (int i = 0; i < 2; i++) lp->mode_deltas[i] = (int8_t)lf[14 + i];
    svt_av1_loop_filter_init(pcs);   /* the sharpness limits and the bilinear level table, as the encoder does once per picture (EbDlfProcess.c) */
    pic->bit_depth = EB_8BIT; pic->width = pic->max_width = (uint16_t)w; pic->height = pic->max_height = (uint16_t)h;
    pic->stride_y = (uint16_t)w; pic->stride_cb = pic->stride_cr = (uint16_t)(w / 2);
    pic->buffer_y = (uint8_t *)calloc((size_t)w * h, 1); pic->buffer_cb = (uint8_t *)calloc((size_t)w * h / 4, 1); pic->buffer_cr = (uint8_t *)calloc((size_t)w * h / 4, 1);
    memset(&g_dlf_rec, 0, sizeof(g_dlf_rec));
    g_dlf_rec.base[0] = pic->buffer_y; g_dlf_rec.base[1] = pic->buffer_cb; g_dlf_rec.base[2] = pic->buffer_cr;
    for (int p = 0; p < 3; p++) {
        const int pw = w >> (p > 0), ph = h >> (p > 0);
        g_dlf_rec.stride[p] = pw; g_dlf_rec.uw[p] = (pw + 3) >> 2; g_dlf_rec.uh[p] = (ph + 3) >> 2;
        g_dlf_rec.ev[p] = ev[p]; g_dlf_rec.eh[p] = eh[p];
        memset(ev[p], 0, sizeof(uint16_t) * g_dlf_rec.uw[p] * g_dlf_rec.uh[p]); memset(eh[p], 0, sizeof(uint16_t) * g_dlf_rec.uw[p] * g_dlf_rec.uh[p]);
    }
    g_dlf_rec.thr = ppcs->lf_info.lfthr;
    void *saved[8] = {(void *)svt_aom_lpf_vertical_4, (void *)svt_aom_lpf_vertical_6, (void *)svt_aom_lpf_vertical_8, (void *)svt_aom_lpf_vertical_14,
                      (void *)svt_aom_lpf_horizontal_4, (void *)svt_aom_lpf_horizontal_6, (void *)svt_aom_lpf_horizontal_8, (void *)svt_aom_lpf_horizontal_14};
    svt_aom_lpf_vertical_4 = rec_v4; svt_aom_lpf_vertical_6 = rec_v6; svt_aom_lpf_vertical_8 = rec_v8; svt_aom_lpf_vertical_14 = rec_v14;
    svt_aom_lpf_horizontal_4 = rec_h4; svt_aom_lpf_horizontal_6 = rec_h6; svt_aom_lpf_horizontal_8 = rec_h8; svt_aom_lpf_horizontal_14 = rec_h14;
    svt_av1_loop_filter_frame(pic, pcs, 0, 3);
    svt_aom_lpf_vertical_4 = saved[0]; svt_aom_lpf_vertical_6 = saved[1]; svt_aom_lpf_vertical_8 = saved[2]; svt_aom_lpf_vertical_14 = saved[3];
    svt_aom_lpf_horizontal_4 = saved[4]; svt_aom_lpf_horizontal_6 = saved[5]; svt_aom_lpf_horizontal_8 = saved[6]; svt_aom_lpf_horizontal_14 = saved[7];
    if (lvl_out)
        for (int p = 0; p < 3; p++)
            for (int d = 0; d < 2; d++)
                for (int r = 0; r < 8; r++)
                    for (int m = 0; m < 2; m++) lvl_out[((p * 2 + d) * 8 + r) * 2 + m] = ppcs->lf_info.lvl[p][0][d][r][m];
    const int bad = g_dlf_rec.bad;
    free(pic->buffer_y); free(pic->buffer_cb); free(pic->buffer_cr); free(pic); free(grid); free(mi); free(wr); free(scs); free(ppcs); free(pcs);
    return bad;
}
