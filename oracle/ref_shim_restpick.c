/*
 * ref_shim_restpick.c — test infrastructure, part of oracle/_ref (the REAL reference built in place by oracle/Makefile.ref).
 *
 * The restoration search's inner functions are `static` in the reference (Encoder/Codec/EbRestorationPick.c:
 * finer_search_pixel_proj_error :353, search_selfguided_restoration :583).  This translation unit REPLACES that source file in the
 * build: it includes it textually from where it lies under $(REF) (nothing is copied into the repository) and adds two exported
 * wrappers, so the oracle's restatement of the per-unit self-guided search can be pinned to the reference's own code.
 */
#include "EbRestoration.h"   /* declared before the counting macro below exists */
static int shim_probe_count;  /* try_restoration_unit_seg (:137) is static: its one call of svt_av1_loop_restoration_filter_unit (:153) is counted instead */
#define svt_av1_loop_restoration_filter_unit(...) (shim_probe_count++, svt_av1_loop_restoration_filter_unit(__VA_ARGS__))
#include "EbRestorationPick.c"
#undef svt_av1_loop_restoration_filter_unit

/* finer_search_pixel_proj_error with the parameter set given by index; xqd is in/out; returns the error */
int64_t ref_shim_sgr_finer_search(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8,
                                  int32_t dat_stride, int32_t use_highbitdepth, int32_t *flt0, int32_t flt0_stride, int32_t *flt1,
                                  int32_t flt1_stride, int32_t start_step, int32_t *xqd, int32_t ep) {
    return finer_search_pixel_proj_error(src8, width, height, src_stride, dat8, dat_stride, use_highbitdepth, flt0, flt0_stride, flt1,
                                         flt1_stride, start_step, xqd, &eb_sgr_params[ep]);
}

/* search_selfguided_restoration for one restoration unit: out[0] = ep, out[1..2] = xqd; rstbuf must hold 2 * RESTORATION_UNITPELS_MAX
 * int32 (ref_shim_sgr_rstbuf_ints()).  The RTCD tables must have been set up. */
void ref_shim_sgr_search_unit(const uint8_t *dat8, int32_t width, int32_t height, int32_t dat_stride, const uint8_t *src8,
                              int32_t src_stride, int32_t use_highbitdepth, int32_t bit_depth, int32_t pu_width, int32_t pu_height,
                              int32_t *rstbuf, int32_t ref_ep0, int32_t ref_ep1, int32_t step, int32_t *out) {
    int8_t  ref_ep[2] = {(int8_t)ref_ep0, (int8_t)ref_ep1};
    int32_t cnt[SGRPROJ_PARAMS] = {0};
    const SgrprojInfo r = search_selfguided_restoration(dat8, width, height, dat_stride, src8, src_stride, use_highbitdepth, bit_depth,
                                                        pu_width, pu_height, rstbuf, ref_ep, cnt, (int8_t)step);
    out[0] = r.ep; out[1] = r.xqd[0]; out[2] = r.xqd[1];
}
int32_t ref_shim_sgr_rstbuf_ints(void) { return 2 * RESTORATION_UNITPELS_MAX; }

/* search_wiener_seg between the statistics and the refinement (:1388-1407), the reference's own functions: 1 = vfilter / hfilter hold the initial filter and it beats
 * the identity filter, 2 = it does not, 0 = the decomposition failed.  M and H are modified in place by nothing here (the reference works on copies). */
int32_t ref_shim_wiener_unit_init(int32_t wiener_win, int64_t *M, int64_t *H, int16_t *vfilter, int16_t *hfilter) {
    int32_t    vfilterd[WIENER_WIN], hfilterd[WIENER_WIN];
    WienerInfo wi;
    memset(&wi, 0, sizeof(wi));
    if (!wiener_decompose_sep_sym(wiener_win, M, H, vfilterd, hfilterd)) return 0;
    finalize_sym_filter(wiener_win, vfilterd, wi.vfilter);
    finalize_sym_filter(wiener_win, hfilterd, wi.hfilter);
    memcpy(vfilter, wi.vfilter, 8 * sizeof(int16_t)); memcpy(hfilter, wi.hfilter, 8 * sizeof(int16_t));
    return compute_score(wiener_win, M, H, wi.vfilter, wi.hfilter) > 0 ? 2 : 1;
}

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * finer_tile_search_wiener_seg (:1092) for every active unit of one plane — the function the device's wiener_walk_kernel and the CPU test double's restatement
 * (oracle/hip_mock.c) stand for.  Arguments like svt_hip_wiener_walk_units_dev (include/svt_hip.h): dgd = sample (0, 0) of the CDEF output (3 writable border samples
 * around it), dbl = the deblocked plane (stripe boundaries), src = the source; ss = 0 luma-like plane, 1 = chroma-like plane of a 4:2:0 picture; unit_wiener[u][2][8]
 * (vertical, horizontal taps) in / out; err[u], probes[u] out.  Only builds the structs the reference's functions take. */
void save_tile_row_boundary_lines(uint8_t *src, int32_t src_stride, int32_t src_width, int32_t src_height, int32_t use_highbd, int32_t plane, Av1Common *cm,
                                  int32_t after_cdef, RestorationStripeBoundaries *boundaries);
typedef struct { RestSearchCtxt *rsc; int16_t *unit_wiener; const uint8_t *active; int32_t win; int64_t *err; uint32_t *probes; } ShimWienerCtx;
static void shim_wiener_visitor(const RestorationTileLimits *limits, const Av1PixelRect *tile_rect, int32_t unit_idx, void *priv) {
    ShimWienerCtx *c = (ShimWienerCtx *)priv;
    if (c->probes) c->probes[unit_idx] = 0;
    if (!c->active[unit_idx]) return;
    RestorationUnitInfo rui;
    memset(&rui, 0, sizeof(rui));
    rui.restoration_type = RESTORE_WIENER;
    memcpy(rui.wiener_info.vfilter, c->unit_wiener + 16 * unit_idx, 8 * sizeof(int16_t));
    memcpy(rui.wiener_info.hfilter, c->unit_wiener + 16 * unit_idx + 8, 8 * sizeof(int16_t));
    shim_probe_count = 0;
    c->err[unit_idx] = finer_tile_search_wiener_seg(c->rsc, limits, tile_rect, &rui, c->win);
    if (c->probes) c->probes[unit_idx] = (uint32_t)shim_probe_count;
    memcpy(c->unit_wiener + 16 * unit_idx, rui.wiener_info.vfilter, 8 * sizeof(int16_t));
    memcpy(c->unit_wiener + 16 * unit_idx + 8, rui.wiener_info.hfilter, 8 * sizeof(int16_t));
}
static int shim_wn_units(int unit_size, int size) { int n = (size + (unit_size >> 1)) / unit_size; return n > 1 ? n : 1; }   /* count_units_in_tile */
int32_t ref_shim_wiener_finer_search_plane(int32_t pix_bytes, int32_t bd, void *dgd, int32_t stride, int32_t pw, int32_t ph, int32_t unit_size, int32_t ss, void *dbl,
                                           int32_t dbl_stride, void *src, int32_t src_stride, int16_t *unit_wiener, const uint8_t *active, int32_t wiener_win, int64_t *err,
                                           uint32_t *probes) {
    const int highbd = pix_bytes == 2, plane = ss ? 1 : 0, frame_w = pw << ss, frame_h = ph << ss;
    Av1Common *cm = (Av1Common *)calloc(1, sizeof(Av1Common));
    cm->frm_size.frame_width = (uint16_t)frame_w; cm->frm_size.frame_height = (uint16_t)frame_h;
    cm->frm_size.superres_upscaled_width = (uint16_t)frame_w; cm->frm_size.superres_upscaled_height = (uint16_t)frame_h;
    cm->frm_size.superres_denominator = 8;
    cm->subsampling_x = 1; cm->subsampling_y = 1; cm->bit_depth = bd; cm->use_highbitdepth = highbd;
    cm->mi_rows = (frame_h + 3) >> 2; cm->mi_cols = (frame_w + 3) >> 2;
    RestorationInfo *rsi = &cm->rst_info[plane];
    rsi->restoration_unit_size = unit_size; rsi->frame_restoration_type = RESTORE_WIENER;
    rsi->horz_units_per_tile = shim_wn_units(unit_size, pw); rsi->vert_units_per_tile = shim_wn_units(unit_size, ph);
    rsi->units_per_tile = rsi->horz_units_per_tile * rsi->vert_units_per_tile;
    /* svt_av1_alloc_restoration_buffers (Common/Codec/EbRestoration.c:1872-1930) */
    const int num_stripes = (RESTORATION_UNIT_OFFSET + (cm->mi_rows << 2) + 63) / 64, bstride = (pw + 2 * RESTORATION_EXTRA_HORZ + 31) & ~31;
    const size_t bsize = (size_t)num_stripes * bstride * RESTORATION_CTX_VERT << highbd;
    rsi->boundaries.stripe_boundary_above = (uint8_t *)calloc(bsize + 64, 1);
    rsi->boundaries.stripe_boundary_below = (uint8_t *)calloc(bsize + 64, 1);
    rsi->boundaries.stripe_boundary_stride = bstride; rsi->boundaries.stripe_boundary_size = (int32_t)bsize;
    uint8_t *dgd8 = highbd ? CONVERT_TO_BYTEPTR(dgd) : (uint8_t *)dgd, *src8 = highbd ? CONVERT_TO_BYTEPTR(src) : (uint8_t *)src;
    save_tile_row_boundary_lines((uint8_t *)dbl, dbl_stride, pw, ph, highbd, plane, cm, 0, &rsi->boundaries);
    save_tile_row_boundary_lines((uint8_t *)dgd, stride, pw, ph, highbd, plane, cm, 1, &rsi->boundaries);
    svt_extend_frame(dgd8, pw, ph, stride, RESTORATION_BORDER, RESTORATION_BORDER, highbd);
    void *trial = calloc((size_t)stride * (ph + 16), (size_t)pix_bytes);
    uint8_t *trial8 = highbd ? CONVERT_TO_BYTEPTR(trial) : (uint8_t *)trial;
    Yv12BufferConfig fts, srcb, dstb;
    memset(&fts, 0, sizeof(fts)); memset(&srcb, 0, sizeof(srcb)); memset(&dstb, 0, sizeof(dstb));
    fts.buffers[plane] = dgd8; fts.strides[plane > 0] = stride; fts.crop_widths[plane > 0] = pw; fts.crop_heights[plane > 0] = ph;
    srcb.buffers[plane] = src8; srcb.strides[plane > 0] = src_stride; srcb.crop_widths[plane > 0] = pw; srcb.crop_heights[plane > 0] = ph;
    dstb.buffers[plane] = trial8; dstb.strides[plane > 0] = stride; dstb.crop_widths[plane > 0] = pw; dstb.crop_heights[plane > 0] = ph;
    RestSearchCtxt rsc;
    memset(&rsc, 0, sizeof(rsc));
    rsc.src = &srcb; rsc.dst = &dstb; rsc.cm = cm; rsc.plane = plane; rsc.plane_width = pw; rsc.plane_height = ph; rsc.org_frame_to_show = &fts;
    rsc.dgd_buffer = dgd8; rsc.dgd_stride = stride; rsc.src_buffer = src8; rsc.src_stride = src_stride; rsc.tile_stripe0 = 0;
    if (posix_memalign((void **)&rsc.tmpbuf, 32, RESTORATION_TMPBUF_SIZE)) return -1;
    ShimWienerCtx c = {&rsc, unit_wiener, active, wiener_win, err, probes};
    av1_foreach_rest_unit_in_frame(cm, plane, NULL, shim_wiener_visitor, &c);
    free(rsc.tmpbuf); free(trial); free(rsi->boundaries.stripe_boundary_above); free(rsi->boundaries.stripe_boundary_below); free(cm);
    return 0;
}
