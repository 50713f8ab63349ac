/*
 * ref_shim_restpick.c — test infrastructure, part of oracle/_ref (the REAL reference built in place by oracle/Makefile.ref).
 *
 * The restoration search's inner functions are `static` in the reference (Encoder/Codec/EbRestorationPick.c:
 * finer_search_pixel_proj_error :353, search_selfguided_restoration :583).  This translation unit REPLACES that source file in the
 * build: it includes it textually from where it lies under $(REF) (nothing is copied into the repository) and adds two exported
 * wrappers, so the oracle's restatement of the per-unit self-guided search can be pinned to the reference's own code.
 */
#include "EbRestorationPick.c"

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
