/* svt_hip_lf_bridge.c — see svt_hip_lf_bridge.h.  Reference-side glue: host code only, every pixel operation is a batched svt_hip_* call. */
#include <stdlib.h>
#include <string.h>
#include "svt_hip_lf_bridge.h"
#include "EbDeblockingCommon.h"
#include "EbDeblockingFilter.h"
#include "EbRestoration.h"
#include "EbCdef.h"
#include "EbReferenceObject.h"
#include "EbUtility.h"

#define HIP_TRY(call) do { if ((call) != SVT_HIP_OK) return EB_ErrorUndefined; } while (0)   /* caller falls back to the C loop */
#define LF_BORDER 3                                                                           /* RESTORATION_BORDER */

static int log2i(int v) { int l = 0; while ((1 << l) < v) l++; return l; }
static size_t plane_bytes(const SvtHipLfPicture *p, int pl) { return (size_t)p->stride[pl] * (size_t)((p->h >> (pl > 0)) + 2 * LF_BORDER) * (size_t)p->pix_bytes; }
static void *plane_origin(const SvtHipLfPicture *p, void *base, int pl) { return (uint8_t *)base + ((size_t)LF_BORDER * p->stride[pl] + LF_BORDER) * (size_t)p->pix_bytes; }

static EbPictureBufferDesc *recon_of(PictureControlSet *pcs, int is_16bit) {
    if (pcs->parent_pcs_ptr->is_used_as_reference_flag == EB_TRUE) {
        EbReferenceObject *ro = (EbReferenceObject *)pcs->parent_pcs_ptr->reference_picture_wrapper_ptr->object_ptr;
        return is_16bit ? ro->reference_picture16bit : ro->reference_picture;
    }
    return is_16bit ? pcs->recon_picture16bit_ptr : pcs->recon_picture_ptr;
}

EbErrorType svt_hip_lf_picture_ctor(SvtHipCtx *hip, SvtHipLfPicture *p, int w, int h, int is_16bit, int bd) {
    memset(p, 0, sizeof(*p));
    p->pix_bytes = is_16bit ? 2 : 1; p->bd = bd; p->w = w; p->h = h;
    const int nfb = ((w + 63) / 64) * ((h + 63) / 64), mi_cols = (w + 3) / 4, mi_rows = (h + 3) / 4;
    for (int pl = 0; pl < 3; pl++) {
        const int pw = w >> (pl > 0), ph = h >> (pl > 0);
        p->stride[pl] = (pw + 2 * LF_BORDER + 63) & ~63; p->src_stride[pl] = (pw + 63) & ~63;
        HIP_TRY(svt_hip_malloc(hip, &p->d_recon[pl], plane_bytes(p, pl))); HIP_TRY(svt_hip_malloc(hip, &p->d_cdef[pl], plane_bytes(p, pl)));
        HIP_TRY(svt_hip_malloc(hip, &p->d_rest[pl], plane_bytes(p, pl)));
        HIP_TRY(svt_hip_malloc(hip, &p->d_src[pl], (size_t)p->src_stride[pl] * ph * p->pix_bytes));
        p->units_w[pl] = (pw + 3) / 4; p->units_h[pl] = (ph + 3) / 4;
        for (int d = 0; d < 2; d++) {
            p->h_edges[pl][d] = (uint16_t *)malloc(sizeof(uint16_t) * p->units_w[pl] * p->units_h[pl]);
            HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_edges[pl][d], sizeof(uint16_t) * p->units_w[pl] * p->units_h[pl]));
            if (!p->h_edges[pl][d]) return EB_ErrorInsufficientResources;
        }
        const int max_units = ((pw + 31) / 32) * ((ph + 31) / 32);     /* smallest restoration unit is 64 -> count_units_in_tile rounds to nearest */
        HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_unit_ep[pl], max_units)); HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_unit_xqd[pl], max_units * 8));
        HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_unit_wiener[pl], max_units * 32));
    }
    p->h_skip8 = (uint8_t *)malloc((size_t)(w / 8) * (h / 8)); p->h_mi = (SvtHipDlfModeInfo *)calloc((size_t)mi_cols * mi_rows, sizeof(SvtHipDlfModeInfo));
    if (!p->h_skip8 || !p->h_mi) return EB_ErrorInsufficientResources;
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_skip8, (size_t)(w / 8) * (h / 8)));
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_mse, sizeof(uint64_t) * 2 * nfb * 64));
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_dir, (size_t)nfb * 64)); HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_var, sizeof(int32_t) * nfb * 64));
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_y_strength, nfb)); HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_uv_strength, nfb));
    return EB_ErrorNone;
}

void svt_hip_lf_picture_dctor(SvtHipCtx *hip, SvtHipLfPicture *p) {
    for (int pl = 0; pl < 3; pl++) {
        svt_hip_free(hip, p->d_recon[pl]); svt_hip_free(hip, p->d_cdef[pl]); svt_hip_free(hip, p->d_rest[pl]); svt_hip_free(hip, p->d_src[pl]);
        for (int d = 0; d < 2; d++) { free(p->h_edges[pl][d]); svt_hip_free(hip, p->d_edges[pl][d]); }
        svt_hip_free(hip, p->d_unit_ep[pl]); svt_hip_free(hip, p->d_unit_xqd[pl]); svt_hip_free(hip, p->d_unit_wiener[pl]);
    }
    free(p->h_skip8); free(p->h_mi);
    svt_hip_free(hip, p->d_skip8); svt_hip_free(hip, p->d_mse); svt_hip_free(hip, p->d_dir); svt_hip_free(hip, p->d_var);
    svt_hip_free(hip, p->d_y_strength); svt_hip_free(hip, p->d_uv_strength);
    memset(p, 0, sizeof(*p));
}

static uint8_t *pic_plane(const EbPictureBufferDesc *pic, int pl, int pix_bytes, int *stride) {
    const int ss = pl > 0;
    uint8_t *base = pl == 0 ? pic->buffer_y : (pl == 1 ? pic->buffer_cb : pic->buffer_cr);
    *stride = pl == 0 ? pic->stride_y : (pl == 1 ? pic->stride_cb : pic->stride_cr);
    return base + ((size_t)(pic->origin_y >> ss) * *stride + (pic->origin_x >> ss)) * (size_t)pix_bytes;
}

EbErrorType svt_hip_lf_upload(SvtHipCtx *hip, SvtHipLfPicture *p, const EbPictureBufferDesc *pic, void *const d_dst[3]) {
    for (int pl = 0; pl < 3; pl++) {
        int st;
        const uint8_t *s = pic_plane(pic, pl, p->pix_bytes, &st);
        const int pw = p->w >> (pl > 0), ph = p->h >> (pl > 0);
        const int is_src = d_dst[pl] == p->d_src[pl];
        const int dstride = is_src ? p->src_stride[pl] : p->stride[pl];
        uint8_t *d = is_src ? (uint8_t *)d_dst[pl] : (uint8_t *)plane_origin(p, d_dst[pl], pl);
        for (int y = 0; y < ph; y++)   /* row copies; a production patch pins the picture buffers and issues one 2-D copy */
            HIP_TRY(svt_hip_memcpy_h2d(hip, d + (size_t)y * dstride * p->pix_bytes, s + (size_t)y * st * p->pix_bytes, (size_t)pw * p->pix_bytes));
    }
    return EB_ErrorNone;
}

EbErrorType svt_hip_lf_download(SvtHipCtx *hip, const SvtHipLfPicture *p, void *const d_src[3], EbPictureBufferDesc *pic) {
    for (int pl = 0; pl < 3; pl++) {
        int st;
        uint8_t *d = pic_plane(pic, pl, p->pix_bytes, &st);
        const int pw = p->w >> (pl > 0), ph = p->h >> (pl > 0);
        const uint8_t *s = (const uint8_t *)plane_origin(p, d_src[pl], pl);
        for (int y = 0; y < ph; y++)
            HIP_TRY(svt_hip_memcpy_d2h(hip, d + (size_t)y * st * p->pix_bytes, s + (size_t)y * p->stride[pl] * p->pix_bytes, (size_t)pw * p->pix_bytes));
    }
    return EB_ErrorNone;
}

/* ---------------------------------------------------------------- deblocking ---------------------------------------------------------
 * SvtHipDlfModeInfo per 4x4 unit from the mode-info grid = what set_lpf_parameters / get_transform_size read (EbDeblockingFilter.c:134-319) */
static void fill_mode_info(SvtHipLfPicture *p, PictureControlSet *pcs) {
    PictureParentControlSet *ppcs = pcs->parent_pcs_ptr;
    FrameHeader *frm_hdr = &ppcs->frm_hdr;
    const LoopFilterInfoN *lfi_n = &ppcs->lf_info;
    const int mi_cols = (p->w + 3) / 4, mi_rows = (p->h + 3) / 4;
    for (int r = 0; r < mi_rows; r++)
        for (int c = 0; c < mi_cols; c++) {
            const MbModeInfo *mbmi = &pcs->mi_grid_base[r * pcs->mi_stride + c]->mbmi;
            SvtHipDlfModeInfo *o = &p->h_mi[r * mi_cols + c];
            const BlockSize bs = mbmi->block_mi.sb_type;
            const int inter = is_inter_block_no_intrabc(mbmi->block_mi.ref_frame[0]);
            TxSize ts = inter ? tx_depth_to_tx_size[0][bs] : tx_depth_to_tx_size[mbmi->tx_depth][bs];
            if (inter && !mbmi->block_mi.skip) ts = tx_depth_to_tx_size[mbmi->tx_depth][bs];
            const TxSize uv = av1_get_max_uv_txsize(bs, 1, 1);
            o->tx_w_log2 = (uint8_t)log2i(tx_size_wide[ts]); o->tx_h_log2 = (uint8_t)log2i(tx_size_high[ts]);
            o->uv_tx_w_log2 = (uint8_t)log2i(tx_size_wide[uv]); o->uv_tx_h_log2 = (uint8_t)log2i(tx_size_high[uv]);
            o->bw_log2 = (uint8_t)log2i(block_size_wide[bs]); o->bh_log2 = (uint8_t)log2i(block_size_high[bs]);
            o->skip_inter = (uint8_t)(mbmi->block_mi.skip && inter);
            const PredictionMode mode = mbmi->block_mi.mode == INTRA_MODE_4x4 ? DC_PRED : mbmi->block_mi.mode;
            for (int pl = 0; pl < 3; pl++)
                for (int dir = 0; dir < 2; dir++)
                    o->level[pl][dir] = frm_hdr->delta_lf_params.delta_lf_present
                        ? get_filter_level_delta_lf(frm_hdr, dir, pl, ppcs->curr_delta_lf, 0, mode, mbmi->block_mi.ref_frame[0])
                        : lfi_n->lvl[pl][0][dir][mbmi->block_mi.ref_frame[0]][mode_lf_lut[mode]];
        }
}

EbErrorType svt_hip_dlf_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs) {
    FrameHeader *frm_hdr = &pcs->parent_pcs_ptr->frm_hdr;
    svt_av1_loop_filter_frame_init(frm_hdr, &pcs->parent_pcs_ptr->lf_info, 0, 3);       /* svt_av1_loop_filter_frame does this first (:722) */
    fill_mode_info(p, pcs);
    const int mi_cols = (p->w + 3) / 4, mi_rows = (p->h + 3) / 4;
    for (int pl = 0; pl < 3; pl++) {
        /* plane skipped when its frame level is 0, like loop_filter_sb's checks (:640-655) */
        if (pl == 0 && !frm_hdr->loop_filter_params.filter_level[0] && !frm_hdr->loop_filter_params.filter_level[1]) continue;
        if (pl == 1 && !frm_hdr->loop_filter_params.filter_level_u) continue;
        if (pl == 2 && !frm_hdr->loop_filter_params.filter_level_v) continue;
        const int pw = p->w >> (pl > 0), ph = p->h >> (pl > 0);
        HIP_TRY(svt_hip_dlf_build_edges(p->h_mi, mi_cols, mi_rows, pl, pl > 0, pl > 0, pw, ph, p->h_edges[pl][0], p->h_edges[pl][1]));
        const size_t eb = sizeof(uint16_t) * p->units_w[pl] * p->units_h[pl];
        HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_edges[pl][0], p->h_edges[pl][0], eb)); HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_edges[pl][1], p->h_edges[pl][1], eb));
        HIP_TRY(svt_hip_deblock_plane_dev(hip, plane_origin(p, p->d_recon[pl], pl), p->pix_bytes, p->stride[pl], p->bd, p->d_edges[pl][0], p->d_edges[pl][1],
                                          p->units_w[pl], p->units_h[pl], frm_hdr->loop_filter_params.sharpness_level));
    }
    return EB_ErrorNone;
}

/* ---------------------------------------------------------------- CDEF ---------------------------------------------------------------- */
static void fill_skip8(SvtHipLfPicture *p, PictureControlSet *pcs) {     /* is_8x8_block_skip (EbEncCdef.c:242-250) for every 8x8 block */
    const int c8 = p->w / 8, r8 = p->h / 8;
    for (int r = 0; r < r8; r++)
        for (int c = 0; c < c8; c++) {
            int skip = 1;
            for (int y = 0; y < 2; y++)
                for (int x = 0; x < 2; x++) skip &= (int)pcs->mi_grid_base[(2 * r + y) * pcs->mi_stride + 2 * c + x]->mbmi.block_mi.skip;
            p->h_skip8[r * c8 + c] = (uint8_t)skip;
        }
}

EbErrorType svt_hip_cdef_search_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs) {
    const int nfb = ((p->w + 63) / 64) * ((p->h + 63) / 64);
    const int pri_damping = 3 + (pcs->parent_pcs_ptr->frm_hdr.quantization_params.base_q_idx >> 6);   /* EbCdefProcess.c:121 */
    fill_skip8(p, pcs);
    HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_skip8, p->h_skip8, (size_t)(p->w / 8) * (p->h / 8)));
    const void *rec[3], *src[3];
    for (int pl = 0; pl < 3; pl++) { rec[pl] = plane_origin(p, p->d_recon[pl], pl); src[pl] = p->d_src[pl]; }
    HIP_TRY(svt_hip_cdef_search_frame_dev(hip, p->pix_bytes, rec, p->stride, src, p->src_stride, p->w, p->h, p->d_skip8, pri_damping, p->bd, p->d_mse, p->d_dir, p->d_var));
    /* pcs->mse_seg[pli][fb][gi]: [2] arrays of nfb x TOTAL_STRENGTHS uint64, the layout of the device table */
    HIP_TRY(svt_hip_memcpy_d2h(hip, pcs->mse_seg[0], p->d_mse, sizeof(uint64_t) * nfb * 64));
    HIP_TRY(svt_hip_memcpy_d2h(hip, pcs->mse_seg[1], p->d_mse + (size_t)nfb * 64, sizeof(uint64_t) * nfb * 64));
    return EB_ErrorNone;
}

EbErrorType svt_hip_cdef_apply_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs) {
    FrameHeader *frm_hdr = &pcs->parent_pcs_ptr->frm_hdr;
    const int nhfb = (p->w + 63) / 64, nvfb = (p->h + 63) / 64, nfb = nhfb * nvfb;
    uint8_t *ys = (uint8_t *)malloc(nfb), *uvs = (uint8_t *)malloc(nfb);
    if (!ys || !uvs) { free(ys); free(uvs); return EB_ErrorInsufficientResources; }
    for (int fbr = 0; fbr < nvfb; fbr++)
        for (int fbc = 0; fbc < nhfb; fbc++) {     /* the strength index finish_cdef_search stored in the fb's first mode-info (EbEncCdef.c:415-425) */
            const int8_t idx = pcs->mi_grid_base[MI_SIZE_64X64 * fbr * pcs->mi_stride + MI_SIZE_64X64 * fbc]->mbmi.cdef_strength;
            ys[fbr * nhfb + fbc] = idx < 0 ? 0 : (uint8_t)frm_hdr->cdef_params.cdef_y_strength[idx];
            uvs[fbr * nhfb + fbc] = idx < 0 ? 0 : (uint8_t)frm_hdr->cdef_params.cdef_uv_strength[idx];
        }
    int rc = svt_hip_memcpy_h2d(hip, p->d_y_strength, ys, nfb) | svt_hip_memcpy_h2d(hip, p->d_uv_strength, uvs, nfb);
    free(ys); free(uvs);
    if (rc != SVT_HIP_OK) return EB_ErrorUndefined;
    const void *in[3]; void *out[3];
    for (int pl = 0; pl < 3; pl++) {
        in[pl] = plane_origin(p, p->d_recon[pl], pl); out[pl] = plane_origin(p, p->d_cdef[pl], pl);
        HIP_TRY(svt_hip_memcpy_d2d(hip, p->d_cdef[pl], p->d_recon[pl], plane_bytes(p, pl)));    /* unfiltered blocks keep the deblocked samples */
    }
    /* direction / variance of the search are reused (same pre-CDEF picture) */
    HIP_TRY(svt_hip_cdef_apply_frame_dev(hip, p->pix_bytes, in, out, p->stride, p->w, p->h, p->d_skip8, p->d_y_strength, p->d_uv_strength,
                                         frm_hdr->cdef_params.cdef_damping, p->bd, p->d_dir, p->d_var));
    return EB_ErrorNone;
}

/* ---------------------------------------------------------------- loop restoration ---------------------------------------------------- */
EbErrorType svt_hip_rest_apply_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs) {
    Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
    for (int pl = 0; pl < 3; pl++) {
        const RestorationInfo *rsi = &cm->rst_info[pl];
        const int pw = p->w >> (pl > 0), ph = p->h >> (pl > 0), n = rsi->units_per_tile;
        if (rsi->frame_restoration_type == RESTORE_NONE) {
            HIP_TRY(svt_hip_memcpy_d2d(hip, p->d_rest[pl], p->d_cdef[pl], plane_bytes(p, pl)));
            continue;
        }
        uint8_t *ep = (uint8_t *)malloc(n); int32_t *xqd = (int32_t *)malloc(sizeof(int32_t) * 2 * n); int16_t *wn = (int16_t *)calloc((size_t)n * 16, sizeof(int16_t));
        if (!ep || !xqd || !wn) { free(ep); free(xqd); free(wn); return EB_ErrorInsufficientResources; }
        for (int u = 0; u < n; u++) {
            const RestorationUnitInfo *rui = &rsi->unit_info[u];
            ep[u] = rui->restoration_type == RESTORE_SGRPROJ ? (uint8_t)rui->sgrproj_info.ep : (rui->restoration_type == RESTORE_WIENER ? 254 : 255);
            xqd[2 * u] = rui->sgrproj_info.xqd[0]; xqd[2 * u + 1] = rui->sgrproj_info.xqd[1];
            memcpy(wn + 16 * u, rui->wiener_info.vfilter, 8 * sizeof(int16_t)); memcpy(wn + 16 * u + 8, rui->wiener_info.hfilter, 8 * sizeof(int16_t));
        }
        int rc = svt_hip_memcpy_h2d(hip, p->d_unit_ep[pl], ep, n) | svt_hip_memcpy_h2d(hip, p->d_unit_xqd[pl], xqd, sizeof(int32_t) * 2 * n) |
                 svt_hip_memcpy_h2d(hip, p->d_unit_wiener[pl], wn, sizeof(int16_t) * 16 * n);
        free(ep); free(xqd); free(wn);
        if (rc != SVT_HIP_OK) return EB_ErrorUndefined;
        /* the CDEF picture's 3-sample border (svt_extend_frame, EbRestoration.c:1306) */
        HIP_TRY(svt_hip_generate_padding_dev(hip, plane_origin(p, p->d_cdef[pl], pl), p->pix_bytes, p->stride[pl], pw, ph, LF_BORDER, LF_BORDER));
        HIP_TRY(svt_hip_lr_apply_plane_dev(hip, p->pix_bytes, p->bd, plane_origin(p, p->d_cdef[pl], pl), p->stride[pl], plane_origin(p, p->d_rest[pl], pl), p->stride[pl],
                                           pw, ph, rsi->restoration_unit_size, pl > 0, plane_origin(p, p->d_recon[pl], pl), p->stride[pl], p->d_unit_ep[pl],
                                           p->d_unit_xqd[pl], p->d_unit_wiener[pl]));
    }
    return EB_ErrorNone;
}

/* rest_kernel, search half: in place of every search_sgrproj_seg call of restoration_seg_search (EbRestorationPick.c:1277-1317, per unit:
 * search_selfguided_restoration + try_restoration_unit_seg).  One svt_hip_sgr_search_units_picture call gives the (ep, xqd) of every unit
 * of the three planes; the units are then filtered with exactly those parameters (stripe rules as in svt_av1_loop_restoration_filter_unit)
 * and their SSE against the source is what try_restoration_unit_seg -> sse_restoration_unit (:58-135) returns.  Results land in
 * pcs->parent_pcs_ptr->rusi_picture[plane][unit] (sgrproj, sse[RESTORE_SGRPROJ]) and cm->sg_frame_ep_cnt, i.e. where search_sgrproj_finish (:1319) and
 * rest_finish_search read them.  p->d_cdef must hold the CDEF output, p->d_recon the deblocked picture, p->d_src the source. */
EbErrorType svt_hip_sgr_search_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs) {
    Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
    /* the set window of search_selfguided_restoration (:596-607) */
    static const int8_t k_step[5] = {16, 0, 1, 4, 16};   /* get_sg_step (:693-704) */
    const int8_t step = k_step[cm->sg_filter_mode >= 1 && cm->sg_filter_mode <= 4 ? cm->sg_filter_mode : 0];
    const int8_t *re = cm->sg_ref_frame_ep;
    const int none = re[0] < 0 && re[1] < 0;
    const int mid = none ? 0 : (re[1] < 0 ? re[0] : (re[0] < 0 ? re[1] : (re[0] + re[1]) / 2));
    const int start_ep = none ? 0 : (mid - step > 0 ? mid - step : 0), end_ep = none ? 16 : (mid + step < 16 ? mid + step : 16);
    uint32_t mask = 0;
    for (int ep = start_ep; ep < end_ep; ep++) mask |= 1u << ep;
    if (!mask) return EB_ErrorBadParameter;

    SvtHipSgrSearchPlane job[3];
    int32_t *xqd[3] = {0}; int64_t *err[3] = {0}; uint8_t *best[3] = {0};
    EbErrorType ret = EB_ErrorNone;
    for (int pl = 0; pl < 3; pl++) {
        const RestorationInfo *rsi = &cm->rst_info[pl];
        const int pw = p->w >> (pl > 0), ph = p->h >> (pl > 0), n = rsi->units_per_tile;
        xqd[pl] = (int32_t *)malloc(sizeof(int32_t) * 32 * n); err[pl] = (int64_t *)malloc(sizeof(int64_t) * 16 * n); best[pl] = (uint8_t *)malloc(n);
        if (!xqd[pl] || !err[pl] || !best[pl]) { ret = EB_ErrorInsufficientResources; goto done; }
        if (svt_hip_generate_padding_dev(hip, plane_origin(p, p->d_cdef[pl], pl), p->pix_bytes, p->stride[pl], pw, ph, LF_BORDER, LF_BORDER) != SVT_HIP_OK) { ret = EB_ErrorUndefined; goto done; }
        job[pl].d_dgd = plane_origin(p, p->d_cdef[pl], pl); job[pl].stride = p->stride[pl];
        job[pl].d_src = p->d_src[pl]; job[pl].src_stride = p->src_stride[pl];
        job[pl].pw = pw; job[pl].ph = ph; job[pl].unit_size = rsi->restoration_unit_size; job[pl].ss_y = pl > 0;
        job[pl].ep_mask = mask; job[pl].xqd_out = xqd[pl]; job[pl].err_out = err[pl]; job[pl].best_ep = best[pl];
    }
    if (svt_hip_sgr_search_units_picture(hip, p->pix_bytes, p->bd, 3, job, NULL) != SVT_HIP_OK) { ret = EB_ErrorUndefined; goto done; }

    for (int pl = 0; pl < 3; pl++) {
        const RestorationInfo *rsi = &cm->rst_info[pl];
        const int pw = p->w >> (pl > 0), ph = p->h >> (pl > 0), n = rsi->units_per_tile, us = rsi->restoration_unit_size;
        RestUnitSearchInfo *rusi = pcs->parent_pcs_ptr->rusi_picture[pl];
        uint8_t *ep = (uint8_t *)malloc(n); int32_t *uq = (int32_t *)malloc(sizeof(int32_t) * 2 * n);
        SvtHipBlkPair *rect = (SvtHipBlkPair *)malloc(sizeof(SvtHipBlkPair) * n); uint64_t *sse = (uint64_t *)malloc(sizeof(uint64_t) * n);
        void *d_rect = NULL, *d_sse = NULL;
        int ok = ep && uq && rect && sse;
        if (ok) {
            /* unit rectangles of foreach_rest_unit_in_tile (EbRestoration.c:1369-1411) */
            const int ext = us * 3 / 2, voff = 8 >> (pl > 0), hunits = rsi->horz_units_per_tile;
            int y0 = 0, i = 0;
            while (y0 < ph) {
                const int rem_h = ph - y0, h = rem_h < ext ? rem_h : us;
                int v0 = y0 - voff > 0 ? y0 - voff : 0, v1 = y0 + h;
                if (v1 < ph) v1 -= voff;
                int x0 = 0, j = 0;
                while (x0 < pw) {
                    const int rem_w = pw - x0, w = rem_w < ext ? rem_w : us, u = i * hunits + j;
                    ep[u] = best[pl][u];
                    uq[2 * u] = xqd[pl][(u * 16 + ep[u]) * 2]; uq[2 * u + 1] = xqd[pl][(u * 16 + ep[u]) * 2 + 1];
                    rusi[u].sgrproj.ep = ep[u]; rusi[u].sgrproj.xqd[0] = uq[2 * u]; rusi[u].sgrproj.xqd[1] = uq[2 * u + 1];
                    cm->sg_frame_ep_cnt[ep[u]]++;
                    rect[u].a_x = rect[u].b_x = x0; rect[u].a_y = rect[u].b_y = v0; rect[u].w = (uint16_t)w; rect[u].h = (uint16_t)(v1 - v0);
                    x0 += w; j++;
                }
                y0 += h; i++;
            }
            ok = svt_hip_malloc(hip, &d_rect, sizeof(SvtHipBlkPair) * n) == SVT_HIP_OK && svt_hip_malloc(hip, &d_sse, sizeof(uint64_t) * n) == SVT_HIP_OK &&
                 svt_hip_memcpy_h2d(hip, p->d_unit_ep[pl], ep, n) == SVT_HIP_OK && svt_hip_memcpy_h2d(hip, p->d_unit_xqd[pl], uq, sizeof(int32_t) * 2 * n) == SVT_HIP_OK &&
                 svt_hip_memcpy_h2d(hip, d_rect, rect, sizeof(SvtHipBlkPair) * n) == SVT_HIP_OK &&
                 /* try_restoration_unit_seg: the unit filtered for real (stripe context from the deblocked picture), then its SSE */
                 svt_hip_sgr_apply_plane_dev(hip, p->pix_bytes, p->bd, plane_origin(p, p->d_cdef[pl], pl), p->stride[pl], plane_origin(p, p->d_rest[pl], pl), p->stride[pl],
                                             pw, ph, us, pl > 0, plane_origin(p, p->d_recon[pl], pl), p->stride[pl], p->d_unit_ep[pl], p->d_unit_xqd[pl]) == SVT_HIP_OK &&
                 svt_hip_block_sse_batch_dev(hip, p->pix_bytes, p->d_src[pl], p->src_stride[pl], plane_origin(p, p->d_rest[pl], pl), p->stride[pl],
                                             (const SvtHipBlkPair *)d_rect, n, (uint64_t *)d_sse) == SVT_HIP_OK &&
                 svt_hip_memcpy_d2h(hip, sse, d_sse, sizeof(uint64_t) * n) == SVT_HIP_OK;
            if (ok)
                for (int u = 0; u < n; u++) rusi[u].sse[RESTORE_SGRPROJ] = (int64_t)sse[u];
        }
        if (d_rect) svt_hip_free(hip, d_rect);
        if (d_sse) svt_hip_free(hip, d_sse);
        free(ep); free(uq); free(rect); free(sse);
        if (!ok) { ret = EB_ErrorUndefined; goto done; }
    }
done:
    for (int pl = 0; pl < 3; pl++) { free(xqd[pl]); free(err[pl]); free(best[pl]); }
    return ret;
}
