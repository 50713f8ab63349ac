/* svt_hip_tf_bridge.c — see svt_hip_tf_bridge.h.  Host code only; the pixel work is one svt_hip_tf_filter_frame_dev launch per central picture. */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "svt_hip_tf_bridge.h"
#include "svt_hip_hooks.h"
#include "EbLog.h"

#define HIP_TRY(call) do { if ((call) != SVT_HIP_OK) return EB_ErrorUndefined; } while (0)   /* caller falls back to the C loop */

EbErrorType svt_hip_tf_window_ctor(SvtHipCtx *hip, SvtHipTfWindow *w, int n_frames, int index_center, int width, int height, int is_16bit, int ss_x, int ss_y) {
    memset(w, 0, sizeof(*w));
    if (n_frames < 1 || n_frames > SVT_HIP_TF_MAX_REFS) return EB_ErrorBadParameter;
    w->n_frames = n_frames; w->index_center = index_center;
    w->blk_cols = (width + 63) / 64; w->blk_rows = (height + 63) / 64;      /* blk_cols / blk_rows, EbTemporalFiltering.c:2076-2079 */
    const size_t nblk = (size_t)w->blk_cols * w->blk_rows;
    const int pb = is_16bit ? 2 : 1;
    w->pred_stride[0] = w->blk_cols * 64; w->pred_stride[1] = w->pred_stride[2] = (w->blk_cols * 64) >> ss_x;
    for (int f = 0; f < n_frames; f++) {
        if (f == index_center) continue;
        w->h_blocks[f] = (SvtHipTfBlk64 *)calloc(nblk, sizeof(SvtHipTfBlk64));
        if (!w->h_blocks[f]) return EB_ErrorInsufficientResources;
        HIP_TRY(svt_hip_hooks_malloc(hip, (void **)&w->d_blocks[f], nblk * sizeof(SvtHipTfBlk64)));
        for (int p = 0; p < 3; p++) {
            const size_t rows = (size_t)(w->blk_rows * 64) >> (p ? ss_y : 0);
            HIP_TRY(svt_hip_hooks_malloc(hip, &w->d_pred[f][p], (size_t)w->pred_stride[p] * rows * pb));
        }
    }
    HIP_TRY(svt_hip_hooks_malloc(hip, (void **)&w->d_sse, 2 * sizeof(uint64_t)));
    return EB_ErrorNone;
}

void svt_hip_tf_window_dctor(SvtHipCtx *hip, SvtHipTfWindow *w) {
    for (int f = 0; f < SVT_HIP_TF_MAX_REFS; f++) {
        free(w->h_blocks[f]);
        svt_hip_hooks_free(hip, w->d_blocks[f]);
        for (int p = 0; p < 3; p++) svt_hip_hooks_free(hip, w->d_pred[f][p]);
    }
    svt_hip_hooks_free(hip, w->d_sse);
    memset(w, 0, sizeof(*w));
}

void svt_hip_tf_record_block(SvtHipTfWindow *w, int frame_index, uint32_t blk_row, uint32_t blk_col, const MeContext *c) {
    SvtHipTfBlk64 *b = &w->h_blocks[frame_index][(size_t)blk_row * w->blk_cols + blk_col];
    for (int i = 0; i < 16; i++) { b->mv16_x[i] = c->tf_16x16_mv_x[i]; b->mv16_y[i] = c->tf_16x16_mv_y[i]; b->err16[i] = c->tf_16x16_block_error[i]; }
    for (int i = 0; i < 4; i++) {
        b->mv32_x[i] = c->tf_32x32_mv_x[i]; b->mv32_y[i] = c->tf_32x32_mv_y[i]; b->err32[i] = c->tf_32x32_block_error[i];
        b->split[i] = c->tf_32x32_block_split_flag[i];
    }
}

EbErrorType svt_hip_tf_flush_picture(SvtHipCtx *hip, SvtHipTfWindow *w, const MeContext *c, int is_16bit, int bd, void *const d_src[3], const int src_stride[3],
                                     void *const d_dst[3], const int dst_stride[3], int ss_x, int ss_y, const double *noise_levels, int decay_control,
                                     uint64_t *filtered_sse, uint64_t *filtered_sse_uv) {
    SvtHipTfRef refs[SVT_HIP_TF_MAX_REFS];
    memset(refs, 0, sizeof(refs));
    const size_t nblk = (size_t)w->blk_cols * w->blk_rows;
    for (int f = 0; f < w->n_frames; f++) {
        if (f == w->index_center) continue;                      /* blocks == NULL: apply_filtering_central */
        if (!w->blocks_on_device[f]) HIP_TRY(svt_hip_memcpy_h2d(hip, w->d_blocks[f], w->h_blocks[f], nblk * sizeof(SvtHipTfBlk64)));
        for (int p = 0; p < 3; p++) { refs[f].pred[p] = w->d_pred[f][p]; refs[f].pred_stride[p] = w->pred_stride[p]; }
        refs[f].blocks = w->d_blocks[f];
    }
    HIP_TRY(svt_hip_tf_filter_frame_dev(hip, is_16bit ? 2 : 1, bd, (const void *const *)d_src, src_stride, d_dst, dst_stride, w->blk_cols * 64, w->blk_rows * 64, ss_x,
                                        ss_y, c->tf_chroma, refs, w->n_frames, noise_levels, decay_control, c->min_frame_size, w->d_sse));
    uint64_t sse[2];
    HIP_TRY(svt_hip_memcpy_d2h(hip, sse, w->d_sse, sizeof(sse)));
    *filtered_sse = sse[0]; *filtered_sse_uv = sse[1];
    return EB_ErrorNone;
}

/* ------------------------------------------------------------------ hook "tf": one TF segment */
struct SvtHipTfSeg {
    SvtHipTfWindow w;                 /* geometry of the segment's rectangle (blk_cols x blk_rows), block records per frame */
    uint32_t       col0, row0;
    int            is_highbd, ss_x, ss_y, decay_control, ctor_done;
    uint8_t       *h_pred[SVT_HIP_TF_MAX_REFS][3]; /* host staging of the predictor pictures (the reference's tf_inter_prediction output); hook "tf_subpel": unused */
    size_t         plane_bytes[3];
    /* hook "tf_subpel": the (block, frame) pairs whose sub-pel searches and prediction run on the device at the flush */
    int                        subpel, mi_cols, mi_rows, tf_hp;
    uint64_t                   th16;
    SvtHipTfSubpelBlk         *jobs[SVT_HIP_TF_MAX_REFS];
    int                        n_jobs[SVT_HIP_TF_MAX_REFS], n_host[SVT_HIP_TF_MAX_REFS];   /* pairs recorded for the device / predicted by the reference's C */
    const EbPictureBufferDesc *ref_pic[SVT_HIP_TF_MAX_REFS];                                /* geometry of the frame's reference picture */
    const uint8_t             *ref_plane[SVT_HIP_TF_MAX_REFS][3];                           /* its planes (8-bit buffers, or altref_buffer_highbd) */
};

SvtHipTfSeg *svt_hip_tf_seg_begin(int n_frames, int index_center, uint32_t col0, uint32_t col1, uint32_t row0, uint32_t row1, int is_highbd, int ss_x, int ss_y) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_TF) || n_frames < 1 || n_frames > SVT_HIP_TF_MAX_REFS || col1 <= col0 || row1 <= row0) return NULL;
    SvtHipTfSeg *s = (SvtHipTfSeg *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->col0 = col0; s->row0 = row0; s->is_highbd = is_highbd; s->ss_x = ss_x; s->ss_y = ss_y;
    s->subpel = svt_hip_hook_enabled(SVT_HIP_HOOK_TF_SUBPEL) && ss_x == 1 && ss_y == 1;   /* av1_inter_prediction's chroma path is 4:2:0 */
    SvtHipTfWindow *w = &s->w;
    w->n_frames = n_frames; w->index_center = index_center;
    w->blk_cols = (int)(col1 - col0); w->blk_rows = (int)(row1 - row0);
    w->pred_stride[0] = w->blk_cols * 64; w->pred_stride[1] = w->pred_stride[2] = (w->blk_cols * 64) >> ss_x;
    const size_t nblk = (size_t)w->blk_cols * w->blk_rows;
    const int    pb = is_highbd ? 2 : 1;
    for (int p = 0; p < 3; p++) s->plane_bytes[p] = (size_t)w->pred_stride[p] * ((size_t)(w->blk_rows * 64) >> (p ? ss_y : 0)) * pb;
    for (int f = 0; f < n_frames; f++) {
        if (f == index_center) continue;
        w->h_blocks[f] = (SvtHipTfBlk64 *)calloc(nblk, sizeof(SvtHipTfBlk64));
        int ok = w->h_blocks[f] != NULL;
        for (int p = 0; p < 3 && ok; p++) ok = (s->h_pred[f][p] = (uint8_t *)calloc(1, s->plane_bytes[p])) != NULL;
        if (ok && s->subpel) ok = (s->jobs[f] = (SvtHipTfSubpelBlk *)calloc(nblk, sizeof(SvtHipTfSubpelBlk))) != NULL;
        if (!ok) { svt_hip_tf_seg_end(s); return NULL; }
    }
    return s;
}

void svt_hip_tf_seg_end(SvtHipTfSeg *s) {
    if (!s) return;
    SvtHipCtx *hip = s->ctor_done ? svt_hip_hooks_lock_any() : NULL;
    for (int f = 0; f < SVT_HIP_TF_MAX_REFS; f++) {
        free(s->w.h_blocks[f]); free(s->jobs[f]);
        for (int p = 0; p < 3; p++) free(s->h_pred[f][p]);
        if (hip) {
            svt_hip_hooks_free(hip, s->w.d_blocks[f]);
            for (int p = 0; p < 3; p++) svt_hip_hooks_free(hip, s->w.d_pred[f][p]);
        }
    }
    if (hip) { svt_hip_hooks_free(hip, s->w.d_sse); svt_hip_hooks_unlock_any(); }
    free(s);
}

/* hook "tf_subpel", after the motion search of (frame, block): 1 = the pair is recorded — tf_32x32 / tf_16x16_sub_pel_search, derive_tf_32x32_block_split_flag and
 * tf_inter_prediction of it run on the device in svt_hip_tf_seg_flush (one launch per window frame), the caller skips them */
int svt_hip_tf_seg_subpel(SvtHipTfSeg *s, int frame_index, uint32_t blk_row, uint32_t blk_col, const MeContext *c, const PictureParentControlSet *pcs_central,
                          const PictureParentControlSet *pcs_ref, const EbPictureBufferDesc *pic_ref, uint32_t sb_origin_x, uint32_t sb_origin_y) {
    if (!s || !s->subpel || frame_index == s->w.index_center || !s->jobs[frame_index] || !c->p_best_mv32x32 || !c->p_best_mv16x16) return 0;
    const uint32_t r = blk_row - s->row0, col = blk_col - s->col0;
    if (s->n_host[frame_index]) return 0;   /* one frame is predicted either here or by the C code, not both */
    SvtHipTfSubpelBlk *j = &s->jobs[frame_index][s->n_jobs[frame_index]++];
    j->x = (int32_t)sb_origin_x; j->y = (int32_t)sb_origin_y;
    j->dst_x = (int32_t)col * 64; j->dst_y = (int32_t)r * 64;
    j->blk_index = (int32_t)(r * (uint32_t)s->w.blk_cols + col);
    for (int i = 0; i < 4; i++) j->mv32[i] = c->p_best_mv32x32[i];
    for (int i = 0; i < 16; i++) j->mv16[i] = c->p_best_mv16x16[i];
    s->mi_cols = pcs_central->av1_cm->mi_cols; s->mi_rows = pcs_central->av1_cm->mi_rows;
    s->th16 = c->tf_block_32x32_16x16_th; s->tf_hp = c->tf_hp;
    s->ref_pic[frame_index] = pic_ref;
    for (int p = 0; p < 3; p++) {
        const uint8_t *b = p == 0 ? pic_ref->buffer_y : (p == 1 ? pic_ref->buffer_cb : pic_ref->buffer_cr);
        s->ref_plane[frame_index][p] = s->is_highbd ? (const uint8_t *)pcs_ref->altref_buffer_highbd[p] : b;
    }
    return 1;
}

void svt_hip_tf_seg_block(SvtHipTfSeg *s, int frame_index, uint32_t blk_row, uint32_t blk_col, const MeContext *c, EbByte *pred, uint16_t **pred_16bit,
                          const uint32_t *stride_pred, int decay_control) {
    s->decay_control = decay_control;   /* the same for every block of the picture (resolution class and QP, EbTemporalFiltering.c:2313-2320) */
    if (frame_index == s->w.index_center) return;   /* apply_filtering_central reads the central picture itself */
    if (s->n_jobs[frame_index]) return;             /* recorded by svt_hip_tf_seg_subpel: predictor and TF fields are produced on the device */
    s->n_host[frame_index]++;
    const uint32_t r = blk_row - s->row0, col = blk_col - s->col0;
    svt_hip_tf_record_block(&s->w, frame_index, r, col, c);
    const int pb = s->is_highbd ? 2 : 1;
    for (int p = 0; p < (c->tf_chroma ? 3 : 1); p++) {
        const int      bw = p ? 64 >> s->ss_x : 64, bh = p ? 64 >> s->ss_y : 64;
        const uint8_t *src = s->is_highbd ? (const uint8_t *)pred_16bit[p] : pred[p];
        uint8_t       *dst = s->h_pred[frame_index][p] + ((size_t)r * bh * s->w.pred_stride[p] + (size_t)col * bw) * pb;
        for (int y = 0; y < bh; y++) memcpy(dst + (size_t)y * s->w.pred_stride[p] * pb, src + (size_t)y * stride_pred[p] * pb, (size_t)bw * pb);
    }
}

#define TF_TRY(x) do { if (ret == EB_ErrorNone && (x) != SVT_HIP_OK) ret = EB_ErrorUndefined; } while (0)
/* The reference picture of window frame f: only the rows the recorded vectors can reach travel (a segment is a band of block rows), every plane with its
 * full padded width; the entry point gets the planes' pointers at picture sample (0, 0). */
static EbErrorType tf_subpel_frame(SvtHipCtx *hip, SvtHipTfSeg *s, int f, const MeContext *c, void *const d_src[3], const int sstride[3], int bd) {
    const EbPictureBufferDesc *rp = s->ref_pic[f];
    const int                  pb = s->is_highbd ? 2 : 1, np = c->tf_chroma ? 3 : 1, n = s->n_jobs[f];
    EbErrorType                ret = EB_ErrorNone;
    /* luma rows: the block displaced by its integer vectors, +- (7/8 sample of refinement + the 8-tap support), and the block's own rows (the clamp of
     * clamp_mv_to_umv_border_sb pulls a vector towards the picture, never away from it) */
    int y_lo = INT32_MAX, y_hi = INT32_MIN;
    for (int k = 0; k < n; k++) {
        const SvtHipTfSubpelBlk *j = &s->jobs[f][k];
        for (int i = 0; i < 20; i++) {
            const uint32_t word = i < 4 ? j->mv32[i] : j->mv16[i - 4];
            const int      my = (int16_t)(word >> 16) >> 2;   /* quarter-pel word of an integer vector */
            const int      lo = j->y + (my < 0 ? my : 0) - 8, hi = j->y + 63 + (my > 0 ? my : 0) + 8;
            if (lo < y_lo) y_lo = lo;
            if (hi > y_hi) y_hi = hi;
        }
    }
    const int stride3[3] = {rp->stride_y, rp->stride_cb, rp->stride_cr};
    void     *d_band[3] = {NULL, NULL, NULL}, *d_jobs = NULL;
    const void *d_ref[3] = {NULL, NULL, NULL};
    const uint8_t *d_res[3] = {NULL, NULL, NULL};
    for (int p = 0; p < np; p++) {
        const int ss = p ? 1 : 0, org_x = rp->origin_x >> ss, org_y = rp->origin_y >> ss, rows = (rp->height >> ss) + 2 * org_y;
        int lo = (y_lo >> ss) - 2, hi = (y_hi >> ss) + 2;   /* chroma: the halved range with a margin for the halved-position rounding */
        if (lo < -org_y) lo = -org_y;
        if (hi > rows - org_y - 1) hi = rows - org_y - 1;
        if (hi < lo) { ret = EB_ErrorUndefined; break; }   /* the clean-up below still runs */
        /* the whole plane resident (SVT_HIP_RESIDENT, svt_hip_hooks.c: 8-bit pictures, announced by picture analysis and the end of their own filtering): no band */
        d_res[p] = ret == EB_ErrorNone ? (const uint8_t *)svt_hip_resident_acquire(hip, s->ref_plane[f][p], (size_t)rows * stride3[p] * pb) : NULL;
        if (d_res[p]) { d_ref[p] = d_res[p] + ((size_t)org_y * stride3[p] + org_x) * pb; continue; }
        const size_t bytes = (size_t)(hi - lo + 1) * stride3[p] * pb;
        TF_TRY(svt_hip_hooks_malloc(hip, &d_band[p], bytes + 64));
        TF_TRY(svt_hip_memcpy_h2d(hip, d_band[p], s->ref_plane[f][p] + (size_t)(org_y + lo) * stride3[p] * pb, bytes));
        if (ret == EB_ErrorNone) d_ref[p] = (const uint8_t *)d_band[p] + ((ptrdiff_t)(-lo) * stride3[p] + org_x) * pb;
    }
    TF_TRY(svt_hip_hooks_malloc(hip, &d_jobs, sizeof(SvtHipTfSubpelBlk) * (size_t)n));
    TF_TRY(svt_hip_memcpy_h2d(hip, d_jobs, s->jobs[f], sizeof(SvtHipTfSubpelBlk) * (size_t)n));
    TF_TRY(svt_hip_tf_subpel_frame_dev(hip, pb, bd, (const void *const *)d_src, sstride, d_ref, stride3, s->w.d_pred[f], s->w.pred_stride, s->mi_cols, s->mi_rows,
                                       s->th16, s->tf_hp, c->tf_chroma, (const SvtHipTfSubpelBlk *)d_jobs, n, s->w.d_blocks[f]));
    if (ret == EB_ErrorNone && svt_hip_memcpy_d2h(hip, &s->w.h_blocks[f][0], s->w.d_blocks[f], sizeof(SvtHipTfBlk64)) != SVT_HIP_OK) ret = EB_ErrorUndefined;   /* completes the launch before the band is freed */
    if (ret != EB_ErrorNone) (void)svt_hip_sync(hip);   /* a launch may still be reading the bands / planes released below */
    for (int p = 0; p < 3; p++) {
        svt_hip_hooks_free(hip, d_band[p]);
        if (d_res[p]) svt_hip_resident_release(s->ref_plane[f][p]);
    }
    svt_hip_hooks_free(hip, d_jobs);
    return ret;
}

EbErrorType svt_hip_tf_seg_flush(SvtHipTfSeg *s, const MeContext *c, EbByte *src_start, uint16_t **src16_start, const uint32_t *stride, int bd, const double *noise_levels,
                                 uint64_t *filtered_sse, uint64_t *filtered_sse_uv) {
    SvtHipTfWindow *w = &s->w;
    const int       pb = s->is_highbd ? 2 : 1, np = c->tf_chroma ? 3 : 1;
    const size_t    nblk = (size_t)w->blk_cols * w->blk_rows;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx      *hip = svt_hip_hooks_lock_any();
    if (!hip) { svt_hip_hooks_count(SVT_HIP_HOOK_TF, 0); return EB_ErrorUndefined; }
    EbErrorType ret = EB_ErrorNone;
    void       *d_src[3] = {NULL, NULL, NULL};
    uint8_t    *host[3];
    int         sstride[3];
    s->ctor_done = 1;
    TF_TRY(svt_hip_hooks_malloc(hip, (void **)&w->d_sse, 2 * sizeof(uint64_t)));
    /* the segment's rectangle of the central picture (chroma is only read when tf_chroma is on) */
    for (int p = 0; p < 3; p++) {
        const int bw = p ? 64 >> s->ss_x : 64, bh = p ? 64 >> s->ss_y : 64;
        host[p] = (s->is_highbd ? (uint8_t *)src16_start[p] : src_start[p]) + ((size_t)s->row0 * bh * stride[p] + (size_t)s->col0 * bw) * pb;
        sstride[p] = w->pred_stride[p];
        TF_TRY(svt_hip_hooks_malloc(hip, &d_src[p], s->plane_bytes[p]));
        TF_TRY(svt_hip_memcpy2d_h2d(hip, d_src[p], (size_t)sstride[p] * pb, host[p], (size_t)stride[p] * pb, (size_t)w->blk_cols * bw * pb, (size_t)w->blk_rows * bh));
    }
    int n_subpel = 0;
    for (int f = 0; f < w->n_frames && ret == EB_ErrorNone; f++) {
        if (f == w->index_center) continue;
        TF_TRY(svt_hip_hooks_malloc(hip, (void **)&w->d_blocks[f], nblk * sizeof(SvtHipTfBlk64)));
        for (int p = 0; p < 3; p++) {
            TF_TRY(svt_hip_hooks_malloc(hip, &w->d_pred[f][p], s->plane_bytes[p]));
            if (p < np && !s->n_jobs[f]) TF_TRY(svt_hip_memcpy_h2d(hip, w->d_pred[f][p], s->h_pred[f][p], s->plane_bytes[p]));
        }
        if (s->n_jobs[f] && ret == EB_ErrorNone) {   /* hook "tf_subpel": sub-pel searches + prediction of this frame's blocks, on the device */
            if ((size_t)s->n_jobs[f] != nblk) { ret = EB_ErrorUndefined; break; }   /* every block of the segment visits every frame once */
            ret = tf_subpel_frame(hip, s, f, c, d_src, sstride, bd);
            w->blocks_on_device[f] = 1;
            n_subpel++;
        }
    }
    if (ret == EB_ErrorNone)
        ret = svt_hip_tf_flush_picture(hip, w, c, s->is_highbd, bd, d_src, sstride, d_src, sstride, s->ss_x, s->ss_y, noise_levels, s->decay_control, filtered_sse,
                                       filtered_sse_uv);
    /* the filtered rectangle comes back into host temporaries first and reaches the central picture only when every plane has arrived: a failed copy of a later
     * plane must not leave a half-filtered picture for the C loop that then reruns the segment */
    uint8_t *back[3] = {NULL, NULL, NULL};
    for (int p = 0; p < np; p++) {   /* get_final_filtered_pixels writes chroma only when tf_chroma is on (:1969, :2011) */
        if (ret == EB_ErrorNone && !(back[p] = (uint8_t *)malloc(s->plane_bytes[p]))) ret = EB_ErrorInsufficientResources;
        TF_TRY(svt_hip_memcpy_d2h(hip, back[p], d_src[p], s->plane_bytes[p]));
    }
    for (int p = 0; p < np && ret == EB_ErrorNone; p++) {
        const int bw = p ? 64 >> s->ss_x : 64, bh = p ? 64 >> s->ss_y : 64;
        for (int y = 0; y < w->blk_rows * bh; y++)
            memcpy(host[p] + (size_t)y * stride[p] * pb, back[p] + (size_t)y * sstride[p] * pb, (size_t)w->blk_cols * bw * pb);
    }
    for (int p = 0; p < 3; p++) free(back[p]);
    for (int p = 0; p < 3; p++) svt_hip_hooks_free(hip, d_src[p]);
    if (ret != EB_ErrorNone) SVT_LOG("temporal filter segment on the device failed (%s): C loop for this segment\n", svt_hip_last_error(hip));
    svt_hip_hooks_unlock_any();
    svt_hip_hooks_log("tf: segment of %d x %d blocks, %d frames, one launch", w->blk_cols, w->blk_rows, w->n_frames);
    if (n_subpel) {
        svt_hip_hooks_log("tf_subpel: sub-pel searches and prediction of %d window frames on the device, %d blocks each (no predictor upload)", n_subpel, (int)nblk);
        svt_hip_hooks_count(SVT_HIP_HOOK_TF_SUBPEL, ret == EB_ErrorNone);
    }
    svt_hip_hooks_count(SVT_HIP_HOOK_TF, ret == EB_ErrorNone);
    svt_hip_hooks_time(SVT_HIP_HOOK_TF, t0);
    return ret;
}

int svt_hip_tf_hook_noise(const void *src, int pix_bytes, int bd, int width, int height, int stride, double *sigma) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_TF) || width < 3 || height < 3) return 0;
    SvtHipCtx *hip = svt_hip_hooks_lock_any();
    if (!hip) return 0;
    void   *d_plane = NULL, *d_out = NULL;
    int64_t out[2] = {0, 0};
    int     rc = svt_hip_hooks_malloc(hip, &d_plane, (size_t)width * height * pix_bytes);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_out, sizeof(out));
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy2d_h2d(hip, d_plane, (size_t)width * pix_bytes, src, (size_t)stride * pix_bytes, (size_t)width * pix_bytes, (size_t)height);
    if (rc == SVT_HIP_OK) rc = svt_hip_tf_estimate_noise_dev(hip, d_plane, pix_bytes, bd, width, height, width, (int64_t *)d_out);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, out, d_out, sizeof(out));
    svt_hip_hooks_free(hip, d_plane); svt_hip_hooks_free(hip, d_out);
    svt_hip_hooks_unlock_any();
    if (rc != SVT_HIP_OK) return 0;
    *sigma = svt_hip_tf_noise_sigma(out[0], out[1]);
    svt_hip_hooks_log("tf: noise of a %d x %d plane = %f", width, height, *sigma);
    return 1;
}
