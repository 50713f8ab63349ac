/* svt_hip_tf_bridge.c — see svt_hip_tf_bridge.h.  Host code only; the pixel work is one svt_hip_tf_filter_frame_dev launch per central picture. */
#include <stdlib.h>
#include <string.h>
#include "svt_hip_tf_bridge.h"

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
        HIP_TRY(svt_hip_malloc(hip, (void **)&w->d_blocks[f], nblk * sizeof(SvtHipTfBlk64)));
        for (int p = 0; p < 3; p++) {
            const size_t rows = (size_t)(w->blk_rows * 64) >> (p ? ss_y : 0);
            HIP_TRY(svt_hip_malloc(hip, &w->d_pred[f][p], (size_t)w->pred_stride[p] * rows * pb));
        }
    }
    HIP_TRY(svt_hip_malloc(hip, (void **)&w->d_sse, 2 * sizeof(uint64_t)));
    return EB_ErrorNone;
}

void svt_hip_tf_window_dctor(SvtHipCtx *hip, SvtHipTfWindow *w) {
    for (int f = 0; f < SVT_HIP_TF_MAX_REFS; f++) {
        free(w->h_blocks[f]);
        svt_hip_free(hip, w->d_blocks[f]);
        for (int p = 0; p < 3; p++) svt_hip_free(hip, w->d_pred[f][p]);
    }
    svt_hip_free(hip, w->d_sse);
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
        HIP_TRY(svt_hip_memcpy_h2d(hip, w->d_blocks[f], w->h_blocks[f], nblk * sizeof(SvtHipTfBlk64)));
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
