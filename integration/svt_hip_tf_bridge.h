/*
 * svt_hip_tf_bridge.h — reference-side glue for the alt-ref temporal filter: produce_temporally_filtered_pic
 * (Source/Lib/Encoder/Codec/EbTemporalFiltering.c:2012-2412) driven through svt_hip_tf_filter_frame_dev.
 *
 * Compiled INTO libSvtAv1Enc (includes the reference's headers).  Two users:
 *   - the hooked encoder (hook "tf", integration/patch_reference.py): svt_hip_tf_seg_* below — one TF segment at a time, the reference's own
 *     motion search and tf_inter_prediction, the pixel side (central / plane-wise filter, normalisation) in one launch per segment;
 *   - a picture-level driver with device-side prediction (svt_hip_tf_window_* / record_block / flush_picture), the form a resident pipeline uses.
 *
 * The per-64x64-block loop keeps its motion search (motion_estimate_sb, tf_32x32 / tf_16x16_sub_pel_search, derive_tf_32x32_block_split_flag:
 * host logic on top of the ME / sub-pel entry points) but
 *   - instead of tf_inter_prediction into the 64x64 `pred` block buffer it predicts into a per-frame predictor PICTURE on the device
 *     (svt_hip_subpel_predict_batch_dev with the block's MVs), and calls svt_hip_tf_record_block(), which copies the MeContext TF fields;
 *   - apply_filtering_central / apply_filtering_block_plane_wise / get_final_filtered_pixels are dropped: after the last block of the
 *     picture, svt_hip_tf_flush_picture() filters the whole picture over the whole window in one launch.
 */
#ifndef SVT_HIP_TF_BRIDGE_H
#define SVT_HIP_TF_BRIDGE_H

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbMotionEstimationContext.h"
#include "svt_hip.h"

typedef struct SvtHipTfWindow {
    int            n_frames, index_center;       /* past_altref_nframes + future_altref_nframes + 1, position of the central picture */
    int            blk_cols, blk_rows;            /* 64x64 blocks of the (64-aligned) picture */
    SvtHipTfBlk64 *h_blocks[SVT_HIP_TF_MAX_REFS]; /* host staging, one array per window frame */
    SvtHipTfBlk64 *d_blocks[SVT_HIP_TF_MAX_REFS];
    void          *d_pred[SVT_HIP_TF_MAX_REFS][3]; /* predictor pictures (device), same geometry as the central picture */
    int            pred_stride[3];
    uint64_t      *d_sse;                          /* filtered_sse, filtered_sse_uv */
    uint8_t        blocks_on_device[SVT_HIP_TF_MAX_REFS]; /* d_blocks[f] was written by svt_hip_tf_subpel_frame_dev: h_blocks[f] is not uploaded */
} SvtHipTfWindow;

EbErrorType svt_hip_tf_window_ctor(SvtHipCtx *hip, SvtHipTfWindow *w, int n_frames, int index_center, int width, int height, int is_16bit, int ss_x, int ss_y);
void        svt_hip_tf_window_dctor(SvtHipCtx *hip, SvtHipTfWindow *w);

/* after derive_tf_32x32_block_split_flag for (frame_index, 64x64 block): keep what the plane-wise filter reads from MeContext */
void svt_hip_tf_record_block(SvtHipTfWindow *w, int frame_index, uint32_t blk_row, uint32_t blk_col, const MeContext *context_ptr);

/* once per central picture; d_src / d_dst: the central picture on the device (may alias), strides in samples.  noise_levels / decay_control as
 * computed by svt_av1_init_temporal_filtering (:2786-2870); filtered_sse / filtered_sse_uv receive get_final_filtered_pixels' sums. */
EbErrorType svt_hip_tf_flush_picture(SvtHipCtx *hip, SvtHipTfWindow *w, const MeContext *context_ptr, int is_16bit, int bd, void *const d_src[3],
                                     const int src_stride[3], void *const d_dst[3], const int dst_stride[3], int ss_x, int ss_y, const double *noise_levels,
                                     int decay_control, uint64_t *filtered_sse, uint64_t *filtered_sse_uv);


/* ------------------------------------------------------------------ hook "tf": one TF segment of produce_temporally_filtered_pic
 * (Source/Lib/Encoder/Codec/EbTemporalFiltering.c:2038-2412).  The block loop keeps Step 1 (motion_estimate_sb, tf_32x32 / tf_16x16_sub_pel_search,
 * derive_tf_32x32_block_split_flag, tf_inter_prediction) and hands every (frame, 64x64 block) predictor + the MeContext TF fields to
 * svt_hip_tf_seg_block instead of running Step 2 (apply_filtering_central / apply_filtering_block_plane_wise) and get_final_filtered_pixels;
 * svt_hip_tf_seg_flush then filters the segment's rectangle over the whole window on the device and writes it into the central picture.
 * Nothing of the central picture changes before the flush succeeded, so a failure simply reruns the segment's unchanged C loop. */
typedef struct SvtHipTfSeg SvtHipTfSeg;
/* NULL = hook off (or no memory): the caller runs its unchanged loop.  [col0, col1) x [row0, row1) = the segment's 64x64 blocks. */
SvtHipTfSeg *svt_hip_tf_seg_begin(int n_frames, int index_center, uint32_t col0, uint32_t col1, uint32_t row0, uint32_t row1, int is_highbd, int ss_x, int ss_y);
void         svt_hip_tf_seg_block(SvtHipTfSeg *s, int frame_index, uint32_t blk_row, uint32_t blk_col, const MeContext *context_ptr, EbByte *pred,
                                  uint16_t **pred_16bit, const uint32_t *stride_pred, int decay_control);
/* hook "tf_subpel", right after the motion search of (frame, block) — motion_estimate_sb or the last pass of the batched search: 1 = recorded, the caller skips
 * tf_32x32_sub_pel_search, tf_16x16_sub_pel_search, derive_tf_32x32_block_split_flag and tf_inter_prediction (EbTemporalFiltering.c:2272-2315), which then run
 * on the device inside svt_hip_tf_seg_flush — one svt_hip_tf_subpel_frame_dev launch per window frame — and leave predictor and TF fields where the filter
 * launch reads them; 0 = hook off / not 4:2:0: the caller runs them as before.  pcs_ref / pic_ref: list_picture_control_set_ptr / list_input_picture_ptr
 * [frame_index]. */
int          svt_hip_tf_seg_subpel(SvtHipTfSeg *s, int frame_index, uint32_t blk_row, uint32_t blk_col, const MeContext *context_ptr,
                                   const PictureParentControlSet *pcs_central, const PictureParentControlSet *pcs_ref, const EbPictureBufferDesc *pic_ref,
                                   uint32_t sb_origin_x, uint32_t sb_origin_y);
/* src_start / src16_start: the central picture's planes at sample (0, 0) (8-bit planes, or altref_buffer_highbd), stride[] in samples */
EbErrorType  svt_hip_tf_seg_flush(SvtHipTfSeg *s, const MeContext *context_ptr, EbByte *src_start, uint16_t **src16_start, const uint32_t *stride, int bd,
                                  const double *noise_levels, uint64_t *filtered_sse, uint64_t *filtered_sse_uv);
void         svt_hip_tf_seg_end(SvtHipTfSeg *s);
/* estimate_noise / estimate_noise_highbd (EbTemporalFiltering.c:2416, :2451), first statement of both: 1 = *sigma holds the result of
 * svt_hip_tf_estimate_noise_dev + svt_hip_tf_noise_sigma, 0 = hook off or a failure (the function continues with its own loop). */
int          svt_hip_tf_hook_noise(const void *src, int pix_bytes, int bd, int width, int height, int stride, double *sigma);

#endif
