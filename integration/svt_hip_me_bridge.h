/*
 * svt_hip_me_bridge.h — reference-side glue for SURVEY 8(f) rank 1: the open-loop ME process loop
 * (Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c:831-963 -> motion_estimate_sb -> integer_search_sb,
 * EbMotionEstimation.c:1868) driven through the batched ABI of include/svt_hip.h.
 *
 * This file is meant to be compiled INTO libSvtAv1Enc (it includes the reference's headers); it is not part of
 * libsvtav1_hip.so.  tests/test_integration_compiles.py syntax-checks it against /root/reference when that tree exists.
 *
 * The per-SB loop becomes three phases per picture:
 *   1. per SB (unchanged host code): HME, search-centre selection, window arithmetic of integer_search_sb
 *      (:1922-2066) -- but instead of calling open_loop_me_fullpel_search_sblock (:2185) the patched
 *      integer_search_sb calls svt_hip_me_record_window();
 *   2. once per picture: svt_hip_me_flush_picture() = one svt_hip_me_fullpel_frame launch per (list, reference);
 *   3. per SB (unchanged host code from me_prune_ref, :2957, onwards) after svt_hip_me_fetch_sb() has put the SB's
 *      85 SADs / MVs back into MeContext::p_sb_best_sad / p_sb_best_mv, exactly where the C kernels leave them.
 */
#ifndef SVT_HIP_ME_BRIDGE_H
#define SVT_HIP_ME_BRIDGE_H

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbMotionEstimationContext.h"
#include "svt_hip.h"

typedef struct SvtHipMePicture {
    uint32_t        n_sb;
    uint32_t        n_list, n_ref;   /* MAX_NUM_OF_REF_PIC_LIST x MAX_REF_IDX slots */
    SvtHipSbSearch *win;             /* [list][ref][n_sb]; width == 0: reference not searched for that SB (do_ref == 0) */
    uint32_t       *best_sad;        /* [list][ref][n_sb][85] */
    uint32_t       *best_mv;
} SvtHipMePicture;

EbErrorType svt_hip_me_picture_ctor(SvtHipMePicture *p, const PictureParentControlSet *pcs);
void        svt_hip_me_picture_dctor(SvtHipMePicture *p);

/* phase 1: called from integer_search_sb in place of open_loop_me_fullpel_search_sblock */
void svt_hip_me_record_window(SvtHipMePicture *p, uint32_t sb_index, uint32_t sb_origin_x, uint32_t sb_origin_y, uint32_t list_index,
                              uint32_t ref_pic_index, int16_t x_search_area_origin, int16_t y_search_area_origin,
                              int16_t search_area_width, int16_t search_area_height);

/* phase 2: every recorded (list, reference) of the picture; returns EB_ErrorNone or EB_ErrorUndefined (caller re-runs the C loop) */
EbErrorType svt_hip_me_flush_picture(SvtHipCtx *hip, SvtHipMePicture *p, const EbPictureBufferDesc *src_padded,
                                     EbPictureBufferDesc *const ref_padded[MAX_NUM_OF_REF_PIC_LIST][MAX_REF_IDX], EbBool sub_sad);

/* phase 3: results of one SB back into the per-thread context */
void svt_hip_me_fetch_sb(const SvtHipMePicture *p, uint32_t sb_index, uint32_t list_index, uint32_t ref_pic_index, MeContext *context_ptr);

#endif
