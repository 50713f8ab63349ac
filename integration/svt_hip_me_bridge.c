/* svt_hip_me_bridge.c — see svt_hip_me_bridge.h.  Host orchestration in C; all arithmetic happens in libsvtav1_hip.so. */
#include <stdlib.h>
#include <string.h>
#include "svt_hip_me_bridge.h"
#include "EbLog.h"

#define SLOT(p, l, r) (((size_t)(l) * (p)->n_ref + (r)) * (p)->n_sb)

EbErrorType svt_hip_me_picture_ctor(SvtHipMePicture *p, const PictureParentControlSet *pcs) {
    memset(p, 0, sizeof(*p));
    p->n_sb   = pcs->sb_total_count;
    p->n_list = MAX_NUM_OF_REF_PIC_LIST;
    p->n_ref  = MAX_REF_IDX;
    const size_t slots = (size_t)p->n_list * p->n_ref * p->n_sb;
    p->win      = (SvtHipSbSearch *)calloc(slots, sizeof(SvtHipSbSearch));
    p->best_sad = (uint32_t *)malloc(slots * SQUARE_PU_COUNT * sizeof(uint32_t));
    p->best_mv  = (uint32_t *)malloc(slots * SQUARE_PU_COUNT * sizeof(uint32_t));
    if (!p->win || !p->best_sad || !p->best_mv) {
        svt_hip_me_picture_dctor(p);
        return EB_ErrorInsufficientResources;
    }
    return EB_ErrorNone;
}

void svt_hip_me_picture_dctor(SvtHipMePicture *p) {
    free(p->win); free(p->best_sad); free(p->best_mv);
    memset(p, 0, sizeof(*p));
}

void svt_hip_me_record_window(SvtHipMePicture *p, uint32_t sb_index, uint32_t sb_origin_x, uint32_t sb_origin_y, uint32_t list_index,
                              uint32_t ref_pic_index, int16_t x_search_area_origin, int16_t y_search_area_origin,
                              int16_t search_area_width, int16_t search_area_height) {
    SvtHipSbSearch *w = &p->win[SLOT(p, list_index, ref_pic_index) + sb_index];
    w->sb_x = (int32_t)sb_origin_x;
    w->sb_y = (int32_t)sb_origin_y;
    w->x_origin = x_search_area_origin;   /* relative to the SB, like the reference's variables of the same name */
    w->y_origin = y_search_area_origin;
    w->width  = search_area_width;
    w->height = search_area_height;
}

EbErrorType svt_hip_me_flush_picture(SvtHipCtx *hip, SvtHipMePicture *p, const EbPictureBufferDesc *src_padded,
                                     EbPictureBufferDesc *const ref_padded[MAX_NUM_OF_REF_PIC_LIST][MAX_REF_IDX], EbBool sub_sad) {
    for (uint32_t l = 0; l < p->n_list; l++)
        for (uint32_t r = 0; r < p->n_ref; r++) {
            const EbPictureBufferDesc *ref = ref_padded[l][r];
            const size_t               s   = SLOT(p, l, r);
            uint32_t                   any = 0;
            for (uint32_t i = 0; i < p->n_sb; i++) any |= (uint32_t)(p->win[s + i].width > 0);
            if (!ref || !any)
                continue;
            /* source and reference pictures share geometry in the ME process (EbMotionEstimationProcess.c:800-830);
             * SBs whose window has width 0 are returned with MAX_SAD_VALUE by the library and ignored by phase 3 */
            const int rc = svt_hip_me_fullpel_frame(hip, src_padded->buffer_y, ref->buffer_y, src_padded->stride_y,
                                                    src_padded->height + 2 * src_padded->origin_y, src_padded->origin_x, src_padded->origin_y,
                                                    &p->win[s], (int)p->n_sb, sub_sad ? 1 : 0, &p->best_sad[s * SQUARE_PU_COUNT],
                                                    &p->best_mv[s * SQUARE_PU_COUNT]);
            if (rc != SVT_HIP_OK) {
                SVT_LOG("svt_hip_me_fullpel_frame failed (%s): falling back to the C search for this picture\n", svt_hip_last_error(hip));
                return EB_ErrorUndefined;   /* error convention: never through a kernel pointer, the caller keeps its C loop */
            }
        }
    return EB_ErrorNone;
}

void svt_hip_me_fetch_sb(const SvtHipMePicture *p, uint32_t sb_index, uint32_t list_index, uint32_t ref_pic_index, MeContext *context_ptr) {
    const size_t o = (SLOT(p, list_index, ref_pic_index) + sb_index) * SQUARE_PU_COUNT;
    memcpy(context_ptr->p_sb_best_sad[list_index][ref_pic_index], &p->best_sad[o], SQUARE_PU_COUNT * sizeof(uint32_t));
    memcpy(context_ptr->p_sb_best_mv[list_index][ref_pic_index], &p->best_mv[o], SQUARE_PU_COUNT * sizeof(uint32_t));
    /* the pointers integer_search_sb keeps into these arrays (EbMotionEstimation.c:2080-2110) stay valid: same storage */
}
