/* svt_hip_me_bridge.c — open-loop ME glue: the SB loop of motion_estimation_kernel
 * (Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c:831-963) run in two passes around ONE svt_hip_me_fullpel_frame launch per
 * (reference list, reference picture) of the segment.  Host orchestration only — every SAD is computed by libsvtav1_hip.so.
 *
 *   pass 0, per SB : motion_estimate_sb up to and including integer_search_sb (EbMotionEstimation.c:2912-2953): HME, reference pruning,
 *                    the search-window arithmetic (:1922-2066) and check_00_center stay the reference's code; where integer_search_sb
 *                    would call open_loop_me_fullpel_search_sblock (:2130) it calls svt_hip_me_record() instead.  The only per-SB state
 *                    the rest of motion_estimate_sb reads is MeContext::hme_results (me_prune_ref :2145, construct_me_candidate_array
 *                    :2825) and p_sb_best_sad / p_sb_best_mv, so hme_results is saved per SB.
 *   flush          : the recorded windows of one (list, ref) = one launch; results [n][85] SAD / MV words.
 *   pass 1, per SB : hme_results restored, the 85 SADs / MVs copied into p_sb_best_sad / p_sb_best_mv[list][ref] — exactly the arrays the C
 *                    kernels update in place — then motion_estimate_sb from me_prune_ref on (:2955-3040), unchanged.
 * A HIP failure marks the batch failed and pass 1 simply runs the whole unchanged motion_estimate_sb per SB (error convention).
 */
#include <stdlib.h>
#include <string.h>
#include "svt_hip_hooks.h"
#include "EbLog.h"

typedef struct {
    uint32_t   sb_index;
    HmeResults hme[MAX_NUM_OF_REF_PIC_LIST][REF_LIST_MAX_DEPTH];
} MeSbState;

struct SvtHipMeBatch {
    uint32_t                   cap, n0, n1;    /* SB slots, SBs seen in pass 0 / pass 1 */
    int                        failed, sub_sad;
    MeSbState                 *sb;             /* [cap] */
    const EbPictureBufferDesc *ref_pic[MAX_NUM_OF_REF_PIC_LIST][MAX_REF_IDX];
    uint8_t                   *has;            /* [list][ref][cap]: a window was recorded */
    SvtHipSbSearch            *win;            /* [list][ref][cap] */
    uint32_t                  *best_sad, *best_mv; /* [list][ref][cap][85] */
};
#define SLOT(b, l, r) ((((size_t)(l)) * MAX_REF_IDX + (r)) * (b)->cap)

static __thread SvtHipMeBatch *tls_batch; /* the batch that is collecting windows on this thread (pass 0 only) */

SvtHipMeBatch *svt_hip_me_batch_begin(PictureParentControlSet *pcs, MeContext *me_ctx, uint32_t n_sb) {
    (void)pcs;
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_ME) || !n_sb || me_ctx->me_type == ME_MCTF) return NULL;
    SvtHipMeBatch *b = (SvtHipMeBatch *)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->cap = n_sb;
    const size_t slots = (size_t)MAX_NUM_OF_REF_PIC_LIST * MAX_REF_IDX * n_sb;
    b->sb = (MeSbState *)calloc(n_sb, sizeof(MeSbState));
    b->has = (uint8_t *)calloc(slots, 1);
    b->win = (SvtHipSbSearch *)calloc(slots, sizeof(SvtHipSbSearch));
    b->best_sad = (uint32_t *)malloc(slots * SQUARE_PU_COUNT * sizeof(uint32_t));
    b->best_mv = (uint32_t *)malloc(slots * SQUARE_PU_COUNT * sizeof(uint32_t));
    if (!b->sb || !b->has || !b->win || !b->best_sad || !b->best_mv) {
        svt_hip_me_batch_end(b);
        return NULL;
    }
    return b;
}

void svt_hip_me_batch_end(SvtHipMeBatch *b) {
    if (!b) return;
    free(b->sb); free(b->has); free(b->win); free(b->best_sad); free(b->best_mv);
    free(b);
}

int svt_hip_me_record(MeContext *me_ctx, uint32_t sb_origin_x, uint32_t sb_origin_y, uint32_t list_index, uint32_t ref_pic_index,
                      const EbPictureBufferDesc *ref_pic, int16_t x_search_area_origin, int16_t y_search_area_origin,
                      int16_t search_area_width, int16_t search_area_height) {
    SvtHipMeBatch *b = tls_batch;
    if (!b) return 0;
    const size_t    s = SLOT(b, list_index, ref_pic_index) + b->n0;
    SvtHipSbSearch *w = &b->win[s];
    w->sb_x = (int32_t)sb_origin_x;
    w->sb_y = (int32_t)sb_origin_y;
    w->x_origin = x_search_area_origin; /* relative to the SB, like the reference's variables of the same name */
    w->y_origin = y_search_area_origin;
    w->width = search_area_width;
    w->height = search_area_height;
    b->has[s] = 1;
    b->ref_pic[list_index][ref_pic_index] = ref_pic;
    b->sub_sad = me_ctx->me_search_method == SUB_SAD_SEARCH;
    return 1;
}

void svt_hip_me_batch_flush(SvtHipMeBatch *b, const EbPictureBufferDesc *src_padded) {
    SvtHipSbSearch *wins = (SvtHipSbSearch *)malloc(sizeof(SvtHipSbSearch) * b->cap);
    uint32_t       *idx = (uint32_t *)malloc(sizeof(uint32_t) * b->cap);
    uint32_t       *sad = (uint32_t *)malloc(sizeof(uint32_t) * SQUARE_PU_COUNT * b->cap);
    uint32_t       *mv = (uint32_t *)malloc(sizeof(uint32_t) * SQUARE_PU_COUNT * b->cap);
    if (!wins || !idx || !sad || !mv) b->failed = 1;
    for (uint32_t l = 0; l < MAX_NUM_OF_REF_PIC_LIST && !b->failed; l++)
        for (uint32_t r = 0; r < MAX_REF_IDX && !b->failed; r++) {
            const size_t s = SLOT(b, l, r);
            uint32_t     n = 0;
            for (uint32_t i = 0; i < b->n0; i++)
                if (b->has[s + i]) { wins[n] = b->win[s + i]; idx[n++] = i; }
            if (!n) continue;
            const EbPictureBufferDesc *ref = b->ref_pic[l][r];
            /* source and reference are the padded luma pictures of two EbPaReferenceObjects of the same sequence: same geometry */
            if (!ref || ref->stride_y != src_padded->stride_y || ref->origin_x != src_padded->origin_x || ref->origin_y != src_padded->origin_y ||
                (src_padded->stride_y & 3)) { b->failed = 1; break; }
            SvtHipCtx *hip = svt_hip_hooks_lock();
            int        rc = hip ? svt_hip_me_fullpel_frame(hip, src_padded->buffer_y, ref->buffer_y, src_padded->stride_y,
                                                           src_padded->height + 2 * src_padded->origin_y, src_padded->origin_x,
                                                           src_padded->origin_y, wins, (int)n, b->sub_sad, sad, mv)
                                : SVT_HIP_ERR_NO_DEVICE;
            if (rc != SVT_HIP_OK)
                SVT_LOG("svt_hip_me_fullpel_frame failed (%s): C search for this segment\n", hip ? svt_hip_last_error(hip) : "no context");
            if (hip) svt_hip_hooks_unlock();
            if (rc != SVT_HIP_OK) { b->failed = 1; break; }
            for (uint32_t k = 0; k < n; k++) {
                memcpy(&b->best_sad[(s + idx[k]) * SQUARE_PU_COUNT], &sad[(size_t)k * SQUARE_PU_COUNT], SQUARE_PU_COUNT * sizeof(uint32_t));
                memcpy(&b->best_mv[(s + idx[k]) * SQUARE_PU_COUNT], &mv[(size_t)k * SQUARE_PU_COUNT], SQUARE_PU_COUNT * sizeof(uint32_t));
            }
            svt_hip_hooks_log("me: list %u ref %u, %u SB windows in one launch", l, r, n);
        }
    free(wins); free(idx); free(sad); free(mv);
    svt_hip_hooks_count(SVT_HIP_HOOK_ME, !b->failed);
}

int svt_hip_me_batch_sb(SvtHipMeBatch *b, int pass, PictureParentControlSet *pcs, uint32_t sb_index, uint32_t sb_origin_x,
                        uint32_t sb_origin_y, MeContext *me_ctx, EbPictureBufferDesc *input_ptr) {
    if (pass == 0) {
        if (b->n0 >= b->cap) { b->failed = 1; return 0; }
        b->sb[b->n0].sb_index = sb_index;
        tls_batch = b;
        motion_estimate_sb_hip(pcs, sb_index, sb_origin_x, sb_origin_y, me_ctx, input_ptr, 0);
        tls_batch = NULL;
        memcpy(b->sb[b->n0].hme, me_ctx->hme_results, sizeof(b->sb[b->n0].hme));
        b->n0++;
        return 0;
    }
    const uint32_t i = b->n1++;
    if (b->failed || i >= b->n0 || b->sb[i].sb_index != sb_index) {
        motion_estimate_sb_hip(pcs, sb_index, sb_origin_x, sb_origin_y, me_ctx, input_ptr, -1); /* the unchanged C path */
        return 1;
    }
    memcpy(me_ctx->hme_results, b->sb[i].hme, sizeof(b->sb[i].hme));
    memset(me_ctx->p_sb_best_mv, 0, sizeof(me_ctx->p_sb_best_mv)); /* motion_estimate_sb's initialisation (:2938-2939) */
    for (uint32_t l = 0; l < MAX_NUM_OF_REF_PIC_LIST; l++)
        for (uint32_t r = 0; r < MAX_REF_IDX; r++) {
            const size_t s = SLOT(b, l, r) + i;
            if (!b->has[s]) continue;
            memcpy(me_ctx->p_sb_best_sad[l][r], &b->best_sad[s * SQUARE_PU_COUNT], SQUARE_PU_COUNT * sizeof(uint32_t));
            memcpy(me_ctx->p_sb_best_mv[l][r], &b->best_mv[s * SQUARE_PU_COUNT], SQUARE_PU_COUNT * sizeof(uint32_t));
        }
    motion_estimate_sb_hip(pcs, sb_index, sb_origin_x, sb_origin_y, me_ctx, input_ptr, 1);
    return 1;
}
