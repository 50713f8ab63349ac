/* svt_hip_me_bridge.c — open-loop ME glue: the SB loop of motion_estimation_kernel
 * (Source/Lib/Encoder/Codec/EbMotionEstimationProcess.c:831-963) run in passes around batched launches: one svt_hip_sad_loop_batch launch
 * per hierarchical-ME level and reference picture, one svt_hip_me_fullpel_frame launch per (reference list, reference picture) of the
 * segment.  Host orchestration only — every SAD is computed by libsvtav1_hip.so.
 *
 * Passes over the SBs of a segment (hook "hme" on, hook "me" on; motion_estimate_sb_hip's hip_phase in brackets):
 *   L0 [10], L1 [11], L2 [12] : hme_level0/1/2_sb (EbMotionEstimation.c:2204, :2333, :2444) with the reference's own search-region arithmetic;
 *                    where hme_level_0/1/2 would call svt_sad_loop_kernel (:998, :1146, :1291) the call is recorded (svt_hip_hme_sad_loop).
 *                    The flush before the next pass runs every recorded search of the level in one launch per reference picture and applies
 *                    what the reference does after the call (SAD doubling of the sub-sampled search, centre + window origin, x4 / x2 / x1).
 *                    MeContext is shared by the SBs of the segment, so the results stay in the batch and are written to the context's
 *                    x/y_hme_levelN_search_center / hme_levelN_sad entries of an SB right before that SB's next pass.
 *   centre + integer windows [2] : set_final_seach_centre_sb (:2575), hme_prune_ref_and_adjust_sr, integer_search_sb (:1922-2066) with
 *                    check_00_center etc. unchanged; where integer_search_sb would call open_loop_me_fullpel_search_sblock (:2130) it
 *                    calls svt_hip_me_record().  The only per-SB state the rest of motion_estimate_sb reads is MeContext::hme_results
 *                    (me_prune_ref :2145, construct_me_candidate_array :2825) and p_sb_best_sad / p_sb_best_mv, so hme_results is saved.
 *   tail [1]       : hme_results restored, the 85 SADs / MVs copied into p_sb_best_sad / p_sb_best_mv[list][ref] — exactly the arrays the C
 *                    kernels update in place — then motion_estimate_sb from me_prune_ref on (:2955-3040), unchanged.
 * With only "me" on the passes are [0] (everything up to the integer windows) and [1]; with only "hme" on they are L0, L1, L2 and [3]
 * (set_final_seach_centre_sb to the end, integer search by the reference's kernels).
 * A HIP failure marks the batch failed and the last pass simply runs the whole unchanged motion_estimate_sb per SB (error convention,
 * SURVEY 8(b)).
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "svt_hip_hooks.h"
#include "EbLog.h"

typedef struct {
    uint32_t   sb_index;
    HmeResults hme[MAX_NUM_OF_REF_PIC_LIST][REF_LIST_MAX_DEPTH];
} MeSbState;

typedef struct {
    const EbPictureBufferDesc *ref;        /* the (decimated) reference picture the window lies in */
    SvtHipSadLoop              job;        /* src_* in the packed source blocks of the level, ref_* in samples of ref->buffer_y */
    uint8_t                    sub, shift; /* sub-sampled search (SAD doubled); centres scaled by 1 << shift */
    int16_t                    x_origin, y_origin, x0, y0; /* window origin; the centre before the call (kept when no candidate wins) */
    uint64_t                  *out_sad;    /* into MeContext: hme_levelN_sad / x,y_hme_levelN_search_center[list][ref][region] */
    int16_t                   *out_x, *out_y;
    uint64_t                   sad;        /* results, post-processed */
    int16_t                    x, y;
} HmeJob;

#define MAX_PASSES 5
struct SvtHipMeBatch {
    uint32_t                   cap, n0, cur;   /* SB slots, SBs seen in the first pass, slot of the SB being processed */
    uint32_t                   seen[MAX_PASSES];
    int                        n_pass, phase[MAX_PASSES];
    int                        failed, sub_sad;
    MeSbState                 *sb;             /* [cap] */
    const EbPictureBufferDesc **win_ref;       /* [list][ref][cap]: the reference picture of each recorded window (the TF loop's slots differ in it) */
    int                        hook_me, hook_hme; /* which counters the flushes feed (SVT_HIP_HOOK_ME / _HME, or SVT_HIP_HOOK_TF_ME for both) */
    uint8_t                   *has;            /* [list][ref][cap]: a window was recorded */
    SvtHipSbSearch            *win;            /* [list][ref][cap] */
    uint32_t                  *best_sad, *best_mv; /* [list][ref][cap][85] */
    /* hierarchical ME */
    HmeJob                    *job;
    uint32_t                   n_job, job_cap, level_first[4]; /* jobs of level L: [level_first[L], level_first[L + 1]) */
    uint32_t                  *sb_first[3], *sb_count[3];     /* [level][slot]: the jobs of one SB are contiguous */
    uint8_t                   *src[3];         /* [level][cap][64][64]: the source block of every SB as the calls read it */
    int                        hme_launches;
    void                      *d_src, *d_ref, *d_job, *d_sad, *d_xy;
    size_t                     d_cap[5];
};
#define SLOT(b, l, r) ((((size_t)(l)) * MAX_REF_IDX + (r)) * (b)->cap)

static __thread SvtHipMeBatch *tls_batch; /* the batch that is collecting integer-search windows on this thread */
static __thread SvtHipMeBatch *tls_hme;   /* the batch that is collecting hierarchical-ME searches on this thread */

int svt_hip_me_batch_passes(const SvtHipMeBatch *b) { return b ? b->n_pass : 1; }

static SvtHipMeBatch *batch_new(uint32_t n_sb, int me, int hme, int hook_me, int hook_hme) {
    SvtHipMeBatch *b = (SvtHipMeBatch *)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->cap = n_sb;
    b->hook_me = hook_me; b->hook_hme = hook_hme;
    if (hme) { b->phase[0] = 10; b->phase[1] = 11; b->phase[2] = 12; b->n_pass = 3; }
    if (me) { b->phase[b->n_pass++] = hme ? 2 : 0; b->phase[b->n_pass++] = 1; }
    else b->phase[b->n_pass++] = 3;
    const size_t slots = (size_t)MAX_NUM_OF_REF_PIC_LIST * MAX_REF_IDX * n_sb;
    b->sb = (MeSbState *)calloc(n_sb, sizeof(MeSbState));
    int ok = b->sb != NULL;
    if (me) {
        b->has = (uint8_t *)calloc(slots, 1);
        b->win = (SvtHipSbSearch *)calloc(slots, sizeof(SvtHipSbSearch));
        b->win_ref = (const EbPictureBufferDesc **)calloc(slots, sizeof(*b->win_ref));
        b->best_sad = (uint32_t *)malloc(slots * SQUARE_PU_COUNT * sizeof(uint32_t));
        b->best_mv = (uint32_t *)malloc(slots * SQUARE_PU_COUNT * sizeof(uint32_t));
        ok = ok && b->has && b->win && b->win_ref && b->best_sad && b->best_mv;
    }
    for (int l = 0; l < 3 && hme; l++) {
        b->sb_first[l] = (uint32_t *)calloc(n_sb, sizeof(uint32_t));
        b->sb_count[l] = (uint32_t *)calloc(n_sb, sizeof(uint32_t));
        b->src[l] = (uint8_t *)calloc((size_t)n_sb * 64 * 64, 1);
        ok = ok && b->sb_first[l] && b->sb_count[l] && b->src[l];
    }
    if (!ok) {
        svt_hip_me_batch_end(b);
        return NULL;
    }
    return b;
}

SvtHipMeBatch *svt_hip_me_batch_begin(PictureParentControlSet *pcs, MeContext *me_ctx, uint32_t n_sb) {
    (void)pcs;
    const int me = svt_hip_hook_enabled(SVT_HIP_HOOK_ME), hme = svt_hip_hook_enabled(SVT_HIP_HOOK_HME) && me_ctx->enable_hme_flag;
    if ((!me && !hme) || !n_sb || me_ctx->me_type == ME_MCTF) return NULL;
    return batch_new(n_sb, me, hme, SVT_HIP_HOOK_ME, SVT_HIP_HOOK_HME);
}

/* The temporal filter's motion search (hook "tf_me"): the slots are the (64x64 block, window frame) pairs of a TF segment, each with its own
 * reference picture; hierarchical levels and integer search are both batched.  enable_hme: MeContext::enable_hme_flag as the TF loop sets it. */
SvtHipMeBatch *svt_hip_me_batch_begin_tf(int enable_hme, uint32_t n_slots) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_TF_ME) || !n_slots) return NULL;
    return batch_new(n_slots, 1, enable_hme != 0, SVT_HIP_HOOK_TF_ME, SVT_HIP_HOOK_TF_ME);
}

void svt_hip_me_batch_end(SvtHipMeBatch *b) {
    if (!b) return;
    if (b->d_src || b->d_ref || b->d_job || b->d_sad || b->d_xy) {
        SvtHipCtx *hip = svt_hip_hooks_lock_any();
        if (hip) {
            svt_hip_hooks_free(hip, b->d_src); svt_hip_hooks_free(hip, b->d_ref); svt_hip_hooks_free(hip, b->d_job); svt_hip_hooks_free(hip, b->d_sad); svt_hip_hooks_free(hip, b->d_xy);
            svt_hip_hooks_unlock_any();
        }
    }
    for (int l = 0; l < 3; l++) { free(b->sb_first[l]); free(b->sb_count[l]); free(b->src[l]); }
    free(b->job);
    free(b->sb); free(b->has); free(b->win); free((void *)b->win_ref); free(b->best_sad); free(b->best_mv);
    free(b);
}

/* ------------------------------------------------------------------ integer search windows */
int svt_hip_me_record(MeContext *me_ctx, uint32_t sb_origin_x, uint32_t sb_origin_y, uint32_t list_index, uint32_t ref_pic_index,
                      const EbPictureBufferDesc *ref_pic, int16_t x_search_area_origin, int16_t y_search_area_origin,
                      int16_t search_area_width, int16_t search_area_height) {
    SvtHipMeBatch *b = tls_batch;
    if (!b) return 0;
    const size_t    s = SLOT(b, list_index, ref_pic_index) + b->cur;
    SvtHipSbSearch *w = &b->win[s];
    w->sb_x = (int32_t)sb_origin_x;
    w->sb_y = (int32_t)sb_origin_y;
    w->x_origin = x_search_area_origin; /* relative to the SB, like the reference's variables of the same name */
    w->y_origin = y_search_area_origin;
    w->width = search_area_width;
    w->height = search_area_height;
    b->has[s] = 1;
    b->win_ref[s] = ref_pic;
    b->sub_sad = me_ctx->me_search_method == SUB_SAD_SEARCH;
    return 1;
}

static int dev_need(SvtHipCtx *hip, void **d, size_t *cap, size_t bytes);
/* svt_hip_me_fullpel_frame on planes that are on the device already: [windows | SADs | MVs] in one staging block of the batch, one upload, one download.  The
 * checks and the choice of the strip-walking instance are the host entry's (svt_hip_api.cpp: negative areas are refused, > 65 536 candidates select the instance). */
static int integer_search_resident(SvtHipCtx *hip, SvtHipMeBatch *b, const uint8_t *d_src, const uint8_t *d_ref, const EbPictureBufferDesc *p,
                                   const SvtHipSbSearch *wins, uint32_t n, uint32_t *sad, uint32_t *mv) {
    int big = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (wins[i].width < 0 || wins[i].height < 0) return SVT_HIP_ERR_BAD_ARG;
        big |= (int)wins[i].width * (int)wins[i].height > 65536;
    }
    if (!n) return SVT_HIP_OK;
    const size_t nres = (size_t)n * SQUARE_PU_COUNT * sizeof(uint32_t);
    const size_t off_sad = (sizeof(SvtHipSbSearch) * (size_t)n + 255) & ~(size_t)255, off_mv = off_sad + ((nres + 255) & ~(size_t)255);
    int          rc = dev_need(hip, &b->d_job, &b->d_cap[2], off_mv + nres + 256);
    uint8_t     *d = (uint8_t *)b->d_job;
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d, wins, sizeof(SvtHipSbSearch) * (size_t)n);
    int big_before = 1;
    if (rc == SVT_HIP_OK) rc = svt_hip_me_get_big_windows(hip, &big_before);
    if (rc == SVT_HIP_OK) rc = svt_hip_me_set_big_windows(hip, big);
    if (rc == SVT_HIP_OK) {
        rc = svt_hip_me_fullpel_frame_dev(hip, d_src, d_ref, p->stride_y, p->origin_x, p->origin_y, (const SvtHipSbSearch *)d, (int)n, b->sub_sad,
                                          (uint32_t *)(d + off_sad), (uint32_t *)(d + off_mv));
        (void)svt_hip_me_set_big_windows(hip, big_before);   /* whatever the context was configured with */
    }
    /* the caller's two result arrays are one block laid out like the device's (flush_integer): one download */
    if (rc == SVT_HIP_OK && (uint8_t *)mv == (uint8_t *)sad + (off_mv - off_sad)) return svt_hip_memcpy_d2h(hip, sad, d + off_sad, off_mv - off_sad + nres);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, sad, d + off_sad, nres);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, mv, d + off_mv, nres);
    return rc;
}

static void flush_integer(SvtHipMeBatch *b, const EbPictureBufferDesc *src_padded) {
    SvtHipSbSearch *wins = (SvtHipSbSearch *)malloc(sizeof(SvtHipSbSearch) * b->cap);
    uint32_t       *idx = (uint32_t *)malloc(sizeof(uint32_t) * b->cap);
    uint32_t       *sad = (uint32_t *)malloc(2 * (sizeof(uint32_t) * SQUARE_PU_COUNT * b->cap + 256));   /* [SADs | pad to 256 | MVs] of a launch, see integer_search_resident */
    if (!wins || !idx || !sad) b->failed = 1;
    for (uint32_t l = 0; l < MAX_NUM_OF_REF_PIC_LIST && !b->failed; l++)
        for (uint32_t r = 0; r < MAX_REF_IDX && !b->failed; r++) {
            const size_t s = SLOT(b, l, r);
            for (uint32_t first = 0; first < b->n0 && !b->failed; first++) {   /* one launch per distinct reference picture among the slots */
                if (!b->has[s + first]) continue;
                const EbPictureBufferDesc *ref = b->win_ref[s + first];
                uint32_t                   n = 0, seen = 0;
                for (uint32_t i = 0; i < first; i++) seen |= b->has[s + i] && b->win_ref[s + i] == ref;
                if (seen) continue;
                for (uint32_t i = first; i < b->n0; i++)
                    if (b->has[s + i] && b->win_ref[s + i] == ref) { wins[n] = b->win[s + i]; idx[n++] = i; }
                /* source and reference are the padded luma pictures of two EbPaReferenceObjects of the same sequence: same geometry */
                if (!ref || ref->stride_y != src_padded->stride_y || ref->origin_x != src_padded->origin_x || ref->origin_y != src_padded->origin_y ||
                    (src_padded->stride_y & 3)) {
                    svt_hip_hooks_log("me: reference picture geometry differs from the source picture: C path for this segment");
                    b->failed = 1;
                    break;
                }
                uint32_t  *mv = (uint32_t *)((uint8_t *)sad + (((size_t)n * SQUARE_PU_COUNT * sizeof(uint32_t) + 255) & ~(size_t)255));
                SvtHipCtx *hip = svt_hip_hooks_lock_any();
                /* both planes resident (SVT_HIP_RESIDENT, svt_hip_hooks.c): only the windows travel; else the row band the windows touch is uploaded per call */
                const size_t   plane_bytes = (size_t)src_padded->stride_y * (size_t)(src_padded->height + 2 * src_padded->origin_y);
                const uint8_t *d_src = hip ? (const uint8_t *)svt_hip_resident_acquire(hip, src_padded->buffer_y, plane_bytes) : NULL;
                const uint8_t *d_ref = d_src ? (const uint8_t *)svt_hip_resident_acquire(hip, ref->buffer_y, plane_bytes) : NULL;
                int            rc = !hip ? SVT_HIP_ERR_NO_DEVICE
                    : d_ref      ? integer_search_resident(hip, b, d_src, d_ref, src_padded, wins, n, sad, mv)
                                 : svt_hip_me_fullpel_frame(hip, src_padded->buffer_y, ref->buffer_y, src_padded->stride_y,
                                                            src_padded->height + 2 * src_padded->origin_y, src_padded->origin_x, src_padded->origin_y, wins, (int)n,
                                                            b->sub_sad, sad, mv);
                if (d_src && rc != SVT_HIP_OK) (void)svt_hip_sync(hip);      /* after a failure a launch may still be reading the planes */
                if (d_ref) svt_hip_resident_release(ref->buffer_y);   /* the results are back: the launch is over */
                if (d_src) svt_hip_resident_release(src_padded->buffer_y);
                if (rc != SVT_HIP_OK)
                    SVT_LOG("svt_hip_me_fullpel_frame failed (%s): C search for this segment\n", hip ? svt_hip_last_error(hip) : "no context");
                if (hip) svt_hip_hooks_unlock_any();
                if (rc != SVT_HIP_OK) { b->failed = 1; break; }
                for (uint32_t k = 0; k < n; k++) {
                    memcpy(&b->best_sad[(s + idx[k]) * SQUARE_PU_COUNT], &sad[(size_t)k * SQUARE_PU_COUNT], SQUARE_PU_COUNT * sizeof(uint32_t));
                    memcpy(&b->best_mv[(s + idx[k]) * SQUARE_PU_COUNT], &mv[(size_t)k * SQUARE_PU_COUNT], SQUARE_PU_COUNT * sizeof(uint32_t));
                }
                svt_hip_hooks_log("me: list %u ref %u, %u windows of one reference picture in one launch", l, r, n);
            }
        }
    free(wins); free(idx); free(sad);
    svt_hip_hooks_count(b->hook_me, !b->failed);
}

/* ------------------------------------------------------------------ hierarchical ME */
int svt_hip_hme_sad_loop(int level, const EbPictureBufferDesc *ref_pic, int16_t x_search_area_origin, int16_t y_search_area_origin,
                         uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height, uint32_t block_width,
                         uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center, uint32_t src_stride_raw,
                         int16_t search_area_width, int16_t search_area_height) {
    SvtHipMeBatch *b = tls_hme;
    if (!b || b->failed) return 0;
    const uint32_t step = src_stride_raw ? ref_stride / src_stride_raw : 0; /* 2 = every other line (hme_search_method != FULL_SAD_SEARCH) */
    const size_t   off = (size_t)(ref - ref_pic->buffer_y);
    if ((step != 1 && step != 2) || step * src_stride_raw != ref_stride || src_stride_raw != ref_pic->stride_y || block_width > 64 ||
        block_height * step > 64 || !block_width || !block_height || search_area_width < 0 || search_area_height < 0 || ref < ref_pic->buffer_y) {
        b->failed = 1; /* a call shape the batch does not describe: the whole segment goes back to the C path */
        svt_hip_hooks_log("hme: level %d call shape outside the batch (block %u x %u, strides %u / %u / %u, area %d x %d): C path for this segment",
                          level, block_width, block_height, src_stride, ref_stride, src_stride_raw, search_area_width, search_area_height);
        return 0;
    }
    if (b->n_job == b->job_cap) {
        const uint32_t cap = b->job_cap ? 2 * b->job_cap : 16 * b->cap + 64;
        HmeJob        *j = (HmeJob *)realloc(b->job, sizeof(HmeJob) * cap);
        if (!j) { b->failed = 1; return 0; }
        b->job = j;
        b->job_cap = cap;
    }
    const uint32_t i = b->cur;
    if (!b->sb_count[level][i]) {
        /* the block as this call reads it: row r of the call at row r * step of the slot, so that row_step = step addresses both operands alike */
        uint8_t *dst = b->src[level] + (size_t)i * 64 * 64;
        for (uint32_t r = 0; r < block_height; r++) memcpy(dst + (size_t)r * step * 64, src + (size_t)r * src_stride, block_width);
        b->sb_first[level][i] = b->n_job;
    }
    b->sb_count[level][i]++;
    HmeJob *j = &b->job[b->n_job++];
    memset(j, 0, sizeof(*j));
    j->ref = ref_pic;
    j->job.src_x = 0;
    j->job.src_y = (int32_t)i * 64;
    j->job.ref_x = (int32_t)(off % ref_pic->stride_y);
    j->job.ref_y = (int32_t)(off / ref_pic->stride_y);
    j->job.bw = (int16_t)block_width;
    j->job.bh = (int16_t)(block_height * step);
    j->job.sa_w = search_area_width;
    j->job.sa_h = search_area_height;
    j->job.row_step = (int16_t)step;
    j->sub = step == 2;
    j->shift = (uint8_t)(2 - level);
    j->x_origin = x_search_area_origin;
    j->y_origin = y_search_area_origin;
    j->x0 = *x_search_center;
    j->y0 = *y_search_center;
    j->out_sad = best_sad;
    j->out_x = x_search_center;
    j->out_y = y_search_center;
    return 1;
}

static int dev_need(SvtHipCtx *hip, void **d, size_t *cap, size_t bytes) {
    if (*cap >= bytes) return SVT_HIP_OK;
    if (*d) svt_hip_hooks_free(hip, *d);
    *d = NULL;
    *cap = 0;
    const int rc = svt_hip_hooks_malloc(hip, d, bytes + bytes / 4);
    if (rc == SVT_HIP_OK) *cap = bytes + bytes / 4;
    return rc;
}

/* what hme_level_0 (:1016-1023), hme_level_1 (:1165-1172) and hme_level_2 (:1309-1314) do with svt_sad_loop_kernel's outputs, in the
 * reference's int16 arithmetic; the centres are written only when a candidate wins (EbComputeSAD_C.c:73) */
static void hme_finish(HmeJob *j, uint32_t sad, int found, int16_t fx, int16_t fy) {
    const int16_t x = found ? fx : j->x0, y = found ? fy : j->y0;
    j->sad = j->sub ? (uint64_t)sad * 2 : sad;
    j->x = (int16_t)((int16_t)(x + j->x_origin) * (1 << j->shift));
    j->y = (int16_t)((int16_t)(y + j->y_origin) * (1 << j->shift));
}

#define HME_TRY(x) do { if (rc == SVT_HIP_OK) rc = (x); } while (0)
typedef struct {
    const EbPictureBufferDesc *ref;
    uint32_t                   first, n;   /* its searches: jobs[first .. first + n) */
    int                        y_lo, y_hi; /* reference rows its windows touch */
    const uint8_t             *d_res;      /* the resident plane, or NULL: the rows travel per launch */
} HmeGroup;
/* One level of a segment: the searches grouped by reference picture (one launch each), ONE upload [searches | initial SADs] and ONE download [SADs | centres]
 * for all of them (it was a pair per reference picture: 3 - 4 round trips per level where a picture has 3 - 4 references). */
static void flush_hme_level(SvtHipMeBatch *b, int level) {
    const uint32_t first = b->level_first[level], end = b->level_first[level + 1];
    if (first == end || b->failed) return;
    const uint32_t n_all = end - first;
    SvtHipSadLoop *jobs = (SvtHipSadLoop *)malloc((sizeof(SvtHipSadLoop) + sizeof(uint32_t)) * n_all);
    uint8_t       *back = (uint8_t *)malloc((sizeof(uint32_t) + 2 * sizeof(int16_t)) * n_all);
    uint32_t      *sel = (uint32_t *)malloc(sizeof(uint32_t) * n_all);
    uint8_t       *done = (uint8_t *)calloc(n_all, 1);
    HmeGroup      *grp = (HmeGroup *)calloc(n_all, sizeof(HmeGroup));
    SvtHipCtx     *hip = (jobs && back && sel && done && grp) ? svt_hip_hooks_lock_any() : NULL;
    int            rc = hip ? SVT_HIP_OK : SVT_HIP_ERR_NO_DEVICE;
    uint32_t       n_grp = 0, n_tot = 0;
    for (uint32_t g = 0; g < n_all && rc == SVT_HIP_OK; g++) {
        if (done[g]) continue;
        HmeGroup *q = &grp[n_grp];
        q->ref = b->job[first + g].ref; /* one launch per reference picture */
        q->first = n_tot; q->n = 0; q->y_lo = INT_MAX; q->y_hi = -1; q->d_res = NULL;
        for (uint32_t k = g; k < n_all; k++) {
            HmeJob *j = &b->job[first + k];
            if (j->ref != q->ref) continue;
            done[k] = 1;
            if (j->job.sa_w < 1 || j->job.sa_h < 1) continue;   /* an empty search area: no candidate, resolved below */
            sel[n_tot] = k;
            jobs[n_tot++] = j->job;
            q->n++;
            if (j->job.ref_y < q->y_lo) q->y_lo = j->job.ref_y;
            const int last = j->job.ref_y + j->job.sa_h - 1 + j->job.bh - 1;
            if (last > q->y_hi) q->y_hi = last;
        }
        if (!q->n) continue;
        if (q->y_hi >= q->ref->height + 2 * q->ref->origin_y) { rc = SVT_HIP_ERR_UNSUPPORTED; break; }
        n_grp++;
    }
    if (rc == SVT_HIP_OK && n_tot) {
        HME_TRY(dev_need(hip, &b->d_src, &b->d_cap[0], (size_t)b->n0 * 64 * 64 + 64));
        HME_TRY(svt_hip_memcpy_h2d(hip, b->d_src, b->src[level], (size_t)b->n0 * 64 * 64));
        /* the (decimated) reference plane resident (SVT_HIP_RESIDENT): the searches address it as recorded; else only the rows the segment's windows touch travel
         * (a segment is a band of SB rows) and the searches are re-based to the band */
        uint32_t n_acq = 0;
        for (; n_acq < n_grp && rc == SVT_HIP_OK; n_acq++) {
            HmeGroup *q = &grp[n_acq];
            q->d_res = (const uint8_t *)svt_hip_resident_acquire(hip, q->ref->buffer_y, (size_t)(q->ref->height + 2 * q->ref->origin_y) * q->ref->stride_y);
            if (!q->d_res)
                for (uint32_t k = 0; k < q->n; k++) jobs[q->first + k].ref_y -= q->y_lo;
        }
        const size_t job_bytes = sizeof(SvtHipSadLoop) * n_tot, sad_bytes = sizeof(uint32_t) * n_tot, xy_bytes = sizeof(int16_t) * 2 * n_tot;
        uint32_t    *sad0 = (uint32_t *)((uint8_t *)jobs + job_bytes);   /* jobs has room for n_all >= n_tot entries of 28 bytes plus their SADs: see the allocation */
        for (uint32_t k = 0; k < n_tot; k++) sad0[k] = 0xffffffu;
        HME_TRY(dev_need(hip, &b->d_job, &b->d_cap[2], job_bytes + sad_bytes + xy_bytes));
        uint8_t *d_sad = (uint8_t *)b->d_job + job_bytes, *d_xy = d_sad + sad_bytes;
        HME_TRY(svt_hip_memcpy_h2d(hip, b->d_job, jobs, job_bytes + sad_bytes));
        int band_in_use = 0;
        for (uint32_t g = 0; g < n_grp && rc == SVT_HIP_OK; g++) {
            const HmeGroup *q = &grp[g];
            if (!q->d_res) {
                const size_t ref_bytes = (size_t)(q->y_hi - q->y_lo + 1) * q->ref->stride_y;
                if (band_in_use) HME_TRY(svt_hip_sync(hip));   /* the previous launch reads the band buffer */
                HME_TRY(dev_need(hip, &b->d_ref, &b->d_cap[1], ref_bytes + 2 * (size_t)q->ref->stride_y + 64));
                HME_TRY(svt_hip_memcpy_h2d(hip, b->d_ref, q->ref->buffer_y + (size_t)q->y_lo * q->ref->stride_y, ref_bytes));
                band_in_use = 1;
            }
            HME_TRY(svt_hip_sad_loop_batch_dev(hip, (const uint8_t *)b->d_src, 64, q->d_res ? q->d_res : (const uint8_t *)b->d_ref, q->ref->stride_y,
                                               (const SvtHipSadLoop *)b->d_job + q->first, (int)q->n, (uint32_t *)d_sad + q->first, (int16_t *)d_xy + 2 * (size_t)q->first));
        }
        HME_TRY(svt_hip_memcpy_d2h(hip, back, d_sad, sad_bytes + xy_bytes));
        /* downloaded: the launches are over; after a failure the context is drained first, a launch may still be reading a plane */
        if (rc != SVT_HIP_OK && n_acq) (void)svt_hip_sync(hip);
        for (uint32_t g = 0; g < n_acq; g++)
            if (grp[g].d_res) svt_hip_resident_release(grp[g].ref->buffer_y);
        if (rc == SVT_HIP_OK) {
            const uint32_t *sad = (const uint32_t *)back;
            const int16_t  *xy = (const int16_t *)(back + sad_bytes);
            for (uint32_t k = 0; k < n_tot; k++) hme_finish(&b->job[first + sel[k]], sad[k], sad[k] != 0xffffffu, xy[2 * k], xy[2 * k + 1]);
            for (uint32_t g = 0; g < n_grp; g++) {
                b->hme_launches++;
                svt_hip_hooks_log("hme: level %d, %u searches of one reference picture in one launch (%d reference rows %s)", level, grp[g].n, grp[g].y_hi - grp[g].y_lo + 1,
                                  grp[g].d_res ? "read from the resident plane" : "uploaded");
            }
        }
    }
    /* A level whose search area the configuration switched down to 0 x 0: svt_sad_loop_kernel leaves the centres as they are, so the level's
     * arithmetic runs on what the previous search through the same pointers left there -- the previous block of this segment, in the reference's
     * block-by-block order; for the segment's first block, what was in the context when the block was recorded. */
    for (uint32_t k = 0; k < n_all && rc == SVT_HIP_OK; k++) {
        HmeJob *j = &b->job[first + k];
        if (j->job.sa_w >= 1 && j->job.sa_h >= 1) continue;
        for (uint32_t p = k; p-- > 0;)
            if (b->job[first + p].out_x == j->out_x) { j->x0 = b->job[first + p].x; j->y0 = b->job[first + p].y; break; }
        hme_finish(j, 0xffffffu, 0, 0, 0);
    }
    if (rc != SVT_HIP_OK) {
        SVT_LOG("hierarchical ME level %d on the device failed (%s): C search for this segment\n", level, hip ? svt_hip_last_error(hip) : "no context");
        b->failed = 1;
    }
    if (hip) svt_hip_hooks_unlock_any();
    free(jobs); free(back); free(sel); free(done); free(grp);
}

/* the level results of SB slot i into the (shared) context, as the reference's calls would have left them */
static void apply_hme(const SvtHipMeBatch *b, uint32_t i, int levels) {
    for (int l = 0; l < levels; l++)
        for (uint32_t k = 0; k < b->sb_count[l][i]; k++) {
            const HmeJob *j = &b->job[b->sb_first[l][i] + k];
            *j->out_sad = j->sad;
            *j->out_x = j->x;
            *j->out_y = j->y;
        }
}

void svt_hip_me_batch_flush(SvtHipMeBatch *b, int next_pass, const EbPictureBufferDesc *src_padded) {
    if (!b || next_pass < 1 || next_pass >= b->n_pass) return;
    const int prev = b->phase[next_pass - 1];
    const long long t0 = svt_hip_hooks_now_ns();
    if (prev >= 10) {
        b->level_first[prev - 10 + 1] = b->n_job;
        flush_hme_level(b, prev - 10);
        if (prev == 12 && b->n_job && b->hook_hme != b->hook_me) svt_hip_hooks_count(b->hook_hme, !b->failed);
        svt_hip_hooks_time(b->hook_hme, t0);   /* the report's svt_hip_hook_time lines: what the batched launches (uploads, launch, download, synchronisation) cost the calling thread */
    } else if (prev == 0 || prev == 2) {
        flush_integer(b, src_padded);
        svt_hip_hooks_time(b->hook_me, t0);
    }
}

int svt_hip_me_batch_sb(SvtHipMeBatch *b, int pass, PictureParentControlSet *pcs, uint32_t sb_index, uint32_t sb_origin_x,
                        uint32_t sb_origin_y, MeContext *me_ctx, EbPictureBufferDesc *input_ptr) {
    return svt_hip_me_batch_slot(b, pass, pcs, sb_index, sb_index, sb_origin_x, sb_origin_y, me_ctx, input_ptr);
}

/* key identifies the slot across the passes (the SB index in the ME loop; block and frame in the TF loop), sb_index is motion_estimate_sb's argument */
int svt_hip_me_batch_slot(SvtHipMeBatch *b, int pass, PictureParentControlSet *pcs, uint32_t key, uint32_t sb_index, uint32_t sb_origin_x,
                          uint32_t sb_origin_y, MeContext *me_ctx, EbPictureBufferDesc *input_ptr) {
    const int      ph = b->phase[pass], last = pass == b->n_pass - 1;
    const uint32_t i = b->seen[pass]++;
    if (pass == 0) {
        if (i >= b->cap) { b->failed = 1; svt_hip_hooks_log("me: more blocks than the batch was opened for (%u): C path", b->cap); }
        else { b->sb[i].sb_index = key; b->n0 = i + 1; }
    } else if (i >= b->n0 || b->sb[i].sb_index != key) {
        if (!b->failed) svt_hip_hooks_log("me: pass %d visits block %u out of order: C path", pass, key);
        b->failed = 1;
    }
    if (b->failed) {
        if (last) motion_estimate_sb_hip(pcs, sb_index, sb_origin_x, sb_origin_y, me_ctx, input_ptr, -1); /* the unchanged C path */
        return last;
    }
    b->cur = i;
    if (ph == 11) apply_hme(b, i, 1);
    else if (ph == 12) apply_hme(b, i, 2);
    else if (ph == 2 || ph == 3) apply_hme(b, i, 3);
    else if (ph == 1) {
        memcpy(me_ctx->hme_results, b->sb[i].hme, sizeof(b->sb[i].hme));
        memset(me_ctx->p_sb_best_mv, 0, sizeof(me_ctx->p_sb_best_mv)); /* motion_estimate_sb's initialisation (:2938-2939) */
        for (uint32_t l = 0; l < MAX_NUM_OF_REF_PIC_LIST; l++)
            for (uint32_t r = 0; r < MAX_REF_IDX; r++) {
                const size_t s = SLOT(b, l, r) + i;
                if (!b->has[s]) continue;
                memcpy(me_ctx->p_sb_best_sad[l][r], &b->best_sad[s * SQUARE_PU_COUNT], SQUARE_PU_COUNT * sizeof(uint32_t));
                memcpy(me_ctx->p_sb_best_mv[l][r], &b->best_mv[s * SQUARE_PU_COUNT], SQUARE_PU_COUNT * sizeof(uint32_t));
            }
    }
    tls_hme = ph >= 10 ? b : NULL;
    tls_batch = (ph == 0 || ph == 2) ? b : NULL;
    motion_estimate_sb_hip(pcs, sb_index, sb_origin_x, sb_origin_y, me_ctx, input_ptr, ph);
    tls_hme = tls_batch = NULL;
    if (ph == 0 || ph == 2) memcpy(b->sb[i].hme, me_ctx->hme_results, sizeof(b->sb[i].hme));
    return last;
}
