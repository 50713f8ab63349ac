/*
 * svt_hip_hooks.h — the functions the patched reference process loops call (integration/patch_reference.py shows every call site).
 *
 * Reference-side glue, compiled INTO libSvtAv1Enc together with svt_hip_me_bridge.c / svt_hip_lf_bridge.c; it is not part of
 * libsvtav1_hip.so.  Every hook keeps the reference's error convention: a hook that is disabled or fails returns
 * "not handled" and the caller runs its unchanged C loop, so a HIP failure can never surface through a kernel pointer.
 *
 * Which hooks are active is a run-time choice, so that a bitstream mismatch bisects to a stage:
 *   SVT_HIP_HOOKS = comma list of  pa, tf, tf_me, tf_subpel, hme, me, cdef_finish, dlf, dlf_search, cdef_search, cdef_apply, sgr_search, wiener_stats, wiener_try, rest_apply  |  all  |  none
 *                   (+ the opt-in hooks, which "all" does not select: md_pre, md_tx, md_subpel, encdec_tx, encdec_sb)
 *   SVT_HIP_RTCD  = comma list of per-call dispatch-table entries to replace by their svt_*_hip wrapper
 *                   (include/svt_hip_rtcd.h), e.g. "svt_sad_loop_kernel,svt_av1_selfguided_restoration"  |  all
 *   SVT_HIP_DEVICE = GPU ordinal (default 0);  SVT_HIP_VERBOSE=1 logs every hooked call.
 *   SVT_HIP_CONTEXTS = contexts of the source-side bridges' pool (default 4), SVT_HIP_ALLOC_CACHE_MB = their block cache (default 1024),
 *   SVT_HIP_RESIDENT=0 = source-side planes are uploaded per segment again (default since round 4: they stay on the device between their writes; SVT_HIP_RESIDENT_MB, default 16384) — svt_hip_hooks.c
 * Unset / empty SVT_HIP_HOOKS = none: the patched encoder then IS the reference encoder.
 */
#ifndef SVT_HIP_HOOKS_H
#define SVT_HIP_HOOKS_H

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbPictureBufferDesc.h"
#include "EbMotionEstimationContext.h"
#include "svt_hip.h"
#include "svt_hip_resident.h"

enum {
    SVT_HIP_HOOK_ME = 0,       /* integer full search of every SB of an ME segment: motion_estimation_kernel (EbMotionEstimationProcess.c:831-963) */
    SVT_HIP_HOOK_DLF,          /* svt_av1_loop_filter_frame in dlf_kernel (EbDlfProcess.c:212) */
    SVT_HIP_HOOK_DLF_SEARCH,   /* svt_av1_pick_filter_level(LPF_PICK_FROM_FULL_IMAGE) (EbDlfProcess.c:203) */
    SVT_HIP_HOOK_CDEF_SEARCH,  /* cdef_seg_search[16bit] of every segment (EbCdefProcess.c:511-514) */
    SVT_HIP_HOOK_CDEF_APPLY,   /* svt_av1_cdef_frame / av1_cdef_frame16bit (EbCdefProcess.c:531-533) */
    SVT_HIP_HOOK_SGR_SEARCH,   /* search_sgrproj_seg of every unit (EbRestorationPick.c:1277, via restoration_seg_search :1537) */
    SVT_HIP_HOOK_WIENER_STATS, /* svt_av1_compute_stats[_highbd] of every unit in search_wiener_seg (EbRestorationPick.c:1347) */
    SVT_HIP_HOOK_REST_APPLY,   /* svt_av1_loop_restoration_filter_frame (EbRestProcess.c:548) */
    SVT_HIP_HOOK_WIENER_TRY,   /* every try_restoration_unit_seg probe of finer_tile_search_wiener_seg (EbRestorationPick.c:1092, :137) */
    SVT_HIP_HOOK_WIENER_SEARCH, /* every search_wiener_seg of the picture at once: statistics, initial filters, the tap refinement of all units in lockstep (:1347);
                                 * takes precedence over wiener_stats / wiener_try (those hook the same work unit by unit) */
    SVT_HIP_HOOK_HME,          /* every svt_sad_loop_kernel search of hme_level_0 / 1 / 2 of an ME segment, one launch per level and reference picture
                                * (EbMotionEstimation.c:998, :1146, :1291) */
    SVT_HIP_HOOK_TF,           /* Step 2 + get_final_filtered_pixels of every TF segment: produce_temporally_filtered_pic (EbTemporalFiltering.c:2038-2412),
                                * glue in svt_hip_tf_bridge.c */
    SVT_HIP_HOOK_PA,           /* picture analysis: the HME pyramids and the per-SB mean / variance pyramid (EbPictureAnalysisProcess.c:3312, :3606, :2929) */
    SVT_HIP_HOOK_TF_ME,        /* the motion search of the temporal filter: HME levels and integer search of every (block, frame) of a TF segment batched like
                                * the open-loop ME (EbTemporalFiltering.c:2264, motion_estimate_sb with ME_MCTF) */
    SVT_HIP_HOOK_CDEF_FINISH,  /* joint_strength_search_dual of finish_cdef_search: the strength-pair selection steps back to back on the device (EbEncCdef.c:1140, :1258) */
    SVT_HIP_HOOK_MD_TX,        /* mode decision: the forward transforms of a transform block for every type tx_type_search tries, one launch (EbProductCodingLoop.c:4258).
                                * Opt-in: not part of SVT_HIP_HOOKS=all (a launch per transform block of every candidate is the slow way to use a GPU; it exists to
                                * put the mode-decision side of the path behind the batched ABI, bit-identically) */
    SVT_HIP_HOOK_TF_SUBPEL,    /* the temporal filter's sub-pel stage: tf_32x32 / tf_16x16_sub_pel_search, derive_tf_32x32_block_split_flag and tf_inter_prediction of every
                                * (block, frame) pair of a TF segment, one launch per window frame (EbTemporalFiltering.c:2272-2315); the predictors stay on the device for
                                * hook "tf", which it needs (without "tf" the reference's C code runs) */
    SVT_HIP_HOOK_ENCDEC_TX,    /* encode pass: the forward transforms of every transform block (luma + chroma) of an inter-coded block in one launch, ahead of the block's
                                * transform loops (av1_encode_decode, EbCodingLoop.c:2997-3560; av1_encode_loop's av1_estimate_transform calls :379, :533, :585 read the
                                * results).  Opt-in like md_tx: a launch per coded block */
    SVT_HIP_HOOK_MD_SUBPEL,    /* mode decision's sub-pel refinement: the eight neighbours of a round of svt_av1_find_best_sub_pixel_tree (mcomp.c:350, svt_first_level_check
                                * :186) predicted and measured in one launch pair; the tree's control flow stays the reference's.  Opt-in */
    SVT_HIP_HOOK_ENCDEC_SB,    /* encode pass, one launch per SUPERBLOCK: the plain-translation inter blocks of a superblock's final partition are predicted ahead of its block
                                * loop (av1_encode_decode, EbCodingLoop.c:2262) and the forward transforms of all their transform blocks run in one launch.  Opt-in */
    SVT_HIP_HOOK_MD_PRE,       /* mode decision, one launch per PICTURE before its mode decision starts: the stage-0 luma distortion (fast_loop_core, EbProductCodingLoop.c:907)
                                * of every (superblock, square PU, reference picture) at its open-loop ME vector; fast_loop_core reads the table and skips the prediction,
                                * full_loop_core predicts the survivors (svt_hip_md_bridge.c).  Opt-in */
    SVT_HIP_HOOK_COUNT
};

/* svt_av1_enc_init, right after setup_common_rtcd_internal / setup_rtcd_internal (EbEncHandle.c:1144-1145) and before the derived
 * tables (:1147): reads the environment, creates the context, installs the requested per-call wrappers. */
/* target_socket: EbSvtAv1EncConfiguration::target_socket of the instance (EbSvtAv1Enc.h:648; -1 = no preference).  The GPU ordinal is SVT_HIP_DEVICE when
 * set, else target_socket when >= 0 (eight encoder instances started with --socket 0..7 land on eight GPUs), else 0. */
void svt_hip_hooks_enc_init(int target_socket);
/* svt_av1_enc_deinit_handle, after svt_av1_enc_component_de_init: restores the dispatch pointers SVT_HIP_RTCD replaced, releases the bridges' device memory
 * (loop-filter picture pool, staging buffers, block cache) and the contexts; the next svt_hip_hooks_enc_init starts afresh */
void svt_hip_hooks_enc_deinit(void);
/* svt_av1_enc_deinit_handle, before svt_av1_enc_component_de_init: the picture buffers that were page-locked in place (SVT_HIP_PIN) are released while they exist */
void svt_hip_hooks_enc_predeinit(void);
/* fewer, larger ME / TF segments when their hooks are on (svt_hip_hooks.c) */
void svt_hip_hooks_segments(uint32_t luma_width, uint32_t luma_height, uint32_t *me_cols, uint32_t *me_rows, uint32_t *tf_cols, uint32_t *tf_rows, uint32_t *cdef_cols,
                            uint32_t *cdef_rows, uint32_t *rest_cols, uint32_t *rest_rows);
int  svt_hip_hooks_pin_enabled(void);
/* wall time of a hook call, for the report at exit ("svt_hip_hook_time <name> calls= wall_ms="): t0 = svt_hip_hooks_now_ns() at entry */
long long svt_hip_hooks_now_ns(void);
void      svt_hip_hooks_time(int which, long long t0_ns);
void svt_hip_lf_bridge_release(SvtHipCtx *hip);   /* svt_hip_lf_bridge.c */
void svt_hip_lf_bridge_unpin(SvtHipCtx *hip, int keep_pinning);   /* svt_hip_lf_bridge.c: the reconstructed pictures' host buffers stop being page-locked; keep_pinning: later pictures are registered again */
void svt_hip_md_bridge_release(SvtHipCtx *hip);
void svt_hip_md_bridge_quiesce(void);   /* md_pre's own context: no new issue, nothing in flight (svt_hip_hooks_enc_predeinit) */
void svt_hip_md_bridge_resume(void);   /* svt_hip_md_bridge.c */
int  svt_hip_hook_enabled(int which);
int  svt_hip_hooks_device(void);   /* the GPU ordinal the hooks' contexts were made on (bridges that keep a context of their own) */
/* the context every hook launches on, with the lock that serialises the process threads on it (NULL: no device / init failed) */
SvtHipCtx *svt_hip_hooks_lock(void);
void       svt_hip_hooks_unlock(void);
/* a context of the pool the source-side bridges share (SVT_HIP_CONTEXTS, default 4; 0 = the main context): for calls that keep no device state between two
 * calls.  lock_any / unlock_any pair up on one thread; unlock_any drains the context's stream first */
/* device memory through the hooks' block cache (power-of-two size classes, SVT_HIP_ALLOC_CACHE_MB); pointers of svt_hip_malloc may be passed to the free too */
int        svt_hip_hooks_malloc(SvtHipCtx *hip, void **p, size_t bytes);
void       svt_hip_hooks_free(SvtHipCtx *hip, void *p);
/* resident planes (default; SVT_HIP_RESIDENT=0 turns them off): the table is svt_hip_resident.h (announce / acquire / release on host pointers); these announce the planes of the reference's
 * pictures from the patched reference (and the picture-analysis hook) right after they have been written */
void svt_hip_hooks_resident_note_picture(const EbPictureBufferDesc *pic);   /* the whole padded luma plane of pic */
void svt_hip_hooks_resident_note_pa(const PictureParentControlSet *pcs, const EbPictureBufferDesc *padded, const EbPictureBufferDesc *quarter,
                                    const EbPictureBufferDesc *sixteenth, int with_padded);
SvtHipCtx *svt_hip_hooks_lock_any(void);
void       svt_hip_hooks_unlock_any(void);
void       svt_hip_hooks_log(const char *fmt, ...);
/* statistics for the tests: how many times each hook really ran on the device / fell back */
void svt_hip_hooks_count(int which, int handled);
void svt_hip_hooks_report(void);

/* ------------------------------------------------------------------ open-loop ME (svt_hip_me_bridge.c) */
typedef struct SvtHipMeBatch SvtHipMeBatch;
/* motion_estimation_kernel, before the SB loop of a segment.  NULL = hooks "hme" and "me" off: the caller runs its unchanged loop. */
SvtHipMeBatch *svt_hip_me_batch_begin(PictureParentControlSet *pcs, MeContext *me_ctx, uint32_t n_sb);
/* the same for the temporal filter's block loop (hook "tf_me"): one slot per (64x64 block, window frame) of the TF segment */
SvtHipMeBatch *svt_hip_me_batch_begin_tf(int enable_hme, uint32_t n_slots);
/* The SB loop runs svt_hip_me_batch_passes() times (1 for NULL); before every pass but the first svt_hip_me_batch_flush launches what the
 * previous pass recorded (one hierarchical-ME level, or the integer-search windows).  svt_hip_me_batch_sb runs this pass's part of
 * motion_estimate_sb for one SB and returns 1 when the SB is complete (last pass), 0 when more passes follow (svt_hip_me_bridge.c). */
int  svt_hip_me_batch_passes(const SvtHipMeBatch *b);
int  svt_hip_me_batch_sb(SvtHipMeBatch *b, int pass, PictureParentControlSet *pcs, uint32_t sb_index, uint32_t sb_origin_x,
                         uint32_t sb_origin_y, MeContext *me_ctx, EbPictureBufferDesc *input_ptr);
int  svt_hip_me_batch_slot(SvtHipMeBatch *b, int pass, PictureParentControlSet *pcs, uint32_t key, uint32_t sb_index, uint32_t sb_origin_x,
                           uint32_t sb_origin_y, MeContext *me_ctx, EbPictureBufferDesc *input_ptr);
void svt_hip_me_batch_flush(SvtHipMeBatch *b, int next_pass, const EbPictureBufferDesc *src_padded);
void svt_hip_me_batch_end(SvtHipMeBatch *b);
/* hme_level_0 / 1 / 2, in front of their svt_sad_loop_kernel call, whose twelve arguments follow the four describing the call site:
 * 1 = recorded for the batch (the caller returns; the bridge applies the SAD doubling / centre scaling that follows the call),
 * 0 = no batch is collecting on this thread (temporal-filter ME, hook off, a failed batch): the caller searches as before. */
int  svt_hip_hme_sad_loop(int level, const EbPictureBufferDesc *ref_pic, int16_t x_search_area_origin, int16_t y_search_area_origin,
                          uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height, uint32_t block_width,
                          uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center, uint32_t src_stride_raw,
                          int16_t search_area_width, int16_t search_area_height);
/* integer_search_sb, in place of open_loop_me_fullpel_search_sblock (EbMotionEstimation.c:2130): 1 = recorded for the batch, 0 = no
 * batch is collecting on this thread (temporal-filter ME, hook off): the caller searches as before. */
int  svt_hip_me_record(MeContext *me_ctx, uint32_t sb_origin_x, uint32_t sb_origin_y, uint32_t list_index, uint32_t ref_pic_index,
                       const EbPictureBufferDesc *ref_pic, int16_t x_search_area_origin, int16_t y_search_area_origin,
                       int16_t search_area_width, int16_t search_area_height);
/* the patched motion_estimate_sb: hip_phase -1 = the whole function (what motion_estimate_sb() still is), 0 = up to and including
 * integer_search_sb, 1 = from me_prune_ref on, 10 / 11 / 12 = hme_level0 / 1 / 2_sb only, 2 = set_final_seach_centre_sb up to and including
 * integer_search_sb, 3 = set_final_seach_centre_sb to the end */
EbErrorType motion_estimate_sb_hip(PictureParentControlSet *pcs_ptr, uint32_t sb_index, uint32_t sb_origin_x, uint32_t sb_origin_y,
                                   MeContext *context_ptr, EbPictureBufferDesc *input_ptr, int hip_phase);

/* ------------------------------------------------------------------ in-loop filters (svt_hip_lf_bridge.c)
 * Every function returns EB_ErrorNone when the device did the work and the reference's objects hold the result; anything else =
 * not handled, the caller runs the C code it replaces. */
struct DlfContext;
/* dlf_kernel: svt_av1_pick_filter_level(context, src, pcs, LPF_PICK_FROM_FULL_IMAGE) */
EbErrorType svt_hip_hook_dlf_pick_level(PictureControlSet *pcs);
/* dlf_kernel: svt_av1_loop_filter_frame(recon, pcs, 0, 3) */
EbErrorType svt_hip_hook_dlf_frame(EbPictureBufferDesc *recon, PictureControlSet *pcs);
/* dlf_kernel, after the deblocked picture is final (before svt_av1_loop_restoration_save_boundary_lines): keeps it on the device for
 * the CDEF / restoration hooks (the host overwrites it in place later). */
void        svt_hip_hook_after_dlf(PictureControlSet *pcs);
/* cdef_kernel, once per picture in place of all cdef_seg_search calls (when every segment has arrived) */
EbErrorType svt_hip_hook_cdef_search(PictureControlSet *pcs);
/* cdef_kernel: svt_av1_cdef_frame / av1_cdef_frame16bit */
EbErrorType svt_hip_hook_cdef_apply(PictureControlSet *pcs);
/* rest_kernel, once per picture (all segments arrived) before rest_finish_search: the search_sgrproj_seg results of every unit */
EbErrorType svt_hip_hook_sgr_search(PictureControlSet *pcs);
/* search_wiener_seg: M / H of one unit from the picture-level statistics pass (computed on first use per picture) */
EbErrorType svt_hip_hook_wiener_stats(PictureControlSet *pcs, int plane, int wiener_win, int unit_idx, int64_t *M, int64_t *H);
/* rest_kernel: svt_av1_loop_restoration_filter_frame(cm->frame_to_show, cm, 0) */
EbErrorType svt_hip_hook_rest_apply(PictureControlSet *pcs);
/* one probe of the Wiener tap refinement: the unit [h_start, h_end) x [v_start, v_end) of `plane` filtered with `wi`, *err = its SSE against the source */
/* called before the search_wiener_seg loop of every segment and plane: the first one to arrive searches the whole picture */
EbErrorType svt_hip_hook_wiener_search(PictureControlSet *pcs);
/* in the patched EbRestorationPick.c (its static helpers): 0 = decomposition failed, 1 = initial filter in *wi, refine it, 2 = the filter does not beat identity */
int svt_hip_wiener_unit_init(int32_t wiener_win, int64_t *M, int64_t *H, WienerInfo *wi);
EbErrorType svt_hip_hook_wiener_try(PictureControlSet *pcs, int plane, int h_start, int h_end, int v_start, int v_end, const WienerInfo *wi, int64_t *err);
/* rest_kernel, when the picture leaves the filter stages: the final reconstruction comes back (deferred pictures), its device state is released */
void        svt_hip_hook_picture_done(PictureControlSet *pcs);
/* deferred pictures (svt_hip_lf_bridge.c): 1 = the host work `which` can be skipped — 0 svt_av1_loop_restoration_save_boundary_lines(.., 0) in dlf_kernel,
 * 1 save_boundary_lines(.., 1) + the three svt_extend_frame in cdef_kernel, 2 get_own_recon of a restoration segment, 3 svt_extend_frame of the segment's copy */
int         svt_hip_hook_skip_host_prep(PictureControlSet *pcs, int which);
/* rest_kernel, first thing of a restoration segment: 1 = the picture-level searches are done on the device and get_own_recon can be skipped */
int         svt_hip_hook_rest_begin(PictureControlSet *pcs);

/* finish_cdef_search, in place of joint_strength_search_dual (svt_hip_lf_bridge.c): 1 = best_lev0 / best_lev1 / *tot_mse hold the device result */
int svt_hip_hook_cdef_joint_search(int32_t *best_lev0, int32_t *best_lev1, int32_t nb_strengths, uint64_t (**mse)[64], int32_t sb_count, int32_t start_gi,
                                   int32_t end_gi, uint64_t *tot_mse);

/* finish_cdef_search, in place of everything it does with the two distortion tables (the four searches, the count of strength pairs by RDCOST, every filter
 * block's pair; EbEncCdef.c:1258-1298): 1 = *nb_strength_bits, y_strength / uv_strength [1 << bits] and selected[sb_count] hold the device result */
int svt_hip_hook_cdef_finish(uint64_t (**mse)[64], int32_t sb_count, int32_t start_gi, int32_t end_gi, uint64_t lambda, int32_t *nb_strength_bits, int32_t *y_strength,
                             int32_t *uv_strength, int32_t *selected);

/* ------------------------------------------------------------------ mode decision (svt_hip_md_bridge.c): tx_type_search's forward transforms, one launch per block */
struct EncDecContext;
int  svt_hip_hook_encdec_tx_begin(struct EncDecContext *ctx, const EbPictureBufferDesc *pred, int is_16bit);
int  svt_hip_hook_encdec_tx_fetch(int plane, int txb, int tx_size, int tx_type, int32_t *coeff);
void svt_hip_hook_encdec_tx_end(void);
void svt_hip_hook_encdec_tx_stats(long *blocks, long *calls);
struct SuperBlock;
int  svt_hip_hook_encdec_sb_begin(SequenceControlSet *scs, PictureControlSet *pcs, struct SuperBlock *sb, uint32_t sb_addr, uint32_t sb_origin_x, uint32_t sb_origin_y,
                                  struct EncDecContext *ctx, EbPictureBufferDesc *recon, int is_16bit);
int  svt_hip_hook_encdec_sb_predicted(const BlkStruct *blk);
void svt_hip_hook_encdec_sb_end(void);
void svt_hip_hook_encdec_sb_stats(long *superblocks, long *launches, long *blocks, long *calls);
long svt_hip_hook_encdec_sb_kernels(void);   /* kernel launches behind those entry-point calls (one per 16 (plane, transform size) pairs of a superblock) */
/* svt_hip_hook_md_subpel_begin(const SUBPEL_SEARCH_VAR_PARAMS *, const MV *centre, int hstep, const SubpelMvLimits *) is declared in the patched mcomp.c (its
 * argument types live in mcomp.h, which includes this header's dependencies the other way round) */
int  svt_hip_hook_md_subpel_fetch(const MV *mv, unsigned int *err, unsigned int *sse);
void svt_hip_hook_md_subpel_end(void);
/* hook "md_pre" (svt_hip_md_bridge.c): mode_decision_configuration_kernel before it posts the picture; rest_kernel after pad_ref_and_set_flags; fast_loop_core in front of
 * its prediction (1 = *sad is the luma distortion, no prediction now); full_loop_core where it decides about the inter prediction (1 = the stage-0 prediction is still owed) */
struct ModeDecisionContext;
struct ModeDecisionCandidateBuffer;
void svt_hip_hook_md_pre_picture(PictureControlSet *pcs);
void svt_hip_hook_md_pre_note_ref(PictureControlSet *pcs);
int  svt_hip_hook_md_pre_lookup(PictureControlSet *pcs, struct ModeDecisionContext *ctx, struct ModeDecisionCandidateBuffer *cb, uint32_t *sad);
/* 2 (SVT_HIP_MD_PRE_VERIFY=1, tests) = a table hit the caller computes itself as well and reports to svt_hip_hook_md_pre_verify */
void svt_hip_hook_md_pre_verify(PictureControlSet *pcs, struct ModeDecisionContext *ctx, struct ModeDecisionCandidateBuffer *cb, uint32_t table_sad, uint32_t ref_sad);
long svt_hip_hook_md_pre_mismatches(void);   /* -1 = not verifying */
int  svt_hip_hook_md_pre_take(const struct ModeDecisionCandidateBuffer *cb, int predicted_late);
/* md_subpel_search, around svt_av1_find_best_sub_pixel_tree: the probes (svt_upsampled_pref_error, mcomp.c:102) of this search are looked up in the picture's sub-pel grid */
void svt_hip_hook_md_pre_subpel_begin(PictureControlSet *pcs, struct ModeDecisionContext *ctx, int list_idx, int ref_idx, int subpel_search_type, int mvx8, int mvy8);
void svt_hip_hook_md_pre_subpel_end(void);
void svt_hip_hook_md_pre_subpel_verify(const MV *mv, unsigned int err, unsigned int sse);   /* the patched svt_upsampled_pref_error, after computing a probe itself (self-check mode) */
void svt_hip_hook_md_pre_subpel_stats(long *pictures, long *probes, long *served);
void svt_hip_hook_md_pre_compound_stats(long *pictures, long *served);   /* pictures with a table of compound-average candidates, candidates served from it */
void svt_hip_hook_md_pre_misses(long *out, int n);   /* inter candidates of fast_loop_core not served, by reason */
double svt_hip_hook_md_pre_device_ms(void);          /* SVT_HIP_MD_PRE_TIMING=1: device time of the pictures' launches and copies; -1 = not measured */
void svt_hip_hook_md_pre_stats(long *pictures, long *launches, long *jobs, long *min_jobs, long *calls, long *inter, long *hits, long *late, long *declined, double *ms);
int  svt_hip_hook_md_tx_begin(const int16_t *resid, uint32_t stride, int tx_size, int coeff_shape, uint32_t type_mask);
int  svt_hip_hook_md_tx_fetch(int tx_size, int tx_type, int32_t *coeff);
void svt_hip_hook_md_tx_end(void);

/* ------------------------------------------------------------------ picture analysis (svt_hip_pa_bridge.c); EB_ErrorNone = handled */
EbErrorType svt_hip_hook_pa_downsample(PictureParentControlSet *pcs, EbPictureBufferDesc *padded, EbPictureBufferDesc *quarter, EbPictureBufferDesc *sixteenth,
                                       int filtered);
EbErrorType svt_hip_hook_pa_variance(SequenceControlSet *scs, PictureParentControlSet *pcs, EbPictureBufferDesc *padded);

/* the temporal-filter hook's declarations (svt_hip_tf_seg_*) */
#include "svt_hip_tf_bridge.h"

#endif
