/* svt_hip_pa_bridge.c — picture-analysis glue, hook "pa": the HME pyramids and the per-SB mean / variance pyramid of
 * picture_analysis_kernel (Source/Lib/Encoder/Codec/EbPictureAnalysisProcess.c:3867) as picture-level launches.
 *
 *   svt_hip_hook_pa_downsample : downsample_decimation_input_picture (:3312) / downsample_filtering_input_picture (:3606) — the luma plane is uploaded
 *                                once, decimation_2d / downsample_2d (:193, :223) run as svt_hip_downsample_2d_dev, the 1/4 and 1/16 pictures come
 *                                back into the reference's buffers, whose borders the reference's own generate_padding then fills (host, unchanged).
 *   svt_hip_hook_pa_variance   : every compute_block_mean_compute_variance call (:1005) of compute_picture_spatial_statistics (:2929) in one
 *                                svt_hip_variance_pyramid_dev launch over the padded luma picture; pcs->y_mean / pcs->variance[sb][85] receive the
 *                                result, the chroma means and the picture average stay with the reference's loop.
 * Host orchestration only.  Anything unexpected returns "not handled" and the reference's loop runs (error convention).
 */
#include <stdlib.h>
#include <string.h>
#include "svt_hip_hooks.h"
#include "EbLog.h"
#include "EbPictureAnalysisProcess.h"
#include "EbMcp.h"

#define PA_TRY(x) do { if (rc == SVT_HIP_OK) rc = (x); } while (0)

EbErrorType svt_hip_hook_pa_downsample(PictureParentControlSet *pcs, EbPictureBufferDesc *padded, EbPictureBufferDesc *quarter, EbPictureBufferDesc *sixteenth,
                                       int filtered) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_PA)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    /* Resident planes (svt_hip_hooks.c): the padded picture is complete when downsample_decimation_input_picture is entered (the first of the two calls, always
     * made) — announced here, so that this hook, the variance hook and every later ME / TF segment share one upload; the pyramids are announced by the patched
     * function when this hook returns (they are written below).  A failure below leaves the announcement to the C path's end of function. */
    if (!filtered) svt_hip_hooks_resident_note_pa(pcs, padded, NULL, NULL, 1);
    const int hme = pcs->enable_hme_flag || pcs->tf_enable_hme_flag;
    const int lvl1 = pcs->enable_hme_level1_flag || pcs->tf_enable_hme_level1_flag, lvl0 = pcs->enable_hme_level0_flag || pcs->tf_enable_hme_level0_flag;
    /* which pictures the reference function produces (:3316-3358 decimation: 1/16 always; :3610-3670 filtering: only inside the HME flags) */
    const int do_q = hme && lvl1, do_s = filtered ? (hme && lvl0) : 1;
    if (!do_q && !do_s) return EB_ErrorNone;
    /* The reference writes the 1/4 (1/16) picture at origin_x + origin_x * stride and reads the 1/4 picture back at origin_x + origin_y * stride (:3327-3329,
     * :3636-3647); the two agree — and the device copy of the 1/4 picture equals what the reference would read — only for square origins, the only case it creates. */
    if ((do_q && quarter->origin_x != quarter->origin_y) || (do_s && sixteenth->origin_x != sixteenth->origin_y)) return EB_ErrorUndefined;
    const int  w = padded->width, h = padded->height;
    SvtHipCtx *hip = svt_hip_hooks_lock_any();
    if (!hip) return EB_ErrorUndefined;
    void *d_in = NULL, *d_q = NULL, *d_s = NULL;
    int   rc = SVT_HIP_OK;
    /* the luma plane: the resident copy read in place (origin and stride of the host picture), or the picture's interior uploaded compactly */
    const size_t   plane_off = padded->origin_x + (size_t)padded->origin_y * padded->stride_y;
    const uint8_t *d_res = (const uint8_t *)svt_hip_resident_acquire(hip, padded->buffer_y, (size_t)padded->stride_y * (size_t)(padded->height + 2 * padded->origin_y));
    const uint8_t *in = d_res ? d_res + plane_off : NULL;
    const int      in_stride = d_res ? padded->stride_y : w;
    if (!d_res) {
        PA_TRY(svt_hip_hooks_malloc(hip, &d_in, (size_t)w * h));
        PA_TRY(svt_hip_memcpy2d_h2d(hip, d_in, (size_t)w, padded->buffer_y + plane_off, padded->stride_y, (size_t)w, (size_t)h));
        in = (const uint8_t *)d_in;
    }
    PA_TRY(svt_hip_hooks_malloc(hip, &d_q, (size_t)(w / 2) * (h / 2) + 64));
    PA_TRY(svt_hip_hooks_malloc(hip, &d_s, (size_t)(w / 4) * (h / 4) + 64));
    /* the destination offset is the reference's own expression (origin_x for the row as well, :3327-3329) */
    if (do_q) {
        PA_TRY(svt_hip_downsample_2d_dev(hip, in, in_stride, w, h, (uint8_t *)d_q, w / 2, 2, filtered));
        PA_TRY(svt_hip_memcpy2d_d2h(hip, quarter->buffer_y + quarter->origin_x + (size_t)quarter->origin_x * quarter->stride_y, quarter->stride_y, d_q, (size_t)(w / 2),
                                    (size_t)(w / 2), (size_t)(h / 2)));
    }
    if (do_s) {
        if (filtered && lvl1)   /* 2x2 average of the 1/4 picture (:3636-3647): its width / height, read where the reference reads it (origin_y for the row here) */
            PA_TRY(svt_hip_downsample_2d_dev(hip, (const uint8_t *)d_q, w / 2, quarter->width, quarter->height, (uint8_t *)d_s, w / 4, 2, 1));
        else
            PA_TRY(svt_hip_downsample_2d_dev(hip, in, in_stride, w, h, (uint8_t *)d_s, w / 4, 4, filtered));
        PA_TRY(svt_hip_memcpy2d_d2h(hip, sixteenth->buffer_y + sixteenth->origin_x + (size_t)sixteenth->origin_x * sixteenth->stride_y, sixteenth->stride_y, d_s,
                                    (size_t)(w / 4), (size_t)(w / 4), (size_t)(h / 4)));
    }
    if (d_res) {   /* the downloads above completed the launches; after a failure the context is drained first */
        if (rc != SVT_HIP_OK) (void)svt_hip_sync(hip);
        svt_hip_resident_release(padded->buffer_y);
    }
    svt_hip_hooks_free(hip, d_in); svt_hip_hooks_free(hip, d_q); svt_hip_hooks_free(hip, d_s);
    if (rc != SVT_HIP_OK) SVT_LOG("picture-analysis pyramids on the device failed (%s): C path\n", svt_hip_last_error(hip));
    svt_hip_hooks_unlock_any();
    svt_hip_hooks_count(SVT_HIP_HOOK_PA, rc == SVT_HIP_OK);
    svt_hip_hooks_time(SVT_HIP_HOOK_PA, t0);
    if (rc != SVT_HIP_OK) return EB_ErrorUndefined;
    if (do_q) generate_padding(&quarter->buffer_y[0], quarter->stride_y, quarter->width, quarter->height, quarter->origin_x, quarter->origin_y);
    if (do_s) generate_padding(&sixteenth->buffer_y[0], sixteenth->stride_y, sixteenth->width, sixteenth->height, sixteenth->origin_x, sixteenth->origin_y);
    svt_hip_hooks_log("pa: %s pyramids of a %d x %d picture (1/4 %d, 1/16 %d)", filtered ? "filtered" : "decimated", w, h, do_q, do_s);
    return EB_ErrorNone;
}

EbErrorType svt_hip_hook_pa_variance(SequenceControlSet *scs, PictureParentControlSet *pcs, EbPictureBufferDesc *padded) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_PA)) return EB_ErrorUndefined;
    const uint32_t n_sb = pcs->sb_total_count;
    if (!n_sb || scs->sb_sz != 64) return EB_ErrorUndefined;
    /* SBs in raster order on a grid of sb_cols columns (sb_params_array is built that way, EbPictureControlSet.c sb_params_init) */
    const int sb_cols = (padded->width + 63) / 64, sb_rows = (int)(n_sb / sb_cols);
    if ((uint32_t)(sb_cols * sb_rows) != n_sb) return EB_ErrorUndefined;
    for (uint32_t i = 0; i < n_sb; i++)
        if (pcs->sb_params_array[i].origin_x != (i % sb_cols) * 64 || pcs->sb_params_array[i].origin_y != (i / sb_cols) * 64) return EB_ErrorUndefined;
    /* the kernel reads whole 64x64 SBs: the padded picture has them (pad_picture_to_multiple_of_sb_dimensions ran before, :3975) */
    const int    pw = sb_cols * 64, ph = sb_rows * 64, stride = (pw + 7) & ~7;
    if (padded->origin_x + pw > padded->stride_y || padded->origin_y + ph > (int)(padded->height + 2 * padded->origin_y)) return EB_ErrorUndefined;
    uint8_t  *mean = (uint8_t *)malloc((size_t)n_sb * 85);
    uint16_t *var = (uint16_t *)malloc((size_t)n_sb * 85 * sizeof(uint16_t));
    SvtHipCtx *hip = (mean && var) ? svt_hip_hooks_lock_any() : NULL;
    int        rc = hip ? SVT_HIP_OK : SVT_HIP_ERR_NO_DEVICE;
    void      *d_in = NULL, *d_mean = NULL, *d_var = NULL;
    /* the kernel reads 8-byte words at 8-sample columns: the resident plane serves in place when the picture's origin and stride keep them aligned */
    const size_t   plane_off = padded->origin_x + (size_t)padded->origin_y * padded->stride_y;
    const uint8_t *d_res = (hip && !(plane_off & 7) && !(padded->stride_y & 7))
        ? (const uint8_t *)svt_hip_resident_acquire(hip, padded->buffer_y, (size_t)padded->stride_y * (size_t)(padded->height + 2 * padded->origin_y)) : NULL;
    if (!d_res) {
        PA_TRY(svt_hip_hooks_malloc(hip, &d_in, (size_t)stride * ph));
        PA_TRY(svt_hip_memcpy2d_h2d(hip, d_in, (size_t)stride, padded->buffer_y + plane_off, padded->stride_y, (size_t)pw, (size_t)ph));
    }
    PA_TRY(svt_hip_hooks_malloc(hip, &d_mean, (size_t)n_sb * 85));
    PA_TRY(svt_hip_hooks_malloc(hip, &d_var, (size_t)n_sb * 85 * sizeof(uint16_t)));
    PA_TRY(svt_hip_variance_pyramid_dev(hip, d_res ? d_res + plane_off : (const uint8_t *)d_in, d_res ? padded->stride_y : stride, sb_cols, (int)n_sb,
                                        scs->block_mean_calc_prec == BLOCK_MEAN_PREC_FULL, (uint8_t *)d_mean, (uint16_t *)d_var));
    PA_TRY(svt_hip_memcpy_d2h(hip, mean, d_mean, (size_t)n_sb * 85));
    PA_TRY(svt_hip_memcpy_d2h(hip, var, d_var, (size_t)n_sb * 85 * sizeof(uint16_t)));
    if (d_res) {
        if (rc != SVT_HIP_OK) (void)svt_hip_sync(hip);
        svt_hip_resident_release(padded->buffer_y);
    }
    if (hip) {
        svt_hip_hooks_free(hip, d_in); svt_hip_hooks_free(hip, d_mean); svt_hip_hooks_free(hip, d_var);
        if (rc != SVT_HIP_OK) SVT_LOG("variance pyramid on the device failed (%s): C path\n", svt_hip_last_error(hip));
        svt_hip_hooks_unlock_any();
    }
    if (rc == SVT_HIP_OK)
        for (uint32_t i = 0; i < n_sb; i++) {   /* [0] 64x64, [1..4] 32x32, [5..20] 16x16, [21..84] 8x8: the raster-scan indices of pcs->y_mean / variance */
            memcpy(pcs->y_mean[i], mean + (size_t)i * 85, 85);
            memcpy(pcs->variance[i], var + (size_t)i * 85, 85 * sizeof(uint16_t));
        }
    free(mean); free(var);
    svt_hip_hooks_count(SVT_HIP_HOOK_PA, rc == SVT_HIP_OK);
    svt_hip_hooks_log("pa: mean / variance pyramid of %u SBs in one launch", n_sb);
    return rc == SVT_HIP_OK ? EB_ErrorNone : EB_ErrorUndefined;
}
