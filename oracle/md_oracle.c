/*
 * md_oracle.c — CPU restatement of the mode-decision stage-0 leaf work that the picture-level precompute (svt_hip_md_fullpel_sad_picture_dev) batches.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 *
 * fast_loop_core (Encoder/Codec/EbProductCodingLoop.c:907) = prediction (:926, inter_pu_prediction_av1, Encoder/Codec/EbEncInterPrediction.c:6178 ->
 * av1_inter_prediction :4040 -> svt_inter_predictor, Common/Codec/EbInterPrediction.c:1368) + luma distortion (:953, svt_nxm_sad_kernel_sub_sampled =
 * svt_nxm_sad_kernel_helper_c, Encoder/Codec/aom_dsp_rtcd.c:372 / Encoder/C_DEFAULT/EbComputeSAD_C.c:208: every row, no sub-sampling).  With a full-pel vector
 * svt_inter_predictor picks convolve[0][0][0] = svt_av1_convolve_2d_copy_sr (Common/Codec/EbInterPrediction.c:861-866 via :1419), a plain copy.
 */
#include "svt_oracle.h"
#include <string.h>

/* Common/Codec/convolve.c svt_av1_convolve_2d_copy_sr_c: dst[y][x] = src[y][x] */
static void copy_sr(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int w, int h) {
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * dst_stride, src + (size_t)y * src_stride, (size_t)w);
}

/* One (PU, reference) of the table: predict the PU at (x, y) from `ref` (sample (0, 0) of the reference plane) with the full-pel vector (mx, my), measure it
 * against the source.  The caller guarantees that both blocks exist. */
uint32_t orc_md_fullpel_candidate(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int x, int y, int w, int h, int mx, int my) {
    uint8_t pred[64 * 64];
    copy_sr(ref + (ptrdiff_t)(y + my) * ref_stride + (x + mx), ref_stride, pred, w, w, h);
    return orc_nxm_sad(src + (ptrdiff_t)y * src_stride + x, (uint32_t)src_stride, pred, (uint32_t)w, (uint32_t)h, (uint32_t)w);
}

/* The whole table, same layout and "not computed" rule as svt_hip_md_fullpel_sad_picture_dev (include/svt_hip.h): pus[i] = {x, y, w, h} relative to the
 * 64x64 superblock; refs[r] = sample (0, 0) of reference r, ref_box[r] = {x_min, y_min, x_max, y_max} of its allocation; mv / sad [n_sb][n_pus][n_refs]. */
void orc_md_fullpel_sad_picture(const uint8_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const uint8_t (*pus)[4], int n_refs,
                                const uint8_t *const *refs, const int *ref_stride, const int (*ref_box)[4], const uint32_t *mv, uint32_t *sad) {
    for (int sb = 0; sb < n_sb; sb++)
        for (int p = 0; p < n_pus; p++)
            for (int r = 0; r < n_refs; r++) {
                const size_t slot = ((size_t)sb * n_pus + p) * n_refs + r;
                const int x = (sb % sb_cols) * 64 + pus[p][0], y = (sb / sb_cols) * 64 + pus[p][1], w = pus[p][2], h = pus[p][3];
                const int mx = (int16_t)(mv[slot] & 0xffff), my = (int16_t)(mv[slot] >> 16);
                const int rx = x + mx, ry = y + my;
                if (mx == -32768 || x + w > pic_w || y + h > pic_h || rx < ref_box[r][0] || ry < ref_box[r][1] || rx + w + 4 > ref_box[r][2] || ry + h > ref_box[r][3]) {
                    sad[slot] = 0xffffffffu;
                    continue;
                }
                sad[slot] = orc_md_fullpel_candidate(src, src_stride, refs[r], ref_stride[r], x, y, w, h, mx, my);
            }
}
