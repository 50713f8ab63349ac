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

/* A COMPOUND-AVERAGE candidate of two full-pel vectors (NEW_NEWMV with MD_COMP_AVG, Encoder/Codec/EbModeDecision.c:3408-3540): av1_inter_prediction calls
 * svt_inter_predictor twice with is_compound conv_params (get_conv_params_no_round: round_0 = 3, round_1 = COMPOUND_ROUND1_BITS = 7, dst = a CONV_BUF_TYPE plane), full-pel
 * vectors pick convolve[0][0][1] = svt_av1_jnt_convolve_2d_copy (Common/Codec/convolve.c): the first call stores (sample << bits) + round_offset, the second (do_average,
 * no distance weights for compound_idx = 1) averages with it, removes the offset and rounds by `bits` into the 8-bit prediction.  Then the same SAD as above. */
uint32_t orc_md_fullpel_avg_candidate(const uint8_t *src, int src_stride, const uint8_t *ref0, int ref0_stride, const uint8_t *ref1, int ref1_stride, int x, int y, int w, int h,
                                      int mx0, int my0, int mx1, int my1) {
    const int bd = 8, round_0 = 3, round_1 = 7, bits = 2 * 7 - round_0 - round_1, offset_bits = bd + 2 * 7 - round_0;
    const int round_offset = (1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1));
    static __thread uint16_t tmp[64 * 64];
    uint8_t pred[64 * 64];
    const uint8_t *a = ref0 + (ptrdiff_t)(y + my0) * ref0_stride + (x + mx0), *b = ref1 + (ptrdiff_t)(y + my1) * ref1_stride + (x + mx1);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) tmp[i * w + j] = (uint16_t)((a[(ptrdiff_t)i * ref0_stride + j] << bits) + round_offset);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const uint16_t res = (uint16_t)((b[(ptrdiff_t)i * ref1_stride + j] << bits) + round_offset);
            int32_t t = (tmp[i * w + j] + res) >> 1;
            t -= round_offset;
            t = (t + ((1 << bits) >> 1)) >> bits;   /* ROUND_POWER_OF_TWO */
            pred[i * w + j] = (uint8_t)(t < 0 ? 0 : (t > 255 ? 255 : t));
        }
    return orc_nxm_sad(src + (ptrdiff_t)y * src_stride + x, (uint32_t)src_stride, pred, (uint32_t)w, (uint32_t)h, (uint32_t)w);
}
/* the table of svt_hip_md_fullpel_avg_sad_picture_dev: [n_sb][n_pus][n_pairs], pairs[i] = {c0, c1} columns of the vector table */
void orc_md_fullpel_avg_sad_picture(const uint8_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const uint8_t (*pus)[4], int n_refs,
                                    const uint8_t *const *refs, const int *ref_stride, const int (*ref_box)[4], const uint32_t *mv, int n_pairs, const uint8_t (*pairs)[2],
                                    uint32_t *sad) {
    for (int sb = 0; sb < n_sb; sb++)
        for (int p = 0; p < n_pus; p++)
            for (int q = 0; q < n_pairs; q++) {
                const size_t base = ((size_t)sb * n_pus + p) * n_refs, slot = ((size_t)sb * n_pus + p) * n_pairs + q;
                const int c0 = pairs[q][0], c1 = pairs[q][1];
                const int x = (sb % sb_cols) * 64 + pus[p][0], y = (sb / sb_cols) * 64 + pus[p][1], w = pus[p][2], h = pus[p][3];
                const int mx0 = (int16_t)(mv[base + c0] & 0xffff), my0 = (int16_t)(mv[base + c0] >> 16), mx1 = (int16_t)(mv[base + c1] & 0xffff), my1 = (int16_t)(mv[base + c1] >> 16);
                int ok = mx0 != -32768 && mx1 != -32768 && x + w <= pic_w && y + h <= pic_h;
                ok = ok && x + mx0 >= ref_box[c0][0] && y + my0 >= ref_box[c0][1] && x + mx0 + w + 4 <= ref_box[c0][2] && y + my0 + h <= ref_box[c0][3];
                ok = ok && x + mx1 >= ref_box[c1][0] && y + my1 >= ref_box[c1][1] && x + mx1 + w + 4 <= ref_box[c1][2] && y + my1 + h <= ref_box[c1][3];
                sad[slot] = ok ? orc_md_fullpel_avg_candidate(src, src_stride, refs[c0], ref_stride[c0], refs[c1], ref_stride[c1], x, y, w, h, mx0, my0, mx1, my1) : 0xffffffffu;
            }
}

/* One probe of the sub-pel refinement: svt_upsampled_pref_error (Encoder/Codec/mcomp.c:102-156) = svt_aom_upsampled_pred (Encoder/C_DEFAULT/variance.c:212-269, orc_upsampled_pred)
 * into a scratch block + the square block's variance (orc_variance = svt_aom_variance{W}x{H}_c).  (mvx8, mvy8): the probed vector in eighth-samples. */
void     orc_upsampled_pred(const uint8_t *ref, int ref_stride, uint8_t *dst, int w, int h, int subpel_x_q3, int subpel_y_q3, int bank);
uint32_t orc_variance(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse);
uint32_t orc_md_subpel_probe(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int x, int y, int s, int mvx8, int mvy8, int bank, uint32_t *sse) {
    uint8_t pred[64 * 64];
    orc_upsampled_pred(ref + (ptrdiff_t)(y + (mvy8 >> 3)) * ref_stride + x + (mvx8 >> 3), ref_stride, pred, s, s, mvx8 & 7, mvy8 & 7, bank);
    return orc_variance(pred, s, src + (ptrdiff_t)y * src_stride + x, src_stride, s, s, sse);
}
/* the table of svt_hip_md_subpel_grid_picture_dev: [n_sb][n_pus][n_refs][49][2] */
void orc_md_subpel_grid_picture(const uint8_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const uint8_t (*pus)[4], int n_refs,
                                const uint8_t *const *refs, const int *ref_stride, const int (*ref_box)[4], const uint32_t *mv, int bank, uint32_t *out) {
    for (int sb = 0; sb < n_sb; sb++)
        for (int p = 0; p < n_pus; p++)
            for (int r = 0; r < n_refs; r++) {
                const size_t slot = ((size_t)sb * n_pus + p) * n_refs + r;
                uint32_t *o = out + slot * 98;
                const int x = (sb % sb_cols) * 64 + pus[p][0], y = (sb / sb_cols) * 64 + pus[p][1], s = pus[p][2];
                const int mx = (int16_t)(mv[slot] & 0xffff), my = (int16_t)(mv[slot] >> 16), wx = x + mx - 4, wy = y + my - 4;
                if (mx == -32768 || pus[p][3] != s || (s != 8 && s != 16 && s != 32 && s != 64) || x + s > pic_w || y + s > pic_h || wx < ref_box[r][0] || wy < ref_box[r][1] ||
                    wx + s + 12 > ref_box[r][2] || wy + s + 8 > ref_box[r][3]) {
                    for (int i = 0; i < 98; i++) o[i] = 0xffffffffu;
                    continue;
                }
                for (int gy = 0; gy < 7; gy++)
                    for (int gx = 0; gx < 7; gx++)
                        o[2 * (7 * gy + gx)] = orc_md_subpel_probe(src, src_stride, refs[r], ref_stride[r], x, y, s, 8 * mx + 2 * gx - 6, 8 * my + 2 * gy - 6, bank, &o[2 * (7 * gy + gx) + 1]);
            }
}


/* ---- the same two tables on 16-bit planes (a 10-bit encode: hbd_mode_decision 1 / 2).  fast_loop_core predicts with the high-bit-depth copy (svt_av1_highbd_convolve_2d_copy_sr:
 * a plain copy) and measures with sad_16b_kernel (orc_sad_16b); the compound copy is svt_av1_highbd_jnt_convolve_2d_copy (Common/Codec/convolve.c): the same arithmetic as the
 * 8-bit one with bd in the offset and the clip. */
uint32_t orc_md_fullpel_candidate16(const uint16_t *src, int src_stride, const uint16_t *ref, int ref_stride, int x, int y, int w, int h, int mx, int my) {
    static __thread uint16_t pred[64 * 64];
    for (int i = 0; i < h; i++) memcpy(pred + (size_t)i * w, ref + (ptrdiff_t)(y + my + i) * ref_stride + (x + mx), (size_t)w * 2);
    return orc_sad_16b(src + (ptrdiff_t)y * src_stride + x, (uint32_t)src_stride, pred, (uint32_t)w, (uint32_t)h, (uint32_t)w);
}
uint32_t orc_md_fullpel_avg_candidate16(const uint16_t *src, int src_stride, const uint16_t *ref0, int ref0_stride, const uint16_t *ref1, int ref1_stride, int x, int y, int w,
                                        int h, int mx0, int my0, int mx1, int my1, int bd) {
    const int round_0 = 3, round_1 = 7, bits = 2 * 7 - round_0 - round_1, offset_bits = bd + 2 * 7 - round_0;
    const int round_offset = (1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1));
    static __thread uint16_t tmp[64 * 64], pred[64 * 64];
    const uint16_t *a = ref0 + (ptrdiff_t)(y + my0) * ref0_stride + (x + mx0), *b = ref1 + (ptrdiff_t)(y + my1) * ref1_stride + (x + mx1);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) tmp[i * w + j] = (uint16_t)((a[(ptrdiff_t)i * ref0_stride + j] << bits) + round_offset);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const uint16_t res = (uint16_t)((b[(ptrdiff_t)i * ref1_stride + j] << bits) + round_offset);
            int32_t t = (tmp[i * w + j] + res) >> 1;
            t -= round_offset;
            t = (t + ((1 << bits) >> 1)) >> bits;
            pred[i * w + j] = (uint16_t)(t < 0 ? 0 : (t > (1 << bd) - 1 ? (1 << bd) - 1 : t));
        }
    return orc_sad_16b(src + (ptrdiff_t)y * src_stride + x, (uint32_t)src_stride, pred, (uint32_t)w, (uint32_t)h, (uint32_t)w);
}
void orc_md_fullpel_sad_picture16(const uint16_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const uint8_t (*pus)[4], int n_refs,
                                  const uint16_t *const *refs, const int *ref_stride, const int (*ref_box)[4], const uint32_t *mv, uint32_t *sad) {
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
                sad[slot] = orc_md_fullpel_candidate16(src, src_stride, refs[r], ref_stride[r], x, y, w, h, mx, my);
            }
}
void orc_md_fullpel_avg_sad_picture16(const uint16_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const uint8_t (*pus)[4], int n_refs,
                                      const uint16_t *const *refs, const int *ref_stride, const int (*ref_box)[4], const uint32_t *mv, int n_pairs, const uint8_t (*pairs)[2],
                                      int bd, uint32_t *sad) {
    for (int sb = 0; sb < n_sb; sb++)
        for (int p = 0; p < n_pus; p++)
            for (int q = 0; q < n_pairs; q++) {
                const size_t base = ((size_t)sb * n_pus + p) * n_refs, slot = ((size_t)sb * n_pus + p) * n_pairs + q;
                const int c0 = pairs[q][0], c1 = pairs[q][1];
                const int x = (sb % sb_cols) * 64 + pus[p][0], y = (sb / sb_cols) * 64 + pus[p][1], w = pus[p][2], h = pus[p][3];
                const int mx0 = (int16_t)(mv[base + c0] & 0xffff), my0 = (int16_t)(mv[base + c0] >> 16), mx1 = (int16_t)(mv[base + c1] & 0xffff), my1 = (int16_t)(mv[base + c1] >> 16);
                int ok = mx0 != -32768 && mx1 != -32768 && x + w <= pic_w && y + h <= pic_h;
                ok = ok && x + mx0 >= ref_box[c0][0] && y + my0 >= ref_box[c0][1] && x + mx0 + w + 4 <= ref_box[c0][2] && y + my0 + h <= ref_box[c0][3];
                ok = ok && x + mx1 >= ref_box[c1][0] && y + my1 >= ref_box[c1][1] && x + mx1 + w + 4 <= ref_box[c1][2] && y + my1 + h <= ref_box[c1][3];
                sad[slot] = ok ? orc_md_fullpel_avg_candidate16(src, src_stride, refs[c0], ref_stride[c0], refs[c1], ref_stride[c1], x, y, w, h, mx0, my0, mx1, my1, bd) : 0xffffffffu;
            }
}
