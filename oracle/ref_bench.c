/*
 * ref_bench.c — drives the REAL reference kernels (through the reference's own RTCD function pointers, i.e. the AVX2 / AVX-512
 * kernels in the SIMD flavour of oracle/_ref, the C ones in the plain flavour) over the same job lists the HIP stages and the oracle
 * batch functions get.  TEST INFRASTRUCTURE ONLY: compiled into oracle/_ref/libsvtav1_ref*.so by oracle/Makefile.ref; used by
 * bench.py's cpu_baseline leg (kind = "reference") and by tests/test_ref_bench.py, which checks every function here bit-for-bit
 * against the oracle batch function of the same name (so the timed reference work IS the work the GPU does).
 *
 * Every loop below follows the reference loop that issues the same calls (cited per function, file:line under
 * /root/reference/Source/Lib); nothing here re-implements a kernel.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "EbDefinitions.h"
#include "common_dsp_rtcd.h"
#include "aom_dsp_rtcd.h"
#include "EbCdef.h"
#include "EbRestoration.h"
#include "EbTransforms.h"
#include "EbInvTransforms.h"
#include "EbInterPrediction.h"
#include "EbMotionEstimation.h"

#ifdef ARCH_X86_64
CPU_FLAGS get_cpu_flags_to_use();
#else
static CPU_FLAGS get_cpu_flags_to_use() { return 0; }   /* C-only flavour: every pointer resolves to its _c function */
#endif

extern int ref_shim_rtcd_ready;
/* setup_common_rtcd_internal / setup_rtcd_internal with the flags of this host (EbEncHandle.c:4027-4028 does the same at init) */
uint64_t refb_setup(uint64_t mask) {
    const CPU_FLAGS f = get_cpu_flags_to_use() & (CPU_FLAGS)mask;
    setup_common_rtcd_internal(f);
    setup_rtcd_internal(f);
    ref_shim_rtcd_ready = 1;   /* keep ref_shim.c from re-installing the C kernels */
    return (uint64_t)f;
}

/* ---------------------------------------------------------------- integer ME --------------------------------------------------------
 * open_loop_me_fullpel_search_sblock (Encoder/Codec/EbMotionEstimation.c:814-877) for the SBs [begin, end) of a picture; window
 * descriptors and outputs as orc_me_fullpel_frame / svt_hip_me_fullpel_frame_dev (85-PU layout). */
typedef struct { int32_t sb_x, sb_y; int16_t x_origin, y_origin, width, height; } RefbSbSearch;
void refb_me_fullpel_frame(const uint8_t *src, const uint8_t *ref, int stride, int org_x, int org_y, const RefbSbSearch *sbs, int n_sb,
                           int sub_sad, uint32_t *best_sad, uint32_t *best_mv, int begin, int end) {
    (void)n_sb;
    static const uint8_t z16[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};   /* raster 16x16 -> z-order PU index */
    for (int i = begin; i < end; i++) {
        const RefbSbSearch *d = &sbs[i];
        uint8_t *s = (uint8_t *)src + (size_t)(org_y + d->sb_y) * stride + org_x + d->sb_x;
        uint8_t *r = (uint8_t *)ref + (size_t)(org_y + d->sb_y + d->y_origin) * stride + org_x + d->sb_x + d->x_origin;
        uint32_t *bs = best_sad + (size_t)i * 85, *bm = best_mv + (size_t)i * 85;
        for (int k = 0; k < 85; k++) { bs[k] = MAX_SAD_VALUE; bm[k] = 0; }
        uint32_t e16[16][8], e8[64][8], e32[4][8], s16[16], s8[64], s32[4];
        const int w8 = d->width & ~7;
        for (int cy = 0; cy < d->height; cy++) {
            for (int cx = 0; cx < d->width; cx += (cx < w8 ? 8 : 1)) {
                const uint32_t mv = ((uint32_t)(uint16_t)(cy + d->y_origin) << 18) | (uint16_t)((uint16_t)(cx + d->x_origin) << 2);
                uint8_t *rp = r + (size_t)cy * stride + cx;
                if (cx < w8) {      /* :838-866 */
                    svt_ext_all_sad_calculation_8x8_16x16(s, stride, rp, stride, mv, bs + 21, bs + 5, bm + 21, bm + 5, e16, e8, (EbBool)sub_sad);
                    svt_ext_eight_sad_calculation_32x32_64x64(e16, bs + 1, bs, bm + 1, bm, mv, e32);
                } else {            /* leftover columns: open_loop_me_get_eight_search_point_results_block's single-point twin (:700-810) */
                    for (int Y = 0; Y < 4; Y++)
                        for (int X = 0; X < 4; X++) {
                            const int p = z16[Y * 4 + X];
                            svt_ext_sad_calculation_8x8_16x16(s + 16 * Y * stride + 16 * X, stride, rp + 16 * Y * stride + 16 * X, stride, bs + 21 + 4 * p,
                                                              bs + 5 + p, bm + 21 + 4 * p, bm + 5 + p, mv, s16 + p, s8 + 4 * p, (EbBool)sub_sad);
                        }
                    svt_ext_sad_calculation_32x32_64x64(s16, bs + 1, bs, bm + 1, bm, mv, s32);
                }
            }
        }
    }
}

/* ---------------------------------------------------------------- HME ---------------------------------------------------------------
 * one svt_sad_loop_kernel call per job, as hme_level_0/1/2 issue them (EbMotionEstimation.c:998,1146,1291) */
typedef struct { int32_t src_x, src_y, ref_x, ref_y; int16_t bw, bh, sa_w, sa_h, row_step, reserved; } RefbSadLoop;
void refb_sad_loop_batch(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, const RefbSadLoop *jobs, int begin, int end,
                         uint32_t *best_sad, int16_t *best_xy) {
    for (int i = begin; i < end; i++) {
        const RefbSadLoop *j = &jobs[i];
        uint64_t bs = 0xffffff;
        int16_t xc = 0, yc = 0;
        if (j->sa_w > 0 && j->sa_h > 0)
            svt_sad_loop_kernel((uint8_t *)src + (size_t)j->src_y * src_stride + j->src_x, (uint32_t)(src_stride * j->row_step),
                                (uint8_t *)ref + (size_t)j->ref_y * ref_stride + j->ref_x, (uint32_t)(ref_stride * j->row_step),
                                (uint32_t)(j->bh / j->row_step), (uint32_t)j->bw, &bs, &xc, &yc, (uint32_t)ref_stride, j->sa_w, j->sa_h);
        best_sad[i] = (uint32_t)bs;
        if (bs != 0xffffff) { best_xy[2 * i] = xc; best_xy[2 * i + 1] = yc; }
    }
}

/* ---------------------------------------------------------------- sub-pel prediction ------------------------------------------------
 * convolve[subpel_x != 0][subpel_y != 0][0] (Common/Codec/EbInterPrediction.c:1161-1173, :1419) on the block list of
 * svt_hip_subpel_predict_batch_dev (mode 0, 8-bit) */
typedef struct { int32_t src_x, src_y, dst_x, dst_y; uint8_t w, h, bank_x, bank_y, subpel_x, subpel_y, mode, reserved; } RefbConvBlk;
extern const int16_t sub_pel_filters_8[16][8], sub_pel_filters_8smooth[16][8], sub_pel_filters_8sharp[16][8], bilinear_filters[16][8],
    sub_pel_filters_4[16][8], sub_pel_filters_4smooth[16][8];
void refb_subpel_predict_batch(const uint8_t *ref, int ref_stride, uint8_t *dst, int dst_stride, const RefbConvBlk *blks, int begin, int end) {
    const int16_t *banks[6] = {&sub_pel_filters_8[0][0], &sub_pel_filters_8smooth[0][0], &sub_pel_filters_8sharp[0][0],
                               &bilinear_filters[0][0],  &sub_pel_filters_4[0][0],       &sub_pel_filters_4smooth[0][0]};
    for (int i = begin; i < end; i++) {
        const RefbConvBlk *b = &blks[i];
        InterpFilterParams fx = {banks[b->bank_x], 8, 16, (InterpFilter)0}, fy = {banks[b->bank_y], 8, 16, (InterpFilter)0};
        ConvolveParams cp;
        memset(&cp, 0, sizeof(cp));
        cp.round_0 = 3; cp.round_1 = 11;   /* get_conv_params_no_round for 8-bit single reference */
        const uint8_t *s = ref + (size_t)b->src_y * ref_stride + b->src_x;
        uint8_t *o = dst + (size_t)b->dst_y * dst_stride + b->dst_x;
        if (b->subpel_x && b->subpel_y) svt_av1_convolve_2d_sr(s, ref_stride, o, dst_stride, b->w, b->h, &fx, &fy, b->subpel_x, b->subpel_y, &cp);
        else if (b->subpel_x) svt_av1_convolve_x_sr(s, ref_stride, o, dst_stride, b->w, b->h, &fx, &fy, b->subpel_x, b->subpel_y, &cp);
        else if (b->subpel_y) svt_av1_convolve_y_sr(s, ref_stride, o, dst_stride, b->w, b->h, &fx, &fy, b->subpel_x, b->subpel_y, &cp);
        else svt_av1_convolve_2d_copy_sr(s, ref_stride, o, dst_stride, b->w, b->h, &fx, &fy, b->subpel_x, b->subpel_y, &cp);
    }
}

/* ---------------------------------------------------------------- sub-pel refinement probes ---------------------------------------------
 * svt_upsampled_pref_error (Encoder/Codec/mcomp.c:102-156) of a list of candidates: svt_aom_upsampled_pred (Encoder/C_DEFAULT/variance.c:212-269 behind the dispatch
 * pointer) into a scratch block, then the block size's variance function against the source (vfp->vf = svt_aom_variance{W}x{H}).  Job layout = SvtHipUpsampledBlk of
 * svt_hip_upsampled_pred_batch_dev + the source position of the candidate's block; square blocks 8 .. 64 (the bench's BASELINE configs[2] sub-line uses 16x16). */
typedef struct { int32_t ref_off, dst_off; uint8_t w, h, subpel_x_q3, subpel_y_q3, bank, reserved[3]; } RefbUpsBlk;
void refb_upsampled_var_batch(const uint8_t *ref, int ref_stride, const uint8_t *src, int src_stride, const RefbUpsBlk *jobs, const int32_t *src_off, int begin, int end,
                              uint32_t *var, uint32_t *sse) {
    DECLARE_ALIGNED(32, uint8_t, pred[128 * 128]);
    for (int i = begin; i < end; i++) {
        const RefbUpsBlk *b = &jobs[i];
        const int st = b->bank == 3 ? USE_2_TAPS : (b->bank == 4 ? USE_4_TAPS : USE_8_TAPS);
        svt_aom_upsampled_pred(NULL, NULL, 0, 0, NULL, pred, b->w, b->h, b->subpel_x_q3, b->subpel_y_q3, ref + b->ref_off, ref_stride, st);
        unsigned int e = 0;
        const uint8_t *sp = src + src_off[i];   /* the square sizes mode decision's sub-pel refinement probes (fn_ptr[bsize].vf, EbProductCodingLoop.c:2063) */
        var[i] = b->w != b->h ? 0 : b->w == 8 ? svt_aom_variance8x8(pred, 8, sp, src_stride, &e) : b->w == 16 ? svt_aom_variance16x16(pred, 16, sp, src_stride, &e)
                 : b->w == 32 ? svt_aom_variance32x32(pred, 32, sp, src_stride, &e) : b->w == 64 ? svt_aom_variance64x64(pred, 64, sp, src_stride, &e) : 0;
        sse[i] = e;
    }
}

/* ---------------------------------------------------------------- residual -> transform -> quantize -> inverse -> recon ------------
 * the calls of av1_encode_loop / full_loop for one transform block (Encoder/Codec/EbCodingLoop.c:560-700): svt_residual_kernel8bit,
 * av1_estimate_transform (EbTransforms.c:3613), svt_aom_quantize_b (EbFullLoop.c:269-310, no quantisation matrix),
 * av1_inv_transform_recon8bit (Common/Codec/EbInvTransforms.c:3167).  Descriptor / argument layout of orc_txfm_chain_8bit. */
void refb_txfm_chain_8bit(const uint8_t *src, int src_stride, const uint8_t *pred, int pred_stride, uint8_t *recon, int recon_stride,
                          const uint32_t *descs, int begin, int end, int tx_size, const int16_t qp[7][2], const int16_t *const scans[3],
                          const int16_t *const iscans[3], int log_scale, int32_t *qcoeff_out, uint16_t *eob_out) {
    const int W = tx_size_wide[tx_size], H = tx_size_high[tx_size], kw = W > 32 ? 32 : W, kh = H > 32 ? 32 : H, nk = kw * kh;
    int16_t *res = (int16_t *)svt_aom_memalign(64, sizeof(int16_t) * 64 * 64);
    int32_t *co = (int32_t *)svt_aom_memalign(64, sizeof(int32_t) * 64 * 64 * 3), *q = co + 4096, *dq = q + 4096;
    /* the reference's Quants / Dequants rows are int16[8] = {dc, ac x 7} (EbRateControlTasks / svt_av1_build_quantizer); the SIMD kernels load all 8 */
    DECLARE_ALIGNED(16, int16_t, q8[5][8]);
    for (int r = 0; r < 5; r++) for (int k = 0; k < 8; k++) q8[r][k] = qp[r][k ? 1 : 0];
    for (int i = begin; i < end; i++) {
        const int x = descs[i] & 0x3FFF, y = (descs[i] >> 14) & 0x3FFF, tt = descs[i] >> 28;
        svt_residual_kernel8bit((uint8_t *)src + (size_t)y * src_stride + x, src_stride, (uint8_t *)pred + (size_t)y * pred_stride + x, pred_stride, res, W, W, H);
        uint64_t energy = 0;
        av1_estimate_transform(res, W, co, W, (TxSize)tx_size, &energy, 8, (TxType)tt, PLANE_TYPE_Y, DEFAULT_SHAPE);
        const int cls = (W <= 16 && H <= 16) ? (tt < 10 ? 0 : ((tt & 1) ? 2 : 1)) : 0;
        uint16_t eob = 0;
        svt_aom_quantize_b(co, nk, q8[0], q8[1], q8[2], q8[3], q, dq, q8[4], &eob, scans[cls], iscans[cls], NULL, NULL, log_scale);
        if (qcoeff_out) memcpy(qcoeff_out + (size_t)i * nk, q, sizeof(int32_t) * nk);
        if (eob_out) eob_out[i] = eob;
        av1_inv_transform_recon8bit(dq, (uint8_t *)pred + (size_t)y * pred_stride + x, pred_stride, recon + (size_t)y * recon_stride + x, recon_stride,
                                    (TxSize)tx_size, (TxType)tt, PLANE_TYPE_Y, eob, 0);
    }
    svt_aom_free(res); svt_aom_free(co);
}

/* ---------------------------------------------------------------- deblocking --------------------------------------------------------
 * svt_aom_lpf_{vertical,horizontal}_{4,6,8,14} on the edge lists of orc_deblock_plane / svt_hip_deblock_plane_dev (all vertical
 * edges, then all horizontal ones: svt_av1_filter_block_plane_vert / _horz, Encoder/Codec/EbDeblockingFilter.c:390-612). */
void refb_deblock_plane(uint8_t *plane, int stride, const uint16_t *edges_v, const uint16_t *edges_h, int units_w, int units_h, int sharpness) {
    for (int dir = 0; dir < 2; dir++) {
        const uint16_t *e = dir == 0 ? edges_v : edges_h;
        for (int uy = 0; uy < units_h; uy++)
            for (int ux = 0; ux < units_w; ux++) {
                const int len = e[uy * units_w + ux] & 0xff, level = e[uy * units_w + ux] >> 8;
                if (!len) continue;
                int inside = level >> ((sharpness > 0) + (sharpness > 4));             /* update_sharpness, EbDeblockingCommon.c:587-606 */
                if (sharpness > 0 && inside > 9 - sharpness) inside = 9 - sharpness;
                if (inside < 1) inside = 1;
                DECLARE_ALIGNED(16, uint8_t, lim[16]); DECLARE_ALIGNED(16, uint8_t, mblim[16]); DECLARE_ALIGNED(16, uint8_t, hev[16]);
                memset(lim, inside, 16); memset(mblim, 2 * (level + 2) + inside, 16); memset(hev, level >> 4, 16);
                uint8_t *s = plane + (size_t)(4 * uy) * stride + 4 * ux;
                if (dir == 0) {
                    if (len == 4) svt_aom_lpf_vertical_4(s, stride, mblim, lim, hev);
                    else if (len == 6) svt_aom_lpf_vertical_6(s, stride, mblim, lim, hev);
                    else if (len == 8) svt_aom_lpf_vertical_8(s, stride, mblim, lim, hev);
                    else svt_aom_lpf_vertical_14(s, stride, mblim, lim, hev);
                } else {
                    if (len == 4) svt_aom_lpf_horizontal_4(s, stride, mblim, lim, hev);
                    else if (len == 6) svt_aom_lpf_horizontal_6(s, stride, mblim, lim, hev);
                    else if (len == 8) svt_aom_lpf_horizontal_8(s, stride, mblim, lim, hev);
                    else svt_aom_lpf_horizontal_14(s, stride, mblim, lim, hev);
                }
            }
    }
}

/* ---------------------------------------------------------------- CDEF --------------------------------------------------------------
 * Stage one 64x64 filter block of one plane into the 16-bit `in` buffer like cdef_seg_search does (Encoder/Codec/EbCdefProcess.c:223-241:
 * CDEF_VERY_LARGE everywhere, then copy_sb8_16 of the block plus the borders that exist inside the picture). */
static int stage_fb(uint16_t *inbuf, const uint8_t *plane, int stride, int fbr, int fbc, int nvfb, int nhfb, int nvb, int nhb, int dec) {
    for (int i = 0; i < CDEF_INBUF_SIZE; i++) inbuf[i] = CDEF_VERY_LARGE;
    uint16_t *in = inbuf + CDEF_VBORDER * CDEF_BSTRIDE + CDEF_HBORDER;
    const int l2 = MI_SIZE_LOG2 - dec;
    const int yoff = CDEF_VBORDER * (fbr != 0), xoff = CDEF_HBORDER * (fbc != 0);
    const int ysize = (nvb << l2) + CDEF_VBORDER * (fbr + 1 < nvfb) + yoff, xsize = (nhb << l2) + CDEF_HBORDER * (fbc + 1 < nhfb) + xoff;
    const int y0 = ((fbr * MI_SIZE_64X64) << l2) - yoff, x0 = ((fbc * MI_SIZE_64X64) << l2) - xoff;
    svt_copy_rect8_8bit_to_16bit(&in[-yoff * CDEF_BSTRIDE - xoff], CDEF_BSTRIDE, plane + (size_t)y0 * stride + x0, stride, ysize, xsize);
    return 0;
}
static int build_dlist(CdefList *dlist, const uint8_t *skip8, int c8, int fbr, int fbc, int nb_y, int nb_x) {
    int n = 0;                                  /* svt_sb_compute_cdef_list (EbCdef.c:35-78): every non-skip 8x8 block, raster order */
    for (int by = 0; by < nb_y; by++)
        for (int bx = 0; bx < nb_x; bx++)
            if (!skip8[(8 * fbr + by) * c8 + 8 * fbc + bx]) { dlist[n].by = (uint8_t)by; dlist[n].bx = (uint8_t)bx; dlist[n].skip = 0; n++; }
    return n;
}
/* cdef_seg_search (EbCdefProcess.c:80-280), all 64 strengths (pick_method 0), 8-bit 4:2:0; mse[2][nfb][64] like orc_cdef_search_frame */
void refb_cdef_search_frame(const uint8_t *const rec[3], const int rec_stride[3], const uint8_t *const src[3], const int src_stride[3], int w, int h,
                            const uint8_t *skip8, int pri_damping, uint64_t *mse, int fb_begin, int fb_end) {
    const int nhfb = (w + 63) / 64, nvfb = (h + 63) / 64, c8 = w / 8, mi_cols = w / 4, mi_rows = h / 4;
    uint16_t *inbuf = (uint16_t *)svt_aom_memalign(32, sizeof(uint16_t) * CDEF_INBUF_SIZE);
    uint8_t *tmp_dst = (uint8_t *)svt_aom_memalign(32, 1 << (MAX_SB_SIZE_LOG2 * 2));
    uint16_t *in = inbuf + CDEF_VBORDER * CDEF_BSTRIDE + CDEF_HBORDER;
    CdefList dlist[MI_SIZE_128X128 * MI_SIZE_128X128];
    int32_t dir[CDEF_NBLOCKS][CDEF_NBLOCKS] = {{0}}, var[CDEF_NBLOCKS][CDEF_NBLOCKS] = {{0}};
    for (int fb = fb_begin; fb < fb_end; fb++) {
        const int fbr = fb / nhfb, fbc = fb % nhfb;
        const int nhb = AOMMIN(MI_SIZE_64X64, mi_cols - MI_SIZE_64X64 * fbc), nvb = AOMMIN(MI_SIZE_64X64, mi_rows - MI_SIZE_64X64 * fbr);
        const int count = build_dlist(dlist, skip8, c8, fbr, fbc, nvb / 2, nhb / 2);
        if (!count) continue;                                                   /* svt_sb_all_skip */
        int32_t dirinit = 0;
        for (int pli = 0; pli < 3; pli++) {
            const int dec = pli ? 1 : 0;
            stage_fb(inbuf, rec[pli], rec_stride[pli], fbr, fbc, nvfb, nhfb, nvb, nhb, dec);
            const int l2 = MI_SIZE_LOG2 - dec;
            for (int gi = 0; gi < 64; gi++) {
                const int threshold = gi / CDEF_SEC_STRENGTHS, sec = gi % CDEF_SEC_STRENGTHS;
                svt_cdef_filter_fb(tmp_dst, NULL, CDEF_BSTRIDE, in, dec, dec, dir, &dirinit, var, pli, dlist, count, threshold, sec + (sec == 3),
                                   pri_damping, pri_damping, 0);
                const uint64_t m = svt_compute_cdef_dist_8bit(src[pli] + (size_t)((fbr * MI_SIZE_64X64) << l2) * src_stride[pli] + ((fbc * MI_SIZE_64X64) << l2),
                                                              src_stride[pli], tmp_dst, dlist, count, dec ? BLOCK_4X4 : BLOCK_8X8, 0, pli);
                uint64_t *o = mse + ((size_t)(pli ? 1 : 0) * ((size_t)nhfb * nvfb) + fb) * 64 + gi;
                if (pli == 2) *o += m; else *o = m;
            }
        }
    }
    svt_aom_free(inbuf); svt_aom_free(tmp_dst);
}
/* svt_av1_cdef_frame (Encoder/Codec/EbEncCdef.c:292-661) restricted to what it computes: per filter block with a non-zero strength,
 * svt_cdef_filter_fb straight into the output picture.  in[] = pre-CDEF planes, out[] starts as a copy of in[]. */
void refb_cdef_apply_frame(const uint8_t *const in_p[3], uint8_t *const out[3], const int stride[3], int w, int h, const uint8_t *skip8,
                           const uint8_t *y_strength, const uint8_t *uv_strength, int damping, int fb_begin, int fb_end) {
    const int nhfb = (w + 63) / 64, nvfb = (h + 63) / 64, c8 = w / 8, mi_cols = w / 4, mi_rows = h / 4;
    uint16_t *inbuf = (uint16_t *)svt_aom_memalign(32, sizeof(uint16_t) * CDEF_INBUF_SIZE);
    uint16_t *in = inbuf + CDEF_VBORDER * CDEF_BSTRIDE + CDEF_HBORDER;
    CdefList dlist[MI_SIZE_128X128 * MI_SIZE_128X128];
    int32_t dir[CDEF_NBLOCKS][CDEF_NBLOCKS] = {{0}}, var[CDEF_NBLOCKS][CDEF_NBLOCKS] = {{0}};
    for (int fb = fb_begin; fb < fb_end; fb++) {
        const int fbr = fb / nhfb, fbc = fb % nhfb;
        const int lv[2] = {y_strength[fb] / 4, uv_strength[fb] / 4};
        int sc[2] = {y_strength[fb] % 4, uv_strength[fb] % 4};
        sc[0] += sc[0] == 3; sc[1] += sc[1] == 3;
        if (!lv[0] && !sc[0] && !lv[1] && !sc[1]) continue;
        const int nhb = AOMMIN(MI_SIZE_64X64, mi_cols - MI_SIZE_64X64 * fbc), nvb = AOMMIN(MI_SIZE_64X64, mi_rows - MI_SIZE_64X64 * fbr);
        const int count = build_dlist(dlist, skip8, c8, fbr, fbc, nvb / 2, nhb / 2);
        if (!count) continue;
        for (int pli = 0; pli < 3; pli++) {
            const int dec = pli ? 1 : 0, l2 = MI_SIZE_LOG2 - dec;
            stage_fb(inbuf, in_p[pli], stride[pli], fbr, fbc, nvfb, nhfb, nvb, nhb, dec);
            svt_cdef_filter_fb(out[pli] + (size_t)((fbr * MI_SIZE_64X64) << l2) * stride[pli] + ((fbc * MI_SIZE_64X64) << l2), NULL, stride[pli], in, dec, dec,
                               dir, NULL, var, pli, dlist, count, lv[pli ? 1 : 0], sc[pli ? 1 : 0], damping, damping, 0);
        }
    }
    svt_aom_free(inbuf);
}

/* ---------------------------------------------------------------- self-guided restoration search -------------------------------------
 * search_selfguided_restoration (Encoder/Codec/EbRestorationPick.c:583-660) for every restoration unit of a plane and every
 * parameter set of ep_mask: apply_sgr (:554-581, svt_av1_selfguided_restoration per 64-wide processing unit) + svt_get_proj_subspace
 * (:497-538).  xq[unit][16][2] receives the projection coefficients (what the HIP path's sums + host solve produce). */
void refb_sgr_search_plane(const uint8_t *dgd, int stride, const uint8_t *src, int src_stride, const int32_t *limits, int unit_begin, int unit_end,
                           int pu_w, int pu_h, uint32_t ep_mask, int32_t *xq_out) {
    int32_t *flt0 = (int32_t *)svt_aom_memalign(32, sizeof(int32_t) * RESTORATION_UNITPELS_MAX * 2), *flt1 = flt0 + RESTORATION_UNITPELS_MAX;
    for (int u = unit_begin; u < unit_end; u++) {
        const int x0 = limits[4 * u], x1 = limits[4 * u + 1], y0 = limits[4 * u + 2], y1 = limits[4 * u + 3], w = x1 - x0, h = y1 - y0;
        const uint8_t *d = dgd + (size_t)y0 * stride + x0, *s = src + (size_t)y0 * src_stride + x0;
        const int fs = ((w + 7) & ~7) + 8;
        for (int ep = 0; ep < SGRPROJ_PARAMS; ep++) {
            if (!((ep_mask >> ep) & 1)) continue;
            for (int i = 0; i < h; i += pu_h)
                for (int j = 0; j < w; j += pu_w)
                    svt_av1_selfguided_restoration(d + (size_t)i * stride + j, AOMMIN(pu_w, w - j), AOMMIN(pu_h, h - i), stride, flt0 + i * fs + j, flt1 + i * fs + j,
                                                   fs, ep, 8, 0);
            int xq[2] = {0, 0};
            svt_get_proj_subspace(s, w, h, src_stride, d, stride, 0, flt0, fs, flt1, fs, xq, &eb_sgr_params[ep]);
            xq_out[((size_t)u * SGRPROJ_PARAMS + ep) * 2] = xq[0]; xq_out[((size_t)u * SGRPROJ_PARAMS + ep) * 2 + 1] = xq[1];
        }
    }
    svt_aom_free(flt0);
}

/* The COMPLETE per-unit search: search_selfguided_restoration (EbRestorationPick.c:583-671: filters, svt_get_proj_subspace, encode_xq and
 * finer_search_pixel_proj_error for every parameter set, best set) through the exported wrapper of oracle/ref_shim_restpick.c.
 * out[unit][3] = {ep, xqd0, xqd1}.  ref_ep = {-1, -1}: all 16 sets, like the first picture of a sequence. */
void ref_shim_sgr_search_unit(const uint8_t *dat8, int32_t width, int32_t height, int32_t dat_stride, const uint8_t *src8, int32_t src_stride,
                              int32_t use_highbitdepth, int32_t bit_depth, int32_t pu_width, int32_t pu_height, int32_t *rstbuf, int32_t ref_ep0,
                              int32_t ref_ep1, int32_t step, int32_t *out);
int32_t ref_shim_sgr_rstbuf_ints(void);
void refb_sgr_search_units_plane(const uint8_t *dgd, int stride, const uint8_t *src, int src_stride, const int32_t *limits, int unit_begin, int unit_end,
                                 int pu_w, int pu_h, int32_t *out) {
    int32_t *rstbuf = (int32_t *)svt_aom_memalign(32, sizeof(int32_t) * (size_t)ref_shim_sgr_rstbuf_ints());
    for (int u = unit_begin; u < unit_end; u++) {
        const int x0 = limits[4 * u], x1 = limits[4 * u + 1], y0 = limits[4 * u + 2], y1 = limits[4 * u + 3];
        ref_shim_sgr_search_unit(dgd + (size_t)y0 * stride + x0, x1 - x0, y1 - y0, stride, src + (size_t)y0 * src_stride + x0, src_stride, 0, 8, pu_w, pu_h, rstbuf,
                                 -1, -1, 16, out + 3 * (size_t)u);
    }
    svt_aom_free(rstbuf);
}

/* finish_cdef_search's decision on ONE picture's two distortion tables (EbEncCdef.c:1258-1298): the four joint_strength_search_dual sequences (a static
 * function: restated here as the greedy + refinement loop it is, every step through the dispatched svt_search_one_dual pointer, i.e. the reference's
 * AVX2 / AVX-512 kernel where the host has it), then the count of pairs by RDCOST and every filter block's pair (the oracle's restatement of that
 * arithmetic).  out: [0] = bits, [1..8] y strengths, [9..16] uv strengths; sel[sb_count].  One picture is one thread's work in the reference. */
static int refb_finish_tail(const uint64_t *mse0, const uint64_t *mse1, int sb_count, const int32_t (*lev0)[8], const int32_t (*lev1)[8], const uint64_t *tot, uint64_t lambda,
                            int32_t *y, int32_t *uv, int32_t *sel) {   /* EbEncCdef.c:1258-1298 (the same lines orc_cdef_finish restates; this library does not link the oracle) */
    uint64_t best = (uint64_t)1 << 63;
    int bits = 0;
    for (int i = 0; i <= 3; i++) {
        const int nb = 1 << i, total_bits = sb_count * i + nb * CDEF_STRENGTH_BITS * 2, rate_cost = total_bits * (1 << 9);
        const uint64_t dist = tot[i] * 16, cost = ((((uint64_t)rate_cost) * lambda + 256) >> 9) + dist * (1 << 7);
        if (cost < best) { best = cost; bits = i; }
    }
    const int nb = 1 << bits;
    for (int g = 0; g < 8; g++) { y[g] = g < nb ? lev0[bits][g] : 0; uv[g] = g < nb ? lev1[bits][g] : 0; }
    for (int i = 0; i < sb_count; i++) {
        uint64_t bm = (uint64_t)1 << 63;
        int bg = 0;
        for (int g = 0; g < nb; g++) {
            const uint64_t c = mse0[(size_t)i * 64 + y[g]] + mse1[(size_t)i * 64 + uv[g]];
            if (c < bm) { bm = c; bg = g; }
        }
        sel[i] = bg;
    }
    return bits;
}
void refb_cdef_finish(const uint64_t *mse0, const uint64_t *mse1, int sb_count, uint64_t lambda, int32_t *out, int32_t *sel) {
    uint64_t(*mse[2])[64] = {(uint64_t(*)[64])(uintptr_t)mse0, (uint64_t(*)[64])(uintptr_t)mse1};
    int32_t lev0[4][8] = {{0}}, lev1[4][8] = {{0}};
    uint64_t tot[4];
    for (int c = 0; c < 4; c++) {
        const int nb = 1 << c;
        int l0[8] = {0}, l1[8] = {0};
        uint64_t t = (uint64_t)1 << 63;
        for (int i = 0; i < nb; i++) t = svt_search_one_dual(l0, l1, i, mse, sb_count, 0, 64);
        for (int i = 0; i < 4 * nb; i++) {
            for (int j = 0; j < nb - 1; j++) { l0[j] = l0[j + 1]; l1[j] = l1[j + 1]; }
            t = svt_search_one_dual(l0, l1, nb - 1, mse, sb_count, 0, 64);
        }
        for (int i = 0; i < 8; i++) { lev0[c][i] = l0[i]; lev1[c][i] = l1[i]; }
        tot[c] = t;
    }
    out[0] = refb_finish_tail(mse0, mse1, sb_count, (const int32_t(*)[8])lev0, (const int32_t(*)[8])lev1, tot, lambda, out + 1, out + 9, sel);
}

/* ---------------------------------------------------------------- thread pool for bench.py's cpu_baseline ---------------------------
 * refb_parallel runs one of the drivers above over n items on n_threads pthreads (dynamic chunks from an atomic counter, so threads of
 * uneven speed stay busy), `reps` times, and returns the best (reps > 0) or the median (reps < 0, of -reps) wall time in seconds.  Arguments travel as an array of 64-bit slots
 * (pointers / integers), so the Python side needs no struct definitions; for the band stages (deblock, restoration apply) the array
 * holds one 16-slot record per band and an item is a band. */
#include <pthread.h>
#include <time.h>
int ref_shim_lr_apply_plane(int plane, int bd, int highbd, int frame_w, int frame_h, void *dbl, int dbl_stride, void *cdef, int stride,
                            void *dst, int dst_stride, int unit_size, const uint8_t *unit_ep, const int32_t *unit_xqd);
#define P(i, T) ((T)(uintptr_t)a[i])
#define I(i) ((int)a[i])
static void run_range(int stage, const int64_t *a, int b, int e) {
    switch (stage) {
    case 0: refb_me_fullpel_frame(P(0, const uint8_t *), P(1, const uint8_t *), I(2), I(3), I(4), P(5, const RefbSbSearch *), I(6), I(7), P(8, uint32_t *), P(9, uint32_t *), b, e); break;
    case 1: refb_sad_loop_batch(P(0, const uint8_t *), I(1), P(2, const uint8_t *), I(3), P(4, const RefbSadLoop *), b, e, P(5, uint32_t *), P(6, int16_t *)); break;
    case 2: refb_subpel_predict_batch(P(0, const uint8_t *), I(1), P(2, uint8_t *), I(3), P(4, const RefbConvBlk *), b, e); break;
    case 3: {
        const int16_t *sc[3] = {P(10, const int16_t *), P(11, const int16_t *), P(12, const int16_t *)}, *isc[3] = {P(13, const int16_t *), P(14, const int16_t *), P(15, const int16_t *)};
        refb_txfm_chain_8bit(P(0, const uint8_t *), I(1), P(2, const uint8_t *), I(3), P(4, uint8_t *), I(5), P(6, const uint32_t *), b, e, I(7), P(8, const int16_t(*)[2]), sc, isc, I(9), NULL, NULL);
        break;
    }
    case 4: for (int k = b; k < e; k++) { const int64_t *r = a + 16 * k; refb_deblock_plane((uint8_t *)(uintptr_t)r[0], (int)r[1], (const uint16_t *)(uintptr_t)r[2], (const uint16_t *)(uintptr_t)r[3], (int)r[4], (int)r[5], 0); } break;
    case 5: {
        const uint8_t *rec[3] = {P(0, const uint8_t *), P(1, const uint8_t *), P(2, const uint8_t *)}, *src[3] = {P(6, const uint8_t *), P(7, const uint8_t *), P(8, const uint8_t *)};
        const int rs[3] = {I(3), I(4), I(5)}, ss[3] = {I(9), I(10), I(11)};
        refb_cdef_search_frame(rec, rs, src, ss, I(12), I(13), P(14, const uint8_t *), I(15), P(16, uint64_t *), b, e);
        break;
    }
    case 6: {
        const uint8_t *in[3] = {P(0, const uint8_t *), P(1, const uint8_t *), P(2, const uint8_t *)};
        uint8_t *out[3] = {P(6, uint8_t *), P(7, uint8_t *), P(8, uint8_t *)};
        const int st[3] = {I(3), I(4), I(5)};
        refb_cdef_apply_frame(in, out, st, I(9), I(10), P(11, const uint8_t *), P(12, const uint8_t *), P(13, const uint8_t *), I(14), b, e);
        break;
    }
    case 7: refb_sgr_search_plane(P(0, const uint8_t *), I(1), P(2, const uint8_t *), I(3), P(4, const int32_t *), b, e, I(5), I(6), (uint32_t)a[7], P(8, int32_t *)); break;
    case 8: for (int k = b; k < e; k++) {
            const int64_t *r = a + 16 * k;
            ref_shim_lr_apply_plane((int)r[0], 8, 0, (int)r[1], (int)r[2], (void *)(uintptr_t)r[3], (int)r[4], (void *)(uintptr_t)r[5], (int)r[6], (void *)(uintptr_t)r[7], (int)r[8], (int)r[9],
                                    (const uint8_t *)(uintptr_t)r[10], (const int32_t *)(uintptr_t)r[11]);
        } break;
    case 10: for (int k = b; k < e; k++) refb_cdef_finish(P(0, const uint64_t *), P(1, const uint64_t *), I(2), (uint64_t)a[3], P(4, int32_t *) + 17 * k, P(5, int32_t *) + (size_t)I(2) * k); break;   /* item = one picture */
    case 9: refb_sgr_search_units_plane(P(0, const uint8_t *), I(1), P(2, const uint8_t *), I(3), P(4, const int32_t *), b, e, I(5), I(6), P(7, int32_t *)); break;
    default: break;
    }
}
#undef P
#undef I
typedef struct { int stage; const int64_t *args; int n, chunk; volatile int next; pthread_barrier_t start, done; } RefbPool;
static void *pool_worker(void *p_) {
    RefbPool *p = (RefbPool *)p_;
    pthread_barrier_wait(&p->start);
    for (;;) {
        const int b = __sync_fetch_and_add(&p->next, p->chunk);
        if (b >= p->n) break;
        run_range(p->stage, p->args, b, b + p->chunk < p->n ? b + p->chunk : p->n);
    }
    pthread_barrier_wait(&p->done);
    return NULL;
}
/* the clock runs from the moment every thread exists (start barrier) to the moment the last one is out of work (done barrier) */
/* reps > 0: the best of `reps` wall times; reps < 0: the MEDIAN of -reps ("median of >= 5") */
double refb_parallel(int stage, const int64_t *args, int n, int chunk, int n_threads, int reps) {
    double best = -1, all[33];
    const int median = reps < 0;
    if (median) reps = -reps > 33 ? 33 : -reps;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int r = 0; r < reps; r++) {
        RefbPool p;
        p.stage = stage; p.args = args; p.n = n; p.chunk = chunk > 0 ? chunk : 1; p.next = 0;
        pthread_barrier_init(&p.start, NULL, (unsigned)n_threads + 1); pthread_barrier_init(&p.done, NULL, (unsigned)n_threads + 1);
        for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, pool_worker, &p);
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        pthread_barrier_wait(&p.start);
        pthread_barrier_wait(&p.done);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
        pthread_barrier_destroy(&p.start); pthread_barrier_destroy(&p.done);
        const double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (best < 0 || s < best) best = s;
        if (median) all[r] = s;
    }
    free(th);
    if (median) {
        for (int i = 1; i < reps; i++) { const double v = all[i]; int j = i; while (j > 0 && all[j - 1] > v) { all[j] = all[j - 1]; j--; } all[j] = v; }
        return all[reps / 2];
    }
    return best;
}

/* ---------------------------------------------------------------- high bit depth (BASELINE configs[3]) ------------------------------
 * The same kind of drivers for 16-bit pictures: sad_16b_kernel + svt_aom_highbd_10_variance{W}x{H} on block pairs, the 64-point (or any)
 * transform chain through av1_estimate_transform / svt_aom_highbd_quantize_b / av1_inv_transform_recon, and the self-guided search with
 * use_highbitdepth.  (The restoration apply pass is ref_shim_lr_apply_plane with highbd = 1.) */
typedef struct { int32_t a_x, a_y, b_x, b_y; uint16_t w, h; } RefbBlkPair;
void refb_hbd_sad_var_batch(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, const RefbBlkPair *p, int begin, int end, uint32_t *sad, uint32_t *var,
                            uint32_t *sse) {
    for (int i = begin; i < end; i++) {
        uint16_t *pa = (uint16_t *)a + (size_t)p[i].a_y * a_stride + p[i].a_x, *pb = (uint16_t *)b + (size_t)p[i].b_y * b_stride + p[i].b_x;
        sad[i] = sad_16b_kernel(pa, (uint32_t)a_stride, pb, (uint32_t)b_stride, p[i].h, p[i].w);
        unsigned int s = 0, v = 0xFFFFFFFFu;
        const uint8_t *a8 = CONVERT_TO_BYTEPTR(pa), *b8 = CONVERT_TO_BYTEPTR(pb);
        if (p[i].w == 64 && p[i].h == 64) v = svt_aom_highbd_10_variance64x64(a8, a_stride, b8, b_stride, &s);
        else if (p[i].w == 32 && p[i].h == 32) v = svt_aom_highbd_10_variance32x32(a8, a_stride, b8, b_stride, &s);
        else if (p[i].w == 16 && p[i].h == 16) v = svt_aom_highbd_10_variance16x16(a8, a_stride, b8, b_stride, &s);
        else if (p[i].w == 8 && p[i].h == 8) v = svt_aom_highbd_10_variance8x8(a8, a_stride, b8, b_stride, &s);
        var[i] = v; sse[i] = s;
    }
}
void refb_txfm_chain_hbd(const uint16_t *src, int src_stride, const uint16_t *pred, int pred_stride, uint16_t *recon, int recon_stride, const uint32_t *descs, int begin,
                         int end, int tx_size, int bd, const int16_t qp[7][2], const int16_t *const scans[3], const int16_t *const iscans[3], int log_scale,
                         int32_t *qcoeff_out, int32_t *dqcoeff_out, uint16_t *eob_out) {
    const int W = tx_size_wide[tx_size], H = tx_size_high[tx_size], kw = W > 32 ? 32 : W, kh = H > 32 ? 32 : H, nk = kw * kh;
    int16_t *res = (int16_t *)svt_aom_memalign(64, sizeof(int16_t) * 64 * 64);
    int32_t *co = (int32_t *)svt_aom_memalign(64, sizeof(int32_t) * 64 * 64 * 3), *q = co + 4096, *dq = q + 4096;
    DECLARE_ALIGNED(16, int16_t, q8[5][8]);
    for (int r = 0; r < 5; r++) for (int k = 0; k < 8; k++) q8[r][k] = qp[r][k ? 1 : 0];
    for (int i = begin; i < end; i++) {
        const int x = descs[i] & 0x3FFF, y = (descs[i] >> 14) & 0x3FFF, tt = descs[i] >> 28;
        for (int r = 0; r < H; r++)      /* svt_residual_kernel16bit */
            for (int c = 0; c < W; c++) res[r * W + c] = (int16_t)((int)src[(size_t)(y + r) * src_stride + x + c] - (int)pred[(size_t)(y + r) * pred_stride + x + c]);
        uint64_t energy = 0;
        av1_estimate_transform(res, W, co, W, (TxSize)tx_size, &energy, bd, (TxType)tt, PLANE_TYPE_Y, DEFAULT_SHAPE);
        const int cls = (W <= 16 && H <= 16) ? (tt < 10 ? 0 : ((tt & 1) ? 2 : 1)) : 0;
        uint16_t eob = 0;
        svt_aom_highbd_quantize_b(co, nk, q8[0], q8[1], q8[2], q8[3], q, dq, q8[4], &eob, scans[cls], iscans[cls], NULL, NULL, log_scale);
        if (qcoeff_out) memcpy(qcoeff_out + (size_t)i * nk, q, sizeof(int32_t) * nk);
        if (dqcoeff_out) memcpy(dqcoeff_out + (size_t)i * nk, dq, sizeof(int32_t) * nk);
        if (eob_out) eob_out[i] = eob;
        av1_inv_transform_recon(dq, CONVERT_TO_BYTEPTR((uint16_t *)pred + (size_t)y * pred_stride + x), pred_stride,
                                CONVERT_TO_BYTEPTR(recon + (size_t)y * recon_stride + x), recon_stride, (TxSize)tx_size, bd, (TxType)tt, PLANE_TYPE_Y, eob, 0);
    }
    svt_aom_free(res); svt_aom_free(co);
}
void refb_sgr_search_plane_hbd(const uint16_t *dgd, int stride, const uint16_t *src, int src_stride, const int32_t *limits, int unit_begin, int unit_end, int pu_w,
                               int pu_h, uint32_t ep_mask, int bd, int32_t *xq_out) {
    int32_t *flt0 = (int32_t *)svt_aom_memalign(32, sizeof(int32_t) * RESTORATION_UNITPELS_MAX * 2), *flt1 = flt0 + RESTORATION_UNITPELS_MAX;
    for (int u = unit_begin; u < unit_end; u++) {
        const int x0 = limits[4 * u], x1 = limits[4 * u + 1], y0 = limits[4 * u + 2], y1 = limits[4 * u + 3], w = x1 - x0, h = y1 - y0;
        const uint16_t *d = dgd + (size_t)y0 * stride + x0, *s = src + (size_t)y0 * src_stride + x0;
        const int fs = ((w + 7) & ~7) + 8;
        for (int ep = 0; ep < SGRPROJ_PARAMS; ep++) {
            if (!((ep_mask >> ep) & 1)) continue;
            for (int i = 0; i < h; i += pu_h)
                for (int j = 0; j < w; j += pu_w)
                    svt_av1_selfguided_restoration(CONVERT_TO_BYTEPTR(d + (size_t)i * stride + j), AOMMIN(pu_w, w - j), AOMMIN(pu_h, h - i), stride, flt0 + i * fs + j,
                                                   flt1 + i * fs + j, fs, ep, bd, 1);
            int xq[2] = {0, 0};
            svt_get_proj_subspace(CONVERT_TO_BYTEPTR(s), w, h, src_stride, CONVERT_TO_BYTEPTR(d), stride, 1, flt0, fs, flt1, fs, xq, &eb_sgr_params[ep]);
            xq_out[((size_t)u * SGRPROJ_PARAMS + ep) * 2] = xq[0]; xq_out[((size_t)u * SGRPROJ_PARAMS + ep) * 2 + 1] = xq[1];
        }
    }
    svt_aom_free(flt0);
}
