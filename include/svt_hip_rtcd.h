/*
 * svt_hip_rtcd.h — per-call wrappers with the reference's RTCD signatures ("per-call
 * pointer-compatible wrappers ... installed by setup_rtcd_hip()").
 *
 * The production boundary is the batched ABI in svt_hip.h (one call = one kernel class over a frame).
 * These wrappers exist so that (a) the reference's own unit tests / a bring-up build can run every
 * call site through the GPU kernels one block at a time and compare, and (b) a maintainer can switch
 * pointers over incrementally.  Each wrapper stages its host buffers to the device, launches the SAME
 * kernel the batched entry point uses with a batch of one, and copies the result back: exact, slow.
 *
 * Signatures are the ones in Source/Lib/Encoder/Codec/aom_dsp_rtcd.h and
 * Source/Lib/Common/Codec/common_dsp_rtcd.h (line numbers next to each member).  Struct parameters are
 * declared layout-compatible here so the header needs none of the reference's headers.
 *
 * Error convention: a wrapper never reports failure through its signature; if the HIP
 * path fails it logs to stderr and calls the C pointer that was in the table when
 * svt_hip_setup_rtcd() ran (a NULL saved pointer + a HIP failure aborts loudly, never a silent CPU
 * fallback inside the product path: the fallback is the REFERENCE's own function, supplied by the caller).
 * Thread safety: wrappers serialise on one mutex (the reference calls these pointers from many
 * threads); the batched ABI is the concurrent path.
 */
#ifndef SVT_HIP_RTCD_H
#define SVT_HIP_RTCD_H
#include <stddef.h>
#include <stdint.h>
#include "svt_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the functions declared in this header are exported */
#pragma GCC visibility push(default)

/* InterpFilterParams / ConvolveParams, Source/Lib/Common/Codec/EbDefinitions.h:493-498 / :379-392 */
typedef struct {
    const int16_t *filter_ptr;
    uint16_t       taps, subpel_shifts;
    uint8_t        interp_filter; /* InterpFilter (ATTRIBUTE_PACKED enum) */
} SvtHipInterpFilterParams;
typedef struct {
    int32_t   ref, do_average;
    uint16_t *dst; /* ConvBufType* */
    int32_t   dst_stride, round_0, round_1, plane, is_compound, use_jnt_comp_avg, fwd_offset, bck_offset, use_dist_wtd_comp_avg;
} SvtHipConvolveParams;

typedef void (*SvtHipSadLoopFn)(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                                uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center,
                                uint32_t src_stride_raw, int16_t search_area_width, int16_t search_area_height);
typedef uint32_t (*SvtHipNxmSadFn)(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height,
                                   uint32_t width);
typedef uint32_t (*SvtHipSadWxHFn)(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride);
typedef unsigned (*SvtHipVarWxHFn)(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, unsigned *sse);
typedef void (*SvtHipConvolveSrFn)(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride, int32_t w, int32_t h,
                                   SvtHipInterpFilterParams *filter_params_x, SvtHipInterpFilterParams *filter_params_y,
                                   const int32_t subpel_x_q4, const int32_t subpel_y_q4, SvtHipConvolveParams *conv_params);
typedef void (*SvtHipHbdConvolveSrFn)(const uint16_t *src, int32_t src_stride, uint16_t *dst, int32_t dst_stride, int32_t w, int32_t h,
                                      const SvtHipInterpFilterParams *filter_params_x, const SvtHipInterpFilterParams *filter_params_y,
                                      const int32_t subpel_x_q4, const int32_t subpel_y_q4, SvtHipConvolveParams *conv_params, int32_t bd);
typedef void (*SvtHipFwdTxfmFn)(int16_t *input, int32_t *output, uint32_t input_stride, uint8_t transform_type, uint8_t bit_depth);
typedef void (*SvtHipInvTxfmSqFn)(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w,
                                  uint8_t tx_type, int32_t bd);
typedef void (*SvtHipInvTxfmRectFn)(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w,
                                    uint8_t tx_type, uint8_t tx_size, int32_t eob, int32_t bd);
typedef void (*SvtHipInvTxfmRect4Fn)(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w,
                                     uint8_t tx_type, uint8_t tx_size, int32_t bd);
typedef void (*SvtHipSgrFilterFn)(const uint8_t *dgd8, int32_t width, int32_t height, int32_t stride, int32_t *flt0, int32_t *flt1,
                                  int32_t flt_stride, int32_t sgr_params_idx, int32_t bit_depth, int32_t highbd);
typedef void (*SvtHipSgrApplyFn)(const uint8_t *dat, int32_t width, int32_t height, int32_t stride, int32_t eps, const int32_t *xqd,
                                 uint8_t *dst, int32_t dst_stride, int32_t *tmpbuf, int32_t bit_depth, int32_t highbd);

typedef unsigned (*SvtHipObmcSadFn)(const uint8_t *pre, int pre_stride, const int32_t *wsrc, const int32_t *mask);
typedef unsigned (*SvtHipObmcVarFn)(const uint8_t *pre, int pre_stride, const int32_t *wsrc, const int32_t *mask, unsigned *sse);
typedef unsigned (*SvtHipObmcSubpixVarFn)(const uint8_t *pre, int pre_stride, int xoffset, int yoffset, const int32_t *wsrc, const int32_t *mask,
                                          unsigned *sse);
typedef void (*SvtHipBlendMaskFn)(uint8_t *dst, uint32_t dst_stride, const uint8_t *src0, uint32_t src0_stride, const uint8_t *src1,
                                  uint32_t src1_stride, const uint8_t *mask, uint32_t mask_stride, int w, int h, int subw, int subh);
typedef void (*SvtHipBlendHVMaskFn)(uint8_t *dst, uint32_t dst_stride, const uint8_t *src0, uint32_t src0_stride, const uint8_t *src1,
                                    uint32_t src1_stride, const uint8_t *mask, int w, int h);
typedef void (*SvtHipHbdBlendMaskFn)(uint8_t *dst, uint32_t dst_stride, const uint8_t *src0, uint32_t src0_stride, const uint8_t *src1,
                                     uint32_t src1_stride, const uint8_t *mask, uint32_t mask_stride, int w, int h, int subw, int subh, int bd);
typedef void (*SvtHipHbdBlendHVMaskFn)(uint8_t *dst, uint32_t dst_stride, const uint8_t *src0, uint32_t src0_stride, const uint8_t *src1,
                                       uint32_t src1_stride, const uint8_t *mask, int w, int h, int bd);
typedef void (*SvtHipWarpAffineFn)(const int32_t *mat, const uint8_t *ref, int width, int height, int stride, uint8_t *pred, int p_col, int p_row,
                                   int p_width, int p_height, int p_stride, int subsampling_x, int subsampling_y,
                                   SvtHipConvolveParams *conv_params, int16_t alpha, int16_t beta, int16_t gamma, int16_t delta);
typedef void (*SvtHipHbdWarpAffineFn)(const int32_t *mat, const uint16_t *ref, int width, int height, int stride, uint16_t *pred, int p_col,
                                      int p_row, int p_width, int p_height, int p_stride, int subsampling_x, int subsampling_y, int bd,
                                      SvtHipConvolveParams *conv_params, int16_t alpha, int16_t beta, int16_t gamma, int16_t delta);
typedef void (*SvtHipComputeStatsFn)(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end,
                                     int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H);
typedef void (*SvtHipHbdComputeStatsFn)(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end,
                                        int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H,
                                        int32_t bit_depth /* AomBitDepth */);

typedef void (*SvtHipExtAllSadFn)(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8,
                                  uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8],
                                  uint32_t p_eight_sad8x8[64][8], uint8_t sub_sad /* EbBool */);
typedef void (*SvtHipExtEightSadFn)(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                    uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]);
/* TranLow = int32_t, QmVal = uint8_t (EbDefinitions.h:668) */
typedef void (*SvtHipQuantizeBFn)(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                  const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
                                  const int16_t *scan, const int16_t *iscan, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, const int32_t log_scale);
typedef void (*SvtHipQuantizeFpFn)(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                   const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
                                   const int16_t *scan, const int16_t *iscan);
typedef void (*SvtHipHbdQuantizeFpFn)(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr,
                                      const int16_t *quant_ptr, const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr,
                                      const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, const int16_t *iscan, int16_t log_scale);
typedef void (*SvtHipLpfFn)(uint8_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
typedef void (*SvtHipHbdLpfFn)(uint16_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int32_t bd);
typedef int32_t (*SvtHipCdefFindDirFn)(const uint16_t *img, int32_t stride, int32_t *var, int32_t coeff_shift);
typedef void (*SvtHipCdefFilterBlockFn)(uint8_t *dst8, uint16_t *dst16, int32_t dstride, const uint16_t *in, int32_t pri_strength, int32_t sec_strength,
                                        int32_t dir, int32_t pri_damping, int32_t sec_damping, int32_t bsize, int32_t coeff_shift);
typedef void (*SvtHipResidual8Fn)(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride,
                                  uint32_t area_width, uint32_t area_height);
typedef void (*SvtHipResidual16Fn)(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride,
                                   uint32_t area_width, uint32_t area_height);
typedef void (*SvtHipSadx4dFn)(const uint8_t *src_ptr, int src_stride, const uint8_t *const ref_ptr[], int ref_stride, uint32_t *sad_array);
/* MacroBlockD*, AV1Common*, MV* are unused by the reference's function (C_DEFAULT/variance.c:218-222) and opaque here */
typedef void (*SvtHipUpsampledPredFn)(void *xd, const void *cm, int mi_row, int mi_col, const void *mv, uint8_t *comp_pred, int width, int height,
                                      int subpel_x_q3, int subpel_y_q3, const uint8_t *ref, int ref_stride, int subpel_search);
typedef void (*SvtHipIntermVarFn)(uint8_t *input_samples, uint16_t input_stride, uint64_t *mean_of8x8_blocks, uint64_t *mean_of_squared8x8_blocks);
typedef uint64_t (*SvtHipHandleTransformFn)(int32_t *output);
/* TxfmParam, EbDefinitions.h:779-791 (TxType / TxSize / TxSetType are one-byte enums) */
typedef struct {
    uint8_t tx_type, tx_size;
    int32_t lossless, bd, is_hbd;
    uint8_t tx_set_type;
    int32_t eob;
} SvtHipTxfmParam;
typedef void (*SvtHipInvTxfmAddFn)(const int32_t *dqcoeff, uint8_t *dst_r, int32_t stride_r, uint8_t *dst_w, int32_t stride_w, const SvtHipTxfmParam *txfm_param);
typedef void (*SvtHipConvert8To16Fn)(uint8_t *src, uint32_t src_stride, uint16_t *dst, uint32_t dst_stride, uint32_t width, uint32_t height);
typedef void (*SvtHipConvert16To8Fn)(uint16_t *src, uint32_t src_stride, uint8_t *dst, uint32_t dst_stride, uint32_t width, uint32_t height);
typedef void (*SvtHipCPackFn)(const uint8_t *inn_bit_buffer, uint32_t inn_stride, uint8_t *in_compn_bit_buffer, uint32_t out_stride, uint8_t *local_cache,
                              uint32_t width, uint32_t height);
typedef void (*SvtHipPackMsbFn)(uint8_t *in8_bit_buffer, uint32_t in8_stride, uint8_t *inn_bit_buffer, uint16_t *out16_bit_buffer, uint32_t inn_stride,
                                uint32_t out_stride, uint32_t width, uint32_t height);
typedef void (*SvtHipUnpackAvgFn)(uint16_t *ref16_l0, uint32_t ref_l0_stride, uint16_t *ref16_l1, uint32_t ref_l1_stride, uint8_t *dst_ptr, uint32_t dst_stride,
                                  uint32_t width, uint32_t height);
typedef void (*SvtHipUnPack2dFn)(uint16_t *in16_bit_buffer, uint32_t in_stride, uint8_t *out8_bit_buffer, uint8_t *outn_bit_buffer, uint32_t out8_stride,
                                 uint32_t outn_stride, uint32_t width, uint32_t height);
typedef void (*SvtHipUnPack8Fn)(uint16_t *in16_bit_buffer, uint32_t in_stride, uint8_t *out8_bit_buffer, uint32_t out8_stride, uint32_t width, uint32_t height);
typedef void (*SvtHipDiffwtdMaskFn)(uint8_t *mask, uint8_t mask_type, const uint8_t *src0, int src0_stride, const uint8_t *src1, int src1_stride, int h, int w);
typedef void (*SvtHipDiffwtdMaskHbdFn)(uint8_t *mask, uint8_t mask_type, const uint8_t *src0, int src0_stride, const uint8_t *src1, int src1_stride, int h, int w, int bd);
typedef void (*SvtHipDiffwtdMaskD16Fn)(uint8_t *mask, uint8_t mask_type, const uint16_t *src0, int src0_stride, const uint16_t *src1, int src1_stride, int h, int w,
                                       SvtHipConvolveParams *conv_params, int bd);
typedef void (*SvtHipBlendD16Fn)(uint8_t *dst, uint32_t dst_stride, const uint16_t *src0, uint32_t src0_stride, const uint16_t *src1, uint32_t src1_stride,
                                 const uint8_t *mask, uint32_t mask_stride, int w, int h, int subw, int subh, SvtHipConvolveParams *conv_params);
typedef void (*SvtHipHbdBlendD16Fn)(uint8_t *dst, uint32_t dst_stride, const uint16_t *src0, uint32_t src0_stride, const uint16_t *src1, uint32_t src1_stride,
                                    const uint8_t *mask, uint32_t mask_stride, int w, int h, int subw, int subh, SvtHipConvolveParams *conv_params, int bd);
typedef void (*SvtHipHbdMseFn)(const uint8_t *src_ptr, int32_t source_stride, const uint8_t *ref_ptr, int32_t recon_stride, uint32_t *sse);
typedef void (*SvtHipSubtractBlockFn)(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride,
                                      const uint8_t *pred_ptr, ptrdiff_t pred_stride);
typedef void (*SvtHipHbdSubtractBlockFn)(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride,
                                         const uint8_t *pred_ptr, ptrdiff_t pred_stride, int bd);
typedef uint32_t (*SvtHipSad16bFn)(uint16_t *src, uint32_t src_stride, uint16_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);
typedef uint32_t (*SvtHipVarianceHbdFn)(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h, uint32_t *sse);
typedef void (*SvtHipExtSad16Fn)(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16,
                                 uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16, uint32_t *p_sad8x8, uint8_t sub_sad);
typedef void (*SvtHipExtSad3264Fn)(uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                   uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32);
typedef void (*SvtHipCopyRect8To16Fn)(uint16_t *dst, int32_t dstride, const uint8_t *src, int32_t sstride, int32_t v, int32_t h);
typedef struct { uint8_t by, bx, skip; } SvtHipCdefList; /* CdefList, EbDefinitions.h:77-81 */
/* bsize: BlockSize, a one-byte enum (BLOCK_4X4 0, BLOCK_4X8 1, BLOCK_8X4 2, BLOCK_8X8 3) */
typedef uint64_t (*SvtHipCdefDist8Fn)(const uint8_t *dst8, int32_t dstride, const uint8_t *src8, const SvtHipCdefList *dlist, int32_t cdef_count, uint8_t bsize,
                                      int32_t coeff_shift, int32_t pli);
typedef uint64_t (*SvtHipCdefDist16Fn)(const uint16_t *dst, int32_t dstride, const uint16_t *src, const SvtHipCdefList *dlist, int32_t cdef_count, uint8_t bsize,
                                       int32_t coeff_shift, int32_t pli);
typedef uint64_t (*SvtHipSearchOneDualFn)(int *lev0, int *lev1, int nb_strengths, uint64_t (**mse)[64], int sb_count, int start_gi, int end_gi);
typedef void (*SvtHipFullDist32Fn)(int32_t *coeff, uint32_t coeff_stride, int32_t *recon_coeff, uint32_t recon_coeff_stride, uint64_t distortion_result[2],
                                   uint32_t area_width, uint32_t area_height);
typedef void (*SvtHipFullDistCbfZero32Fn)(int32_t *coeff, uint32_t coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height);
typedef uint64_t (*SvtHipSpatialDistFn)(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset, uint32_t recon_stride,
                                        uint32_t area_width, uint32_t area_height);
typedef int64_t (*SvtHipSseFn)(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int width, int height);
typedef int (*SvtHipSatdFn)(const int32_t *coeff, int length);
typedef int64_t (*SvtHipBlockErrorFn)(const int32_t *coeff, const int32_t *dqcoeff, intptr_t block_size, int64_t *ssz);
typedef struct { int32_t r[2], s[2]; } SvtHipSgrParamsType; /* SgrParamsType, EbDefinitions.h:1483-1486 */
typedef void (*SvtHipGetProjSubspaceFn)(const uint8_t *src8, int width, int height, int src_stride, const uint8_t *dat8, int dat_stride, int use_highbitdepth,
                                        int32_t *flt0, int flt0_stride, int32_t *flt1, int flt1_stride, int *xq, const SvtHipSgrParamsType *params);
typedef int64_t (*SvtHipPixelProjErrorFn)(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8, int32_t dat_stride,
                                          int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParamsType *params);
typedef uint64_t (*SvtHipMeanSq8x8Fn)(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height);
typedef uint64_t (*SvtHipSubMean8x8Fn)(uint8_t *input_samples, uint16_t input_stride);
typedef void (*SvtHipConvolve8Fn)(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x, int x_step_q4,
                                  const int16_t *filter_y, int y_step_q4, int w, int h);
typedef void (*SvtHipWienerConvolveFn)(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x,
                                       const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params);
typedef void (*SvtHipHbdWienerConvolveFn)(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x,
                                          const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params, int32_t bd);

/* The 22 block sizes of svt_aom_sad{W}x{H} / svt_aom_variance{W}x{H} in BlockSize order
 * (aom_dsp_rtcd.h:334-336, :524): index = position in this list. */
#define SVT_HIP_RTCD_BLOCK_SIZES(X) /* X(index, W, H) */ \
    X(0, 4, 4) X(1, 4, 8) X(2, 8, 4) X(3, 8, 8) X(4, 8, 16) X(5, 16, 8) X(6, 16, 16) X(7, 16, 32) X(8, 32, 16) X(9, 32, 32) X(10, 32, 64) X(11, 64, 32) X(12, 64, 64) X(13, 64, 128) X(14, 128, 64) X(15, 128, 128) X(16, 4, 16) X(17, 16, 4) X(18, 8, 32) X(19, 32, 8) X(20, 16, 64) X(21, 64, 16)
/* The 14 forward transform sizes without a 64-point dimension, TxSize order (aom_dsp_rtcd.h:129-135); the 64-point
 * sizes only exist fused with svt_handle_transform64x* in the batched entry point (packed 32x32 output). */
#define SVT_HIP_RTCD_FWD_SIZES(X) /* X(index, TxSize, W, H) */ \
    X(0, 0, 4, 4) X(1, 1, 8, 8) X(2, 2, 16, 16) X(3, 3, 32, 32) X(4, 5, 4, 8) X(5, 6, 8, 4) X(6, 7, 8, 16) X(7, 8, 16, 8) X(8, 9, 16, 32) X(9, 10, 32, 16) X(10, 13, 4, 16) X(11, 14, 16, 4) X(12, 15, 8, 32) X(13, 16, 32, 8)

typedef struct SvtHipRtcd {
    SvtHipSadLoopFn       svt_sad_loop_kernel;              /* aom_dsp_rtcd.h:597 */
    SvtHipNxmSadFn        svt_nxm_sad_kernel;               /* :644 */
    SvtHipSadWxHFn        svt_aom_sad[22];                  /* :334 svt_aom_sad{W}x{H}, SVT_HIP_RTCD_BLOCK_SIZES order */
    SvtHipVarWxHFn        svt_aom_variance[22];             /* :524 */
    SvtHipVarWxHFn        svt_aom_highbd_10_variance[22];   /* :568 (uint8_t* = CONVERT_TO_BYTEPTR(uint16_t*)) */
    SvtHipConvolveSrFn    svt_av1_convolve_2d_sr, svt_av1_convolve_x_sr, svt_av1_convolve_y_sr, svt_av1_convolve_2d_copy_sr; /* common_dsp_rtcd.h:197-209 */
    SvtHipHbdConvolveSrFn svt_av1_highbd_convolve_2d_sr, svt_av1_highbd_convolve_x_sr, svt_av1_highbd_convolve_y_sr,
                          svt_av1_highbd_convolve_2d_copy_sr;                                                              /* :219-229 */
    SvtHipFwdTxfmFn       svt_av1_fwd_txfm2d[14];           /* aom_dsp_rtcd.h:129-135, SVT_HIP_RTCD_FWD_SIZES order */
    SvtHipInvTxfmSqFn     svt_av1_inv_txfm2d_add_sq[5];     /* common_dsp_rtcd.h:120-128: 4x4, 8x8, 16x16, 32x32, 64x64 */
    SvtHipInvTxfmRectFn   svt_av1_inv_txfm2d_add_rect;      /* :129-148: rectangular sizes with both sides >= 8 (tx_size, eob arguments) */
    SvtHipInvTxfmRect4Fn  svt_av1_inv_txfm2d_add_rect4;     /* :145-154: 4x8, 8x4, 4x16, 16x4 (tx_size argument, no eob) */
    SvtHipSgrFilterFn     svt_av1_selfguided_restoration;   /* :191 */
    SvtHipSgrApplyFn      svt_apply_selfguided_restoration; /* :187 */
    /* --- SURVEY 8(f) rows (members added at the end: older tables stay layout-compatible) */
    SvtHipObmcSadFn        svt_aom_obmc_sad[22];                 /* aom_dsp_rtcd.h:356-398, SVT_HIP_RTCD_BLOCK_SIZES order */
    SvtHipObmcVarFn        svt_aom_obmc_variance[22];            /* :443-... */
    SvtHipObmcSubpixVarFn  svt_aom_obmc_sub_pixel_variance[22];  /* :399-... */
    SvtHipBlendMaskFn      svt_aom_blend_a64_mask;               /* common_dsp_rtcd.h:73 */
    SvtHipBlendHVMaskFn    svt_aom_blend_a64_hmask, svt_aom_blend_a64_vmask;                           /* :75, :77 */
    SvtHipHbdBlendMaskFn   svt_aom_highbd_blend_a64_mask;        /* uint8_t* arguments carry uint16_t* (no CONVERT_TO_SHORTPTR: EbBlend_a64_mask.c:275) */
    SvtHipHbdBlendHVMaskFn svt_aom_highbd_blend_a64_hmask_8bit, svt_aom_highbd_blend_a64_vmask_8bit;   /* :78-80 */
    SvtHipWarpAffineFn     svt_av1_warp_affine;                  /* non-compound calls; compound ones go to the saved pointer */
    SvtHipHbdWarpAffineFn  svt_av1_highbd_warp_affine;
    SvtHipComputeStatsFn   svt_av1_compute_stats;                /* aom_dsp_rtcd.h:99 */
    SvtHipHbdComputeStatsFn svt_av1_compute_stats_highbd;        /* :103 */
    SvtHipFwdTxfmFn        svt_av1_fwd_txfm2d_N2[14], svt_av1_fwd_txfm2d_N4[14];   /* aom_dsp_rtcd.h:284-350 (squares: svt_av1_fwd_txfm2d_{N}x{N}_N2 / _N4), SVT_HIP_RTCD_FWD_SIZES order */
    /* --- per-call forms of the remaining pointers on the hot path */
    SvtHipExtAllSadFn      svt_ext_all_sad_calculation_8x8_16x16;     /* aom_dsp_rtcd.h:640 */
    SvtHipExtEightSadFn    svt_ext_eight_sad_calculation_32x32_64x64; /* :641 */
    SvtHipQuantizeBFn      svt_aom_quantize_b, svt_aom_highbd_quantize_b;                             /* :252, :254 */
    SvtHipQuantizeFpFn     svt_av1_quantize_fp, svt_av1_quantize_fp_32x32, svt_av1_quantize_fp_64x64; /* :256, :260, :262 */
    SvtHipHbdQuantizeFpFn  svt_av1_highbd_quantize_fp;                                                /* :258 */
    SvtHipLpfFn            svt_aom_lpf_horizontal[4], svt_aom_lpf_vertical[4];               /* common_dsp_rtcd.h:1060-1075, filter lengths 4, 6, 8, 14 */
    SvtHipHbdLpfFn         svt_aom_highbd_lpf_horizontal[4], svt_aom_highbd_lpf_vertical[4]; /* :1044-1059 */
    SvtHipCdefFindDirFn    svt_cdef_find_dir;                  /* :1032 */
    SvtHipCdefFilterBlockFn svt_cdef_filter_block;             /* :1034 */
    SvtHipResidual8Fn      svt_residual_kernel8bit;            /* :169 */
    SvtHipResidual16Fn     svt_residual_kernel16bit;           /* :180 */
    SvtHipSadx4dFn         svt_aom_sadx4d[22];                 /* aom_dsp_rtcd.h:336 svt_aom_sad{W}x{H}x4d, SVT_HIP_RTCD_BLOCK_SIZES order */
    SvtHipUpsampledPredFn  svt_aom_upsampled_pred;             /* :354 */
    SvtHipIntermVarFn      svt_compute_interm_var_four8x8;     /* :650 */
    SvtHipHandleTransformFn svt_handle_transform64[5];         /* :221-230 in header order: 16x64, 32x64, 64x16, 64x32, 64x64 */
    SvtHipInvTxfmAddFn     svt_av1_inv_txfm_add;               /* common_dsp_rtcd.h:156 */
    /* --- the small helpers of the same kernel classes (SURVEY 2's dispatch-table rows) */
    SvtHipSubtractBlockFn    svt_aom_subtract_block;           /* common_dsp_rtcd.h:241 */
    SvtHipHbdSubtractBlockFn svt_aom_highbd_subtract_block;    /* the uint8_t* arguments are plain casts of uint16_t* (EbInterPrediction.c:52) */
    SvtHipSad16bFn         sad_16b_kernel;                     /* aom_dsp_rtcd.h:651 */
    SvtHipVarianceHbdFn    variance_highbd;                    /* :653 */
    SvtHipNxmSadFn         svt_nxm_sad_kernel_sub_sampled;     /* :642 (the same function as svt_nxm_sad_kernel in the C table) */
    SvtHipExtSad16Fn       svt_ext_sad_calculation_8x8_16x16;  /* :630 */
    SvtHipExtSad3264Fn     svt_ext_sad_calculation_32x32_64x64; /* :636 */
    SvtHipCopyRect8To16Fn  svt_copy_rect8_8bit_to_16bit;       /* common_dsp_rtcd.h:1037 */
    SvtHipCdefDist8Fn      svt_compute_cdef_dist_8bit;         /* aom_dsp_rtcd.c:98 */
    SvtHipCdefDist16Fn     svt_compute_cdef_dist_16bit;        /* aom_dsp_rtcd.c:97 */
    SvtHipSearchOneDualFn  svt_search_one_dual;                /* aom_dsp_rtcd.c:363 */
    SvtHipFullDist32Fn     svt_full_distortion_kernel32_bits;  /* common_dsp_rtcd.h; EbPictureOperators.c:156 */
    SvtHipFullDistCbfZero32Fn svt_full_distortion_kernel_cbf_zero32_bits; /* EbPictureOperators.c:212 */
    SvtHipSpatialDistFn    svt_spatial_full_distortion_kernel; /* 8-bit planes */
    SvtHipSpatialDistFn    svt_full_distortion_kernel16_bits;  /* the uint8_t* arguments are uint16_t planes (EbPictureOperators.c:182-207) */
    SvtHipSseFn            svt_aom_sse, svt_aom_highbd_sse;    /* aom_dsp_rtcd.h:93-94 (highbd: plain casts of uint16_t*, EbEncInterPrediction.c:789) */
    SvtHipSatdFn           svt_aom_satd;                       /* common_dsp_rtcd.c:47 */
    SvtHipBlockErrorFn     svt_av1_block_error;                /* common_dsp_rtcd.c:56; exact sums (the C function's `int` products are undefined beyond |coeff| = 46340) */
    SvtHipGetProjSubspaceFn svt_get_proj_subspace;             /* EbRestorationPick.c:448 */
    SvtHipPixelProjErrorFn svt_av1_lowbd_pixel_proj_error, svt_av1_highbd_pixel_proj_error; /* :174, :244 */
    SvtHipMeanSq8x8Fn      svt_compute_mean_square_values_8x8; /* EbPictureAnalysisProcess.c:287 */
    SvtHipSubMean8x8Fn     svt_compute_sub_mean_8x8;           /* :310 */
    SvtHipConvolve8Fn      svt_aom_convolve8_horiz, svt_aom_convolve8_vert; /* common_dsp_rtcd.h:231; convolve.c:286, :298 */
    SvtHipWienerConvolveFn svt_av1_wiener_convolve_add_src;    /* convolve.c:105 */
    SvtHipHbdWienerConvolveFn svt_av1_highbd_wiener_convolve_add_src; /* convolve.c:205 */
    SvtHipHandleTransformFn handle_transform64_N2_N4[5];       /* aom_dsp_rtcd.h:237-245 in header order: 16x64, 32x64, 64x16, 64x32, 64x64 */
    SvtHipVarWxHFn         svt_aom_mse16x16;                   /* :248 (EbPsnr.c:84) */
    SvtHipHbdMseFn         svt_aom_highbd_8_mse16x16;          /* :264 (uint8_t* = CONVERT_TO_BYTEPTR(uint16_t*)) */
    /* --- the picture formats either side of the high-bit-depth path (Common/C_DEFAULT/EbPackUnPack_C.c; svt_hip_picture_format_dev) */
    SvtHipConvert8To16Fn   svt_convert_8bit_to_16bit;
    SvtHipConvert16To8Fn   svt_convert_16bit_to_8bit;
    SvtHipCPackFn          svt_c_pack;                         /* local_cache is not used */
    SvtHipPackMsbFn        svt_compressed_packmsb, svt_pack2d_16_bit_src_mul4;   /* = svt_enc_msb_pack2_d */
    SvtHipUnpackAvgFn      svt_unpack_avg;
    SvtHipUnPack2dFn       svt_un_pack2d_16_bit_src_mul4;      /* = svt_enc_msb_un_pack2_d */
    SvtHipUnPack8Fn        svt_un_pack8_bit_data;
    /* --- one reference of a compound prediction (common_dsp_rtcd.h:211-243): do_average = 0 writes ConvolveParams::dst, 1 averages with it */
    SvtHipConvolveSrFn     svt_av1_jnt_convolve_2d, svt_av1_jnt_convolve_x, svt_av1_jnt_convolve_y, svt_av1_jnt_convolve_2d_copy;
    SvtHipHbdConvolveSrFn  svt_av1_highbd_jnt_convolve_2d, svt_av1_highbd_jnt_convolve_x, svt_av1_highbd_jnt_convolve_y, svt_av1_highbd_jnt_convolve_2d_copy;
    /* --- masked compound: the difference-weighted mask and the blend of the two compound buffers (common_dsp_rtcd.h:113-117; DIFFWTD_MASK_TYPE is a one-byte enum) */
    SvtHipDiffwtdMaskFn    svt_av1_build_compound_diffwtd_mask;
    SvtHipDiffwtdMaskHbdFn svt_av1_build_compound_diffwtd_mask_highbd;   /* the uint8_t* arguments are plain casts of uint16_t* (EbInterPrediction.c:155) */
    SvtHipDiffwtdMaskD16Fn svt_av1_build_compound_diffwtd_mask_d16;
    SvtHipBlendD16Fn       svt_aom_lowbd_blend_a64_d16_mask;
    SvtHipHbdBlendD16Fn    svt_aom_highbd_blend_a64_d16_mask;            /* dst: a plain cast of uint16_t* (EbBlend_a64_mask.c:126) */
} SvtHipRtcd;

/* In: the table holds the C (or SIMD) pointers currently installed (may be NULL).  Out: every member points at the
 * corresponding *_hip wrapper bound to ctx; the incoming pointers are kept as the failure fallbacks. */
int svt_hip_setup_rtcd(SvtHipCtx *ctx, SvtHipRtcd *table);
/* Releases the wrappers' device staging buffers and unbinds their context; only when none of the wrapper pointers is installed any more. */
void svt_hip_rtcd_release(void);
/* Per-wrapper bookkeeping on stderr: "svt_hip_rtcd_calls <wrapper> calls=N" for every wrapper that ran and
 * "svt_hip_rtcd_delegated <table entry> count=N device_failures=M" for every entry that handed a call to the saved pointer (a call outside the
 * kernel's domain, or -- counted separately and logged every time with the error string -- a failed device call).  Returns the delegation total. */
int svt_hip_rtcd_report(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
