/*
 * ref_asm_stubs.c — stand-ins for the reference's NASM sources in the SIMD flavour of oracle/_ref (Makefile.ref `simd`).
 * TEST INFRASTRUCTURE ONLY.  This image has no nasm / yasm, so the few hand-written .asm entry points
 * (Common/ASM_SSE2/{intrapred_sse2,highbd_intrapred_sse2_,subtract_sse2,aom_subpixel_8t_sse2,EbPictureOperators_SSE2,x64RegisterUtil}.asm,
 * Common/ASM_SSSE3/aom_subpixel_bilinear_ssse3.asm, Encoder/ASM_SSE2/highbd_variance_impl_sse2.asm) are provided here in plain C with
 * the same contracts, so that the library links and its RTCD tables can be set up exactly as in a normal x86 build.  None of them is
 * on the path bench.py times (the timed leaves are the AVX2 / AVX-512 intrinsics kernels, compiled from the reference's own sources);
 * the intra-prediction and 8-tap helper entries, which nothing here calls, abort if they are ever reached.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void RunEmms(void) {}                                           /* x64RegisterUtil.asm: `emms` — no x87/MMX state is used here */
uint32_t Log2f_ASM(uint32_t x) { return x ? 31u - (uint32_t)__builtin_clz(x) : 0u; }   /* bsr */

void picture_copy_kernel_sse2(uint8_t *src, uint32_t src_stride, uint8_t *dst, uint32_t dst_stride, uint32_t area_width, uint32_t area_height) {
    for (uint32_t y = 0; y < area_height; y++) memcpy(dst + (size_t)y * dst_stride, src + (size_t)y * src_stride, area_width);
}

void svt_aom_subtract_block_sse2(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride,
                                 const uint8_t *pred_ptr, ptrdiff_t pred_stride) {
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) diff_ptr[r * diff_stride + c] = (int16_t)(src_ptr[r * src_stride + c] - pred_ptr[r * pred_stride + c]);
}

static uint32_t hbd_var(const uint16_t *src, int32_t ss, const uint16_t *ref, int32_t rs, uint32_t *sse, int32_t *sum, int n) {
    int32_t s = 0; uint32_t q = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) { const int d = src[y * ss + x] - ref[y * rs + x]; s += d; q += (uint32_t)(d * d); }
    *sum = s; *sse = q;
    return 0;
}
uint32_t svt_aom_highbd_calc4x4var_sse2(const uint16_t *s, int32_t ss, const uint16_t *r, int32_t rs, uint32_t *sse, int32_t *sum) { return hbd_var(s, ss, r, rs, sse, sum, 4); }
uint32_t svt_aom_highbd_calc8x8var_sse2(const uint16_t *s, int32_t ss, const uint16_t *r, int32_t rs, uint32_t *sse, int32_t *sum) { return hbd_var(s, ss, r, rs, sse, sum, 8); }
uint32_t svt_aom_highbd_calc16x16var_sse2(const uint16_t *s, int32_t ss, const uint16_t *r, int32_t rs, uint32_t *sse, int32_t *sum) { return hbd_var(s, ss, r, rs, sse, sum, 16); }

#define UNREACHED(name) void name(void) { fprintf(stderr, "oracle/_ref SIMD flavour: %s (NASM source) is not built\n", #name); abort(); }
UNREACHED(svt_aom_v_predictor_4x4_sse2) UNREACHED(svt_aom_v_predictor_8x8_sse2) UNREACHED(svt_aom_v_predictor_16x16_sse2)
UNREACHED(svt_aom_h_predictor_4x4_sse2) UNREACHED(svt_aom_h_predictor_8x8_sse2) UNREACHED(svt_aom_h_predictor_16x16_sse2)
UNREACHED(svt_aom_dc_predictor_4x4_sse2) UNREACHED(svt_aom_dc_predictor_8x8_sse2) UNREACHED(svt_aom_dc_predictor_16x16_sse2)
UNREACHED(svt_aom_dc_top_predictor_4x4_sse2) UNREACHED(svt_aom_dc_top_predictor_8x8_sse2) UNREACHED(svt_aom_dc_top_predictor_16x16_sse2)
UNREACHED(svt_aom_dc_left_predictor_4x4_sse2) UNREACHED(svt_aom_dc_left_predictor_8x8_sse2) UNREACHED(svt_aom_dc_left_predictor_16x16_sse2)
UNREACHED(svt_aom_dc_128_predictor_4x4_sse2) UNREACHED(svt_aom_dc_128_predictor_8x8_sse2) UNREACHED(svt_aom_dc_128_predictor_16x16_sse2)
UNREACHED(svt_aom_highbd_v_predictor_4x4_sse2) UNREACHED(svt_aom_highbd_v_predictor_8x8_sse2)
UNREACHED(svt_aom_highbd_dc_predictor_4x4_sse2) UNREACHED(svt_aom_highbd_dc_predictor_8x8_sse2)
UNREACHED(svt_aom_filter_block1d4_v8_sse2)
UNREACHED(svt_aom_filter_block1d4_h2_ssse3) UNREACHED(svt_aom_filter_block1d4_v2_ssse3)
UNREACHED(svt_aom_filter_block1d8_h2_ssse3) UNREACHED(svt_aom_filter_block1d8_v2_ssse3)
UNREACHED(svt_aom_filter_block1d16_h2_ssse3) UNREACHED(svt_aom_filter_block1d16_v2_ssse3)
