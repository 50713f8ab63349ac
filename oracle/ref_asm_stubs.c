/*
 * ref_asm_stubs.c — stand-ins for the reference's NASM sources in the SIMD flavour of oracle/_ref (Makefile.ref `simd`).
 * TEST INFRASTRUCTURE ONLY.  This image has no nasm / yasm, so the few hand-written .asm entry points
 * (Common/ASM_SSE2/{intrapred_sse2,highbd_intrapred_sse2_,subtract_sse2,aom_subpixel_8t_sse2,EbPictureOperators_SSE2,x64RegisterUtil}.asm,
 * Common/ASM_SSSE3/aom_subpixel_bilinear_ssse3.asm, Encoder/ASM_SSE2/highbd_variance_impl_sse2.asm) are provided here in plain C with
 * the same contracts, so that the library links and its RTCD tables can be set up exactly as in a normal x86 build.  None of them is
 * on the path bench.py times (the timed leaves are the AVX2 / AVX-512 intrinsics kernels, compiled from the reference's own sources).
 * The intra predictors forward to the reference's own *_c functions; the 1-D filter helpers behind svt_aom_convolve8_{horiz,vert}_avx2
 * (FUN_CONV_1D, Common/ASM_AVX2/aom_subpixel_8t_intrin_avx2.c:81-170: the 4-wide vertical 8-tap leaf and the 2-tap leaves) are written out:
 * sum of taps, + 64, >> 7, clip — the result the reference's unit tests require of the assembly.  With them the SIMD flavour also runs whole
 * encodes (oracle/Makefile.enc `simd`: SvtAv1EncApp_simd), which tests/test_encode_e2e.py compares with the C build bit for bit.
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

/* intrapred_sse2.asm / highbd_intrapred_sse2_.asm: same contracts as the C predictors they accelerate */
#define FWD8(name, n)                                                                                                   \
    void name##_##n##x##n##_c(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);                \
    void name##_##n##x##n##_sse2(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left) { name##_##n##x##n##_c(dst, stride, above, left); }
#define FWD8_ALL(name) FWD8(name, 4) FWD8(name, 8) FWD8(name, 16)
FWD8_ALL(svt_aom_v_predictor) FWD8_ALL(svt_aom_h_predictor) FWD8_ALL(svt_aom_dc_predictor) FWD8_ALL(svt_aom_dc_top_predictor)
FWD8_ALL(svt_aom_dc_left_predictor) FWD8_ALL(svt_aom_dc_128_predictor)
#define FWD16(name, n)                                                                                                              \
    void name##_##n##x##n##_c(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int32_t bd);             \
    void name##_##n##x##n##_sse2(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int32_t bd) { name##_##n##x##n##_c(dst, stride, above, left, bd); }
FWD16(svt_aom_highbd_v_predictor, 4) FWD16(svt_aom_highbd_v_predictor, 8) FWD16(svt_aom_highbd_dc_predictor, 4) FWD16(svt_aom_highbd_dc_predictor, 8)

/* aom_subpixel_8t_sse2.asm / aom_subpixel_bilinear_ssse3.asm: one column strip of a 1-D convolution, rounding (+ 64) >> 7, clip to 8 bits.
 * v8: src_ptr is the row of the first tap (the caller passes src - 3 rows); h2 / v2: the two middle taps filter[3], filter[4] on src[x], src[x + 1 | + pitch]. */
static inline uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void svt_aom_filter_block1d4_v8_sse2(const uint8_t *src_ptr, ptrdiff_t src_pitch, uint8_t *output_ptr, ptrdiff_t out_pitch, uint32_t output_height, const int16_t *filter) {
    for (uint32_t y = 0; y < output_height; y++)
        for (int x = 0; x < 4; x++) {
            int sum = 0;
            for (int k = 0; k < 8; k++) sum += src_ptr[(ptrdiff_t)(y + k) * src_pitch + x] * filter[k];
            output_ptr[(ptrdiff_t)y * out_pitch + x] = clip8((sum + 64) >> 7);
        }
}
#define TAP2(name, w, step)                                                                                                                                    \
    void name(const uint8_t *src_ptr, ptrdiff_t src_pitch, uint8_t *output_ptr, ptrdiff_t out_pitch, uint32_t output_height, const int16_t *filter) {          \
        for (uint32_t y = 0; y < output_height; y++)                                                                                                           \
            for (int x = 0; x < w; x++)                                                                                                                        \
                output_ptr[(ptrdiff_t)y * out_pitch + x] =                                                                                                     \
                    clip8((src_ptr[(ptrdiff_t)y * src_pitch + x] * filter[3] + src_ptr[(ptrdiff_t)y * src_pitch + x + (step)] * filter[4] + 64) >> 7);          \
    }
TAP2(svt_aom_filter_block1d4_h2_ssse3, 4, 1) TAP2(svt_aom_filter_block1d8_h2_ssse3, 8, 1) TAP2(svt_aom_filter_block1d16_h2_ssse3, 16, 1)
TAP2(svt_aom_filter_block1d4_v2_ssse3, 4, src_pitch) TAP2(svt_aom_filter_block1d8_v2_ssse3, 8, src_pitch) TAP2(svt_aom_filter_block1d16_v2_ssse3, 16, src_pitch)
