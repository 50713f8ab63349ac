/*
 * format_oracle.c — CPU restatement of the picture-format conversions on either side of the high-bit-depth path (the reference keeps 10-bit
 * pictures as an 8-bit MSB plane + a 2-bit plane, the kernels work on 16-bit samples).  TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 * Pinned by tests/test_oracle_vs_ref.py against Common/C_DEFAULT/EbPackUnPack_C.c (file:line below are in that file).
 *   mode 0  svt_enc_msb_pack2_d        :18   out16 = in8 << 2 | (inn >> 6) & 3
 *   mode 1  svt_compressed_packmsb_c   :41   the same with the 2-bit plane packed 4 samples per byte (first sample in the top bits)
 *   mode 2  svt_enc_msb_un_pack2_d     :105  out8 = in16 >> 2, outn = (uint8_t)(in16 << 6) (outn optional)
 *   mode 3  svt_convert_8bit_to_16bit_c :176
 *   mode 4  svt_convert_16bit_to_8bit_c :183
 *   mode 5  svt_c_pack_c               :77   unpacked 2-bit plane -> 4 samples per byte
 *   mode 6  svt_unpack_avg_c           :137  out8 = ((a16 >> 2 & 255) + (b16 >> 2 & 255) + 1) >> 1
 * Strides in samples of the respective plane (mode 1 / 5: bytes of the packed plane); width a multiple of 4 for modes 1 and 5.
 */
#include <stdint.h>
#include <stddef.h>
#include "svt_oracle.h"

void orc_picture_format(int mode, const void *in0, int in0_stride, const void *in1, int in1_stride, void *out0, int out0_stride, void *out1, int out1_stride,
                        int w, int h) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            switch (mode) {
            case 0: ((uint16_t *)out0)[(size_t)y * out0_stride + x] = (uint16_t)((((const uint8_t *)in0)[(size_t)y * in0_stride + x] << 2) | ((((const uint8_t *)in1)[(size_t)y * in1_stride + x] >> 6) & 3)); break;
            case 1: { const uint8_t four = ((const uint8_t *)in1)[(size_t)y * in1_stride + (x >> 2)];
                      ((uint16_t *)out0)[(size_t)y * out0_stride + x] = (uint16_t)((((const uint8_t *)in0)[(size_t)y * in0_stride + x] << 2) | ((four >> (6 - 2 * (x & 3))) & 3)); break; }
            case 2: { const uint16_t v = ((const uint16_t *)in0)[(size_t)y * in0_stride + x];
                      ((uint8_t *)out0)[(size_t)y * out0_stride + x] = (uint8_t)(v >> 2);
                      if (out1) ((uint8_t *)out1)[(size_t)y * out1_stride + x] = (uint8_t)(v << 6); break; }
            case 3: ((uint16_t *)out0)[(size_t)y * out0_stride + x] = ((const uint8_t *)in0)[(size_t)y * in0_stride + x]; break;
            case 4: ((uint8_t *)out0)[(size_t)y * out0_stride + x] = (uint8_t)((const uint16_t *)in0)[(size_t)y * in0_stride + x]; break;
            case 5: if ((x & 3) == 0) { const uint8_t *p = (const uint8_t *)in0 + (size_t)y * in0_stride + x;
                      ((uint8_t *)out0)[(size_t)y * out0_stride + (x >> 2)] = (uint8_t)((p[0] & 0xC0) | ((p[1] >> 2) & 0x30) | ((p[2] >> 4) & 0x0C) | ((p[3] >> 6) & 0x03)); } break;
            case 6: { const int a = (uint8_t)(((const uint16_t *)in0)[(size_t)y * in0_stride + x] >> 2), b = (uint8_t)(((const uint16_t *)in1)[(size_t)y * in1_stride + x] >> 2);
                      ((uint8_t *)out0)[(size_t)y * out0_stride + x] = (uint8_t)((a + b + 1) >> 1); break; }
            default: break;
            }
        }
}

/* generate_padding / generate_padding16_bit (Common/Codec/EbMcp.c:112-160, :166-214): replicate the picture edges into a border of pad_w columns and pad_h
 * rows (rows first get their left / right border, then whole padded rows are copied up and down: every border sample = the nearest picture sample).
 * `plane` points at picture sample (0, 0); the reference passes the buffer start and the padded stride. */
void orc_generate_padding(void *plane, int pix_bytes, int stride, int w, int h, int pad_w, int pad_h) {
    for (int y = -pad_h; y < h + pad_h; y++)
        for (int x = -pad_w; x < w + pad_w; x++) {
            if (x >= 0 && x < w && y >= 0 && y < h) continue;
            const int sx = x < 0 ? 0 : (x >= w ? w - 1 : x), sy = y < 0 ? 0 : (y >= h ? h - 1 : y);
            if (pix_bytes == 1) ((uint8_t *)plane)[(ptrdiff_t)y * stride + x] = ((uint8_t *)plane)[(ptrdiff_t)sy * stride + sx];
            else ((uint16_t *)plane)[(ptrdiff_t)y * stride + x] = ((uint16_t *)plane)[(ptrdiff_t)sy * stride + sx];
        }
}
