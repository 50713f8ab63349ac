// format.hip — picture-format conversions on either side of the high-bit-depth path; gfx950.
//
// The reference stores 10-bit pictures as an 8-bit MSB plane plus a 2-bit plane (unpacked: one byte per sample with the two bits on top;
// or compressed: four samples per byte) and converts to / from 16-bit samples around the kernels that need them.  Replaces
// (Common/C_DEFAULT/EbPackUnPack_C.c, dispatched through common_dsp_rtcd.h / aom_dsp_rtcd.h):
//   :18 svt_enc_msb_pack2_d, :41 svt_compressed_packmsb, :105 svt_enc_msb_un_pack2_d, :176 svt_convert_8bit_to_16bit,
//   :183 svt_convert_16bit_to_8bit, :77 svt_c_pack, :137 svt_unpack_avg.
// Pure HBM streams: one thread handles 4 consecutive samples of a row (dword of bytes / two dwords of 16-bit samples).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svt_hip_internal.h"

namespace {

template <int MODE>
__global__ void __launch_bounds__(256)
picture_format_kernel(const void* __restrict__ in0, int s0, const void* __restrict__ in1, int s1, void* __restrict__ out0, int t0, void* __restrict__ out1, int t1, int w,
                      int h) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= w || y >= h) return;
    const int n = min(4, w - x0);
    if (MODE == 1) {        // compressed 2-bit plane: one byte per 4 samples
        const uint8_t four = ((const uint8_t*)in1)[(size_t)y * s1 + (x0 >> 2)];
        for (int k = 0; k < n; k++)
            ((uint16_t*)out0)[(size_t)y * t0 + x0 + k] = (uint16_t)((((const uint8_t*)in0)[(size_t)y * s0 + x0 + k] << 2) | ((four >> (6 - 2 * k)) & 3));
        return;
    }
    if (MODE == 5) {        // four unpacked 2-bit samples -> one byte
        const uint8_t* p = (const uint8_t*)in0 + (size_t)y * s0 + x0;
        ((uint8_t*)out0)[(size_t)y * t0 + (x0 >> 2)] = (uint8_t)((p[0] & 0xC0) | ((p[1] >> 2) & 0x30) | ((p[2] >> 4) & 0x0C) | ((p[3] >> 6) & 0x03));
        return;
    }
    for (int k = 0; k < n; k++) {
        const size_t x = (size_t)x0 + k;
        if (MODE == 0) ((uint16_t*)out0)[(size_t)y * t0 + x] = (uint16_t)((((const uint8_t*)in0)[(size_t)y * s0 + x] << 2) | ((((const uint8_t*)in1)[(size_t)y * s1 + x] >> 6) & 3));
        else if (MODE == 2) {
            const uint16_t v = ((const uint16_t*)in0)[(size_t)y * s0 + x];
            ((uint8_t*)out0)[(size_t)y * t0 + x] = (uint8_t)(v >> 2);
            if (out1) ((uint8_t*)out1)[(size_t)y * t1 + x] = (uint8_t)(v << 6);
        } else if (MODE == 3) ((uint16_t*)out0)[(size_t)y * t0 + x] = ((const uint8_t*)in0)[(size_t)y * s0 + x];
        else if (MODE == 4) ((uint8_t*)out0)[(size_t)y * t0 + x] = (uint8_t)((const uint16_t*)in0)[(size_t)y * s0 + x];
        else if (MODE == 6) {
            const int a = (uint8_t)(((const uint16_t*)in0)[(size_t)y * s0 + x] >> 2), b = (uint8_t)(((const uint16_t*)in1)[(size_t)y * s1 + x] >> 2);
            ((uint8_t*)out0)[(size_t)y * t0 + x] = (uint8_t)((a + b + 1) >> 1);
        }
    }
}


// generate_padding / generate_padding16_bit (Common/Codec/EbMcp.c:112, :166): every border sample = the nearest picture sample.  One thread per border
// sample of a row; rows -pad_h .. h + pad_h - 1 (interior samples are skipped: the border only ever reads the picture, so one pass suffices).
template <typename PIX>
__global__ void __launch_bounds__(256)
generate_padding_kernel(PIX* __restrict__ plane, int stride, int w, int h, int pad_w, int pad_h) {
    const int y = (int)blockIdx.y - pad_h;
    const int sy = min(max(y, 0), h - 1);
    const bool inner_row = y >= 0 && y < h;
    const int n = inner_row ? 2 * pad_w : w + 2 * pad_w;   // border samples of this row
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int x = inner_row ? (i < pad_w ? i - pad_w : w + (i - pad_w)) : i - pad_w;
        const int sx = min(max(x, 0), w - 1);
        plane[(ptrdiff_t)y * stride + x] = plane[(ptrdiff_t)sy * stride + sx];
    }
}

}  // namespace

extern "C" int svt_hip_launch_picture_format(hipStream_t st, int mode, const void* in0, int s0, const void* in1, int s1, void* out0, int t0, void* out1, int t1, int w,
                                             int h) {
    if (w <= 0 || h <= 0) return 0;
    const dim3 grid((w + 1023) / 1024, h), block(256);
#define L(M) hipLaunchKernelGGL((picture_format_kernel<M>), grid, block, 0, st, in0, s0, in1, s1, out0, t0, out1, t1, w, h)
    switch (mode) {
    case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break; case 6: L(6); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef L
    return (int)hipGetLastError();
}

extern "C" int svt_hip_launch_generate_padding(hipStream_t st, void* plane, int pix_bytes, int stride, int w, int h, int pad_w, int pad_h) {
    if (w <= 0 || h <= 0 || (pad_w <= 0 && pad_h <= 0)) return 0;
    const dim3 grid(min((w + 2 * pad_w + 255) / 256, 16), h + 2 * pad_h), block(256);
    if (pix_bytes == 1) hipLaunchKernelGGL(generate_padding_kernel<uint8_t>, grid, block, 0, st, (uint8_t*)plane, stride, w, h, pad_w, pad_h);
    else hipLaunchKernelGGL(generate_padding_kernel<uint16_t>, grid, block, 0, st, (uint16_t*)plane, stride, w, h, pad_w, pad_h);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(format)
