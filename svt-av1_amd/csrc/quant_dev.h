// quant_dev.h — the four scalar quantizers of the reference as one device function (shared by the fused forward-transform kernels of
// txfm2d.hip and the stand-alone quantizer of percall.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/svt_hip.h"

__device__ __forceinline__ int rpot(int v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); }

// one coefficient through the selected quantizer (flat quant matrix); returns the unsigned level
__device__ __forceinline__ int32_t quant_one(const SvtHipQuantParams& qp, int32_t c, int ac, int32_t& dq_abs) {
    const int ls = qp.log_scale;
    const int32_t absc = c < 0 ? -c : c;
    int32_t level = 0;
    dq_abs = 0;
    if (qp.variant == 0) {        // svt_aom_quantize_b_c_ii, EbFullLoop.c:37-93
        if (absc >= rpot(qp.zbin[ac], ls)) {
            int64_t tmp = (int64_t)absc + rpot(qp.round[ac], ls);
            tmp = tmp > 32767 ? 32767 : tmp;
            tmp *= 32;
            level = (int32_t)(((((tmp * qp.quant[ac]) >> 16) + tmp) * qp.quant_shift[ac]) >> (16 - ls + 5));
            dq_abs = (int32_t)((uint32_t)level * (uint32_t)qp.dequant[ac]) >> ls;
        }
    } else if (qp.variant == 1) { // svt_aom_highbd_quantize_b_c, :171-225
        if (absc >= rpot(qp.zbin[ac], ls)) {
            const int64_t tmpw = ((int64_t)absc + rpot(qp.round[ac], ls)) * 32;
            const int64_t tmp2 = ((tmpw * qp.quant[ac]) >> 16) + tmpw;
            level = (int32_t)((tmp2 * qp.quant_shift[ac]) >> (16 - ls + 5));
            dq_abs = (int32_t)((uint32_t)level * (uint32_t)qp.dequant[ac]) >> ls;
        }
    } else if (qp.variant == 2) { // quantize_fp_helper_c (round = round_fp, quant = quant_fp), :314-377
        if (((int64_t)absc << (1 + ls)) >= qp.dequant[ac]) {
            int64_t a = (int64_t)absc + rpot(qp.round[ac], ls);
            a = a > 32767 ? 32767 : a;
            level = (int32_t)((a * qp.quant[ac]) >> (16 - ls));
            dq_abs = (int32_t)((uint32_t)level * (uint32_t)qp.dequant[ac]) >> ls;
        }
    } else {                      // highbd_quantize_fp_helper_c, :467-532
        if ((int32_t)((uint32_t)absc << (1 + ls)) >= qp.dequant[ac]) {
            const int64_t tmp = (int64_t)absc + rpot(qp.round[ac], ls);
            level = (int32_t)((tmp * qp.quant[ac]) >> (16 - ls));
            dq_abs = (int32_t)((uint32_t)level * (uint32_t)qp.dequant[ac]) >> ls;
        }
    }
    return level;
}
