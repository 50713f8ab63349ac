/*
 * quant_oracle.c — CPU restatement of SVT-AV1's quantize / dequantize kernels (flat quant matrix,
 * which is the only case on this path: Encoder/Codec/EbModeDecisionConfigurationProcess.c:299-301).
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 *
 * Every coefficient is quantised independently of the others in all four variants (the reference's
 * "pre-scan" passes only skip work); eob = 1 + last scan position with a non-zero level.
 */
#include "svt_oracle.h"
#include <string.h>

static inline int rpot(int v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); } /* ROUND_POWER_OF_TWO */
static inline int64_t clamp64i(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* variant 0: svt_aom_quantize_b_c_ii        (Encoder/Codec/EbFullLoop.c:37-93)   8-bit path
 * variant 1: svt_aom_highbd_quantize_b_c    (:171-225)
 * variant 2: quantize_fp_helper_c           (:314-377, qm == NULL branch)  svt_av1_quantize_fp{,_32x32,_64x64}_c
 * variant 3: highbd_quantize_fp_helper_c    (:467-532, qm == NULL branch)  svt_av1_highbd_quantize_fp_c
 * For variants 2/3 pass round = round_fp_qtx and quant = quant_fp_qtx like the facades do (:603-711). */
void orc_quantize(int variant, const int32_t *coeff, int n, const int16_t *zbin, const int16_t *round,
                  const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                  const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, int log_scale) {
    int eob = -1;
    memset(qcoeff, 0, sizeof(int32_t) * n);
    memset(dqcoeff, 0, sizeof(int32_t) * n);
    for (int i = 0; i < n; i++) {
        const int rc = scan[i], ac = rc != 0;
        const int32_t c = coeff[rc];
        const int sign = c < 0 ? -1 : 0;
        const int32_t absc = (c ^ sign) - sign;
        int32_t level = 0, dq = 0;
        if (variant == 0) {
            const int zb = rpot(zbin[ac], log_scale);
            if (absc >= zb) { /* (abs*wt >= zbin<<5) with wt = 32 */
                int64_t tmp = clamp64i((int64_t)absc + rpot(round[ac], log_scale), INT16_MIN, INT16_MAX);
                tmp *= 32;
                level = (int32_t)(((((tmp * quant[ac]) >> 16) + tmp) * quant_shift[ac]) >> (16 - log_scale + 5));
                /* (tmp32 * dequant) is an int multiply in the reference, then an arithmetic shift */
                dq = (int32_t)((int32_t)((uint32_t)level * (uint32_t)(int32_t)dequant[ac]) >> log_scale);
            }
        } else if (variant == 1) {
            const int zb = rpot(zbin[ac], log_scale);
            if (c >= zb || c <= -zb) {
                const int64_t tmp1 = (int64_t)absc + rpot(round[ac], log_scale);
                const int64_t tmpw = tmp1 * 32;
                const int64_t tmp2 = ((tmpw * quant[ac]) >> 16) + tmpw;
                level = (int32_t)((tmp2 * quant_shift[ac]) >> (16 - log_scale + 5));
                dq = (int32_t)((int32_t)((uint32_t)level * (uint32_t)(int32_t)dequant[ac]) >> log_scale);
            }
        } else if (variant == 2) {
            if (((int64_t)absc << (1 + log_scale)) >= (int32_t)dequant[ac]) {
                const int64_t a = clamp64i((int64_t)absc + rpot(round[ac], log_scale), INT16_MIN, INT16_MAX);
                level = (int)((a * quant[ac]) >> (16 - log_scale));
                if (level) dq = (int32_t)((int32_t)((uint32_t)level * (uint32_t)(int32_t)dequant[ac]) >> log_scale);
            }
        } else {
            if ((int32_t)((uint32_t)absc << (1 + log_scale)) >= dequant[ac]) {
                const int64_t tmp = (int64_t)absc + rpot(round[ac], log_scale);
                level = (int)((tmp * quant[ac]) >> (16 - log_scale));
                dq = (int32_t)((int32_t)((uint32_t)level * (uint32_t)(int32_t)dequant[ac]) >> log_scale);
            }
        }
        qcoeff[rc]  = (level ^ sign) - sign;
        dqcoeff[rc] = (dq ^ sign) - sign;
        if (level) eob = i;
    }
    *eob_out = (uint16_t)(eob + 1);
}

/* cul_level + dc sign as derived after quantisation (Encoder/Codec/EbFullLoop.c:1595-1608;
 * set_dc_sign, COEFF_CONTEXT_BITS = 6, COEFF_CONTEXT_MASK = 63). */
int32_t orc_cul_level(const int32_t *qcoeff, const int16_t *scan, int eob) {
    int32_t cul = 0;
    for (int c = 0; c < eob; c++) {
        const int32_t v = qcoeff[scan[c]];
        cul += v < 0 ? -v : v;
    }
    if (cul > 63) cul = 63;
    if (qcoeff[0] < 0) cul |= 1 << 6;
    else if (qcoeff[0] > 0) cul += 2 << 6;
    return cul;
}

/* Coefficient-domain distortion of one block (flat, n coefficients): out[0] = sum (c - r)^2 (r == NULL: sum c^2), out[1] = sum c^2,
 * out[2] = sum |c|.  svt_full_distortion_kernel32_bits_c / _cbf_zero32_bits_c (Common/Codec/EbPictureOperators.c:156,212),
 * svt_av1_block_error_c and svt_aom_satd_c (Common/Codec/common_dsp_rtcd.c:56,47). */
void orc_coeff_distortion(const int32_t *coeff, const int32_t *recon, int n, uint64_t out[3]) {
    uint64_t res = 0, pred = 0, satd = 0;
    for (int i = 0; i < n; i++) {
        const int64_t c = coeff[i], d = recon ? c - recon[i] : c;
        res += (uint64_t)(d * d); pred += (uint64_t)(c * c); satd += (uint64_t)(c < 0 ? -c : c);
    }
    out[0] = res; out[1] = pred; out[2] = satd;
}
