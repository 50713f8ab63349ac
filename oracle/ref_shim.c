/*
 * ref_shim.c — tiny accessor file compiled INTO oracle/_ref/libsvtav1_ref.so (never into the
 * product, never on the GPU box): exposes constant tables that the reference keeps `static` in its
 * headers (scan orders) or inside encoder objects (quantizer tables), so that tests and
 * tests/golden/make_golden.py can feed the oracle / HIP path with the reference's real tables.
 * It includes the reference headers from /root/reference at build time only.
 */
#include <stdint.h>
#include <string.h>
#include "EbDefinitions.h"
#include "EbCoefficients.h"
#include "EbPictureControlSet.h"
#include "EbFullLoop.h"
#include "EbInvTransforms.h"

void svt_av1_build_quantizer(AomBitDepth bit_depth, int32_t y_dc_delta_q, int32_t u_dc_delta_q, int32_t u_ac_delta_q,
                             int32_t v_dc_delta_q, int32_t v_ac_delta_q, Quants *const quants, Dequants *const deq);

int ref_shim_scan(int tx_size, int tx_type, const int16_t **scan, const int16_t **iscan) {
    *scan  = av1_scan_orders[tx_size][tx_type].scan;
    *iscan = av1_scan_orders[tx_size][tx_type].iscan;
    return av1_get_max_eob((TxSize)tx_size);
}
int ref_shim_tx_scale(int tx_size) { return av1_get_tx_scale_tab[tx_size]; }
void ref_shim_flip(int tx_type, int *ud, int *lr) { get_flip_cfg((TxType)tx_type, ud, lr); }

/* out[k][0..1] = {dc, ac} of: 0 zbin, 1 round, 2 quant, 3 quant_shift, 4 dequant, 5 round_fp, 6 quant_fp
 * for plane 0 (Y), 1 (U), 2 (V); tables as built at encoder init with zero delta-q
 * (Encoder/Codec/EbModeDecisionConfigurationProcess.c:205-287). */
void ref_shim_qparams(int bd, int qindex, int plane, int16_t out[7][2]) {
    static Quants   q[2];
    static Dequants d[2];
    static int      ready[2] = {0, 0};
    const int       b        = bd == 8 ? 0 : 1;
    if (!ready[b]) {
        svt_av1_build_quantizer(bd == 8 ? AOM_BITS_8 : AOM_BITS_10, 0, 0, 0, 0, 0, &q[b], &d[b]);
        ready[b] = 1;
    }
    const int16_t *src[7];
    if (plane == 0) {
        src[0] = q[b].y_zbin[qindex]; src[1] = q[b].y_round[qindex]; src[2] = q[b].y_quant[qindex];
        src[3] = q[b].y_quant_shift[qindex]; src[4] = d[b].y_dequant_qtx[qindex];
        src[5] = q[b].y_round_fp[qindex]; src[6] = q[b].y_quant_fp[qindex];
    } else if (plane == 1) {
        src[0] = q[b].u_zbin[qindex]; src[1] = q[b].u_round[qindex]; src[2] = q[b].u_quant[qindex];
        src[3] = q[b].u_quant_shift[qindex]; src[4] = d[b].u_dequant_qtx[qindex];
        src[5] = q[b].u_round_fp[qindex]; src[6] = q[b].u_quant_fp[qindex];
    } else {
        src[0] = q[b].v_zbin[qindex]; src[1] = q[b].v_round[qindex]; src[2] = q[b].v_quant[qindex];
        src[3] = q[b].v_quant_shift[qindex]; src[4] = d[b].v_dequant_qtx[qindex];
        src[5] = q[b].v_round_fp[qindex]; src[6] = q[b].v_quant_fp[qindex];
    }
    for (int k = 0; k < 7; k++) { out[k][0] = src[k][0]; out[k][1] = src[k][1]; }
}
