/* svt_hip_md_bridge.c — mode decision, hook "md_tx": the forward transforms of ONE transform block for EVERY transform type tx_type_search is going to
 * try (EbProductCodingLoop.c:4258-4560) in one launch.  The search quantises (with RDOQ: the entropy-context dependent part, host), inverse-transforms and
 * costs the types one after the other; what they share is the residual, and av1_estimate_transform of that residual for up to 16 types is the batchable
 * piece: svt_hip_hook_md_tx_begin uploads the residual once and runs svt_hip_fwd_txfm_quant_batch_dev over one descriptor per type, the loop's
 * av1_estimate_transform calls then read their coefficients from the thread's cache (svt_hip_hook_md_tx_fetch).  Reference-side glue (C, compiled into
 * libSvtAv1Enc); the transform itself is the library's. */
#include <stdlib.h>
#include <string.h>
#include "svt_hip_hooks.h"
#include "EbDefinitions.h"
#include "EbTransforms.h"

#define MD_TX_MAX 32   /* transform blocks of more than 32 samples per side only try DCT_DCT (:4349-4356): no batch */
static __thread struct {
    int      valid, tx_size, n_coeff, n;
    uint32_t mask;
    int8_t   slot[TX_TYPES];
    int32_t  coeff[TX_TYPES][MD_TX_MAX * MD_TX_MAX];
} tls_tx;
/* device staging shared by the MD threads, touched only with the hooks lock held */
static void *d_src, *d_pred, *d_desc, *d_coeff;

/* 1: the cache holds the coefficients of every type in `mask`.  resid: the candidate's luma residual at the transform block (int16, `stride` samples per
 * row); tx_size: TxSize; coeff_shape: EB_TRANS_COEFF_SHAPE of the pass (pf_ctrls.pf_shape). */
int svt_hip_hook_md_tx_begin(const int16_t *resid, uint32_t stride, int tx_size, int coeff_shape, uint32_t mask) {
    tls_tx.valid = 0;
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_MD_TX) || tx_size < 0 || tx_size >= TX_SIZES_ALL) return 0;
    const int w = tx_size_wide[tx_size], h = tx_size_high[tx_size];
    int n = 0;
    for (int t = 0; t < TX_TYPES; t++) n += (mask >> t) & 1;
    if (n < 2 || w > MD_TX_MAX || h > MD_TX_MAX) return 0;   /* a single type: nothing to batch, the reference's call stays */
    /* the batched entry point forms the residual itself from two pixel planes: r = max(r, 0) - max(-r, 0) */
    uint16_t hs[MD_TX_MAX * MD_TX_MAX], hp[MD_TX_MAX * MD_TX_MAX];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int r = resid[(size_t)y * stride + x];
            hs[y * w + x] = (uint16_t)(r > 0 ? r : 0); hp[y * w + x] = (uint16_t)(r < 0 ? -r : 0);
        }
    uint32_t desc[TX_TYPES];
    n = 0;
    for (int t = 0; t < TX_TYPES; t++) {
        tls_tx.slot[t] = -1;
        if ((mask >> t) & 1) { tls_tx.slot[t] = (int8_t)n; desc[n++] = SVT_HIP_TX_DESC(0, 0, t); }
    }
    SvtHipQuantParams qp;
    memset(&qp, 0, sizeof(qp));
    qp.coeff_shape = coeff_shape;
    SvtHipCtx *hip = svt_hip_hooks_lock();
    if (!hip) return 0;
    int rc = SVT_HIP_OK;
    if (!d_src) {
        rc = svt_hip_malloc(hip, &d_src, sizeof(hs));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_pred, sizeof(hp));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_desc, sizeof(desc));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_coeff, sizeof(tls_tx.coeff));
        if (rc != SVT_HIP_OK) { svt_hip_free(hip, d_src); svt_hip_free(hip, d_pred); svt_hip_free(hip, d_desc); svt_hip_free(hip, d_coeff); d_src = d_pred = d_desc = d_coeff = NULL; }
    }
    const size_t pb = (size_t)w * h * sizeof(uint16_t), cb = (size_t)w * h * sizeof(int32_t);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_src, hs, pb);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_pred, hp, pb);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_desc, desc, sizeof(uint32_t) * (size_t)n);
    if (rc == SVT_HIP_OK)
        rc = svt_hip_fwd_txfm_quant_batch_dev(hip, tx_size, 2, d_src, w, d_pred, w, (const uint32_t *)d_desc, n, &qp, NULL, (int32_t *)d_coeff, NULL, NULL, NULL, NULL, NULL);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, tls_tx.coeff, d_coeff, cb * (size_t)n);
    svt_hip_hooks_unlock();
    svt_hip_hooks_count(SVT_HIP_HOOK_MD_TX, rc == SVT_HIP_OK);
    if (rc != SVT_HIP_OK) return 0;
    tls_tx.valid = 1; tls_tx.tx_size = tx_size; tls_tx.n_coeff = w * h; tls_tx.n = n; tls_tx.mask = mask;
    return 1;
}

/* av1_estimate_transform of the block begun above for one type: 1 = coeff holds the device result (three_quad_energy is 0 for these sizes) */
int svt_hip_hook_md_tx_fetch(int tx_size, int tx_type, int32_t *coeff) {
    if (!tls_tx.valid || tls_tx.tx_size != tx_size || tx_type < 0 || tx_type >= TX_TYPES || tls_tx.slot[tx_type] < 0) return 0;
    memcpy(coeff, (const uint8_t *)tls_tx.coeff + (size_t)tls_tx.slot[tx_type] * tls_tx.n_coeff * sizeof(int32_t), (size_t)tls_tx.n_coeff * sizeof(int32_t));
    return 1;
}
void svt_hip_hook_md_tx_end(void) { tls_tx.valid = 0; }
