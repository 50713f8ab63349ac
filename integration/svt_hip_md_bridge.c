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

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * Encode pass, hook "encdec_tx": the forward transforms of EVERY transform block of one inter-coded block — luma and both chroma planes, all
 * transform blocks of the block's depth — in one launch, before the block's transform loops run (av1_encode_decode, EbCodingLoop.c:2997-3560, whose
 * av1_encode_loop / av1_encode_loop_16bit calls :3069, :3393 do residual -> av1_estimate_transform -> av1_quantize_inv_quantize per transform block, :379-596 /
 * :760-975).  The prediction of an inter block is complete before its first transform block and the reconstruction of one transform block never touches the
 * samples of another, so every residual is known up front; quantisation (RDOQ, entropy contexts) stays the reference's, block by block.  A transform block
 * whose luma ends up without coefficients switches its type to DCT_DCT (luma in the second loop, chroma right away, :441-449), so both the chosen type and
 * DCT_DCT are computed where they differ.  Transform blocks with a 64-sample side keep the reference's call (they come with the discarded-energy sum and a
 * re-packed layout); the cache is keyed by (plane, transform block, size, type) and dropped at the end of the block. */
#include "EbEncDecProcess.h"
#include "EbCodingUnit.h"

#define ED_MAX_JOBS (3 * MAX_TXB_COUNT * 2)
#define ED_MAX_COEFF (128 * 128 * 3)   /* a 128x128 block, luma + chroma, both types: well below this with sides <= 32 */
static __thread struct {
    int      valid, n;
    struct { int8_t plane, txb, tx_size, tx_type; int32_t off, count; } e[ED_MAX_JOBS];
    int32_t  coeff[ED_MAX_COEFF];
} tls_ed;
static void *d_ed_src, *d_ed_pred, *d_ed_desc, *d_ed_coeff;   /* shared staging, hooks lock held */
static long  g_ed_blocks, g_ed_tx;                            /* inter blocks batched / av1_estimate_transform calls they replaced (svt_hip_hooks_report) */

void svt_hip_hook_encdec_tx_stats(long *blocks, long *calls) { *blocks = g_ed_blocks; *calls = g_ed_tx; }

int svt_hip_hook_encdec_tx_begin(EncDecContext *ctx, const EbPictureBufferDesc *pred, int is_16bit) {
    tls_ed.valid = 0;
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_ENCDEC_TX)) return 0;
    const BlkStruct *blk = ctx->blk_ptr;
    const BlockGeom *g = ctx->blk_geom;
    const int        d = blk->tx_depth, tot = g->txb_count[d], is_inter = 1;
    /* the residual of every transform block, as the encode loops form it (8-bit: input picture vs the prediction in the reconstruction buffer, :315-338;
     * 16-bit: the superblock's 16-bit input buffer, :677-700), split into two non-negative planes for the batched entry point (src - pred) */
    static __thread uint16_t hs[ED_MAX_COEFF / 2], hp[ED_MAX_COEFF / 2];
    uint32_t desc[ED_MAX_JOBS];
    SvtHipFwdTxJob jobs[ED_MAX_JOBS];
    int      n = 0, n_pix = 0, n_coeff = 0;
    for (int t = 0; t < tot; t++) {
        const int uv_pass = d && t ? 0 : 1;
        const uint32_t ox = ctx->blk_origin_x + g->tx_org_x[is_inter][d][t] - g->origin_x, oy = ctx->blk_origin_y + g->tx_org_y[is_inter][d][t] - g->origin_y;
        const uint32_t rx = (ox >> 3) << 3, ry = (oy >> 3) << 3;
        for (int p = 0; p < ((g->has_uv && uv_pass) ? 3 : 1); p++) {
            const int w = p ? g->tx_width_uv[d][t] : g->tx_width[d][t], h = p ? g->tx_height_uv[d][t] : g->tx_height[d][t];
            const int tx_size = p ? g->txsize_uv[d][t] : g->txsize[d][t];
            if (w > 32 || h > 32) continue;
            if (n_pix + w * h > ED_MAX_COEFF / 2) return 0;
            /* sample (x, y) of source and prediction */
            uint16_t *ps = hs + n_pix, *pp = hp + n_pix;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    int s, q;
                    if (!is_16bit) {
                        const EbPictureBufferDesc *in = ctx->input_samples;
                        if (p == 0) {
                            s = in->buffer_y[(size_t)(oy + in->origin_y + y) * in->stride_y + ox + in->origin_x + x];
                            q = pred->buffer_y[(size_t)(pred->origin_y + oy + y) * pred->stride_y + pred->origin_x + ox + x];
                        } else {
                            const uint8_t *ib = p == 1 ? in->buffer_cb : in->buffer_cr, *pb = p == 1 ? pred->buffer_cb : pred->buffer_cr;
                            const uint32_t is = p == 1 ? in->stride_cb : in->stride_cr, pst = p == 1 ? pred->stride_cb : pred->stride_cr;
                            s = ib[(size_t)(((ry + in->origin_y) >> 1) + y) * is + ((rx + in->origin_x) >> 1) + x];
                            q = pb[(size_t)(((pred->origin_y + ry) >> 1) + y) * pst + ((pred->origin_x + rx) >> 1) + x];
                        }
                    } else {
                        const EbPictureBufferDesc *in = ctx->input_sample16bit_buffer;
                        const uint32_t tx = g->tx_org_x[is_inter][d][t], ty = g->tx_org_y[is_inter][d][t];
                        if (p == 0) {
                            s = ((const uint16_t *)in->buffer_y)[(size_t)(ty + y) * in->stride_y + tx + x];
                            q = ((const uint16_t *)pred->buffer_y)[(size_t)(pred->origin_y + oy + y) * pred->stride_y + pred->origin_x + ox + x];
                        } else {
                            const uint16_t *ib = (const uint16_t *)(p == 1 ? in->buffer_cb : in->buffer_cr), *pb = (const uint16_t *)(p == 1 ? pred->buffer_cb : pred->buffer_cr);
                            const uint32_t is = p == 1 ? in->stride_cb : in->stride_cr, pst = p == 1 ? pred->stride_cb : pred->stride_cr;
                            s = ib[(size_t)(ROUND_UV(ty) / 2 + y) * is + ROUND_UV(tx) / 2 + x];
                            q = pb[(size_t)(((pred->origin_y + ry) >> 1) + y) * pst + ((pred->origin_x + rx) >> 1) + x];
                        }
                    }
                    ps[y * w + x] = (uint16_t)s; pp[y * w + x] = (uint16_t)q;
                }
            const int type0 = blk->txb_array[t].transform_type[p ? PLANE_TYPE_UV : PLANE_TYPE_Y];
            for (int k = 0; k < 2; k++) {
                const int type = k ? DCT_DCT : type0;
                if (k && type0 == DCT_DCT) break;
                if (n >= ED_MAX_JOBS || n_coeff + w * h > ED_MAX_COEFF) return 0;
                tls_ed.e[n].plane = (int8_t)p; tls_ed.e[n].txb = (int8_t)t; tls_ed.e[n].tx_size = (int8_t)tx_size; tls_ed.e[n].tx_type = (int8_t)type;
                tls_ed.e[n].off = n_coeff; tls_ed.e[n].count = w * h;
                desc[n] = SVT_HIP_TX_DESC(0, 0, type);
                memset(&jobs[n], 0, sizeof(jobs[n]));
                jobs[n].tx_size = tx_size; jobs[n].nblk = 1; jobs[n].src_stride = w; jobs[n].pred_stride = w;
                jobs[n].d_src = (const void *)(size_t)n_pix;      /* offsets for now: the device base is added under the lock */
                jobs[n].d_coeff = (int32_t *)(size_t)n_coeff;
                jobs[n].qp.coeff_shape = ctx->md_context->pf_ctrls.pf_shape;
                n++; n_coeff += w * h;
            }
            n_pix += w * h;
        }
    }
    if (!n) return 0;
    SvtHipCtx *hip = svt_hip_hooks_lock();
    if (!hip) return 0;
    int rc = SVT_HIP_OK;
    if (!d_ed_src) {
        rc = svt_hip_malloc(hip, &d_ed_src, sizeof(hs));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_ed_pred, sizeof(hp));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_ed_desc, sizeof(desc));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_ed_coeff, sizeof(tls_ed.coeff));
        if (rc != SVT_HIP_OK) { svt_hip_free(hip, d_ed_src); svt_hip_free(hip, d_ed_pred); svt_hip_free(hip, d_ed_desc); svt_hip_free(hip, d_ed_coeff); d_ed_src = d_ed_pred = d_ed_desc = d_ed_coeff = NULL; }
    }
    for (int i = 0; i < n && rc == SVT_HIP_OK; i++) {
        const size_t po = (size_t)jobs[i].d_src, co = (size_t)jobs[i].d_coeff;
        jobs[i].d_src = (const uint16_t *)d_ed_src + po; jobs[i].d_pred = (const uint16_t *)d_ed_pred + po;
        jobs[i].d_descs = (const uint32_t *)d_ed_desc + i; jobs[i].d_coeff = (int32_t *)d_ed_coeff + co;
    }
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_ed_src, hs, sizeof(uint16_t) * (size_t)n_pix);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_ed_pred, hp, sizeof(uint16_t) * (size_t)n_pix);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_ed_desc, desc, sizeof(uint32_t) * (size_t)n);
    if (rc == SVT_HIP_OK) rc = svt_hip_fwd_txfm_quant_multi_dev(hip, 2, jobs, n);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, tls_ed.coeff, d_ed_coeff, sizeof(int32_t) * (size_t)n_coeff);
    if (rc == SVT_HIP_OK) g_ed_blocks++;
    svt_hip_hooks_unlock();
    svt_hip_hooks_count(SVT_HIP_HOOK_ENCDEC_TX, rc == SVT_HIP_OK);
    if (rc != SVT_HIP_OK) return 0;
    tls_ed.n = n; tls_ed.valid = 1;
    return 1;
}

/* av1_estimate_transform of transform block `txb` of plane `plane` inside av1_encode_loop[_16bit]: 1 = coeff holds the device result */
int svt_hip_hook_encdec_tx_fetch(int plane, int txb, int tx_size, int tx_type, int32_t *coeff) {
    if (!tls_ed.valid) return 0;
    for (int i = 0; i < tls_ed.n; i++)
        if (tls_ed.e[i].plane == plane && tls_ed.e[i].txb == txb && tls_ed.e[i].tx_size == tx_size && tls_ed.e[i].tx_type == tx_type) {
            memcpy(coeff, tls_ed.coeff + tls_ed.e[i].off, sizeof(int32_t) * (size_t)tls_ed.e[i].count);
            __sync_fetch_and_add(&g_ed_tx, 1);
            return 1;
        }
    return 0;
}
void svt_hip_hook_encdec_tx_end(void) { tls_ed.valid = 0; }

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * Mode decision's sub-pel refinement, hook "md_subpel": one round of svt_av1_find_best_sub_pixel_tree (mcomp.c:350; md_subpel_search, EbProductCodingLoop.c:2063)
 * evaluates the four axis neighbours of its centre at the round's step and then one diagonal (svt_first_level_check, mcomp.c:186-250) — each
 * svt_upsampled_pref_error (:102) = svt_aom_upsampled_pred + the block size's variance function.  The candidates of a round are independent, so
 * svt_hip_hook_md_subpel_begin predicts and measures all eight neighbours (the four diagonals cover whichever one the comparison picks) in one launch pair:
 * the reference window (block + 8 taps + the one-sample spread of the candidates) and the source block travel once, svt_hip_upsampled_pred_batch_dev predicts
 * the list, svt_hip_block_variance_batch_dev measures it, and the round's svt_upsampled_pref_error calls read (variance, sse) from the thread's cache.  The tree
 * itself — motion-vector costs, the strict "<" of svt_check_better, the diagonal rule, the second-level probes around the new best vector — stays the
 * reference's control flow.  Opt-in like md_tx. */
#include "mcomp.h"
static __thread struct { int valid, n; MV mv[8]; uint32_t var[8], sse[8]; } tls_sp;
static void *d_sp_ref, *d_sp_src, *d_sp_pred, *d_sp_job, *d_sp_out;   /* shared staging, hooks lock held */
#define SP_W 144   /* pitch of the staged reference window: 128 + 8 taps + 1, rounded up */

int svt_hip_hook_md_subpel_begin(const SUBPEL_SEARCH_VAR_PARAMS *vp, const MV *centre, int hstep, const SubpelMvLimits *lim) {
    tls_sp.valid = 0;
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_MD_SUBPEL)) return 0;
    const int w = vp->w, h = vp->h, st = (int)vp->subpel_search_type;
    const int bank = st == 1 ? 3 : (st == 2 ? 4 : (st == 3 ? 0 : -1));   /* USE_2_TAPS / USE_4_TAPS / USE_8_TAPS (EbDefinitions.h:487-490, variance.c:200-209) */
    if (bank < 0 || w < 4 || h < 4 || w > 128 || h > 128 || hstep < 1 || hstep > 4) return 0;
    const struct svt_buf_2d *rb = vp->ms_buffers.ref, *sb = vp->ms_buffers.src;
    SvtHipUpsampledBlk job[8];
    SvtHipBlkPair      pair[8];
    int                n = 0;
    /* integer positions of the candidates lie in {r0, r0 + 1} x {c0, c0 + 1} (the step is below one sample) */
    const int r0 = (centre->row - hstep) >> 3, c0 = (centre->col - hstep) >> 3;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            if (!dx && !dy) continue;
            const MV mv = {(int16_t)(centre->row + dy * hstep), (int16_t)(centre->col + dx * hstep)};
            if (!svt_av1_is_subpelmv_in_range(lim, mv)) continue;
            memset(&job[n], 0, sizeof(job[n]));
            job[n].ref_off = (((mv.row >> 3) - r0) + 3) * SP_W + ((mv.col >> 3) - c0) + 3;
            job[n].dst_off = n * w * h;
            job[n].w = (uint8_t)w; job[n].h = (uint8_t)h; job[n].subpel_x_q3 = (uint8_t)(mv.col & 7); job[n].subpel_y_q3 = (uint8_t)(mv.row & 7); job[n].bank = (uint8_t)bank;
            pair[n].a_x = 0; pair[n].a_y = n * h; pair[n].b_x = 0; pair[n].b_y = 0; pair[n].w = (uint16_t)w; pair[n].h = (uint16_t)h;
            tls_sp.mv[n] = mv;
            n++;
        }
    if (!n) return 0;
    static __thread uint8_t win[SP_W * (128 + 9)], src[128 * 128];
    const int ww = w + 9, wh = h + 9;
    const uint8_t *r = rb->buf + (ptrdiff_t)(r0 - 3) * rb->stride + (c0 - 3);
    for (int y = 0; y < wh; y++) memcpy(win + (size_t)y * SP_W, r + (ptrdiff_t)y * rb->stride, (size_t)ww);
    for (int y = 0; y < h; y++) memcpy(src + (size_t)y * w, sb->buf + (ptrdiff_t)y * sb->stride, (size_t)w);
    SvtHipCtx *hip = svt_hip_hooks_lock();
    if (!hip) return 0;
    int rc = SVT_HIP_OK;
    if (!d_sp_ref) {
        rc = svt_hip_malloc(hip, &d_sp_ref, sizeof(win) + 64);
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_sp_src, sizeof(src));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_sp_pred, 8 * sizeof(src));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_sp_job, sizeof(job) + sizeof(pair));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &d_sp_out, 16 * sizeof(uint32_t));
        if (rc != SVT_HIP_OK) { svt_hip_free(hip, d_sp_ref); svt_hip_free(hip, d_sp_src); svt_hip_free(hip, d_sp_pred); svt_hip_free(hip, d_sp_job); svt_hip_free(hip, d_sp_out); d_sp_ref = d_sp_src = d_sp_pred = d_sp_job = d_sp_out = NULL; }
    }
    uint8_t  both[sizeof(job) + sizeof(pair)];
    uint32_t out[16];
    memcpy(both, job, sizeof(job)); memcpy(both + sizeof(job), pair, sizeof(pair));
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_sp_ref, win, (size_t)SP_W * wh);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_sp_src, src, (size_t)w * h);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d(hip, d_sp_job, both, sizeof(both));
    if (rc == SVT_HIP_OK) rc = svt_hip_upsampled_pred_batch_dev(hip, (const uint8_t *)d_sp_ref, SP_W, (uint8_t *)d_sp_pred, (const SvtHipUpsampledBlk *)d_sp_job, n);
    if (rc == SVT_HIP_OK)
        rc = svt_hip_block_variance_batch_dev(hip, 1, 8, d_sp_pred, w, d_sp_src, w, (const SvtHipBlkPair *)((const uint8_t *)d_sp_job + sizeof(job)), n, (uint32_t *)d_sp_out,
                                              (uint32_t *)d_sp_out + 8);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, out, d_sp_out, sizeof(out));
    svt_hip_hooks_unlock();
    svt_hip_hooks_count(SVT_HIP_HOOK_MD_SUBPEL, rc == SVT_HIP_OK);
    if (rc != SVT_HIP_OK) return 0;
    for (int i = 0; i < n; i++) { tls_sp.var[i] = out[i]; tls_sp.sse[i] = out[8 + i]; }
    tls_sp.n = n; tls_sp.valid = 1;
    return 1;
}
/* svt_upsampled_pref_error of a candidate of the round begun above: 1 = *err / *sse hold the device results */
int svt_hip_hook_md_subpel_fetch(const MV *mv, unsigned int *err, unsigned int *sse) {
    if (!tls_sp.valid) return 0;
    for (int i = 0; i < tls_sp.n; i++)
        if (tls_sp.mv[i].row == mv->row && tls_sp.mv[i].col == mv->col) { *err = tls_sp.var[i]; *sse = tls_sp.sse[i]; return 1; }
    return 0;
}
void svt_hip_hook_md_subpel_end(void) { tls_sp.valid = 0; }

/* the shared staging buffers of the three hooks above (svt_hip_hooks_enc_deinit) */
void svt_hip_md_bridge_release(SvtHipCtx *hip) {
    void **all[] = {&d_src, &d_pred, &d_desc, &d_coeff, &d_ed_src, &d_ed_pred, &d_ed_desc, &d_ed_coeff, &d_sp_ref, &d_sp_src, &d_sp_pred, &d_sp_job, &d_sp_out};
    for (unsigned i = 0; i < sizeof(all) / sizeof(all[0]); i++) { svt_hip_free(hip, *all[i]); *all[i] = NULL; }
}
