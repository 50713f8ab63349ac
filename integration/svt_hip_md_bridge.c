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
/* the transform-block jobs of ONE inter-coded block appended to a batch: source and prediction samples (two 16-bit planes), one job per (plane, transform
 * block, type); 0 = the batch's capacity is exhausted (nothing appended) */
typedef struct { int16_t blk; int8_t plane, txb, tx_size, tx_type; int32_t off, count; } EdEntry;
typedef struct {
    uint16_t *hs, *hp; int pix_cap, n_pix;
    uint32_t *desc; SvtHipFwdTxJob *jobs; EdEntry *e; int job_cap, n;
    int coeff_cap, n_coeff;
} EdBatch;
static int ed_gather_block(EdBatch *B, EncDecContext *ctx, const BlkStruct *blk, const BlockGeom *g, uint32_t blk_origin_x, uint32_t blk_origin_y, const EbPictureBufferDesc *pred,
                           int is_16bit) {
    const int d = blk->tx_depth, tot = g->txb_count[d], is_inter = 1;
    const int n0 = B->n, pix0 = B->n_pix, coeff0 = B->n_coeff;
    /* the residual of every transform block, as the encode loops form it (8-bit: input picture vs the prediction in the reconstruction buffer, :315-338;
     * 16-bit: the superblock's 16-bit input buffer, :677-700), split into two non-negative planes for the batched entry point (src - pred) */
    for (int t = 0; t < tot; t++) {
        const int uv_pass = d && t ? 0 : 1;
        const uint32_t ox = blk_origin_x + g->tx_org_x[is_inter][d][t] - g->origin_x, oy = blk_origin_y + g->tx_org_y[is_inter][d][t] - g->origin_y;
        const uint32_t rx = (ox >> 3) << 3, ry = (oy >> 3) << 3;
        for (int p = 0; p < ((g->has_uv && uv_pass) ? 3 : 1); p++) {
            const int w = p ? g->tx_width_uv[d][t] : g->tx_width[d][t], h = p ? g->tx_height_uv[d][t] : g->tx_height[d][t];
            const int tx_size = p ? g->txsize_uv[d][t] : g->txsize[d][t];
            if (w > 32 || h > 32) continue;
            if (B->n_pix + w * h > B->pix_cap) goto full;
            /* sample (x, y) of source and prediction */
            uint16_t *ps = B->hs + B->n_pix, *pp = B->hp + B->n_pix;
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
                if (B->n >= B->job_cap || B->n_coeff + w * h > B->coeff_cap) goto full;
                EdEntry *e = &B->e[B->n];
                e->blk = (int16_t)blk->mds_idx; e->plane = (int8_t)p; e->txb = (int8_t)t; e->tx_size = (int8_t)tx_size; e->tx_type = (int8_t)type;
                e->off = B->n_coeff; e->count = w * h;
                B->desc[B->n] = SVT_HIP_TX_DESC(0, 0, type);
                SvtHipFwdTxJob *j = &B->jobs[B->n];
                memset(j, 0, sizeof(*j));
                j->tx_size = tx_size; j->nblk = 1; j->src_stride = w; j->pred_stride = w;
                j->d_src = (const void *)(size_t)B->n_pix;      /* offsets for now: the device base is added once the blocks exist */
                j->d_coeff = (int32_t *)(size_t)B->n_coeff;
                j->qp.coeff_shape = ctx->md_context->pf_ctrls.pf_shape;
                B->n++; B->n_coeff += w * h;
            }
            B->n_pix += w * h;
        }
    }
    return 1;
full:
    B->n = n0; B->n_pix = pix0; B->n_coeff = coeff0;
    return 0;
}
/* one launch for the batch: coefficients of every job -> coeff (host).  A pool context and blocks of the hooks' cache: encode-pass threads run side by side. */
static int ed_launch(EdBatch *B, int32_t *coeff) {
    SvtHipCtx *hip = svt_hip_hooks_lock_any();
    if (!hip) return SVT_HIP_ERR_RUNTIME;
    void *d_s = NULL, *d_p = NULL, *d_d = NULL, *d_c = NULL;
    int   rc = svt_hip_hooks_malloc(hip, &d_s, sizeof(uint16_t) * (size_t)B->n_pix);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_p, sizeof(uint16_t) * (size_t)B->n_pix);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_d, sizeof(uint32_t) * (size_t)B->n);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_c, sizeof(int32_t) * (size_t)B->n_coeff);
    for (int i = 0; i < B->n && rc == SVT_HIP_OK; i++) {
        const size_t po = (size_t)B->jobs[i].d_src, co = (size_t)B->jobs[i].d_coeff;
        B->jobs[i].d_src = (const uint16_t *)d_s + po; B->jobs[i].d_pred = (const uint16_t *)d_p + po;
        B->jobs[i].d_descs = (const uint32_t *)d_d + i; B->jobs[i].d_coeff = (int32_t *)d_c + co;
    }
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d_async(hip, d_s, B->hs, sizeof(uint16_t) * (size_t)B->n_pix);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d_async(hip, d_p, B->hp, sizeof(uint16_t) * (size_t)B->n_pix);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d_async(hip, d_d, B->desc, sizeof(uint32_t) * (size_t)B->n);
    if (rc == SVT_HIP_OK) rc = svt_hip_fwd_txfm_quant_multi_dev(hip, 2, B->jobs, B->n);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, coeff, d_c, sizeof(int32_t) * (size_t)B->n_coeff);   /* drains the stream: the staging arrays may be reused */
    if (rc != SVT_HIP_OK) (void)svt_hip_sync(hip);
    svt_hip_hooks_free(hip, d_s); svt_hip_hooks_free(hip, d_p); svt_hip_hooks_free(hip, d_d); svt_hip_hooks_free(hip, d_c);
    svt_hip_hooks_unlock_any();
    return rc;
}

static __thread struct {
    int      valid, n;
    EdEntry  e[ED_MAX_JOBS];
    int32_t  coeff[ED_MAX_COEFF];
} tls_ed;
static long  g_ed_blocks, g_ed_tx;                            /* inter blocks batched / av1_estimate_transform calls they replaced (svt_hip_hooks_report) */

void svt_hip_hook_encdec_tx_stats(long *blocks, long *calls) { *blocks = g_ed_blocks; *calls = g_ed_tx; }

static int sb_begin_block(const BlkStruct *blk);
int svt_hip_hook_encdec_tx_begin(EncDecContext *ctx, const EbPictureBufferDesc *pred, int is_16bit) {
    tls_ed.valid = 0;
    if (sb_begin_block(ctx->blk_ptr)) return 1;   /* hook "encdec_sb": this block's transforms came with its superblock's launch */
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_ENCDEC_TX)) return 0;
    static __thread uint16_t hs[ED_MAX_COEFF / 2], hp[ED_MAX_COEFF / 2];
    uint32_t       desc[ED_MAX_JOBS];
    SvtHipFwdTxJob jobs[ED_MAX_JOBS];
    EdBatch        B = {hs, hp, ED_MAX_COEFF / 2, 0, desc, jobs, tls_ed.e, ED_MAX_JOBS, 0, ED_MAX_COEFF, 0};
    if (!ed_gather_block(&B, ctx, ctx->blk_ptr, ctx->blk_geom, ctx->blk_origin_x, ctx->blk_origin_y, pred, is_16bit) || !B.n) return 0;
    const int rc = ed_launch(&B, tls_ed.coeff);
    if (rc == SVT_HIP_OK) __sync_fetch_and_add(&g_ed_blocks, 1);
    svt_hip_hooks_count(SVT_HIP_HOOK_ENCDEC_TX, rc == SVT_HIP_OK);
    if (rc != SVT_HIP_OK) return 0;
    tls_ed.n = B.n; tls_ed.valid = 1;
    return 1;
}

/* av1_estimate_transform of transform block `txb` of plane `plane` inside av1_encode_loop[_16bit]: 1 = coeff holds the device result */
static int sb_fetch(int plane, int txb, int tx_size, int tx_type, int32_t *coeff);
int svt_hip_hook_encdec_tx_fetch(int plane, int txb, int tx_size, int tx_type, int32_t *coeff) {
    if (sb_fetch(plane, txb, tx_size, tx_type, coeff)) return 1;
    if (!tls_ed.valid) return 0;
    for (int i = 0; i < tls_ed.n; i++)
        if (tls_ed.e[i].plane == plane && tls_ed.e[i].txb == txb && tls_ed.e[i].tx_size == tx_size && tls_ed.e[i].tx_type == tx_type) {
            memcpy(coeff, tls_ed.coeff + tls_ed.e[i].off, sizeof(int32_t) * (size_t)tls_ed.e[i].count);
            __sync_fetch_and_add(&g_ed_tx, 1);
            return 1;
        }
    return 0;
}
static void sb_end_block(void);
void svt_hip_hook_encdec_tx_end(void) { tls_ed.valid = 0; sb_end_block(); }

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * Encode pass, hook "encdec_sb": ONE launch per SUPERBLOCK.  av1_encode_decode (EbCodingLoop.c:1987) predicts an inter block right before it codes it (:2848-3006),
 * so a block's residual only exists once the blocks before it are coded — but an inter prediction with plain translation reads reference PICTURES only, never
 * the picture being coded, and every mode of the superblock is final when the encode pass starts (mode decision ran on the whole superblock first).  So, before
 * the block loop of a superblock (svt_hip_hook_encdec_sb_begin): every final block that is inter-coded with SIMPLE_TRANSLATION motion, no inter-intra, no 4-sample
 * side (those chroma predictions look at neighbouring mode info the loop is still going to write) is predicted into its own area of the reconstruction buffer — the
 * reference's own av1_inter_prediction[_16bit_pipeline] with the arguments the loop would pass and the block's frame-edge distances in the shared MacroBlockD, which the
 * loop's enc_pass_av1_mv_pred would have set (EbAdaptiveMotionVectorPrediction.c:1185-1188) — and the forward transforms of ALL their transform blocks (luma, chroma,
 * chosen type and DCT_DCT) go to the device in one launch.  The loop then skips the prediction of those blocks (svt_hip_hook_encdec_sb_predicted) and its
 * av1_estimate_transform calls read the superblock's cache.  Nobody reads a block's area of the reconstruction buffer before the block is coded (intra prediction
 * works from the neighbour arrays, intra block copy from areas coded earlier), so the early prediction is invisible.  Quantisation with RDOQ (entropy contexts of the
 * neighbouring transform blocks), the inverse transforms and intra blocks stay the reference's, in coding order.  Opt-in. */
#include "EbEncInterPrediction.h"
#include "EbModeDecisionConfigurationProcess.h"
#include "EbReferenceObject.h"
#include "EbUtility.h"
#define SB_MAX_BLK 4432    /* BLOCK_MAX_COUNT_SB_128 */
#define SB_MAX_TU 2560     /* (plane, transform block, type) records of one superblock */
#define SB_LW 128          /* staging planes of a superblock: luma 128 x 128, each chroma plane 64 x 64, 16-bit samples, source and prediction */
#define SB_PIX (SB_LW * SB_LW + 2 * (SB_LW / 2) * (SB_LW / 2))
#define SB_MAX_GROUPS 64   /* distinct (plane, transform size) pairs: one job each */
static __thread struct {
    int       valid, n, cur;   /* cur: mds index of the block whose transform loops run now (-1: none of the batch) */
    int       lo, hi;          /* the current block's entries (a block's entries are contiguous) */
    uint8_t   predicted[SB_MAX_BLK];
    EdEntry   e[SB_MAX_TU];
    uint32_t  tu_xy[SB_MAX_TU]; /* position of the record's transform block inside its staging plane: x | y << 14 */
    int32_t  *coeff;           /* [2 * SB_PIX], allocated at the thread's first superblock */
    uint16_t *hs, *hp;         /* [SB_PIX] each */
} tls_sb;
static long g_sb_launches, g_sb_blocks, g_sb_superblocks, g_sb_tx, g_sb_kernels;

void svt_hip_hook_encdec_sb_stats(long *superblocks, long *launches, long *blocks, long *calls) { *superblocks = g_sb_superblocks; *launches = g_sb_launches; *blocks = g_sb_blocks; *calls = g_sb_tx; }
long svt_hip_hook_encdec_sb_kernels(void) { return g_sb_kernels; }

/* One block's samples into the superblock's staging planes + one record per (plane, transform block, type); 0 = no room for the block's records (nothing kept) */
static int sb_gather_block(int *n_rec, int *n_coeff, EncDecContext *ctx, const BlkStruct *blk, const BlockGeom *g, uint32_t sb_x, uint32_t sb_y, uint32_t blk_origin_x,
                           uint32_t blk_origin_y, const EbPictureBufferDesc *pred, int is_16bit) {
    const int d = blk->tx_depth, tot = g->txb_count[d], is_inter = 1, n0 = *n_rec, c0 = *n_coeff;
    static const int plane_off[3] = {0, SB_LW * SB_LW, SB_LW * SB_LW + (SB_LW / 2) * (SB_LW / 2)};
    for (int t = 0; t < tot; t++) {
        const int uv_pass = d && t ? 0 : 1;
        const uint32_t ox = blk_origin_x + g->tx_org_x[is_inter][d][t] - g->origin_x, oy = blk_origin_y + g->tx_org_y[is_inter][d][t] - g->origin_y;
        const uint32_t rx = (ox >> 3) << 3, ry = (oy >> 3) << 3;
        for (int p = 0; p < ((g->has_uv && uv_pass) ? 3 : 1); p++) {
            const int w = p ? g->tx_width_uv[d][t] : g->tx_width[d][t], h = p ? g->tx_height_uv[d][t] : g->tx_height[d][t];
            const int tx_size = p ? g->txsize_uv[d][t] : g->txsize[d][t];
            if (w > 32 || h > 32) continue;
            const int px = p ? (int)(rx - sb_x) >> 1 : (int)(ox - sb_x), py = p ? (int)(ry - sb_y) >> 1 : (int)(oy - sb_y), pst = p ? SB_LW / 2 : SB_LW;
            if (px < 0 || py < 0 || px + w > pst || py + h > pst) goto full;
            uint16_t *ps = tls_sb.hs + plane_off[p] + py * pst + px, *pp = tls_sb.hp + plane_off[p] + py * pst + px;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    int sv, q;
                    if (!is_16bit) {
                        const EbPictureBufferDesc *in = ctx->input_samples;
                        if (p == 0) {
                            sv = in->buffer_y[(size_t)(oy + in->origin_y + y) * in->stride_y + ox + in->origin_x + x];
                            q = pred->buffer_y[(size_t)(pred->origin_y + oy + y) * pred->stride_y + pred->origin_x + ox + x];
                        } else {
                            const uint8_t *ib = p == 1 ? in->buffer_cb : in->buffer_cr, *pb = p == 1 ? pred->buffer_cb : pred->buffer_cr;
                            const uint32_t is = p == 1 ? in->stride_cb : in->stride_cr, pstr = p == 1 ? pred->stride_cb : pred->stride_cr;
                            sv = ib[(size_t)(((ry + in->origin_y) >> 1) + y) * is + ((rx + in->origin_x) >> 1) + x];
                            q = pb[(size_t)(((pred->origin_y + ry) >> 1) + y) * pstr + ((pred->origin_x + rx) >> 1) + x];
                        }
                    } else {
                        const EbPictureBufferDesc *in = ctx->input_sample16bit_buffer;
                        const uint32_t tx = g->tx_org_x[is_inter][d][t], ty = g->tx_org_y[is_inter][d][t];
                        if (p == 0) {
                            sv = ((const uint16_t *)in->buffer_y)[(size_t)(ty + y) * in->stride_y + tx + x];
                            q = ((const uint16_t *)pred->buffer_y)[(size_t)(pred->origin_y + oy + y) * pred->stride_y + pred->origin_x + ox + x];
                        } else {
                            const uint16_t *ib = (const uint16_t *)(p == 1 ? in->buffer_cb : in->buffer_cr), *pb = (const uint16_t *)(p == 1 ? pred->buffer_cb : pred->buffer_cr);
                            const uint32_t is = p == 1 ? in->stride_cb : in->stride_cr, pstr = p == 1 ? pred->stride_cb : pred->stride_cr;
                            sv = ib[(size_t)(ROUND_UV(ty) / 2 + y) * is + ROUND_UV(tx) / 2 + x];
                            q = pb[(size_t)(((pred->origin_y + ry) >> 1) + y) * pstr + ((pred->origin_x + rx) >> 1) + x];
                        }
                    }
                    ps[y * pst + x] = (uint16_t)sv; pp[y * pst + x] = (uint16_t)q;
                }
            const int type0 = blk->txb_array[t].transform_type[p ? PLANE_TYPE_UV : PLANE_TYPE_Y];
            for (int k = 0; k < 2; k++) {
                const int type = k ? DCT_DCT : type0;
                if (k && type0 == DCT_DCT) break;
                if (*n_rec >= SB_MAX_TU || *n_coeff + w * h > 2 * SB_PIX) goto full;
                EdEntry *e = &tls_sb.e[*n_rec];
                e->blk = (int16_t)blk->mds_idx; e->plane = (int8_t)p; e->txb = (int8_t)t; e->tx_size = (int8_t)tx_size; e->tx_type = (int8_t)type;
                e->off = -1; e->count = w * h;
                tls_sb.tu_xy[*n_rec] = (uint32_t)px | (uint32_t)py << 14;
                (*n_rec)++; *n_coeff += w * h;
            }
        }
    }
    return 1;
full:
    *n_rec = n0; *n_coeff = c0;
    return 0;
}
/* the records grouped into one job per (plane, transform size): the whole superblock is ONE entry-point call, and one kernel launch as long as it has at most 16
 * such pairs (the library packs 16 job lists into a launch) */
static int sb_launch(int n_rec, int n_coeff, int coeff_shape) {
    static __thread uint32_t       desc[SB_MAX_TU];
    static __thread SvtHipFwdTxJob jobs[SB_MAX_GROUPS];
    static const int plane_off[3] = {0, SB_LW * SB_LW, SB_LW * SB_LW + (SB_LW / 2) * (SB_LW / 2)};
    int n_jobs = 0, nd = 0, co = 0;
    for (int p = 0; p < 3; p++)
        for (int ts = 0; ts < TX_SIZES_ALL; ts++) {
            int cnt = 0;
            for (int i = 0; i < n_rec; i++)
                if (tls_sb.e[i].plane == p && tls_sb.e[i].tx_size == ts) {
                    if (!cnt) {
                        if (n_jobs >= SB_MAX_GROUPS) return SVT_HIP_ERR_UNSUPPORTED;
                        memset(&jobs[n_jobs], 0, sizeof(jobs[0]));
                        jobs[n_jobs].tx_size = ts; jobs[n_jobs].src_stride = jobs[n_jobs].pred_stride = p ? SB_LW / 2 : SB_LW;
                        jobs[n_jobs].d_src = (const void *)(size_t)plane_off[p];   /* offsets for now */
                        jobs[n_jobs].d_descs = (const uint32_t *)(size_t)nd; jobs[n_jobs].d_coeff = (int32_t *)(size_t)co;
                        jobs[n_jobs].qp.coeff_shape = coeff_shape;
                    }
                    desc[nd++] = SVT_HIP_TX_DESC(tls_sb.tu_xy[i] & 0x3FFF, tls_sb.tu_xy[i] >> 14, tls_sb.e[i].tx_type);
                    tls_sb.e[i].off = co; co += tls_sb.e[i].count;
                    cnt++;
                }
            if (cnt) jobs[n_jobs++].nblk = cnt;
        }
    SvtHipCtx *hip = svt_hip_hooks_lock_any();
    if (!hip) return SVT_HIP_ERR_RUNTIME;
    void *d_s = NULL, *d_p = NULL, *d_d = NULL, *d_c = NULL;
    int   rc = svt_hip_hooks_malloc(hip, &d_s, sizeof(uint16_t) * SB_PIX);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_p, sizeof(uint16_t) * SB_PIX);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_d, sizeof(uint32_t) * (size_t)nd);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_c, sizeof(int32_t) * (size_t)n_coeff);
    for (int j = 0; j < n_jobs && rc == SVT_HIP_OK; j++) {
        const size_t po = (size_t)jobs[j].d_src;
        jobs[j].d_src = (const uint16_t *)d_s + po; jobs[j].d_pred = (const uint16_t *)d_p + po;
        jobs[j].d_descs = (const uint32_t *)d_d + (size_t)jobs[j].d_descs; jobs[j].d_coeff = (int32_t *)d_c + (size_t)jobs[j].d_coeff;
    }
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d_async(hip, d_s, tls_sb.hs, sizeof(uint16_t) * SB_PIX);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d_async(hip, d_p, tls_sb.hp, sizeof(uint16_t) * SB_PIX);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d_async(hip, d_d, desc, sizeof(uint32_t) * (size_t)nd);
    if (rc == SVT_HIP_OK) rc = svt_hip_fwd_txfm_quant_multi_dev(hip, 2, jobs, n_jobs);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, tls_sb.coeff, d_c, sizeof(int32_t) * (size_t)n_coeff);
    if (rc != SVT_HIP_OK) (void)svt_hip_sync(hip);
    svt_hip_hooks_free(hip, d_s); svt_hip_hooks_free(hip, d_p); svt_hip_hooks_free(hip, d_d); svt_hip_hooks_free(hip, d_c);
    svt_hip_hooks_unlock_any();
    if (rc == SVT_HIP_OK) __sync_fetch_and_add(&g_sb_kernels, (n_jobs + 15) / 16);
    return rc;
}

static int sb_hoistable(const BlkStruct *blk, const BlockGeom *g) {
    return blk->prediction_mode_flag == INTER_MODE && !blk->use_intrabc && blk->prediction_unit_array[0].motion_mode == SIMPLE_TRANSLATION && !blk->is_interintra_used &&
           g->bwidth > 4 && g->bheight > 4;
}
/* the prediction call of av1_encode_decode (:2946-3004) for one block */
static void sb_predict(SequenceControlSet *scs, PictureControlSet *pcs, SuperBlock *sb, EncDecContext *ctx, BlkStruct *blk, const BlockGeom *g, uint32_t org_x, uint32_t org_y,
                       EbPictureBufferDesc *recon, int is_16bit) {
    ModeDecisionContext *md = ctx->md_context;
    const int8_t ref_idx_l0 = md->md_local_blk_unit[g->blkidx_mds].ref_frame_index_l0, ref_idx_l1 = md->md_local_blk_unit[g->blkidx_mds].ref_frame_index_l1;
    MvReferenceFrame rf[2];
    av1_set_ref_frame(rf, blk->prediction_unit_array[0].ref_frame_type);
    const uint8_t list_idx0 = get_list_idx(rf[0]), list_idx1 = rf[1] == NONE_FRAME ? get_list_idx(rf[0]) : get_list_idx(rf[1]);
    EbReferenceObject *ref_obj_0 = ref_idx_l0 >= 0 ? (EbReferenceObject *)pcs->ref_pic_ptr_array[list_idx0][ref_idx_l0]->object_ptr : NULL;
    EbReferenceObject *ref_obj_1 = ref_idx_l1 >= 0 ? (EbReferenceObject *)pcs->ref_pic_ptr_array[list_idx1][ref_idx_l1]->object_ptr : NULL;
    EbPictureBufferDesc *ref0 = ref_obj_0 ? (is_16bit ? ref_obj_0->reference_picture16bit : ref_obj_0->reference_picture) : NULL;
    EbPictureBufferDesc *ref1 = ref_obj_1 ? (is_16bit ? ref_obj_1->reference_picture16bit : ref_obj_1->reference_picture) : NULL;
    const PredictionUnit *pu = blk->prediction_unit_array;
    MvUnit mvu;
    mvu.pred_direction = (uint8_t)pu->inter_pred_direction_index;
    mvu.mv[REF_LIST_0].mv_union = pu->mv[REF_LIST_0].mv_union;
    mvu.mv[REF_LIST_1].mv_union = pu->mv[REF_LIST_1].mv_union;
    /* what generate_av1_mvp_table leaves in the superblock's MacroBlockD for this block: the motion-vector clamp of the predictor reads it */
    MacroBlockD *xd = blk->av1xd;
    const Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
    const int32_t mi_row = org_y >> MI_SIZE_LOG2, mi_col = org_x >> MI_SIZE_LOG2, bw = mi_size_wide[g->bsize], bh = mi_size_high[g->bsize];
    xd->mb_to_top_edge = -((mi_row * MI_SIZE) * 8); xd->mb_to_bottom_edge = ((cm->mi_rows - bh - mi_row) * MI_SIZE) * 8;
    xd->mb_to_left_edge = -((mi_col * MI_SIZE) * 8); xd->mb_to_right_edge = ((cm->mi_cols - bw - mi_col) * MI_SIZE) * 8;
    const uint16_t tile_idx = ctx->tile_index;
    NeighborArrayUnit *nl = is_16bit ? pcs->ep_luma_recon_neighbor_array16bit[tile_idx] : pcs->ep_luma_recon_neighbor_array[tile_idx];
    NeighborArrayUnit *ncb = is_16bit ? pcs->ep_cb_recon_neighbor_array16bit[tile_idx] : pcs->ep_cb_recon_neighbor_array[tile_idx];
    NeighborArrayUnit *ncr = is_16bit ? pcs->ep_cr_recon_neighbor_array16bit[tile_idx] : pcs->ep_cr_recon_neighbor_array[tile_idx];
    if (is_16bit && !(scs->static_config.superres_mode > SUPERRES_NONE))
        av1_inter_prediction_16bit_pipeline(pcs, blk->interp_filters, blk, pu->ref_frame_type, &mvu, 0, pu->motion_mode, 0, 0, blk->compound_idx, &blk->interinter_comp, &sb->tile_info, nl,
                                            ncb, ncr, blk->is_interintra_used, blk->interintra_mode, blk->use_wedge_interintra, blk->interintra_wedge_index, (uint16_t)org_x, (uint16_t)org_y,
                                            g->bwidth, g->bheight, ref0, ref1, recon, (uint16_t)org_x, (uint16_t)org_y, EB_TRUE, (uint8_t)scs->static_config.encoder_bit_depth);
    else
        av1_inter_prediction(pcs, blk->interp_filters, blk, pu->ref_frame_type, &mvu, 0, pu->motion_mode, 0, 0, blk->compound_idx, &blk->interinter_comp, &sb->tile_info, nl, ncb, ncr,
                             blk->is_interintra_used, blk->interintra_mode, blk->use_wedge_interintra, blk->interintra_wedge_index, (uint16_t)org_x, (uint16_t)org_y, g->bwidth, g->bheight,
                             ref0, ref1, recon, (uint16_t)org_x, (uint16_t)org_y, EB_TRUE, (uint8_t)scs->static_config.encoder_bit_depth);
}

/* av1_encode_decode, right before its block loop (:2262): 1 = blocks of this superblock were predicted and their transforms are cached */
int svt_hip_hook_encdec_sb_begin(SequenceControlSet *scs, PictureControlSet *pcs, SuperBlock *sb, uint32_t sb_addr, uint32_t sb_origin_x, uint32_t sb_origin_y, EncDecContext *ctx,
                                 EbPictureBufferDesc *recon, int is_16bit) {
    tls_sb.valid = 0; tls_sb.cur = -1;
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_ENCDEC_SB) || scs->max_block_cnt > SB_MAX_BLK || pcs->slice_type == I_SLICE) return 0;
    if (!tls_sb.coeff) {
        tls_sb.coeff = (int32_t *)malloc(sizeof(int32_t) * 2 * SB_PIX); tls_sb.hs = (uint16_t *)calloc(SB_PIX, sizeof(uint16_t)); tls_sb.hp = (uint16_t *)calloc(SB_PIX, sizeof(uint16_t));
        if (!tls_sb.coeff || !tls_sb.hs || !tls_sb.hp) { free(tls_sb.coeff); free(tls_sb.hs); free(tls_sb.hp); tls_sb.coeff = NULL; tls_sb.hs = tls_sb.hp = NULL; return 0; }
    }
    int n_rec = 0, n_coeff = 0;
    memset(tls_sb.predicted, 0, scs->max_block_cnt);
    ModeDecisionContext *md = ctx->md_context;
    const int sb128 = scs->seq_header.sb_size == BLOCK_128X128;
    int nblk = 0;
    /* the walk of the block loop (:2262-3785): the final partition of the superblock */
    uint32_t blk_it = 0;
    while (blk_it < scs->max_block_cnt) {
        const BlockGeom *g0 = get_blk_geom_mds(blk_it);
        const PartitionType part = md->md_blk_arr_nsq[blk_it].part;
        if (part != PARTITION_SPLIT && pcs->parent_pcs_ptr->sb_geom[sb_addr].block_is_allowed[blk_it]) {
            const int32_t first = (int32_t)blk_it + (int32_t)ns_blk_offset[(int32_t)part], count = (int32_t)ns_blk_num[(int32_t)part];
            for (int32_t d1 = first; d1 < first + count; d1++) {
                const BlockGeom *g = get_blk_geom_mds((uint32_t)d1);
                BlkStruct *blk = &md->md_blk_arr_nsq[d1];
                if (!sb_hoistable(blk, g)) continue;
                const uint32_t org_x = sb_origin_x + g->origin_x, org_y = sb_origin_y + g->origin_y;
                blk->mds_idx = (uint16_t)d1;   /* the loop sets it for the first block of a partition only (:2274); the predictor looks the geometry up by it */
                sb_predict(scs, pcs, sb, ctx, blk, g, org_x, org_y, recon, is_16bit);
                tls_sb.predicted[d1] = 1;   /* predicted: the loop must not predict again, whether or not the transforms fitted the batch */
                nblk++;
                (void)sb_gather_block(&n_rec, &n_coeff, ctx, blk, g, sb_origin_x, sb_origin_y, org_x, org_y, recon, is_16bit);
            }
            blk_it += ns_depth_offset[sb128][g0->depth];
        } else
            blk_it += d1_depth_offset[sb128][g0->depth];
    }
    if (!nblk) return 0;
    __sync_fetch_and_add(&g_sb_superblocks, 1);
    __sync_fetch_and_add(&g_sb_blocks, nblk);
    tls_sb.valid = 1; tls_sb.n = 0;
    if (n_rec) {
        const int rc = sb_launch(n_rec, n_coeff, ctx->md_context->pf_ctrls.pf_shape);
        svt_hip_hooks_count(SVT_HIP_HOOK_ENCDEC_SB, rc == SVT_HIP_OK);
        if (rc == SVT_HIP_OK) { tls_sb.n = n_rec; __sync_fetch_and_add(&g_sb_launches, 1); }   /* not handled: the predictions stand, the transforms run on the host */
    }
    return 1;
}
/* the block loop, in front of the prediction of an inter block: 1 = it is in the reconstruction buffer already */
int svt_hip_hook_encdec_sb_predicted(const BlkStruct *blk) { return tls_sb.valid && blk->mds_idx < SB_MAX_BLK && tls_sb.predicted[blk->mds_idx]; }
void svt_hip_hook_encdec_sb_end(void) { tls_sb.valid = 0; tls_sb.cur = -1; }
static int sb_begin_block(const BlkStruct *blk) {
    tls_sb.cur = -1;
    if (!tls_sb.valid || !tls_sb.n || blk->mds_idx >= SB_MAX_BLK || !tls_sb.predicted[blk->mds_idx]) return 0;
    tls_sb.cur = blk->mds_idx;
    tls_sb.lo = 0;
    while (tls_sb.lo < tls_sb.n && tls_sb.e[tls_sb.lo].blk != tls_sb.cur) tls_sb.lo++;
    tls_sb.hi = tls_sb.lo;
    while (tls_sb.hi < tls_sb.n && tls_sb.e[tls_sb.hi].blk == tls_sb.cur) tls_sb.hi++;
    return tls_sb.hi > tls_sb.lo;
}
static void sb_end_block(void) { tls_sb.cur = -1; }
static int sb_fetch(int plane, int txb, int tx_size, int tx_type, int32_t *coeff) {
    if (tls_sb.cur < 0) return 0;
    for (int i = tls_sb.lo; i < tls_sb.hi; i++) {
        const EdEntry *e = &tls_sb.e[i];
        if (e->plane == plane && e->txb == txb && e->tx_size == tx_size && e->tx_type == tx_type) {
            memcpy(coeff, tls_sb.coeff + e->off, sizeof(int32_t) * (size_t)e->count);
            __sync_fetch_and_add(&g_sb_tx, 1);
            return 1;
        }
    }
    return 0;
}

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
static int pre_grid_fetch(const MV *mv, unsigned int *err, unsigned int *sse);
int svt_hip_hook_md_subpel_fetch(const MV *mv, unsigned int *err, unsigned int *sse) {
    if (pre_grid_fetch(mv, err, sse)) return 1;   /* hook "md_pre": the picture's sub-pel grid */
    if (!tls_sp.valid) return 0;
    for (int i = 0; i < tls_sp.n; i++)
        if (tls_sp.mv[i].row == mv->row && tls_sp.mv[i].col == mv->col) { *err = tls_sp.var[i]; *sse = tls_sp.sse[i]; return 1; }
    return 0;
}
void svt_hip_hook_md_subpel_end(void) { tls_sp.valid = 0; }

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * Mode decision, hook "md_pre": the stage-0 distortions of a whole PICTURE in one launch, made before the picture's mode decision starts.
 *
 * md_stage_0 (EbProductCodingLoop.c:1461) calls fast_loop_core (:907) for every candidate of every block: predict, measure the luma distortion against the source
 * (svt_nxm_sad_kernel_sub_sampled, :953), add the rate.  The candidates the open-loop ME contributes are known before the picture's first block: in the first
 * partitioning pass (PD_PASS_0) at presets above M4 the ME vectors enter stage 0 unrefined (read_refine_me_mvs, :2159, with md_sq / md_nsq / pme / sub-pel levels 0,
 * EbEncDecProcess.c:3050-3093) — full-pel, single reference, plain translation — and a full-pel prediction is a copy of the reference block.  Neither the source,
 * nor the reference pictures (complete: the picture manager starts a picture only when its references are), nor the ME vectors depend on a neighbouring block,
 * so svt_hip_hook_md_pre_picture — called once per picture by mode_decision_configuration_kernel, right before it posts the picture to the mode-decision threads
 * (EbModeDecisionConfigurationProcess.c:1058) — computes the SAD of every (superblock, square PU, reference picture) in ONE launch
 * (svt_hip_md_fullpel_sad_picture_dev) into a table in page-locked host memory.  NOBODY WAITS for it: the configuration thread queues the upload of the vectors, the
 * launches and the downloads of the tables on a context of the hook's own (g_pre_ctx, under g_pre_issue_mu) and returns; behind the last download a sequence word
 * travels to the device and back into the slot's page-locked flags[1].  A table is READY when flags[1] equals the slot's sequence number (pre_done) -- the copies of
 * sad / bisad / grid were queued before the word on the same stream, so they have landed when it has, and the mode-decision threads see them through the same
 * page-locked mapping (the word is read with acquire semantics).  A block that asks before that is a MISS and runs the reference's own code: early blocks of a picture
 * may miss, no block ever waits for the device.  fast_loop_core asks svt_hip_hook_md_pre_lookup: a candidate whose (PU, reference, vector) is in a ready table takes its
 * distortion from there and is NOT predicted.  The rate, the candidate order, every decision stay the reference's.
 *
 * The prediction samples of such a candidate are made later, and only if somebody reads them: full_loop_core (:5820) reuses stage 0's prediction when
 * md_staging_perform_inter_pred is off (always in PD_PASS_0, whose MD_STAGING_MODE_0 has no later prediction; stages 1 / 2 of the other modes) — the patched
 * full_loop_core asks svt_hip_hook_md_pre_take(candidate_buffer) and, if the buffer still carries the mark, runs the reference's own predictor with stage 0's settings
 * right there (the survivors of stage 0: one or two per block instead of every candidate).  The mark is a thread-local direct-mapped set of buffer addresses, cleared
 * whenever fast_loop_core sees the buffer again; a collision is a table miss.  When md_staging_perform_inter_pred is ON the later stage predicts the candidate itself:
 * svt_hip_hook_md_pre_take then only drops the mark and reports "nothing to do".  What the reference's predictor leaves behind besides samples — the warped-motion sample count
 * of the candidate, which the fast cost reads (wm_count_samples, EbEncInterPrediction.c:6285-6297), and ifs_is_regular_last — the patched fast_loop_core does itself on a hit.
 * SVT_HIP_MD_PRE_VERIFY=1 (tests) makes every hit compute the reference's value as well and counts disagreements (svt_hip_hook_md_pre_verify).
 *
 * Exact by construction: the table is keyed by what fast_loop_core is about to compute — block position and size, reference picture, vector — and holds the value
 * the reference's kernels give for it (tests: device vs oracle vs the reference's svt_nxm_sad_kernel / convolve copy; end-to-end bitstream identity).  A miss — sub-pel or
 * the compound types other than the plain average of two ME vectors (svt_hip_md_fullpel_avg_sad_picture_dev serves that one), non-translation candidates, vectors the MV clamp
 * of av1_inter_prediction would move, 128x128 superblocks, scaled references, later passes' refined vectors — is the reference's own code.  A 10-bit encode's fast loop works on 16-bit
 * samples: its tables are made on 16-bit planes (the _hbd_ entry points). */
#include "EbCodingLoop.h"   /* me_idx[]: mode-decision block index -> PU index of the open-loop ME results */
#include "EbModeDecisionProcess.h"
#include "EbMotionEstimationLcuResults.h"
#include <pthread.h>

#define PRE_SLOTS 64
#define PRE_PUS SVT_HIP_SQUARE_PU_COUNT
typedef struct {
    PictureControlSet *pcs;
    uint64_t  picture_number;
    volatile int ready;
    int       n_sb, n_refs;
    int8_t    slot_of[2][4];     /* (list, reference index) -> column of the table, -1 = none */
    uint32_t *mv;                /* [n_sb][85][n_refs]: the vector in the units of the candidates (1/8 sample), x | y << 16; PRE_NONE = no entry */
    uint32_t *sad;               /* [n_sb][85][n_refs], page-locked */
    size_t    cap;               /* entries allocated */
    int       hbd;               /* the two distortion tables were made on the 16-bit planes (a 10-bit encode: mode decision's fast loop works on them); the sub-pel grid is 8-bit either way */
    void     *d_src16;           /* the picture's packed 16-bit luma (svt_enc_msb_pack2_d of the 8-bit plane and the 2-bit plane), made per picture */
    size_t    src16_cap;
    int       n_pairs;           /* compound-average candidates: pairs of table columns that occur as bi-directional ME candidates in this picture */
    uint8_t   pairs[SVT_HIP_MD_MAX_PAIRS][2];
    int8_t    pair_of[SVT_HIP_MD_MAX_REFS][SVT_HIP_MD_MAX_REFS];   /* (column of the first reference, of the second) -> index into pairs, -1 = none */
    uint32_t *bisad;             /* [n_sb][85][n_pairs], page-locked */
    void     *d_bisad;
    size_t    bicap;             /* entries allocated */
    uint32_t *grid;              /* [n_sb][85][n_refs][49][2] (variance, sse) of the sub-pel refinement's probes around the vector, page-locked; NULL: not made */
    size_t    grid_cap;          /* entries (of 98 words) allocated */
    int       grid_bank, grid_ready;
    /* the slot's device side and its completion mark: everything is queued, nobody waits */
    void     *d_mv, *d_sad, *d_grid, *d_seq;
    uint32_t *h_dev_mv;          /* page-locked staging of the device-side vectors */
    volatile uint32_t *flags;    /* page-locked: [0] = the sequence number on its way to the device, [1] = where it comes back once both tables have arrived */
    uint32_t  seq;
    int       pending, n_held;
    const void *held[2 * SVT_HIP_MD_MAX_REFS + 2];   /* resident planes (svt_hip_resident_acquire) the queued launches read: released once the mark is back */
} MdPre;
#define PRE_NONE 0x80008000u
static MdPre           g_pre[PRE_SLOTS];
static pthread_mutex_t g_pre_mu = PTHREAD_MUTEX_INITIALIZER;
static SvtHipMdPu      g_pre_pu[PRE_PUS];
static int             g_pre_pu_ok;   /* 0 not built, 1 built, -1 the tables are not what this code expects */
static long g_pre_pictures, g_pre_launches, g_pre_jobs, g_pre_min_jobs, g_pre_declined;
static long g_pre_bi_pictures;     /* pictures with a pair table */
static long g_pre_grid_pictures;   /* the sub-pel grid: pictures it was made for */
/* The counters of the HOT path -- svt_hip_hook_md_pre_lookup runs once per fast_loop_core candidate on every mode-decision thread, the grid fetch once per sub-pel probe --
 * are striped: a thread counts in the cache line of its own stripe (threads beyond PRE_STRIPES share one: the adds stay atomic), the report sums the stripes.  One shared
 * line per counter made 100+ MD threads take turns on it in the encoder's hottest loop. */
enum { PC_CALLS, PC_INTER, PC_HITS, PC_BI_HITS, PC_LATE, PC_PROBES, PC_PROBE_HITS, PC_MISS0 };
#define PRE_STRIPES 128
#define PC_N (PC_MISS0 + 10 /* PRE_MISS_N, checked below */)
static struct { long c[PC_N]; char pad[(64 - (PC_N * sizeof(long)) % 64) % 64]; } g_pre_stripe[PRE_STRIPES] __attribute__((aligned(64)));
static int g_pre_stripe_next;
static __thread int tls_stripe = -1;
static inline long *pre_counters(void) {
    if (tls_stripe < 0) tls_stripe = __sync_fetch_and_add(&g_pre_stripe_next, 1) % PRE_STRIPES;
    return g_pre_stripe[tls_stripe].c;
}
#define PRE_COUNT(which) __sync_fetch_and_add(&pre_counters()[which], 1)
static long pre_total(int which) { long t = 0; for (int i = 0; i < PRE_STRIPES; i++) t += g_pre_stripe[i].c[which]; return t; }
static int  g_pre_grid_on = -1, g_pre_compound_on = -1, g_pre_grid_mb = -1;   /* SVT_HIP_MD_PRE_SUBPEL=0 / SVT_HIP_MD_PRE_COMPOUND=0 leave the table out */
static __thread struct { const uint32_t *row; int cx, cy; } tls_grid;
static long long g_pre_ns;
static long g_pre_mismatch;
static int  g_pre_timing = -1;   /* SVT_HIP_MD_PRE_TIMING=1: measure the device time per picture (makes the configuration thread wait: a diagnostic, not the product's mode) */
static long g_pre_device_us;
double svt_hip_hook_md_pre_device_ms(void) { return g_pre_timing > 0 ? g_pre_device_us / 1000.0 : -1.0; }
/* why an inter candidate of fast_loop_core was not served from the table */
enum { PRE_MISS_COMPOUND, PRE_MISS_MOTION, PRE_MISS_HBD, PRE_MISS_LATER_PASS, PRE_MISS_SHAPE, PRE_MISS_NO_TABLE, PRE_MISS_REFERENCE, PRE_MISS_VECTOR, PRE_MISS_BORDER, PRE_MISS_MARK, PRE_MISS_N };
typedef char pre_miss_fits_the_stripe[PRE_MISS_N <= PC_N - PC_MISS0 ? 1 : -1];
#define PRE_MISS(why) do { PRE_COUNT(PC_MISS0 + (why)); return 0; } while (0)
void svt_hip_hook_md_pre_misses(long *out, int n) { for (int i = 0; i < n; i++) out[i] = i < PRE_MISS_N ? pre_total(PC_MISS0 + i) : 0; }
static int  g_pre_verify = -1;
static __thread struct { PictureControlSet *pcs; uint64_t pic; MdPre *t; } tls_pre;
#define PRE_MARKS 1024
static __thread const void *tls_mark[PRE_MARKS];
static inline unsigned mark_slot(const void *p) { const uintptr_t a = (uintptr_t)p; return (unsigned)((a >> 6) ^ (a >> 16)) & (PRE_MARKS - 1); }

long svt_hip_hook_md_pre_mismatches(void) { return g_pre_verify > 0 ? g_pre_mismatch : -1; }
void svt_hip_hook_md_pre_compound_stats(long *pictures, long *served) { *pictures = g_pre_bi_pictures; *served = pre_total(PC_BI_HITS); }
void svt_hip_hook_md_pre_subpel_stats(long *pictures, long *probes, long *served) { *pictures = g_pre_grid_pictures; *probes = pre_total(PC_PROBES); *served = pre_total(PC_PROBE_HITS); }
void svt_hip_hook_md_pre_stats(long *pictures, long *launches, long *jobs, long *min_jobs, long *calls, long *inter, long *hits, long *late, long *declined, double *ms) {
    *pictures = g_pre_pictures; *launches = g_pre_launches; *jobs = g_pre_jobs; *min_jobs = g_pre_min_jobs; *calls = pre_total(PC_CALLS); *inter = pre_total(PC_INTER); *hits = pre_total(PC_HITS);
    *late = pre_total(PC_LATE); *declined = g_pre_declined; *ms = g_pre_ns / 1e6;
}

/* PU index of the ME results -> position and size inside a 64x64 superblock, from the reference's own tables (me_idx over the block geometry) */
static void pre_build_pus_once(void);
static pthread_once_t g_pre_pu_once = PTHREAD_ONCE_INIT;
static int pre_build_pus(void) {   /* several configuration threads arrive here together: the table is built by exactly one of them (ThreadSanitizer, round 5) */
    pthread_once(&g_pre_pu_once, pre_build_pus_once);
    return g_pre_pu_ok > 0;
}
static void pre_build_pus_once(void) {
    int seen[PRE_PUS] = {0}, n = 0;
    for (uint32_t i = 0; i < BLOCK_MAX_COUNT_SB_64; i++) {
        const BlockGeom *g = get_blk_geom_mds(i);
        if (g->shape != PART_N || g->bwidth != g->bheight || g->bwidth < 8 || g->bwidth > 64) continue;
        const uint32_t pu = me_idx[i];
        if (pu >= PRE_PUS || seen[pu]) { g_pre_pu_ok = -1; return; }
        seen[pu] = 1; n++;
        g_pre_pu[pu].x = (uint8_t)g->origin_x; g_pre_pu[pu].y = (uint8_t)g->origin_y; g_pre_pu[pu].w = g_pre_pu[pu].h = (uint8_t)g->bwidth;
    }
    g_pre_pu_ok = n == PRE_PUS ? 1 : -1;
}

/* the luma plane of `pic` on the device: its resident copy (announced by the writer: *from_table = 1, release it afterwards) or an upload of the whole padded plane */
static const uint8_t *pre_plane_n(SvtHipCtx *hip, const EbPictureBufferDesc *pic, int pix_bytes, int *from_table, void **tmp);
static const uint8_t *pre_plane(SvtHipCtx *hip, const EbPictureBufferDesc *pic, int *from_table, void **tmp) { return pre_plane_n(hip, pic, 1, from_table, tmp); }
static const uint8_t *pre_plane_n(SvtHipCtx *hip, const EbPictureBufferDesc *pic, int pix_bytes, int *from_table, void **tmp) {
    const size_t bytes = (size_t)pic->stride_y * (size_t)(pic->height + 2 * pic->origin_y) * (size_t)pix_bytes;
    *from_table = 0; *tmp = NULL;
    const void *d = svt_hip_resident_acquire(hip, pic->buffer_y, bytes);
    if (d) { *from_table = 1; return (const uint8_t *)d; }
    if (svt_hip_hooks_malloc(hip, tmp, bytes) != SVT_HIP_OK) { *tmp = NULL; return NULL; }
    if (svt_hip_memcpy_h2d_async(hip, *tmp, pic->buffer_y, bytes) != SVT_HIP_OK) return NULL;
    return (const uint8_t *)*tmp;
}

/* rest_kernel, after pad_ref_and_set_flags (EbRestProcess.c:581): the picture's 8-bit reference planes are final, padding included — the next pictures' tables read the
 * luma plane, so its device copy is made once per reference picture (the resident table), not once per picture that references it */
void svt_hip_hook_md_pre_note_ref(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_MD_PRE) || pcs->parent_pcs_ptr->is_used_as_reference_flag != EB_TRUE || !pcs->parent_pcs_ptr->reference_picture_wrapper_ptr) return;
    const EbReferenceObject *ro = (const EbReferenceObject *)pcs->parent_pcs_ptr->reference_picture_wrapper_ptr->object_ptr;
    if (ro && ro->reference_picture) svt_hip_hooks_resident_note_picture(ro->reference_picture);
    const EbPictureBufferDesc *r16 = ro ? ro->reference_picture16bit : NULL;   /* a 10-bit encode: the planes its mode decision predicts from */
    if (r16 && r16->buffer_y) svt_hip_resident_note(r16->buffer_y, (size_t)r16->stride_y * (size_t)(r16->height + 2 * r16->origin_y) * 2);
}

/* the hook's own context (a stream of its own: its work overlaps everything else and is never drained by another bridge's unlock), made at the first picture */
static SvtHipCtx      *g_pre_ctx;
static pthread_mutex_t g_pre_issue_mu = PTHREAD_MUTEX_INITIALIZER;
static uint32_t        g_pre_seq;
static SvtHipCtx *pre_context(void) {   /* g_pre_issue_mu held */
    if (!g_pre_ctx && svt_hip_init(svt_hip_hooks_device(), &g_pre_ctx) != SVT_HIP_OK) g_pre_ctx = NULL;
    return g_pre_ctx;
}
static int pre_done(const MdPre *t) { return t->flags && t->flags[1] == t->seq; }
/* resident planes of pictures whose queue has run out go back to the table (all = 1: the context has just been drained, every slot is complete) */
static void pre_sweep(SvtHipCtx *hip, int all) {   /* g_pre_issue_mu held */
    (void)hip;
    for (int i = 0; i < PRE_SLOTS; i++) {
        MdPre *t = &g_pre[i];
        if (!t->pending || !(all || pre_done(t))) continue;
        for (int k = 0; k < t->n_held; k++) svt_hip_resident_release(t->held[k]);
        t->n_held = 0; t->pending = 0;
    }
}

void svt_hip_hook_md_pre_picture(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_MD_PRE)) return;
    const long long t0 = svt_hip_hooks_now_ns();
    PictureParentControlSet *ppcs = pcs->parent_pcs_ptr;
    const SequenceControlSet *scs = (const SequenceControlSet *)pcs->scs_wrapper_ptr->object_ptr;
    /* the picture's slot: found by its control set (a pool object: the table of the picture that used it before is dead by now) */
    MdPre *t = NULL;
    pthread_mutex_lock(&g_pre_mu);
    for (int i = 0; i < PRE_SLOTS && !t; i++) if (g_pre[i].pcs == pcs) t = &g_pre[i];
    for (int i = 0; i < PRE_SLOTS && !t; i++) if (!g_pre[i].pcs) { t = &g_pre[i]; t->pcs = pcs; }
    pthread_mutex_unlock(&g_pre_mu);
    if (!t) { __sync_fetch_and_add(&g_pre_declined, 1); return; }
    t->ready = 0;
    __sync_synchronize();
    t->picture_number = pcs->picture_number;
    const EbPictureBufferDesc *in = ppcs->enhanced_picture_ptr;
    int ok = pcs->slice_type != I_SLICE && scs->seq_header.sb_size == BLOCK_64X64 && !ppcs->frame_superres_enabled && in && in->buffer_y && ppcs->pa_me_data &&
             ppcs->max_number_of_pus_per_sb >= PRE_PUS && pre_build_pus();
    /* the reference pictures ME searched, in the order of the ME vector array: list 0 at [0, 4), list 1 at [4, 7) (EbMotionEstimationLcuResults.h:44) */
    const EbPictureBufferDesc *ref_pic[SVT_HIP_MD_MAX_REFS], *ref_pic16[SVT_HIP_MD_MAX_REFS];
    int ref_col[SVT_HIP_MD_MAX_REFS], n_refs = 0;
    /* a 10-bit encode decides on 16-bit samples in the fast loop (hbd_mode_decision 1, and 2 = "dual": only the searches are 8-bit): the two distortion tables are then made
     * from the packed source and reference_picture16bit; without the unpacked 2-bit plane or a 16-bit reference the tables are not made (the reference's path) */
    int hbd = pcs->hbd_mode_decision != 0;
    if (hbd && !(scs->static_config.encoder_bit_depth == EB_10BIT && in && in->buffer_bit_inc_y && in->stride_bit_inc_y)) ok = 0;
    memset(t->slot_of, -1, sizeof(t->slot_of));
    for (int li = 0; li < 2 && ok; li++) {
        const int cnt = li ? ppcs->ref_list1_count_try : ppcs->ref_list0_count_try;
        for (int ri = 0; ri < cnt && ri < (li ? 3 : 4) && ok; ri++) {
            const EbObjectWrapper *w = pcs->ref_pic_ptr_array[li][ri];
            const EbReferenceObject *ro = w ? (const EbReferenceObject *)w->object_ptr : NULL;
            const EbPictureBufferDesc *rp = ro ? ro->reference_picture : NULL;
            if (!rp || !rp->buffer_y || rp->width != in->width || rp->height != in->height) { ok = 0; break; }   /* scaled references keep the reference's path */
            const EbPictureBufferDesc *rp16 = ro->reference_picture16bit;
            if (hbd && (!rp16 || !rp16->buffer_y || rp16->width != rp->width || rp16->height != rp->height || rp16->origin_x != rp->origin_x || rp16->origin_y != rp->origin_y)) { ok = 0; break; }
            t->slot_of[li][ri] = (int8_t)n_refs; ref_pic[n_refs] = rp; ref_pic16[n_refs] = rp16; ref_col[n_refs] = li * 4 + ri; n_refs++;
        }
    }
    if (!ok || !n_refs) { if (pcs->slice_type != I_SLICE) __sync_fetch_and_add(&g_pre_declined, 1); return; }
    const int sb_cols = (ppcs->aligned_width + 63) / 64, n_sb = pcs->sb_total_count;
    const size_t n = (size_t)n_sb * PRE_PUS * (size_t)n_refs;
    /* Everything below is QUEUED on the hook's own context and the configuration thread goes on: no block of any picture ever waits for the device.  The last operation
     * of the queue copies the picture's sequence number into the table's page-locked `done` word (stream order: after both tables have arrived); a lookup that does not
     * find it there yet is a table miss and the reference's own code runs — so the first blocks of a picture that starts immediately may go unserved, nothing else. */
    pthread_mutex_lock(&g_pre_issue_mu);
    SvtHipCtx *hip = pre_context();
    if (!hip) { pthread_mutex_unlock(&g_pre_issue_mu); __sync_fetch_and_add(&g_pre_declined, 1); return; }
    pre_sweep(hip, 0);
    if (t->pending) { (void)svt_hip_sync(hip); pre_sweep(hip, 0); }   /* cannot be: the picture that used the slot has left mode decision long ago */
    int rc = SVT_HIP_OK;
    if (g_pre_grid_on < 0) g_pre_grid_on = !(getenv("SVT_HIP_MD_PRE_SUBPEL") && !atoi(getenv("SVT_HIP_MD_PRE_SUBPEL")));
    const size_t gwords = n * 2 * SVT_HIP_MD_GRID;
    if (g_pre_grid_mb < 0) g_pre_grid_mb = getenv("SVT_HIP_MD_PRE_GRID_MB") ? atoi(getenv("SVT_HIP_MD_PRE_GRID_MB")) : 256;
    const int want_grid = g_pre_grid_on && gwords * sizeof(uint32_t) <= ((size_t)g_pre_grid_mb << 20);   /* pictures whose grid would exceed the cap (SVT_HIP_MD_PRE_GRID_MB, default 256) go without: the table crosses PCIe */
    if (t->cap < n) {   /* the slot's buffers grow with the largest picture it has seen: no allocation per picture */
        if (t->sad) svt_hip_host_free(hip, t->sad);
        if (t->h_dev_mv) svt_hip_host_free(hip, t->h_dev_mv);
        svt_hip_free(hip, t->d_mv); svt_hip_free(hip, t->d_sad);
        free(t->mv);
        t->sad = t->h_dev_mv = NULL; t->d_mv = t->d_sad = NULL; t->cap = 0;
        t->mv = (uint32_t *)malloc(n * sizeof(uint32_t));
        void *h = NULL, *h2 = NULL;
        rc = t->mv ? svt_hip_host_alloc(hip, &h, n * sizeof(uint32_t)) : SVT_HIP_ERR_RUNTIME;
        if (rc == SVT_HIP_OK) rc = svt_hip_host_alloc(hip, &h2, n * sizeof(uint32_t));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &t->d_mv, n * sizeof(uint32_t));
        if (rc == SVT_HIP_OK) rc = svt_hip_malloc(hip, &t->d_sad, n * sizeof(uint32_t));
        t->sad = (uint32_t *)h; t->h_dev_mv = (uint32_t *)h2;
        if (rc == SVT_HIP_OK) t->cap = n;
    }
    if (rc == SVT_HIP_OK && !t->flags) {
        void *h = NULL;
        rc = svt_hip_host_alloc(hip, &h, 2 * sizeof(uint32_t));
        if (rc == SVT_HIP_OK) { t->flags = (volatile uint32_t *)h; t->flags[0] = t->flags[1] = 0; rc = svt_hip_malloc(hip, &t->d_seq, sizeof(uint32_t)); }
    }
    if (rc == SVT_HIP_OK && want_grid && t->grid_cap < n) {
        if (t->grid) svt_hip_host_free(hip, t->grid);
        svt_hip_free(hip, t->d_grid);
        t->grid = NULL; t->d_grid = NULL; t->grid_cap = 0;
        void *h = NULL;
        if (svt_hip_host_alloc(hip, &h, gwords * sizeof(uint32_t)) == SVT_HIP_OK && svt_hip_malloc(hip, &t->d_grid, gwords * sizeof(uint32_t)) == SVT_HIP_OK) { t->grid = (uint32_t *)h; t->grid_cap = n; }
        else if (h) svt_hip_host_free(hip, h);
    }
    /* the vectors: table side in the candidates' units (1/8 sample), device side in whole samples */
    if (rc == SVT_HIP_OK)
        for (int sb = 0; sb < n_sb; sb++) {
            const MeSbResults *mr = ppcs->pa_me_data->me_results[sb];
            for (int pu = 0; pu < PRE_PUS; pu++)
                for (int r = 0; r < n_refs; r++) {
                    const MvCandidate m = mr->me_mv_array[pu * MAX_PA_ME_MV + ref_col[r]];
                    const size_t e = ((size_t)sb * PRE_PUS + pu) * n_refs + r;
                    if ((m.x_mv | m.y_mv) & 3) { t->mv[e] = PRE_NONE; t->h_dev_mv[e] = (uint32_t)(uint16_t)SVT_HIP_MD_NO_MV; continue; }   /* the open-loop search is full-pel */
                    t->mv[e] = (uint32_t)(uint16_t)(m.x_mv * 2) | (uint32_t)(uint16_t)(m.y_mv * 2) << 16;
                    t->h_dev_mv[e] = (uint32_t)(uint16_t)(m.x_mv >> 2) | (uint32_t)(uint16_t)(m.y_mv >> 2) << 16;
                }
        }
    /* the pairs of references the open-loop ME proposes as bi-directional candidates (MeCandidate.direction == 2): mode decision turns each into a NEW_NEWMV candidate whose
     * two vectors are the two references' own ME vectors (EbModeDecision.c:3408-3540) */
    t->n_pairs = 0;
    memset(t->pair_of, -1, sizeof(t->pair_of));
    if (rc == SVT_HIP_OK && g_pre_compound_on < 0) g_pre_compound_on = !(getenv("SVT_HIP_MD_PRE_COMPOUND") && !atoi(getenv("SVT_HIP_MD_PRE_COMPOUND")));
    if (rc == SVT_HIP_OK && g_pre_compound_on > 0 && n_refs > 1)
        for (int sb = 0; sb < n_sb; sb++) {
            const MeSbResults *mr = ppcs->pa_me_data->me_results[sb];
            for (int pu = 0; pu < PRE_PUS; pu++) {
                const MeCandidate *mc = &mr->me_candidate_array[pu * MAX_PA_ME_CAND];
                for (int k = 0; k < mr->total_me_candidate_index[pu] && k < MAX_PA_ME_CAND; k++) {
                    if (mc[k].direction != 2 || mc[k].ref0_list > 1 || mc[k].ref1_list > 1) continue;
                    const int c0 = t->slot_of[mc[k].ref0_list][mc[k].ref_idx_l0], c1 = t->slot_of[mc[k].ref1_list][mc[k].ref_idx_l1];
                    if (c0 < 0 || c1 < 0 || t->pair_of[c0][c1] >= 0 || t->n_pairs >= SVT_HIP_MD_MAX_PAIRS) continue;
                    t->pair_of[c0][c1] = (int8_t)t->n_pairs; t->pairs[t->n_pairs][0] = (uint8_t)c0; t->pairs[t->n_pairs][1] = (uint8_t)c1; t->n_pairs++;
                }
            }
        }
    const size_t nbi = (size_t)n_sb * PRE_PUS * (size_t)t->n_pairs;
    if (rc == SVT_HIP_OK && nbi > t->bicap) {
        if (t->bisad) svt_hip_host_free(hip, t->bisad);
        svt_hip_free(hip, t->d_bisad);
        t->bisad = NULL; t->d_bisad = NULL; t->bicap = 0;
        void *h = NULL;
        if (svt_hip_host_alloc(hip, &h, nbi * sizeof(uint32_t)) == SVT_HIP_OK && svt_hip_malloc(hip, &t->d_bisad, nbi * sizeof(uint32_t)) == SVT_HIP_OK) { t->bisad = (uint32_t *)h; t->bicap = nbi; }
        else { if (h) svt_hip_host_free(hip, h); t->n_pairs = 0; }   /* the pair table is an extra: without it compound candidates stay the reference's */
    }
    void *tmp[2 * SVT_HIP_MD_MAX_REFS + 2] = {0};   /* [0] source, [1 ..] references, [MAX_REFS + 1] the 2-bit plane, [MAX_REFS + 2 ..] 16-bit references */
    int   from_table[2 * SVT_HIP_MD_MAX_REFS + 2] = {0}, any_tmp = 0;
    const uint8_t *d_src = NULL;
    SvtHipMdRefPlane planes[SVT_HIP_MD_MAX_REFS], planes16[SVT_HIP_MD_MAX_REFS];
    const int s16 = (int)ppcs->aligned_width + 64;   /* row pitch of the packed source (samples): the kernels read whole dwords past a row's last sample */
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d_async(hip, t->d_mv, t->h_dev_mv, n * sizeof(uint32_t));
    if (rc == SVT_HIP_OK) { d_src = pre_plane(hip, in, &from_table[0], &tmp[0]); if (!d_src) rc = SVT_HIP_ERR_RUNTIME; }
    for (int r = 0; r < n_refs && rc == SVT_HIP_OK; r++) {
        const EbPictureBufferDesc *rp = ref_pic[r];
        const uint8_t *d = pre_plane(hip, rp, &from_table[1 + r], &tmp[1 + r]);
        if (!d) { rc = SVT_HIP_ERR_RUNTIME; break; }
        planes[r].d_plane = d + (size_t)rp->origin_y * rp->stride_y + rp->origin_x;
        planes[r].stride = rp->stride_y;
        planes[r].x_min = -(int)rp->origin_x; planes[r].y_min = -(int)rp->origin_y;
        planes[r].x_max = (int)rp->width + (int)rp->origin_x; planes[r].y_max = (int)rp->height + (int)rp->origin_y;
    }
    if (rc == SVT_HIP_OK && hbd) {
        const size_t need = (size_t)s16 * ((size_t)ppcs->aligned_height + 1) * 2;
        if (t->src16_cap < need) {
            svt_hip_free(hip, t->d_src16); t->d_src16 = NULL; t->src16_cap = 0;
            if (svt_hip_malloc(hip, &t->d_src16, need) == SVT_HIP_OK) t->src16_cap = need; else rc = SVT_HIP_ERR_RUNTIME;
        }
        const size_t inc_bytes = (size_t)in->stride_bit_inc_y * (size_t)(in->height + 2 * in->origin_y);
        void **inc = &tmp[SVT_HIP_MD_MAX_REFS + 1];
        if (rc == SVT_HIP_OK && svt_hip_hooks_malloc(hip, inc, inc_bytes) != SVT_HIP_OK) { *inc = NULL; rc = SVT_HIP_ERR_RUNTIME; }
        if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_h2d_async(hip, *inc, in->buffer_bit_inc_y, inc_bytes);
        if (rc == SVT_HIP_OK)   /* pack2d_src of mode_decision_sb (EbProductCodingLoop.c:8131) for the whole picture: svt_enc_msb_pack2_d */
            rc = svt_hip_picture_format_dev(hip, 0, d_src + (size_t)in->origin_y * in->stride_y + in->origin_x, in->stride_y,
                                            (const uint8_t *)*inc + (size_t)in->origin_y * in->stride_bit_inc_y + in->origin_x, in->stride_bit_inc_y, t->d_src16, s16, NULL, 0,
                                            (int)ppcs->aligned_width, (int)ppcs->aligned_height);
        for (int r = 0; r < n_refs && rc == SVT_HIP_OK; r++) {
            const EbPictureBufferDesc *rp = ref_pic16[r];
            const uint8_t *d = pre_plane_n(hip, rp, 2, &from_table[SVT_HIP_MD_MAX_REFS + 2 + r], &tmp[SVT_HIP_MD_MAX_REFS + 2 + r]);
            if (!d) { rc = SVT_HIP_ERR_RUNTIME; break; }
            planes16[r] = planes[r];
            planes16[r].d_plane = d + 2 * ((size_t)rp->origin_y * rp->stride_y + rp->origin_x);
            planes16[r].stride = rp->stride_y;
        }
    }
    t->hbd = hbd;
    for (int i = 0; i < 2 * SVT_HIP_MD_MAX_REFS + 2; i++) any_tmp |= tmp[i] != NULL;
    if (g_pre_timing < 0) g_pre_timing = getenv("SVT_HIP_MD_PRE_TIMING") && atoi(getenv("SVT_HIP_MD_PRE_TIMING"));
    const int timing = g_pre_timing > 0 && rc == SVT_HIP_OK && svt_hip_timer_start(hip) == SVT_HIP_OK;   /* diagnostic: the device time of this picture's launches and copies (waits) */
    if (rc == SVT_HIP_OK)
        rc = hbd ? svt_hip_md_fullpel_sad_picture_hbd_dev(hip, (const uint16_t *)t->d_src16, s16, ppcs->aligned_width, ppcs->aligned_height, sb_cols, n_sb, PRE_PUS, g_pre_pu, n_refs, planes16,
                                                          (const uint32_t *)t->d_mv, (uint32_t *)t->d_sad)
                 : svt_hip_md_fullpel_sad_picture_dev(hip, d_src + (size_t)in->origin_y * in->stride_y + in->origin_x, in->stride_y, ppcs->aligned_width, ppcs->aligned_height, sb_cols,
                                                      n_sb, PRE_PUS, g_pre_pu, n_refs, planes, (const uint32_t *)t->d_mv, (uint32_t *)t->d_sad);
    if (rc == SVT_HIP_OK && t->n_pairs) {
        int brc = hbd ? svt_hip_md_fullpel_avg_sad_picture_hbd_dev(hip, (const uint16_t *)t->d_src16, s16, ppcs->aligned_width, ppcs->aligned_height, sb_cols, n_sb, PRE_PUS, g_pre_pu, n_refs,
                                                                   planes16, (const uint32_t *)t->d_mv, t->n_pairs, (const uint8_t(*)[2])t->pairs, (uint32_t *)t->d_bisad)
                      : svt_hip_md_fullpel_avg_sad_picture_dev(hip, d_src + (size_t)in->origin_y * in->stride_y + in->origin_x, in->stride_y, ppcs->aligned_width, ppcs->aligned_height, sb_cols,
                                                               n_sb, PRE_PUS, g_pre_pu, n_refs, planes, (const uint32_t *)t->d_mv, t->n_pairs, (const uint8_t(*)[2])t->pairs, (uint32_t *)t->d_bisad);
        if (brc == SVT_HIP_OK) brc = svt_hip_memcpy_d2h_async(hip, t->bisad, t->d_bisad, nbi * sizeof(uint32_t));
        if (brc != SVT_HIP_OK) t->n_pairs = 0;
    }
    /* ... and the sub-pel refinement's probes (md_subpel_search, :2063): (variance, sse) of the 7 x 7 quarter-pel grid around the same vectors, with the interpolation
     * kernels the picture's final pass searches with (md_subpel_me_level, EbEncDecProcess.c:3088-3097: USE_8_TAPS up to M4, USE_4_TAPS above).  SVT_HIP_MD_PRE_SUBPEL=0
     * leaves it out. */
    t->grid_ready = 0;
    if (rc == SVT_HIP_OK && want_grid && t->grid) {
        t->grid_bank = pcs->enc_mode <= ENC_M4 ? 0 : 4;
        int grc = svt_hip_md_subpel_grid_picture_dev(hip, d_src + (size_t)in->origin_y * in->stride_y + in->origin_x, in->stride_y, ppcs->aligned_width, ppcs->aligned_height, sb_cols,
                                                     n_sb, PRE_PUS, g_pre_pu, n_refs, planes, (const uint32_t *)t->d_mv, t->grid_bank, (uint32_t *)t->d_grid);
        if (grc == SVT_HIP_OK) grc = svt_hip_memcpy_d2h_async(hip, t->grid, t->d_grid, gwords * sizeof(uint32_t));
        t->grid_ready = grc == SVT_HIP_OK;   /* the grid is an extra: without it the sub-pel probes stay the reference's */
    }
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h_async(hip, t->sad, t->d_sad, n * sizeof(uint32_t));
    if (rc == SVT_HIP_OK) {   /* the completion mark: the sequence number travels to the device and back behind everything else of this picture */
        t->seq = ++g_pre_seq ? g_pre_seq : ++g_pre_seq;
        t->flags[0] = t->seq;
        rc = svt_hip_memcpy_h2d_async(hip, t->d_seq, (const void *)&t->flags[0], sizeof(uint32_t));
        if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h_async(hip, (void *)&t->flags[1], t->d_seq, sizeof(uint32_t));
    }
    if (timing) {
        float ms = 0;
        if (svt_hip_timer_stop_ms(hip, &ms) == SVT_HIP_OK) __sync_fetch_and_add(&g_pre_device_us, (long)(ms * 1000.0f));
    }
    t->n_held = 0;
    if (from_table[0]) t->held[t->n_held++] = in->buffer_y;
    for (int r = 0; r < n_refs; r++) if (from_table[1 + r]) t->held[t->n_held++] = ref_pic[r]->buffer_y;
    for (int r = 0; r < n_refs; r++) if (hbd && from_table[SVT_HIP_MD_MAX_REFS + 2 + r]) t->held[t->n_held++] = ref_pic16[r]->buffer_y;
    t->pending = 1;
    if (rc != SVT_HIP_OK || any_tmp) {   /* a failure, or planes that had to be uploaded for this picture alone (resident planes off / over budget): finish here */
        (void)svt_hip_sync(hip);
        for (int i = 0; i < 2 * SVT_HIP_MD_MAX_REFS + 2; i++) svt_hip_hooks_free(hip, tmp[i]);
        pre_sweep(hip, 1);
    }
    if (rc != SVT_HIP_OK) { t->grid_ready = 0; t->n_pairs = 0; }
    svt_hip_hooks_count(SVT_HIP_HOOK_MD_PRE, rc == SVT_HIP_OK);
    if (rc == SVT_HIP_OK) {
        t->n_sb = n_sb; t->n_refs = n_refs;
        __sync_synchronize();
        t->ready = 1;
        __sync_fetch_and_add(&g_pre_pictures, 1); __sync_fetch_and_add(&g_pre_launches, 1 + t->grid_ready + (t->n_pairs > 0)); __sync_fetch_and_add(&g_pre_jobs, (long)n);
        if (t->n_pairs) __sync_fetch_and_add(&g_pre_bi_pictures, 1);
        if (t->grid_ready) __sync_fetch_and_add(&g_pre_grid_pictures, 1);
        if (!g_pre_min_jobs || (long)n < g_pre_min_jobs) g_pre_min_jobs = (long)n;
    }
    pthread_mutex_unlock(&g_pre_issue_mu);
    __sync_fetch_and_add(&g_pre_ns, svt_hip_hooks_now_ns() - t0);
}

static const MdPre *pre_table_of(PictureControlSet *pcs) {
    if (tls_pre.pcs != pcs || tls_pre.pic != pcs->picture_number) {
        tls_pre.pcs = pcs; tls_pre.pic = pcs->picture_number; tls_pre.t = NULL;
        for (int i = 0; i < PRE_SLOTS; i++)
            if (g_pre[i].pcs == pcs) { if (g_pre[i].ready && g_pre[i].picture_number == pcs->picture_number) tls_pre.t = &g_pre[i]; break; }
    }
    return tls_pre.t && pre_done(tls_pre.t) ? tls_pre.t : NULL;   /* not arrived yet: a miss, the reference's code runs */
}
/* md_subpel_search (EbProductCodingLoop.c:2063), around svt_av1_find_best_sub_pixel_tree: the probes of this search — svt_upsampled_pref_error (mcomp.c:102) of vectors
 * within 6/8 sample of the start vector, on quarter-sample positions — are in the picture's grid when the block is a square PU of the open-loop ME, the search starts at
 * that PU's vector for this reference and filters with the kernels the grid was made with.  (mvx8, mvy8): the start vector in eighth-samples, full-pel. */
void svt_hip_hook_md_pre_subpel_begin(PictureControlSet *pcs, ModeDecisionContext *ctx, int list_idx, int ref_idx, int subpel_search_type, int mvx8, int mvy8) {
    tls_grid.row = NULL;
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_MD_PRE)) return;
    const MdPre *t = pre_table_of(pcs);
    if (!t || !t->grid_ready || list_idx < 0 || list_idx > 1 || ref_idx < 0 || ref_idx > 3) return;
    const int bank = subpel_search_type == USE_2_TAPS ? 3 : (subpel_search_type == USE_4_TAPS ? 4 : (subpel_search_type == USE_8_TAPS ? 0 : -1));
    const BlockGeom *g = ctx->blk_geom;
    const uint32_t pu = ctx->me_block_offset, sb = ctx->me_sb_addr;
    const int col = t->slot_of[list_idx][ref_idx];
    if (bank != t->grid_bank || col < 0 || g->shape != PART_N || g->bwidth != g->bheight || pu >= PRE_PUS || sb >= (uint32_t)t->n_sb || g_pre_pu[pu].x != g->origin_x ||
        g_pre_pu[pu].y != g->origin_y || g_pre_pu[pu].w != g->bwidth)
        return;
    const size_t e = ((size_t)sb * PRE_PUS + pu) * (size_t)t->n_refs + (size_t)col;
    if (t->mv[e] != ((uint32_t)(uint16_t)mvx8 | (uint32_t)(uint16_t)mvy8 << 16)) return;   /* e.g. md_sq_motion_search moved the start */
    const uint32_t *row = t->grid + e * (2 * SVT_HIP_MD_GRID);
    if (row[0] == 0xffffffffu && row[1] == 0xffffffffu) return;
    tls_grid.row = row; tls_grid.cx = mvx8; tls_grid.cy = mvy8;
}
void svt_hip_hook_md_pre_subpel_end(void) { tls_grid.row = NULL; }
/* svt_upsampled_pref_error of one probe: 1 = *err / *sse come from the grid (called through svt_hip_hook_md_subpel_fetch) */
static int pre_grid_fetch(const MV *mv, unsigned int *err, unsigned int *sse) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_MD_PRE)) return 0;
    if (g_pre_verify < 0) g_pre_verify = getenv("SVT_HIP_MD_PRE_VERIFY") ? atoi(getenv("SVT_HIP_MD_PRE_VERIFY")) : 0;
    PRE_COUNT(PC_PROBES);
    if (!tls_grid.row) return 0;
    const int dx = mv->col - tls_grid.cx, dy = mv->row - tls_grid.cy;
    if (dx < -6 || dx > 6 || dy < -6 || dy > 6 || ((dx | dy) & 1)) return 0;
    const int i = 7 * ((dy + 6) >> 1) + ((dx + 6) >> 1);
    PRE_COUNT(PC_PROBE_HITS);
    if (g_pre_verify > 0) return 0;   /* SVT_HIP_MD_PRE_VERIFY=1: the reference computes the probe, svt_hip_hook_md_pre_subpel_verify compares */
    *err = tls_grid.row[2 * i]; *sse = tls_grid.row[2 * i + 1];
    return 1;
}
/* svt_upsampled_pref_error, after the reference has computed a probe itself (only reached when the grid did not serve it): in the self-check mode a probe the grid holds
 * must agree with what the reference got */
void svt_hip_hook_md_pre_subpel_verify(const MV *mv, unsigned int err, unsigned int sse) {
    if (g_pre_verify <= 0 || !tls_grid.row) return;
    const int dx = mv->col - tls_grid.cx, dy = mv->row - tls_grid.cy;
    if (dx < -6 || dx > 6 || dy < -6 || dy > 6 || ((dx | dy) & 1)) return;
    const int i = 7 * ((dy + 6) >> 1) + ((dx + 6) >> 1);
    if (tls_grid.row[2 * i] == err && tls_grid.row[2 * i + 1] == sse) return;
    if (__sync_fetch_and_add(&g_pre_mismatch, 1) < 10)
        fprintf(stderr, "svt_hip_md_pre MISMATCH sub-pel probe mv=(%d,%d) start=(%d,%d) grid=(%u,%u) reference=(%u,%u)\n", mv->col, mv->row, tls_grid.cx, tls_grid.cy, tls_grid.row[2 * i],
                tls_grid.row[2 * i + 1], err, sse);
}

/* fast_loop_core, in front of the prediction: 1 = *sad is the candidate's luma distortion (svt_nxm_sad_kernel_sub_sampled of its prediction) and the prediction is
 * NOT to be made now (svt_hip_hook_md_pre_take tells full_loop_core when it is needed after all) */
int svt_hip_hook_md_pre_lookup(PictureControlSet *pcs, ModeDecisionContext *ctx, ModeDecisionCandidateBuffer *cb, uint32_t *sad) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_MD_PRE)) return 0;
    if (g_pre_verify < 0) g_pre_verify = getenv("SVT_HIP_MD_PRE_VERIFY") ? atoi(getenv("SVT_HIP_MD_PRE_VERIFY")) : 0;
    const unsigned ms = mark_slot(cb);
    if (tls_mark[ms] == cb) tls_mark[ms] = NULL;   /* the buffer gets a new candidate: whatever it was marked for is gone */
    PRE_COUNT(PC_CALLS);
    const ModeDecisionCandidate *c = cb->candidate_ptr;
    if (c->type != INTER_MODE || c->use_intrabc) return 0;
    PRE_COUNT(PC_INTER);
    if (!ctx->md_staging_skip_chroma_pred || !ctx->md_staging_skip_interpolation_search) PRE_MISS(PRE_MISS_LATER_PASS);
    if (c->motion_mode != SIMPLE_TRANSLATION || c->is_interintra_used) PRE_MISS(PRE_MISS_MOTION);
    if (c->is_compound && (c->interinter_comp.type != COMPOUND_AVERAGE || c->compound_idx != 1 || c->comp_group_idx != 0)) PRE_MISS(PRE_MISS_COMPOUND);   /* distance-weighted, wedge, difference-weighted */
    const BlockGeom *g = ctx->blk_geom;
    if (g->shape != PART_N || g->bwidth != g->bheight || g->bwidth < 8 || g->bwidth > 64) PRE_MISS(PRE_MISS_SHAPE);
    const MdPre *t = pre_table_of(pcs);
    if (!t) PRE_MISS(PRE_MISS_NO_TABLE);
    if ((ctx->hbd_mode_decision != 0) != (t->hbd != 0)) PRE_MISS(PRE_MISS_HBD);   /* the tables were made on the planes this picture's fast loop does not decide on */
    const uint32_t pu = ctx->me_block_offset, sb = ctx->me_sb_addr;
    if (pu >= PRE_PUS || sb >= (uint32_t)t->n_sb || g_pre_pu[pu].x != g->origin_x || g_pre_pu[pu].y != g->origin_y || g_pre_pu[pu].w != g->bwidth) PRE_MISS(PRE_MISS_SHAPE);
    MvReferenceFrame rf[2];
    av1_set_ref_frame(rf, c->ref_frame_type);
    if (c->is_compound != (rf[1] > INTRA_FRAME)) PRE_MISS(PRE_MISS_REFERENCE);
    if (c->is_compound) {   /* both vectors are the references' own ME vectors, the prediction is their rounded average (svt_hip_md_fullpel_avg_sad_picture_dev) */
        const int l0 = get_list_idx(rf[0]), r0 = get_ref_frame_idx(rf[0]), l1 = get_list_idx(rf[1]), r1 = get_ref_frame_idx(rf[1]);
        if (l0 < 0 || l0 > 1 || r0 < 0 || r0 > 3 || l1 < 0 || l1 > 1 || r1 < 0 || r1 > 3 || c->prediction_direction[0] != 2) PRE_MISS(PRE_MISS_REFERENCE);
        const int c0 = t->slot_of[l0][r0], c1 = t->slot_of[l1][r1];
        if (c0 < 0 || c1 < 0 || !t->n_pairs) PRE_MISS(PRE_MISS_COMPOUND);
        const int q = t->pair_of[c0][c1];
        if (q < 0) PRE_MISS(PRE_MISS_COMPOUND);
        const size_t base = ((size_t)sb * PRE_PUS + pu) * (size_t)t->n_refs, eb = ((size_t)sb * PRE_PUS + pu) * (size_t)t->n_pairs + (size_t)q;
        const int16_t x0 = c->motion_vector_xl0, y0 = c->motion_vector_yl0, x1 = c->motion_vector_xl1, y1 = c->motion_vector_yl1;
        if (t->mv[base + c0] != ((uint32_t)(uint16_t)x0 | (uint32_t)(uint16_t)y0 << 16) || t->mv[base + c1] != ((uint32_t)(uint16_t)x1 | (uint32_t)(uint16_t)y1 << 16) ||
            t->bisad[eb] == 0xffffffffu)
            PRE_MISS(PRE_MISS_VECTOR);
        const int bw = g->bwidth, pic_w = (int)pcs->parent_pcs_ptr->av1_cm->mi_cols * 4, pic_h = (int)pcs->parent_pcs_ptr->av1_cm->mi_rows * 4;
        for (int k = 0; k < 2; k++) {   /* neither vector may be one av1_inter_prediction would clamp */
            const int bx = (int)ctx->blk_origin_x + ((k ? x1 : x0) >> 3), by = (int)ctx->blk_origin_y + ((k ? y1 : y0) >> 3);
            if (bx <= -(bw + 4) || by <= -(bw + 4) || bx >= pic_w + 3 || by >= pic_h + 3) PRE_MISS(PRE_MISS_BORDER);
        }
        if (tls_mark[ms]) PRE_MISS(PRE_MISS_MARK);
        *sad = t->bisad[eb];
        PRE_COUNT(PC_HITS); PRE_COUNT(PC_BI_HITS);
        if (g_pre_verify == 3) { tls_mark[ms] = cb; return 5; }   /* debug: predicted now AND again where the survivors are predicted */
        if (g_pre_verify) return g_pre_verify == 2 ? 3 : 2;   /* 3: the reference predicts as usual and the TABLE's distortion is used (SVT_HIP_MD_PRE_VERIFY=2) */
        tls_mark[ms] = cb;
        return 1;
    }
    const int li = get_list_idx(rf[0]), ri = get_ref_frame_idx(rf[0]);
    if (li < 0 || li > 1 || ri < 0 || ri > 3 || c->prediction_direction[0] != li) PRE_MISS(PRE_MISS_REFERENCE);
    const int col = t->slot_of[li][ri];
    if (col < 0) PRE_MISS(PRE_MISS_REFERENCE);
    const int16_t mx = li ? c->motion_vector_xl1 : c->motion_vector_xl0, my = li ? c->motion_vector_yl1 : c->motion_vector_yl0;
    const size_t e = ((size_t)sb * PRE_PUS + pu) * (size_t)t->n_refs + (size_t)col;
    if (t->mv[e] != ((uint32_t)(uint16_t)mx | (uint32_t)(uint16_t)my << 16) || t->sad[e] == 0xffffffffu) PRE_MISS(PRE_MISS_VECTOR);
    /* av1_inter_prediction clamps the vector so that the block stays within (block size + AOM_INTERP_EXTEND) samples of the picture (clamp_mv_to_umv_border_sb,
     * Common/Codec/EbInterPrediction.h): a vector it would move is not what the table measured */
    const int bx = (int)ctx->blk_origin_x + (mx >> 3), by = (int)ctx->blk_origin_y + (my >> 3), bw = g->bwidth;
    const int pic_w = (int)pcs->parent_pcs_ptr->av1_cm->mi_cols * 4, pic_h = (int)pcs->parent_pcs_ptr->av1_cm->mi_rows * 4;
    if (bx <= -(bw + 4) || by <= -(bw + 4) || bx >= pic_w + 3 || by >= pic_h + 3) PRE_MISS(PRE_MISS_BORDER);
    if (tls_mark[ms]) PRE_MISS(PRE_MISS_MARK);   /* another buffer's mark lives here: no room to remember that this one has no samples yet */
    *sad = t->sad[e];
    PRE_COUNT(PC_HITS);
    if (g_pre_verify == 3) { tls_mark[ms] = cb; return 5; }
    if (g_pre_verify) return g_pre_verify == 2 ? 3 : 2;   /* SVT_HIP_MD_PRE_VERIFY=1: the reference computes the candidate as well and svt_hip_hook_md_pre_verify compares; 2: it predicts and the table's distortion is used */
    tls_mark[ms] = cb;
    return 1;
}
/* SVT_HIP_MD_PRE_VERIFY=1 (tests): fast_loop_core computed the candidate itself after a table hit; the two distortions must agree */
void svt_hip_hook_md_pre_verify(PictureControlSet *pcs, ModeDecisionContext *ctx, ModeDecisionCandidateBuffer *cb, uint32_t table_sad, uint32_t ref_sad) {
    if (table_sad == ref_sad) return;
    const ModeDecisionCandidate *c = cb->candidate_ptr;
    if (__sync_fetch_and_add(&g_pre_mismatch, 1) < 10)
        fprintf(stderr, "svt_hip_md_pre MISMATCH picture=%llu sb=%u pu=%u blk=(%u,%u) %ux%u ref_type=%d dir=%d mv0=(%d,%d) mv1=(%d,%d) table=%u reference=%u pd_pass=%d\n",
                (unsigned long long)pcs->picture_number, ctx->me_sb_addr, ctx->me_block_offset, ctx->blk_origin_x, ctx->blk_origin_y, ctx->blk_geom->bwidth, ctx->blk_geom->bheight,
                c->ref_frame_type, c->prediction_direction[0], c->motion_vector_xl0, c->motion_vector_yl0, c->motion_vector_xl1, c->motion_vector_yl1, table_sad, ref_sad, ctx->pd_pass);
}
/* full_loop_core, where it decides whether to predict an inter candidate: 1 = the buffer's stage-0 prediction was skipped (svt_hip_hook_md_pre_lookup) and nobody has made
 * it since; the mark is dropped either way */
int svt_hip_hook_md_pre_take(const ModeDecisionCandidateBuffer *cb, int predicted_late) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_MD_PRE)) return 0;
    const unsigned ms = mark_slot(cb);
    if (tls_mark[ms] != cb) return 0;
    tls_mark[ms] = NULL;
    if (predicted_late) PRE_COUNT(PC_LATE);
    return 1;
}

/* svt_hip_hooks_enc_predeinit with other instances still encoding: hook "md_pre" queues on a context of its own (host_register + upload of a stale resident plane
 * included, svt_hip_resident_acquire), which the quiesce of the main and pool contexts does not cover.  Lock = no new issue, drained = nothing in flight; the caller
 * unregisters the page-locked ranges and then calls svt_hip_md_bridge_resume. */
void svt_hip_md_bridge_quiesce(void) {
    pthread_mutex_lock(&g_pre_issue_mu);
    if (g_pre_ctx) (void)svt_hip_sync(g_pre_ctx);
}
void svt_hip_md_bridge_resume(void) { pthread_mutex_unlock(&g_pre_issue_mu); }

/* the shared staging buffers of the three hooks above (svt_hip_hooks_enc_deinit) */
void svt_hip_md_bridge_release(SvtHipCtx *hip) {
    void **all[] = {&d_src, &d_pred, &d_desc, &d_coeff, &d_sp_ref, &d_sp_src, &d_sp_pred, &d_sp_job, &d_sp_out};
    for (unsigned i = 0; i < sizeof(all) / sizeof(all[0]); i++) { svt_hip_free(hip, *all[i]); *all[i] = NULL; }
    pthread_mutex_lock(&g_pre_issue_mu);
    if (g_pre_ctx) { (void)svt_hip_sync(g_pre_ctx); pre_sweep(g_pre_ctx, 1); }
    for (int i = 0; i < PRE_SLOTS; i++) {   /* the picture tables of hook "md_pre" */
        if (g_pre[i].sad) svt_hip_host_free(hip, g_pre[i].sad);
        if (g_pre[i].grid) svt_hip_host_free(hip, g_pre[i].grid);
        if (g_pre[i].bisad) svt_hip_host_free(hip, g_pre[i].bisad);
        svt_hip_free(hip, g_pre[i].d_bisad); svt_hip_free(hip, g_pre[i].d_src16);
        if (g_pre[i].h_dev_mv) svt_hip_host_free(hip, g_pre[i].h_dev_mv);
        if (g_pre[i].flags) svt_hip_host_free(hip, (void *)g_pre[i].flags);
        svt_hip_free(hip, g_pre[i].d_mv); svt_hip_free(hip, g_pre[i].d_sad); svt_hip_free(hip, g_pre[i].d_grid); svt_hip_free(hip, g_pre[i].d_seq);
        free(g_pre[i].mv);
        memset(&g_pre[i], 0, sizeof(g_pre[i]));
    }
    if (g_pre_ctx) { svt_hip_destroy(g_pre_ctx); g_pre_ctx = NULL; }
    pthread_mutex_unlock(&g_pre_issue_mu);
}
