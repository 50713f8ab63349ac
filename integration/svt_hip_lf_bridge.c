/* svt_hip_lf_bridge.c — see svt_hip_lf_bridge.h / svt_hip_hooks.h.  Reference-side glue: host code only, every pixel operation is a batched
 * svt_hip_* call. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "svt_hip_hooks.h"
#include "svt_hip_lf_bridge.h"
#include "EbDeblockingCommon.h"
#include "EbDeblockingFilter.h"
#include "EbRestoration.h"
#include "EbCdef.h"
#include "EbReferenceObject.h"
#include "EbUtility.h"
#include "EbLog.h"

int8_t get_sg_step(int8_t sg_filter_mode);   /* Encoder/Codec/EbRestorationPick.c:690 (no header declares it) */
void   svt_av1_loop_restoration_save_boundary_lines(const Yv12BufferConfig *frame, Av1Common *cm, int32_t after_cdef);   /* Common/Codec/EbRestoration.c:1843; its callers declare it themselves (EbDlfProcess.c:24, EbCdefProcess.c:44) */

#define HIP_TRY(call) do { if ((call) != SVT_HIP_OK) return EB_ErrorUndefined; } while (0)   /* caller falls back to the C loop */
#define LF_BORDER 3                                                                           /* RESTORATION_BORDER */

static int log2i(int v) { int l = 0; while ((1 << l) < v) l++; return l; }
static size_t plane_bytes(const SvtHipLfPicture *p, int pl) { return (size_t)p->stride[pl] * (size_t)((p->h >> (pl > 0)) + 2 * LF_BORDER) * (size_t)p->pix_bytes; }
/* Columns left of a device plane's sample (0, 0): the restoration border, rounded up so that (0, 0) sits on a 64-sample boundary.  The runtime's 2-D copy picks its element
 * width from the alignment of both pointers, both pitches and the row length; with the origin 3 bytes into a row it moved a 3840 x 2160 plane one byte per thread — 1.5 ms
 * up (5 GB/s, rocprofv3: copyBufferRect grid 3840 x 2160) against 0.2 ms for the same plane between aligned addresses (tools/copy_probe2.py). */
#define LF_XOFF 64
static void *plane_origin(const SvtHipLfPicture *p, void *base, int pl) { return (uint8_t *)base + ((size_t)LF_BORDER * p->stride[pl] + LF_XOFF) * (size_t)p->pix_bytes; }

static int is_16bit_of(const PictureControlSet *pcs) {
    const SequenceControlSet *scs = (const SequenceControlSet *)pcs->scs_wrapper_ptr->object_ptr;
    return scs->static_config.encoder_bit_depth > EB_8BIT || scs->static_config.is_16bit_pipeline;
}
static EbPictureBufferDesc *recon_of(PictureControlSet *pcs, int is_16bit) {   /* the selection every filter stage repeats (e.g. EbDlfProcess.c:178-191) */
    if (pcs->parent_pcs_ptr->is_used_as_reference_flag == EB_TRUE) {
        EbReferenceObject *ro = (EbReferenceObject *)pcs->parent_pcs_ptr->reference_picture_wrapper_ptr->object_ptr;
        return is_16bit ? ro->reference_picture16bit : ro->reference_picture;
    }
    return is_16bit ? pcs->recon_picture16bit_ptr : pcs->recon_picture_ptr;
}
static EbPictureBufferDesc *source_of(PictureControlSet *pcs, int is_16bit) {   /* picture_sse_calculations :843 / :905, cdef_seg_search :136 */
    return is_16bit ? pcs->input_frame16bit : (EbPictureBufferDesc *)pcs->parent_pcs_ptr->enhanced_picture_ptr;
}

EbErrorType svt_hip_lf_picture_ctor(SvtHipCtx *hip, SvtHipLfPicture *p, int w, int h, int is_16bit, int bd) {
    memset(p, 0, sizeof(*p));
    p->pix_bytes = is_16bit ? 2 : 1; p->bd = bd; p->w = w; p->h = h;
    const int nfb = ((w + 63) / 64) * ((h + 63) / 64), mi_cols = (w + 3) / 4, mi_rows = (h + 3) / 4;
    for (int pl = 0; pl < 3; pl++) {
        const int pw = w >> (pl > 0), ph = h >> (pl > 0);
        p->stride[pl] = (pw + LF_XOFF + LF_BORDER + 63) & ~63; p->src_stride[pl] = (pw + 63) & ~63;
        HIP_TRY(svt_hip_malloc(hip, &p->d_recon[pl], plane_bytes(p, pl))); HIP_TRY(svt_hip_malloc(hip, &p->d_cdef[pl], plane_bytes(p, pl)));
        HIP_TRY(svt_hip_malloc(hip, &p->d_rest[pl], plane_bytes(p, pl))); HIP_TRY(svt_hip_malloc(hip, &p->d_dbl[pl], plane_bytes(p, pl)));
        HIP_TRY(svt_hip_malloc(hip, &p->d_src[pl], (size_t)p->src_stride[pl] * ph * p->pix_bytes));
        p->src[pl] = p->d_src[pl]; p->src_st[pl] = p->src_stride[pl];
        p->units_w[pl] = (pw + 3) / 4; p->units_h[pl] = (ph + 3) / 4;
        for (int d = 0; d < 2; d++) {
            p->h_edges[pl][d] = (uint16_t *)malloc(sizeof(uint16_t) * p->units_w[pl] * p->units_h[pl]);
            HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_edges[pl][d], sizeof(uint16_t) * p->units_w[pl] * p->units_h[pl]));
            if (!p->h_edges[pl][d]) return EB_ErrorInsufficientResources;
        }
        p->max_units[pl] = ((pw + 31) / 32) * ((ph + 31) / 32);     /* smallest restoration unit is 64 -> count_units_in_tile rounds to nearest */
        HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_unit_ep[pl], p->max_units[pl])); HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_unit_xqd[pl], p->max_units[pl] * 8));
        HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_unit_wiener[pl], p->max_units[pl] * 32));
    }
    p->h_skip8 = (uint8_t *)malloc((size_t)(w / 8) * (h / 8)); p->h_mi = (SvtHipDlfModeInfo *)calloc((size_t)mi_cols * mi_rows, sizeof(SvtHipDlfModeInfo));
    p->h_mse = (uint64_t *)malloc(sizeof(uint64_t) * 2 * nfb * 64);
    if (!p->h_skip8 || !p->h_mi || !p->h_mse) return EB_ErrorInsufficientResources;
    if (!(p->h_mi_until = (uint16_t *)malloc(sizeof(uint16_t) * mi_cols)) || !(p->h_skip4 = (uint8_t *)malloc((size_t)mi_cols * mi_rows))) return EB_ErrorInsufficientResources;
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_mi, sizeof(SvtHipDlfModeInfo) * mi_cols * mi_rows));
    p->h_mi_pinned = svt_hip_hooks_pin_enabled() && svt_hip_host_register(hip, p->h_mi, sizeof(SvtHipDlfModeInfo) * mi_cols * mi_rows) == SVT_HIP_OK;
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_skip8, (size_t)(w / 8) * (h / 8)));
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_mse, sizeof(uint64_t) * 2 * nfb * 64));
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_dir, (size_t)nfb * 64)); HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_var, sizeof(int32_t) * nfb * 64));
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_y_strength, nfb)); HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_uv_strength, nfb));
    HIP_TRY(svt_hip_malloc(hip, (void **)&p->d_sse, 64));
    return EB_ErrorNone;
}

void svt_hip_lf_picture_dctor(SvtHipCtx *hip, SvtHipLfPicture *p) {
    for (int pl = 0; pl < 3; pl++) {
        svt_hip_free(hip, p->d_recon[pl]); svt_hip_free(hip, p->d_cdef[pl]); svt_hip_free(hip, p->d_rest[pl]); svt_hip_free(hip, p->d_src[pl]); svt_hip_free(hip, p->d_dbl[pl]);
        for (int d = 0; d < 2; d++) { free(p->h_edges[pl][d]); svt_hip_free(hip, p->d_edges[pl][d]); }
        svt_hip_free(hip, p->d_unit_ep[pl]); svt_hip_free(hip, p->d_unit_xqd[pl]); svt_hip_free(hip, p->d_unit_wiener[pl]);
        free(p->h_wiener_M[pl]); free(p->h_wiener_H[pl]);
    }
    if (p->h_mi_pinned) svt_hip_host_unregister(hip, p->h_mi);
    free(p->h_skip8); free(p->h_mi); free(p->h_mi_until); free(p->h_skip4); free(p->h_mse); svt_hip_free(hip, p->d_mi);
    svt_hip_free(hip, p->d_skip8); svt_hip_free(hip, p->d_mse); svt_hip_free(hip, p->d_dir); svt_hip_free(hip, p->d_var);
    svt_hip_free(hip, p->d_y_strength); svt_hip_free(hip, p->d_uv_strength); svt_hip_free(hip, p->d_sse);
    memset(p, 0, sizeof(*p));
}

/* ---------------------------------------------------------------- per-picture state ----------------------------------------------------
 * Pictures pass dlf_kernel -> cdef_kernel -> rest_kernel in order, several pictures can be in different stages at once — and run side by side: the table is
 * guarded by a mutex of its own that is held only to find / hand out an entry, every entry has its own mutex that a hook holds for the duration of its call
 * (the stages of ONE picture follow each other, and the segments of a stage all ask for the same picture-level result: the first to arrive produces it, the
 * others wait on the entry and find it done), and the device work runs on a context of the hooks' pool (svt_hip_hooks_lock_any; taken AFTER the entry's mutex, so
 * nobody waits for a picture while holding a context).  Entries are pooled: a finished picture's device buffers serve the next one of the same size.
 *
 * Deferred host picture (SVT_HIP_DEFER, default on when every loop-filter hook is): the reconstructed picture stays on the device from deblocking to the
 * restoration filter and comes back ONCE, when the picture leaves the filter stages (svt_hip_hook_picture_done) — the deblocked and the CDEF-filtered versions
 * are never downloaded, and the host-side preparation that only the C restoration path reads (svt_av1_loop_restoration_save_boundary_lines after both stages,
 * svt_extend_frame, the per-segment copy of the picture in get_own_recon) is skipped by the patched process loops (svt_hip_hook_skip_host_prep).  The reference's
 * error convention still holds: a hook that fails after such a skip first brings the host up to date — lf_recover() downloads the versions the C code needs and runs
 * the skipped preparation itself, in the reference's order — and only then reports "not handled". */
enum { ST_SRC = 1, ST_DBL = 2, ST_CDEF = 4, ST_DIRVAR = 8, ST_CDEF_SEARCHED = 16, ST_CDEF_FAILED = 32, ST_SGR_DONE = 64, ST_SGR_FAILED = 128,
       ST_WIENER_DONE = 256, ST_WIENER_FAILED = 512, ST_PADDED = 1024, ST_WNSEARCH_DONE = 2048, ST_WNSEARCH_FAILED = 4096,
       ST_RECON = 8192,        /* d_recon holds the picture as it was before deblocking (uploaded by the level search) */
       ST_HOST_STALE = 16384,  /* the host's recon picture is older than the device's */
       ST_SKIP0 = 32768,       /* save_boundary_lines(.., 0) was skipped on the host */
       ST_SKIP1 = 65536,       /* save_boundary_lines(.., 1) + svt_extend_frame were skipped on the host */
       ST_REST = 131072,       /* d_rest holds restored planes (rest_mask) that the host has not seen */
       ST_MI = 262144,         /* d_mi holds this picture's mode-info grid (uploaded by the level search; its levels are placeholders) */
       ST_SKIP4 = 524288 };    /* h_skip4 holds this picture's skip flag per 4 x 4 unit (a by-product of the mode-info pass; the CDEF stages' skip map comes from it) */
typedef struct {
    PictureControlSet *pcs;     /* NULL: free */
    int                allocated, flags, defer, rest_mask, mu_ready;
    pthread_mutex_t    mu;
    const void        *res_host[3];   /* source planes taken from the resident table (released when the picture is done) */
    SvtHipLfPicture    pic;
} LfState;
#define LF_MAX_IN_FLIGHT 64
static LfState         g_state[LF_MAX_IN_FLIGHT];
static pthread_mutex_t g_tab_mu = PTHREAD_MUTEX_INITIALIZER;
static long            g_lf_up_planes, g_lf_down_planes, g_lf_recoveries, g_lf_deferred, g_lf_src_resident;
static long            g_lf_final_retries, g_lf_final_host_chains, g_lf_final_fatal;   /* lf_final_download */
static long long       g_lf_up_bytes, g_lf_down_bytes;

/* the reconstructed pictures' host buffers, page-locked in place once (they are allocated once per encoder instance: EbReferenceObject / the PCS pool) */
#define LF_PINS 512
static struct { void *p; size_t n; } g_pin[LF_PINS];
static pthread_mutex_t g_pin_mu = PTHREAD_MUTEX_INITIALIZER;
static int             g_pin_off;
static void lf_pin(SvtHipCtx *hip, void *base, size_t bytes) {
    if (!svt_hip_hooks_pin_enabled() || !base || !bytes) return;
    pthread_mutex_lock(&g_pin_mu);
    int have = g_pin_off, slot = -1;
    for (int i = 0; i < LF_PINS && !have; i++) {
        if (g_pin[i].p == base) have = 1;
        else if (!g_pin[i].p && slot < 0) slot = i;
    }
    if (!have && slot >= 0 && svt_hip_host_register(hip, base, bytes) == SVT_HIP_OK) { g_pin[slot].p = base; g_pin[slot].n = bytes; }
    pthread_mutex_unlock(&g_pin_mu);
}
static void lf_pin_picture(SvtHipCtx *hip, const EbPictureBufferDesc *pic, int pix_bytes) {
    lf_pin(hip, pic->buffer_y, (size_t)pic->luma_size * pix_bytes);
    lf_pin(hip, pic->buffer_cb, (size_t)pic->chroma_size * pix_bytes);
    lf_pin(hip, pic->buffer_cr, (size_t)pic->chroma_size * pix_bytes);
}
void svt_hip_lf_bridge_unpin(SvtHipCtx *hip, int keep_pinning) {
    pthread_mutex_lock(&g_pin_mu);
    for (int i = 0; i < LF_PINS; i++)
        if (g_pin[i].p) { (void)svt_hip_host_unregister(hip, g_pin[i].p); g_pin[i].p = NULL; }
    g_pin_off = !keep_pinning;
    pthread_mutex_unlock(&g_pin_mu);
}

static int lf_defer_wanted(const SequenceControlSet *scs) {
    static int env = -1;
    if (env < 0) env = !(getenv("SVT_HIP_DEFER") && !atoi(getenv("SVT_HIP_DEFER")));
    (void)scs;
    return env && svt_hip_hook_enabled(SVT_HIP_HOOK_DLF) && svt_hip_hook_enabled(SVT_HIP_HOOK_CDEF_SEARCH) && svt_hip_hook_enabled(SVT_HIP_HOOK_CDEF_APPLY) &&
           svt_hip_hook_enabled(SVT_HIP_HOOK_SGR_SEARCH) && svt_hip_hook_enabled(SVT_HIP_HOOK_WIENER_SEARCH) && svt_hip_hook_enabled(SVT_HIP_HOOK_REST_APPLY);
}

/* g_tab_mu held: the entry of pcs, or (create) a free one reserved for it; *fresh = its device buffers still have to be (re)built for this geometry */
static LfState *state_find(PictureControlSet *pcs, int create) {
    for (int i = 0; i < LF_MAX_IN_FLIGHT; i++)
        if (g_state[i].pcs == pcs) return &g_state[i];
    if (!create) return NULL;
    const int is_16bit = is_16bit_of(pcs);
    const SequenceControlSet *scs = (const SequenceControlSet *)pcs->scs_wrapper_ptr->object_ptr;
    const EbPictureBufferDesc *rec = recon_of(pcs, is_16bit);
    const int w = rec->width, h = rec->height, bd = scs->static_config.encoder_bit_depth;
    const int pad_r = scs->max_input_pad_right, pad_b = scs->max_input_pad_bottom;
    if ((w & 7) || (h & 7) || (bd != 8 && bd != 10) || scs->subsampling_x != 1 || scs->subsampling_y != 1 || scs->seq_header.color_config.mono_chrome ||
        pad_r < 0 || pad_r >= 8 || pad_b < 0 || pad_b >= 8 || ((w - pad_r) & 1) || ((h - pad_b) & 1) || w != scs->max_input_luma_width || h != scs->max_input_luma_height) {
        svt_hip_hooks_log("loop filter stages: %d x %d (source padded by %d x %d), %d-bit, subsampling %d/%d: not covered, C loops",
                          w, h, pad_r, pad_b, bd, scs->subsampling_x, scs->subsampling_y);
        return NULL;   /* outside what the device path covers (monochrome, 4:2:2 / 4:4:4, 12-bit): the caller keeps its C loop */
    }
    LfState *s = NULL;
    for (int i = 0; i < LF_MAX_IN_FLIGHT && !s; i++)
        if (!g_state[i].pcs && g_state[i].allocated && g_state[i].pic.w == w && g_state[i].pic.h == h && g_state[i].pic.pix_bytes == (is_16bit ? 2 : 1) && g_state[i].pic.bd == bd)
            s = &g_state[i];
    for (int i = 0; i < LF_MAX_IN_FLIGHT && !s; i++)
        if (!g_state[i].pcs && !g_state[i].allocated) s = &g_state[i];
    if (!s) return NULL;
    if (!s->mu_ready) { pthread_mutex_init(&s->mu, NULL); s->mu_ready = 1; }
    s->pcs = pcs; s->flags = 0; s->rest_mask = 0; s->defer = -1;   /* -1: lf_enter completes the entry under its own mutex */
    return s;
}
/* the entry's mutex and a pool context held: builds the device buffers of a reserved entry */
static int state_complete(SvtHipCtx *hip, LfState *s) {
    if (s->defer >= 0) return 1;
    PictureControlSet *pcs = s->pcs;
    const int is_16bit = is_16bit_of(pcs);
    const SequenceControlSet *scs = (const SequenceControlSet *)pcs->scs_wrapper_ptr->object_ptr;
    const EbPictureBufferDesc *rec = recon_of(pcs, is_16bit);
    const int w = rec->width, h = rec->height, bd = scs->static_config.encoder_bit_depth;
    if (!s->allocated) {
        if (svt_hip_lf_picture_ctor(hip, &s->pic, w, h, is_16bit, bd) != EB_ErrorNone) { svt_hip_lf_picture_dctor(hip, &s->pic); return 0; }
        s->allocated = 1;
    }
    for (int pl = 0; pl < 3; pl++) { s->pic.src[pl] = s->pic.d_src[pl]; s->pic.src_st[pl] = s->pic.src_stride[pl]; s->res_host[pl] = NULL; }
    /* A source size that is not a multiple of 8 is coded padded (w, h), but the reference deblocks the last superblock row / column only up to the
     * unpadded extent (EbDeblockingFilter.c:343-367) and restores the cropped frame (link_eb_to_aom_buffer_desc, EbDlfProcess.c:247-251; the
     * restoration units, stripes and the 3-sample extension all follow the crop size); CDEF and the level search's SSE use the coded size. */
    s->pic.cw = w - scs->max_input_pad_right; s->pic.ch = h - scs->max_input_pad_bottom; s->pic.sb_size = scs->seq_header.sb_size == BLOCK_128X128 ? 128 : 64;
    s->defer = lf_defer_wanted(scs);
    if (s->defer) __sync_fetch_and_add(&g_lf_deferred, 1);
    lf_pin_picture(hip, rec, s->pic.pix_bytes);
    return 1;
}
/* -> the picture's entry with its mutex held and a pool context in *hip, or NULL (nothing held) */
static LfState *lf_enter(PictureControlSet *pcs, int create, SvtHipCtx **hip) {
    pthread_mutex_lock(&g_tab_mu);
    LfState *s = state_find(pcs, create);
    pthread_mutex_unlock(&g_tab_mu);
    if (!s) return NULL;
    pthread_mutex_lock(&s->mu);
    if (s->pcs != pcs || !(*hip = svt_hip_hooks_lock_any())) { pthread_mutex_unlock(&s->mu); return NULL; }
    if (!state_complete(*hip, s)) {
        svt_hip_hooks_unlock_any();
        pthread_mutex_lock(&g_tab_mu); s->pcs = NULL; pthread_mutex_unlock(&g_tab_mu);
        pthread_mutex_unlock(&s->mu);
        return NULL;
    }
    return s;
}
static void lf_leave(LfState *s) {
    svt_hip_hooks_unlock_any();   /* drains the context: what this call launched is complete for whichever context the next stage gets */
    pthread_mutex_unlock(&s->mu);
}

void svt_hip_lf_bridge_release(SvtHipCtx *hip) {   /* no picture is in flight any more (svt_hip_hooks_enc_deinit) */
    if (g_lf_up_planes + g_lf_down_planes)
        fprintf(stderr, "svt_hip_lf_pictures deferred=%ld recovered=%ld source_planes_resident=%ld planes_up=%ld up_mb=%.1f planes_down=%ld down_mb=%.1f final_retries=%ld final_host_chains=%ld final_fatal=%ld\n",
                g_lf_deferred, g_lf_recoveries, g_lf_src_resident, g_lf_up_planes, g_lf_up_bytes / 1048576.0, g_lf_down_planes, g_lf_down_bytes / 1048576.0, g_lf_final_retries,
                g_lf_final_host_chains, g_lf_final_fatal);
    for (int i = 0; i < LF_MAX_IN_FLIGHT; i++) {
        if (g_state[i].allocated) svt_hip_lf_picture_dctor(hip, &g_state[i].pic);
        if (g_state[i].mu_ready) pthread_mutex_destroy(&g_state[i].mu);
        memset(&g_state[i], 0, sizeof(g_state[i]));
    }
    svt_hip_lf_bridge_unpin(hip, 0);
    g_pin_off = 0;
}

/* ---------------------------------------------------------------- host <-> device planes ------------------------------------------------
 * Copies of a picture's planes are queued on the context's stream and waited for once (the host side is page-locked: lf_pin_picture). */
static uint8_t *pic_plane(const EbPictureBufferDesc *pic, int pl, int pix_bytes, int *stride) {
    const int ss = pl > 0;
    uint8_t *base = pl == 0 ? pic->buffer_y : (pl == 1 ? pic->buffer_cb : pic->buffer_cr);
    *stride = pl == 0 ? pic->stride_y : (pl == 1 ? pic->stride_cb : pic->stride_cr);
    return base + ((size_t)(pic->origin_y >> ss) * *stride + (pic->origin_x >> ss)) * (size_t)pix_bytes;
}
static EbErrorType upload(SvtHipCtx *hip, SvtHipLfPicture *p, const EbPictureBufferDesc *pic, void *const d_dst[3], int is_src) {
    for (int pl = 0; pl < 3; pl++) {
        int st;
        const uint8_t *s = pic_plane(pic, pl, p->pix_bytes, &st);
        const int pw = p->w >> (pl > 0), ph = p->h >> (pl > 0);
        const int dstride = is_src ? p->src_stride[pl] : p->stride[pl];
        uint8_t *d = is_src ? (uint8_t *)d_dst[pl] : (uint8_t *)plane_origin(p, d_dst[pl], pl);
        HIP_TRY(svt_hip_memcpy2d_h2d_async(hip, d, (size_t)dstride * p->pix_bytes, s, (size_t)st * p->pix_bytes, (size_t)pw * p->pix_bytes, ph));
        __sync_fetch_and_add(&g_lf_up_planes, 1); __sync_fetch_and_add(&g_lf_up_bytes, (long long)pw * ph * p->pix_bytes);
    }
    HIP_TRY(svt_hip_sync(hip));   /* the host picture may change as soon as the hook returns */
    return EB_ErrorNone;
}
/* crop: only the unpadded extent comes back (the restoration filter writes the cropped frame, EbRestoration.c:1330-1350) */
static __thread int tls_download_started;   /* a copy into the host picture has been queued since the caller cleared this: the host planes may hold part of it */
static EbErrorType download(SvtHipCtx *hip, const SvtHipLfPicture *p, void *const d_src[3], EbPictureBufferDesc *pic, int plane_mask, int crop) {
    for (int pl = 0; pl < 3; pl++) {
        if (!(plane_mask & (1 << pl))) continue;
        int st;
        uint8_t *d = pic_plane(pic, pl, p->pix_bytes, &st);
        const int pw = (crop ? p->cw : p->w) >> (pl > 0), ph = (crop ? p->ch : p->h) >> (pl > 0);
        const uint8_t *s = (const uint8_t *)plane_origin(p, d_src[pl], pl);
        tls_download_started = 1;
        HIP_TRY(svt_hip_memcpy2d_d2h_async(hip, d, (size_t)st * p->pix_bytes, s, (size_t)p->stride[pl] * p->pix_bytes, (size_t)pw * p->pix_bytes, ph));
        __sync_fetch_and_add(&g_lf_down_planes, 1); __sync_fetch_and_add(&g_lf_down_bytes, (long long)pw * ph * p->pix_bytes);
    }
    HIP_TRY(svt_hip_sync(hip));
    return EB_ErrorNone;
}
/* The source picture of the filter stages: with resident planes (SVT_HIP_RESIDENT) the 8-bit enhanced picture is usually on the device already — the temporal
 * filter and the motion search read it there — and the stages read that copy in place (picture stride, origin offset); otherwise it is uploaded once per picture. */
static EbErrorType ensure_src(SvtHipCtx *hip, LfState *s) {
    if (s->flags & ST_SRC) return EB_ErrorNone;
    SvtHipLfPicture *p = &s->pic;
    const EbPictureBufferDesc *in = source_of(s->pcs, p->pix_bytes == 2);
    if (p->pix_bytes == 1 && svt_hip_resident_enabled()) {
        const uint8_t *dev[3] = {0};
        int n = 0;
        for (int pl = 0; pl < 3; pl++, n++) {
            const uint8_t *base = pl == 0 ? in->buffer_y : (pl == 1 ? in->buffer_cb : in->buffer_cr);
            const int st = pl == 0 ? in->stride_y : (pl == 1 ? in->stride_cb : in->stride_cr);
            dev[pl] = (const uint8_t *)svt_hip_resident_acquire(hip, base, (size_t)st * (size_t)((in->height >> (pl > 0)) + 2 * (in->origin_y >> (pl > 0))));
            if (!dev[pl]) break;
        }
        if (n == 3) {
            for (int pl = 0; pl < 3; pl++) {
                const int st = pl == 0 ? in->stride_y : (pl == 1 ? in->stride_cb : in->stride_cr);
                s->res_host[pl] = pl == 0 ? in->buffer_y : (pl == 1 ? in->buffer_cb : in->buffer_cr);
                p->src[pl] = (void *)(uintptr_t)(dev[pl] + (size_t)(in->origin_y >> (pl > 0)) * st + (in->origin_x >> (pl > 0)));
                p->src_st[pl] = st;
            }
            __sync_fetch_and_add(&g_lf_src_resident, 3);
            s->flags |= ST_SRC;
            return EB_ErrorNone;
        }
        for (int pl = 0; pl < n; pl++) svt_hip_resident_release(pl == 0 ? in->buffer_y : (pl == 1 ? in->buffer_cb : in->buffer_cr));
    }
    lf_pin_picture(hip, in, p->pix_bytes);
    if (upload(hip, p, in, p->d_src, 1) != EB_ErrorNone) return EB_ErrorUndefined;
    s->flags |= ST_SRC;
    return EB_ErrorNone;
}

/* A hook has failed (or is about to hand its stage back to the C code) after host work was skipped: the host picture and the boundary-line buffers are brought to
 * exactly the state the reference's own flow would have left at this point.  stage 0: the C code that follows reads the deblocked picture (CDEF search / filter);
 * stage 1: it reads the CDEF output (restoration search / filter).  Afterwards the picture is no longer deferred. */
static void lf_recover(SvtHipCtx *hip, LfState *s, int stage) {
    if (!s->defer) return;
    SvtHipLfPicture *p = &s->pic;
    PictureControlSet *pcs = s->pcs;
    const SequenceControlSet *scs = (const SequenceControlSet *)pcs->scs_wrapper_ptr->object_ptr;
    Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
    EbPictureBufferDesc *rec = recon_of(pcs, p->pix_bytes == 2);
    const int highbd = p->pix_bytes == 2;
    __sync_fetch_and_add(&g_lf_recoveries, 1);
    int lost = 0;   /* a picture only the device has could not be brought back, not even at the second attempt */
    if ((s->flags & (ST_HOST_STALE | ST_SKIP0)) && (s->flags & ST_DBL) && download(hip, p, p->d_recon, rec, 7, 0) != EB_ErrorNone) {
        (void)svt_hip_sync(hip);
        lost |= download(hip, p, p->d_recon, rec, 7, 0) != EB_ErrorNone;
    }
    if (s->flags & ST_SKIP0) svt_av1_loop_restoration_save_boundary_lines(cm->frame_to_show, cm, 0);
    if (stage >= 1) {
        if ((s->flags & ST_CDEF) && download(hip, p, p->d_cdef, rec, 7, 0) != EB_ErrorNone) {   /* a border the device has added inside a padded picture is the one svt_extend_frame writes below */
            (void)svt_hip_sync(hip);
            lost |= download(hip, p, p->d_cdef, rec, 7, 0) != EB_ErrorNone;
        }
        if (s->flags & ST_SKIP1) {
            svt_av1_loop_restoration_save_boundary_lines(cm->frame_to_show, cm, 1);
            for (int pl = 0; pl < 3; pl++)
                svt_extend_frame(cm->frame_to_show->buffers[pl], cm->frame_to_show->crop_widths[pl > 0], cm->frame_to_show->crop_heights[pl > 0], cm->frame_to_show->strides[pl > 0],
                                 RESTORATION_BORDER, RESTORATION_BORDER, highbd);
        }
        s->flags &= ~ST_SKIP1;
    }
    if (lost) {   /* the C code that takes over would work on a stale picture: the encoder stops instead (lf_final_download explains) */
        SVT_LOG("svt_hip: a picture could not be brought back from the device for the C filter stages (%s) - stopping the encoder\n", svt_hip_last_error(hip));
        __sync_fetch_and_add(&g_lf_final_fatal, 1);
        if (scs->encode_context_ptr && scs->encode_context_ptr->app_callback_ptr && scs->encode_context_ptr->app_callback_ptr->error_handler)
            scs->encode_context_ptr->app_callback_ptr->error_handler(scs->encode_context_ptr->app_callback_ptr->handle, (uint32_t)EB_ErrorUndefined);
    }
    s->flags &= ~(ST_HOST_STALE | ST_SKIP0);
    s->defer = 0;
    svt_hip_hooks_log("loop filter stages: host picture brought up to date after a failed hook (stage %d)", stage);
}
/* SVT_HIP_LF_FAULT=<hook name>: that hook of the loop-filter bridge reports a failure on every picture AFTER doing (or skipping) its work — the tests' way to drive
 * the recovery path on the device and on the CPU test double */
static int lf_fault(const char *hook) {
    const char *e = getenv("SVT_HIP_LF_FAULT");
    return e && !strcmp(e, hook);
}

/* The patched process loops ask before host work that only the C restoration path reads (which: 0 = save_boundary_lines after deblocking, EbDlfProcess.c:251;
 * 1 = save_boundary_lines + svt_extend_frame after CDEF, EbCdefProcess.c:549-572; 2 = get_own_recon of a restoration segment, EbRestProcess.c:517;
 * 3 = svt_extend_frame of the segment's picture copy, EbRestorationPick.c:1535): 1 = skip it, the picture is deferred and lf_recover() knows what was skipped. */
int svt_hip_hook_skip_host_prep(PictureControlSet *pcs, int which) {
    pthread_mutex_lock(&g_tab_mu);
    LfState *s = state_find(pcs, 0);
    pthread_mutex_unlock(&g_tab_mu);
    if (!s) return 0;
    pthread_mutex_lock(&s->mu);
    const int skip = s->pcs == pcs && s->defer > 0;
    if (skip && which == 0) s->flags |= ST_SKIP0;
    if (skip && which == 1) s->flags |= ST_SKIP1;
    pthread_mutex_unlock(&s->mu);
    return skip;
}

/* ---------------------------------------------------------------- deblocking ---------------------------------------------------------
 * SvtHipDlfModeInfo per 4x4 unit from the mode-info grid = what set_lpf_parameters / get_transform_size read (EbDeblockingFilter.c:134-319).
 * uniform_level > 0: the filter-level search only needs the edge geometry (the probed level replaces every non-zero level). */
static void fill_mode_info(SvtHipLfPicture *p, PictureControlSet *pcs, int uniform_level) {
    PictureParentControlSet *ppcs = pcs->parent_pcs_ptr;
    FrameHeader *frm_hdr = &ppcs->frm_hdr;
    const LoopFilterInfoN *lfi_n = &ppcs->lf_info;
    const int mi_cols = (p->w + 3) / 4, mi_rows = (p->h + 3) / 4;
    /* a coded block is a rectangle of 4 x 4 units with one mode info, aligned to its own width and height: scanning in raster order a block is first met at its top-left
     * unit; its record is derived there, once, and copied over the block's rectangle; until[c] = the first row below the block covering column c, so the block's other rows
     * are stepped over by the width their record carries, without touching the mode-info grid again (a 3840 x 2160 picture has 518 400 units; deriving every one of them
     * cost 7 - 10 ms per call on one host thread, once per unit row of every block 2.5 ms) */
    uint16_t *until = p->h_mi_until;
    memset(until, 0, sizeof(uint16_t) * mi_cols);
    for (int r = 0; r < mi_rows; r++)
        for (int c = 0; c < mi_cols;) {
            SvtHipDlfModeInfo *o = &p->h_mi[r * mi_cols + c];
            if (until[c] > r) {   /* covered by a block that began in a row above: its record is here already */
                const int run = (1 << o->bw_log2) >> 2;
                c = ((c / run) + 1) * run;
                continue;
            }
            const MbModeInfo *mbmi = &pcs->mi_grid_base[r * pcs->mi_stride + c]->mbmi;
            const BlockSize bs = mbmi->block_mi.sb_type;
            const int run = block_size_wide[bs] >> 2, c_next = AOMMIN(((c / run) + 1) * run, mi_cols);
            const int rows = block_size_high[bs] >> 2, r_next = AOMMIN(((r / rows) + 1) * rows, mi_rows);
            const int inter = is_inter_block_no_intrabc(mbmi->block_mi.ref_frame[0]);
            TxSize ts = inter ? tx_depth_to_tx_size[0][bs] : tx_depth_to_tx_size[mbmi->tx_depth][bs];
            if (inter && !mbmi->block_mi.skip) ts = tx_depth_to_tx_size[mbmi->tx_depth][bs];
            const TxSize uv = av1_get_max_uv_txsize(bs, 1, 1);
            o->tx_w_log2 = (uint8_t)log2i(tx_size_wide[ts]); o->tx_h_log2 = (uint8_t)log2i(tx_size_high[ts]);
            o->uv_tx_w_log2 = (uint8_t)log2i(tx_size_wide[uv]); o->uv_tx_h_log2 = (uint8_t)log2i(tx_size_high[uv]);
            o->bw_log2 = (uint8_t)log2i(block_size_wide[bs]); o->bh_log2 = (uint8_t)log2i(block_size_high[bs]);
            o->skip_inter = (uint8_t)(mbmi->block_mi.skip && inter);
            const PredictionMode mode = mbmi->block_mi.mode == INTRA_MODE_4x4 ? DC_PRED : mbmi->block_mi.mode;
            for (int pl = 0; pl < 3; pl++)
                for (int dir = 0; dir < 2; dir++)
                    o->level[pl][dir] = uniform_level ? (uint8_t)uniform_level
                        : frm_hdr->delta_lf_params.delta_lf_present
                        ? get_filter_level_delta_lf(frm_hdr, dir, pl, ppcs->curr_delta_lf, 0, mode, mbmi->block_mi.ref_frame[0])
                        : lfi_n->lvl[pl][0][dir][mbmi->block_mi.ref_frame[0]][mode_lf_lut[mode]];
            for (int k = c + 1; k < c_next; k++) p->h_mi[r * mi_cols + k] = *o;
            for (int rr = r; rr < r_next; rr++) memset(&p->h_skip4[rr * mi_cols + c], mbmi->block_mi.skip & 1, (size_t)(c_next - c));   /* bit 0: what is_8x8_block_skip's `is_skip &= skip` keeps */
            for (int rr = r + 1; rr < r_next; rr++) memcpy(&p->h_mi[rr * mi_cols + c], o, sizeof(*o) * (size_t)(c_next - c));
            for (int k = c; k < c_next; k++) until[k] = (uint16_t)r_next;
            c = c_next;
        }
}
static EbErrorType build_and_upload_edges(SvtHipCtx *hip, SvtHipLfPicture *p, int pl) {
    const int mi_cols = (p->w + 3) / 4, mi_rows = (p->h + 3) / 4, pw = p->w >> (pl > 0), ph = p->h >> (pl > 0);
    const int fw = svt_hip_dlf_filtered_units(p->w, p->w - p->cw, p->sb_size, pl > 0), fh = svt_hip_dlf_filtered_units(p->h, p->h - p->ch, p->sb_size, pl > 0);
    if (fw < 0 || fh < 0) return EB_ErrorUndefined;
    HIP_TRY(svt_hip_dlf_build_edges_crop(p->h_mi, mi_cols, mi_rows, pl, pl > 0, pl > 0, pw, ph, fw, fh, p->h_edges[pl][0], p->h_edges[pl][1]));
    const size_t eb = sizeof(uint16_t) * p->units_w[pl] * p->units_h[pl];
    HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_edges[pl][0], p->h_edges[pl][0], eb)); HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_edges[pl][1], p->h_edges[pl][1], eb));
    return EB_ErrorNone;
}
/* The edge planes of the planes in `mask`, built on the device from the grid in d_mi (upload_grid: h_mi was just refilled); level: see svt_hip_dlf_build_edges_picture_dev.
 * SVT_HIP_DLF_EDGES=host keeps the host builder and the upload of its output (the form before: 2.5 ms per call for a 3840 x 2160 picture). */
static int edges_on_host(void) {
    static int v = -1;
    if (v < 0) { const char *e = getenv("SVT_HIP_DLF_EDGES"); v = e && !strcmp(e, "host"); }
    return v;
}
static EbErrorType build_edges(SvtHipCtx *hip, SvtHipLfPicture *p, int mask, int upload_grid, const int (*level)[2]) {
    const int mi_cols = (p->w + 3) / 4, mi_rows = (p->h + 3) / 4;
    if (edges_on_host()) {
        for (int pl = 0; pl < 3; pl++)
            if ((mask & (1 << pl)) && build_and_upload_edges(hip, p, pl) != EB_ErrorNone) return EB_ErrorUndefined;
        return EB_ErrorNone;
    }
    int pw[3], ph[3], fw[3], fh[3];
    uint16_t *ev[3], *eh[3];
    for (int pl = 0; pl < 3; pl++) {
        pw[pl] = p->w >> (pl > 0); ph[pl] = p->h >> (pl > 0);
        fw[pl] = svt_hip_dlf_filtered_units(p->w, p->w - p->cw, p->sb_size, pl > 0); fh[pl] = svt_hip_dlf_filtered_units(p->h, p->h - p->ch, p->sb_size, pl > 0);
        if (fw[pl] < 0 || fh[pl] < 0) return EB_ErrorUndefined;
        ev[pl] = (mask & (1 << pl)) ? p->d_edges[pl][0] : NULL; eh[pl] = (mask & (1 << pl)) ? p->d_edges[pl][1] : NULL;
    }
    if (upload_grid) HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_mi, p->h_mi, sizeof(SvtHipDlfModeInfo) * mi_cols * mi_rows));
    HIP_TRY(svt_hip_dlf_build_edges_picture_dev(hip, p->d_mi, mi_cols, mi_rows, 1, 1, pw, ph, fw, fh, level, ev, eh));
    return EB_ErrorNone;
}
/* Frame-uniform filter levels: lfi_n->lvl[plane][0][dir][ref][mode] is one number per (plane, direction) unless the frame header carries delta_lf or mode / reference deltas
 * (svt_av1_loop_filter_frame_init, EbDeblockingCommon.c:105-160; the encoder sets mode_ref_delta_enabled = 0, EbResourceCoordinationProcess.c:436) — checked, not assumed. */
static int uniform_levels(PictureControlSet *pcs, int mask, int level[3][2]) {
    PictureParentControlSet *ppcs = pcs->parent_pcs_ptr;
    const LoopFilterInfoN *lfi_n = &ppcs->lf_info;
    if (ppcs->frm_hdr.delta_lf_params.delta_lf_present) return 0;
    for (int pl = 0; pl < 3; pl++)
        for (int dir = 0; dir < 2; dir++) {
            level[pl][dir] = 0;
            if (!(mask & (1 << pl))) continue;   /* a plane that is not filtered: its table is not refreshed (:118-123), and its edges are not built */
            level[pl][dir] = lfi_n->lvl[pl][0][dir][0][0];
            for (int ref = 0; ref < REF_FRAMES; ref++)
                for (int m = 0; m < MAX_MODE_LF_DELTAS; m++)
                    if (lfi_n->lvl[pl][0][dir][ref][m] != level[pl][dir]) return 0;
        }
    return 1;
}

/* svt_av1_pick_filter_level(.., LPF_PICK_FROM_FULL_IMAGE) (EbDeblockingFilter.c:1193, the else branch :1262-1310): three searches
 * (luma with dir = 2, i.e. both directions at the probed level and — as the reference indexes last_frame_filter_level[dir] — started from
 * the previous U level; then U; then V), every probe on the device. */
static EbErrorType dlf_pick_level(SvtHipCtx *hip, LfState *s) {
    SvtHipLfPicture *p = &s->pic;
    PictureControlSet *pcs = s->pcs;
    FrameHeader *frm_hdr = &pcs->parent_pcs_ptr->frm_hdr;
    struct LoopFilter *lf = &frm_hdr->loop_filter_params;
    const long long td0 = svt_hip_hooks_now_ns();
    if (ensure_src(hip, s) != EB_ErrorNone) return EB_ErrorUndefined;
    if (!(s->flags & ST_RECON) && upload(hip, p, recon_of(pcs, p->pix_bytes == 2), p->d_recon, 0) != EB_ErrorNone) return EB_ErrorUndefined;
    s->flags |= ST_RECON;   /* the search filters into d_cdef: d_recon stays the picture as coded, which svt_av1_loop_filter_frame's hook starts from */
    const long long td1 = svt_hip_hooks_now_ns();
    fill_mode_info(p, pcs, 1);
    s->flags |= ST_SKIP4;
    const long long td2 = svt_hip_hooks_now_ns();
    long long t_edges = 0, t_search = 0;
    int best[3];
    const int last[4] = {lf->filter_level[0], lf->filter_level[1], lf->filter_level_u, lf->filter_level_v};
    s->flags &= ~ST_MI;
    if (build_edges(hip, p, 7, 1, NULL) != EB_ErrorNone) return EB_ErrorUndefined;
    if (!edges_on_host()) s->flags |= ST_MI;
    t_edges = svt_hip_hooks_now_ns() - td2;
    {   /* the three planes' searches are independent (svt_av1_pick_filter_level :1281-1300 runs them one after the other): advanced in lockstep, the probes each walk needs
         * next measured in one round trip (svt_hip_dlf_search_levels_picture_dev; SVT_HIP_DLF_SEARCH=planes: one plane after the other, a round trip per probe).  Scratch:
         * d_cdef and d_dbl, both free until the deblocking hook runs */
        const long long te1 = svt_hip_hooks_now_ns();
        static int per_plane = -1;
        if (per_plane < 0) { const char *e = getenv("SVT_HIP_DLF_SEARCH"); per_plane = e && !strcmp(e, "planes"); }
        SvtHipDlfSearchPlane sp[3];
        int64_t err[3] = {0, 0, 0};
        memset(sp, 0, sizeof(sp));
        for (int pl = 0; pl < 3; pl++) {
            SvtHipDlfSearch *q = &sp[pl].q;
            q->plane = pl; q->dir = 2; q->other_level = 0;
            q->start_level = pl == 0 ? last[2] : last[pl + 1];           /* search_filter_level :1044-1049 with dir = 2 / 0 / 0 */
            q->loop_filter_mode = pcs->parent_pcs_ptr->loop_filter_mode;
            q->tx_mode_only_4x4 = frm_hdr->tx_mode == ONLY_4X4;
            q->sharpness = 0;                                            /* lf->sharpness_level = 0 (:1202) */
            sp[pl].d_recon = plane_origin(p, p->d_recon[pl], pl); sp[pl].d_tmp[0] = plane_origin(p, p->d_cdef[pl], pl); sp[pl].d_tmp[1] = plane_origin(p, p->d_dbl[pl], pl);
            sp[pl].stride = p->stride[pl]; sp[pl].plane_w = p->w >> (pl > 0); sp[pl].plane_h = p->h >> (pl > 0);
            sp[pl].d_src = p->src[pl]; sp[pl].src_stride = p->src_st[pl];
            sp[pl].d_edges_v = p->d_edges[pl][0]; sp[pl].d_edges_h = p->d_edges[pl][1]; sp[pl].units_w = p->units_w[pl]; sp[pl].units_h = p->units_h[pl];
        }
        if (!per_plane) HIP_TRY(svt_hip_dlf_search_levels_picture_dev(hip, 3, sp, p->pix_bytes, p->bd, p->d_sse, best, err));
        else
            for (int pl = 0; pl < 3; pl++)
                HIP_TRY(svt_hip_dlf_search_level_dev(hip, &sp[pl].q, sp[pl].d_recon, sp[pl].d_tmp[0], p->pix_bytes, sp[pl].stride, p->bd, sp[pl].plane_w, sp[pl].plane_h, sp[pl].d_src,
                                                     sp[pl].src_stride, sp[pl].d_edges_v, sp[pl].d_edges_h, sp[pl].units_w, sp[pl].units_h, p->d_sse, &best[pl], &err[pl]));
        t_search += svt_hip_hooks_now_ns() - te1;
        for (int pl = 0; pl < 3; pl++) svt_hip_hooks_log("dlf_search: plane %d start %d -> level %d (sse %lld)", pl, sp[pl].q.start_level, best[pl], (long long)err[pl]);
    }
    svt_hip_hooks_log("dlf_search: picture up %.2f ms, mode info (host) %.2f ms, edges %.2f ms, probes %.2f ms", (td1 - td0) / 1e6, (td2 - td1) / 1e6, t_edges / 1e6, t_search / 1e6);
    lf->sharpness_level = 0;
    lf->filter_level[0] = lf->filter_level[1] = best[0];
    lf->filter_level_u = best[1];
    lf->filter_level_v = best[2];
    return EB_ErrorNone;
}

EbErrorType svt_hip_hook_dlf_pick_level(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_DLF_SEARCH)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 1, &hip);
    EbErrorType rc = s ? dlf_pick_level(hip, s) : EB_ErrorUndefined;
    if (s && rc == EB_ErrorNone && lf_fault("dlf_search")) rc = EB_ErrorUndefined;
    if (s) lf_leave(s);
    svt_hip_hooks_count(SVT_HIP_HOOK_DLF_SEARCH, rc == EB_ErrorNone);
    svt_hip_hooks_time(SVT_HIP_HOOK_DLF_SEARCH, t0);
    return rc;
}

/* svt_av1_loop_filter_frame(recon, pcs, 0, 3) (EbDeblockingFilter.c:711) */
static EbErrorType dlf_frame(SvtHipCtx *hip, LfState *s, EbPictureBufferDesc *recon) {
    SvtHipLfPicture *p = &s->pic;
    PictureControlSet *pcs = s->pcs;
    FrameHeader *frm_hdr = &pcs->parent_pcs_ptr->frm_hdr;
    const struct LoopFilter *lf = &frm_hdr->loop_filter_params;
    svt_av1_loop_filter_frame_init(frm_hdr, &pcs->parent_pcs_ptr->lf_info, 0, 3);       /* svt_av1_loop_filter_frame does this first (:722) */
    /* loop_filter_sb (:636-646): both luma levels 0 -> `break`, NO plane is filtered; a chroma plane with level 0 is skipped */
    if (!lf->filter_level[0] && !lf->filter_level[1]) {
        if (s->flags & ST_RECON) s->flags |= ST_DBL;   /* the picture as coded IS the deblocked picture, and it is on the device already */
        return EB_ErrorNone;
    }
    if (!(s->flags & ST_RECON) && upload(hip, p, recon, p->d_recon, 0) != EB_ErrorNone) return EB_ErrorUndefined;
    s->flags &= ~ST_RECON;
    void *pl_ptr[3]; const uint16_t *ev[3], *eh[3];
    int mask = 0;
    for (int pl = 0; pl < 3; pl++) {
        const int on = pl == 0 || (pl == 1 ? lf->filter_level_u : lf->filter_level_v);
        pl_ptr[pl] = on ? plane_origin(p, p->d_recon[pl], pl) : NULL;
        ev[pl] = p->d_edges[pl][0]; eh[pl] = p->d_edges[pl][1];
        if (on) mask |= 1 << pl;
    }
    /* the level search of this picture left its grid on the device, and the frame's levels are one number per plane and direction: the geometry is the same, the
     * levels ride along as arguments (no second pass over the mode info, no upload) */
    const long long te0 = svt_hip_hooks_now_ns();
    int level[3][2];
    const int reuse = (s->flags & ST_MI) && uniform_levels(pcs, mask, level);
    if (!reuse) { fill_mode_info(p, pcs, 0); s->flags |= ST_SKIP4; }
    s->flags &= ~ST_MI;
    if (build_edges(hip, p, mask, !reuse, reuse ? (const int (*)[2])level : NULL) != EB_ErrorNone) return EB_ErrorUndefined;
    const long long te1 = svt_hip_hooks_now_ns();
    static int fused = -1;
    if (fused < 0) fused = !(getenv("SVT_HIP_DLF_FUSED") && !atoi(getenv("SVT_HIP_DLF_FUSED")));
    if (fused) {   /* both directions of all planes in one out-of-place launch; the result takes the place of the picture as coded */
        const void *in[3]; void *out[3]; int pw[3], ph[3];
        for (int pl = 0; pl < 3; pl++) { in[pl] = pl_ptr[pl]; out[pl] = plane_origin(p, p->d_dbl[pl], pl); pw[pl] = p->w >> (pl > 0); ph[pl] = p->h >> (pl > 0); }
        HIP_TRY(svt_hip_deblock_frame_fused_dev(hip, in, out, p->pix_bytes, p->stride, p->bd, pw, ph, ev, eh, p->units_w, p->units_h, lf->sharpness_level));
        for (int pl = 0; pl < 3; pl++)
            if (mask & (1 << pl)) { void *t = p->d_recon[pl]; p->d_recon[pl] = p->d_dbl[pl]; p->d_dbl[pl] = t; }
    } else
        HIP_TRY(svt_hip_deblock_frame_dev(hip, pl_ptr, p->pix_bytes, p->stride, p->bd, ev, eh, p->units_w, p->units_h, lf->sharpness_level));
    if (s->defer) s->flags |= ST_HOST_STALE;   /* the deblocked picture stays on the device (svt_hip_hook_picture_done brings the final one back) */
    else {
        tls_download_started = 0;
        if (download(hip, p, p->d_recon, recon, mask, 0) != EB_ErrorNone) {
            if (!tls_download_started) return EB_ErrorUndefined;   /* nothing was queued: the host picture is the coded one, the C filter takes over */
            /* copies into the host picture were queued: it may hold deblocked samples already, and the C filter must not filter those twice — a second complete
             * pass makes it whole; if the device cannot deliver that either, the encoder stops (lf_final_download explains) */
            (void)svt_hip_sync(hip);
            __sync_fetch_and_add(&g_lf_final_retries, 1);
            if (download(hip, p, p->d_recon, recon, mask, 0) != EB_ErrorNone) {
                SVT_LOG("svt_hip: a deblocked picture could not be brought back from the device (%s) - stopping the encoder\n", svt_hip_last_error(hip));
                __sync_fetch_and_add(&g_lf_final_fatal, 1);
                const SequenceControlSet *scs = (const SequenceControlSet *)pcs->scs_wrapper_ptr->object_ptr;
                if (scs->encode_context_ptr && scs->encode_context_ptr->app_callback_ptr && scs->encode_context_ptr->app_callback_ptr->error_handler)
                    scs->encode_context_ptr->app_callback_ptr->error_handler(scs->encode_context_ptr->app_callback_ptr->handle, (uint32_t)EB_ErrorUndefined);
            }
        }
    }
    s->flags |= ST_DBL;
    svt_hip_hooks_log("dlf: levels %d %d %d %d, planes %d, edges %.2f ms (%s)", lf->filter_level[0], lf->filter_level[1], lf->filter_level_u, lf->filter_level_v, mask,
                      (te1 - te0) / 1e6, reuse ? "the level search's grid" : "mode info refilled");
    return EB_ErrorNone;
}
EbErrorType svt_hip_hook_dlf_frame(EbPictureBufferDesc *recon, PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_DLF)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 1, &hip);
    EbErrorType rc = s ? dlf_frame(hip, s, recon) : EB_ErrorUndefined;
    if (s && rc != EB_ErrorNone) { s->flags &= ~(ST_RECON | ST_DBL | ST_HOST_STALE); s->defer = 0; }   /* the C filter runs on the host picture, which nothing has touched (dlf_frame's download only reports a failure from before its first copy) */
    if (s) lf_leave(s);
    svt_hip_hooks_count(SVT_HIP_HOOK_DLF, rc == EB_ErrorNone);
    svt_hip_hooks_time(SVT_HIP_HOOK_DLF, t0);
    return rc;
}

/* dlf_kernel, when the deblocked picture is final: the later hooks need it on the device (CDEF input; the stripe context rows of the
 * restoration filters come from the deblocked, pre-CDEF picture, which the host overwrites in place). */
void svt_hip_hook_after_dlf(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_CDEF_SEARCH) && !svt_hip_hook_enabled(SVT_HIP_HOOK_CDEF_APPLY) && !svt_hip_hook_enabled(SVT_HIP_HOOK_SGR_SEARCH) &&
        !svt_hip_hook_enabled(SVT_HIP_HOOK_REST_APPLY) && !svt_hip_hook_enabled(SVT_HIP_HOOK_WIENER_STATS) && !svt_hip_hook_enabled(SVT_HIP_HOOK_WIENER_TRY) && !svt_hip_hook_enabled(SVT_HIP_HOOK_WIENER_SEARCH))
        return;
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 1, &hip);
    if (!s) return;
    if (!(s->flags & ST_DBL)) {   /* deblocked by the C code (hook off / not handled), or not at all: the host picture is the current one */
        s->flags &= ~ST_RECON;
        if (upload(hip, &s->pic, recon_of(pcs, s->pic.pix_bytes == 2), s->pic.d_recon, 0) == EB_ErrorNone) s->flags |= ST_DBL;
        else s->defer = 0;
    }
    lf_leave(s);
}

/* ---------------------------------------------------------------- CDEF ---------------------------------------------------------------- */
static void fill_skip8(SvtHipLfPicture *p, PictureControlSet *pcs, int have_skip4) {     /* is_8x8_block_skip (EbEncCdef.c:242-250) for every 8x8 block */
    const int c8 = p->w / 8, r8 = p->h / 8, mi_cols = (p->w + 3) / 4;
    if (have_skip4) {   /* the deblocking hooks' mode-info pass of this picture left the flag of every 4 x 4 unit in a dense map (one look at the mode info per coded block) */
        for (int r = 0; r < r8; r++) {
            const uint8_t *u0 = &p->h_skip4[(size_t)(2 * r) * mi_cols], *u1 = u0 + mi_cols;
            for (int c = 0; c < c8; c++) p->h_skip8[r * c8 + c] = (uint8_t)(u0[2 * c] & u0[2 * c + 1] & u1[2 * c] & u1[2 * c + 1]);
        }
        return;
    }
    for (int r = 0; r < r8; r++)
        for (int c = 0; c < c8; c++) {
            int skip = 1;
            for (int y = 0; y < 2; y++)
                for (int x = 0; x < 2; x++) skip &= (int)pcs->mi_grid_base[(2 * r + y) * pcs->mi_stride + 2 * c + x]->mbmi.block_mi.skip;
            p->h_skip8[r * c8 + c] = (uint8_t)skip;
        }
}

/* all cdef_seg_search[16bit] calls of the picture (EbCdefProcess.c:80-475): pcs->mse_seg[pli][fb][gi] for the strengths of the picture's
 * pick method (gi indexes the REDUCED strength list, get_cdef_filter_strengths, Common/Codec/EbDefinitions.h:1696) */
static EbErrorType cdef_search(SvtHipCtx *hip, LfState *s) {
    SvtHipLfPicture *p = &s->pic;
    PictureControlSet *pcs = s->pcs;
    if (!(s->flags & ST_DBL) || ensure_src(hip, s) != EB_ErrorNone) return EB_ErrorUndefined;
    const int nfb = ((p->w + 63) / 64) * ((p->h + 63) / 64);
    const int pri_damping = 3 + (pcs->parent_pcs_ptr->frm_hdr.quantization_params.base_q_idx >> 6);   /* EbCdefProcess.c:121 */
    fill_skip8(p, pcs, (s->flags & ST_SKIP4) != 0);
    svt_hip_hooks_log("cdef_search: %d x %d, %d filter blocks, primary damping %d", p->w, p->h, nfb, pri_damping);
    HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_skip8, p->h_skip8, (size_t)(p->w / 8) * (p->h / 8)));
    const void *rec[3], *src[3];
    for (int pl = 0; pl < 3; pl++) { rec[pl] = plane_origin(p, p->d_recon[pl], pl); src[pl] = p->src[pl]; }
    HIP_TRY(svt_hip_cdef_search_frame_dev(hip, p->pix_bytes, rec, p->stride, src, p->src_st, p->w, p->h, p->d_skip8, pri_damping, p->bd, p->d_mse, p->d_dir, p->d_var));
    HIP_TRY(svt_hip_memcpy_d2h(hip, p->h_mse, p->d_mse, sizeof(uint64_t) * 2 * nfb * 64));
    const int level = pcs->parent_pcs_ptr->cdef_level;
    const CDEF_PICK_METHOD pick = level == 2 ? CDEF_FAST_SEARCH_LVL1 : level == 3 ? CDEF_FAST_SEARCH_LVL2 : level == 4 ? CDEF_FAST_SEARCH_LVL3 : 0;
    const int c8 = p->w / 8, nhfb = (p->w + 63) / 64, nvfb = (p->h + 63) / 64;
    for (int fb = 0; fb < nfb; fb++) {
        const int fbr = fb / nhfb, fbc = fb % nhfb;
        /* 128 x 128 superblocks: a filter block whose first mode-info belongs to an unsplit 128-wide / 128-high block is searched together with its
         * right / lower neighbour -- one distortion entry, kept in the first of them, for the blocks of the whole 128 x 128 (128 x 64, 64 x 128) area,
         * and the others are passed over (cdef_seg_search, EbCdefProcess.c:181-199; svt_sb_compute_cdef_list with that block size).  The distortion
         * of a list of 8 x 8 blocks is the sum over its blocks, so the merged entry is the sum of the device table's 64 x 64 entries. */
        const ModeInfo *mi0 = pcs->mi_grid_base[MI_SIZE_64X64 * fbr * pcs->mi_stride + MI_SIZE_64X64 * fbc];
        const BlockSize bt = mi0->mbmi.block_mi.sb_type;
        if (((fbc & 1) && (bt == BLOCK_128X128 || bt == BLOCK_128X64)) || ((fbr & 1) && (bt == BLOCK_128X128 || bt == BLOCK_64X128))) continue;
        const int hb_step = (bt == BLOCK_128X128 || bt == BLOCK_128X64) ? 2 : 1, vb_step = (bt == BLOCK_128X128 || bt == BLOCK_64X128) ? 2 : 1;
        /* the reference leaves the entries of an all-skip filter block untouched (svt_sb_all_skip of the FIRST 64 x 64, :201): so do we */
        int all_skip = 1;
        for (int r = fbr * 8; r < fbr * 8 + 8 && r < p->h / 8 && all_skip; r++)
            for (int c = fbc * 8; c < fbc * 8 + 8 && c < c8; c++) all_skip &= p->h_skip8[r * c8 + c];
        if (all_skip) continue;
        for (int gi = 0; gi < nb_cdef_strengths[pick]; gi++) {
            int pri = gi / CDEF_SEC_STRENGTHS, sec = gi % CDEF_SEC_STRENGTHS;
            get_cdef_filter_strengths(pick, &pri, &sec, gi);
            uint64_t m0 = 0, m1 = 0;
            for (int dy = 0; dy < vb_step && fbr + dy < nvfb; dy++)
                for (int dx = 0; dx < hb_step && fbc + dx < nhfb; dx++) {
                    const size_t q = (size_t)(fbr + dy) * nhfb + fbc + dx;
                    m0 += p->h_mse[q * 64 + pri * CDEF_SEC_STRENGTHS + sec];
                    m1 += p->h_mse[((size_t)nfb + q) * 64 + pri * CDEF_SEC_STRENGTHS + sec];
                }
            pcs->mse_seg[0][fb][gi] = m0;
            pcs->mse_seg[1][fb][gi] = m1;
        }
    }
    s->flags |= ST_DIRVAR;
    svt_hip_hooks_log("cdef_search: distortion table of the picture is in pcs->mse_seg");
    return EB_ErrorNone;
}
/* called by every segment of the picture: the first one to arrive searches the whole picture, the others find it done */
EbErrorType svt_hip_hook_cdef_search(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_CDEF_SEARCH)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 0, &hip);
    EbErrorType rc = EB_ErrorUndefined;
    if (s && (s->flags & ST_CDEF_SEARCHED)) rc = EB_ErrorNone;
    else if (s && !(s->flags & ST_CDEF_FAILED)) {
        rc = cdef_search(hip, s);
        if (rc == EB_ErrorNone && lf_fault("cdef_search")) rc = EB_ErrorUndefined;
        s->flags |= rc == EB_ErrorNone ? ST_CDEF_SEARCHED : ST_CDEF_FAILED;
        if (rc != EB_ErrorNone) lf_recover(hip, s, 0);   /* cdef_seg_search reads the deblocked picture on the host */
        svt_hip_hooks_count(SVT_HIP_HOOK_CDEF_SEARCH, rc == EB_ErrorNone);
    }
    if (s) lf_leave(s);
    svt_hip_hooks_time(SVT_HIP_HOOK_CDEF_SEARCH, t0);
    return rc;
}

/* svt_av1_cdef_frame / av1_cdef_frame16bit (EbEncCdef.c:292-1031) */
static EbErrorType cdef_apply(SvtHipCtx *hip, LfState *s) {
    SvtHipLfPicture *p = &s->pic;
    PictureControlSet *pcs = s->pcs;
    if (!(s->flags & ST_DBL)) return EB_ErrorUndefined;
    FrameHeader *frm_hdr = &pcs->parent_pcs_ptr->frm_hdr;
    const int nhfb = (p->w + 63) / 64, nvfb = (p->h + 63) / 64, nfb = nhfb * nvfb;
    uint8_t *ys = (uint8_t *)malloc(nfb), *uvs = (uint8_t *)malloc(nfb);
    if (!ys || !uvs) { free(ys); free(uvs); return EB_ErrorInsufficientResources; }
    int bad = 0;
    for (int fbr = 0; fbr < nvfb; fbr++)
        for (int fbc = 0; fbc < nhfb; fbc++) {     /* the strength index finish_cdef_search stored in the fb's first mode-info (EbEncCdef.c:1292) */
            const ModeInfo *mi = pcs->mi_grid_base[MI_SIZE_64X64 * fbr * pcs->mi_stride + MI_SIZE_64X64 * fbc];
            const int8_t idx = mi ? mi->mbmi.cdef_strength : -1;
            if (idx < 0 || idx >= CDEF_MAX_STRENGTHS) bad = 1;   /* the reference skips such a filter block with an error message (:370-377): leave it to the reference */
            ys[fbr * nhfb + fbc] = idx < 0 ? 0 : (uint8_t)frm_hdr->cdef_params.cdef_y_strength[idx];
            uvs[fbr * nhfb + fbc] = idx < 0 ? 0 : (uint8_t)frm_hdr->cdef_params.cdef_uv_strength[idx];
        }
    int rc = bad ? SVT_HIP_ERR_UNSUPPORTED : (svt_hip_memcpy_h2d(hip, p->d_y_strength, ys, nfb) | svt_hip_memcpy_h2d(hip, p->d_uv_strength, uvs, nfb));
    free(ys); free(uvs);
    if (rc != SVT_HIP_OK) return EB_ErrorUndefined;
    if (!(s->flags & ST_DIRVAR)) {   /* the search hook is off: build the skip map here; directions are computed by the apply call */
        fill_skip8(p, pcs, (s->flags & ST_SKIP4) != 0);
        HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_skip8, p->h_skip8, (size_t)(p->w / 8) * (p->h / 8)));
    }
    const void *in[3]; void *out[3];
    for (int pl = 0; pl < 3; pl++) {
        in[pl] = plane_origin(p, p->d_recon[pl], pl); out[pl] = plane_origin(p, p->d_cdef[pl], pl);   /* the apply kernel writes every sample of the picture (unfiltered blocks passed through) */
    }
    /* direction / variance of the search are reused when it ran here (same pre-CDEF picture) */
    HIP_TRY(svt_hip_cdef_apply_frame_dev(hip, p->pix_bytes, in, out, p->stride, p->w, p->h, p->d_skip8, p->d_y_strength, p->d_uv_strength,
                                         frm_hdr->cdef_params.cdef_damping, p->bd, p->d_dir, (s->flags & ST_DIRVAR) ? p->d_var : NULL));
    if (s->defer) s->flags |= ST_HOST_STALE;
    else if (download(hip, p, p->d_cdef, recon_of(pcs, p->pix_bytes == 2), 7, 0) != EB_ErrorNone) return EB_ErrorUndefined;
    s->flags |= ST_CDEF;
    return EB_ErrorNone;
}
EbErrorType svt_hip_hook_cdef_apply(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_CDEF_APPLY)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 0, &hip);
    EbErrorType rc = s ? cdef_apply(hip, s) : EB_ErrorUndefined;
    if (s && rc == EB_ErrorNone && lf_fault("cdef_apply")) { rc = EB_ErrorUndefined; s->flags &= ~(ST_CDEF | ST_PADDED); }
    if (s && rc != EB_ErrorNone) lf_recover(hip, s, 0);   /* svt_av1_cdef_frame filters the deblocked picture on the host */
    if (s) lf_leave(s);
    svt_hip_hooks_count(SVT_HIP_HOOK_CDEF_APPLY, rc == EB_ErrorNone);
    svt_hip_hooks_time(SVT_HIP_HOOK_CDEF_APPLY, t0);
    return rc;
}

/* ---------------------------------------------------------------- loop restoration ---------------------------------------------------- */
/* the CDEF output on the device (from the CDEF hook, or uploaded from the host picture) with its 3-sample border (svt_extend_frame) */
static EbErrorType ensure_cdef_padded(SvtHipCtx *hip, LfState *s) {
    SvtHipLfPicture *p = &s->pic;
    if (!(s->flags & ST_CDEF)) {   /* CDEF is off for this picture, or ran on the host */
        if (s->flags & ST_HOST_STALE) {    /* ... off: the deblocked picture, which only the device has, is the restoration input */
            for (int pl = 0; pl < 3; pl++) HIP_TRY(svt_hip_memcpy_d2d(hip, p->d_cdef[pl], p->d_recon[pl], plane_bytes(p, pl)));
        } else if (upload(hip, p, recon_of(s->pcs, p->pix_bytes == 2), p->d_cdef, 0) != EB_ErrorNone) return EB_ErrorUndefined;
        s->flags |= ST_CDEF;
    }
    if (!(s->flags & ST_PADDED)) {
        for (int pl = 0; pl < 3; pl++)
            HIP_TRY(svt_hip_generate_padding_dev(hip, plane_origin(p, p->d_cdef[pl], pl), p->pix_bytes, p->stride[pl], p->cw >> (pl > 0), p->ch >> (pl > 0), LF_BORDER, LF_BORDER));   /* svt_extend_frame of the CROPPED frame (EbCdefProcess.c:552-572): inside a padded picture it overwrites coded samples, on the host as well */
        s->flags |= ST_PADDED;
    }
    return EB_ErrorNone;
}
static int rest_geometry_ok(const SvtHipLfPicture *p, const Av1Common *cm) {
    for (int pl = 0; pl < 3; pl++) {
        const RestorationInfo *rsi = &cm->rst_info[pl];
        if (rsi->units_per_tile <= 0 || rsi->units_per_tile > p->max_units[pl] || (rsi->restoration_unit_size != 64 && rsi->restoration_unit_size != 128 && rsi->restoration_unit_size != 256))
            return 0;
    }
    return 1;   /* restoration units ignore tiles (one "tile" = the frame, EbRestoration.c:1413) */
}

/* svt_av1_loop_restoration_filter_frame(cm->frame_to_show, cm, 0) (Common/Codec/EbRestoration.c:1293) */
static EbErrorType rest_apply(SvtHipCtx *hip, LfState *s) {
    SvtHipLfPicture *p = &s->pic;
    PictureControlSet *pcs = s->pcs;
    Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
    if (!(s->flags & ST_DBL) || !rest_geometry_ok(p, cm) || ensure_cdef_padded(hip, s) != EB_ErrorNone) return EB_ErrorUndefined;
    int mask = 0;
    for (int pl = 0; pl < 3; pl++) {
        const RestorationInfo *rsi = &cm->rst_info[pl];
        const int pw = p->cw >> (pl > 0), ph = p->ch >> (pl > 0), n = rsi->units_per_tile;
        if (rsi->frame_restoration_type == RESTORE_NONE) continue;    /* the plane is left alone (:1322-1323) */
        uint8_t *ep = (uint8_t *)malloc(n); int32_t *xqd = (int32_t *)malloc(sizeof(int32_t) * 2 * n); int16_t *wn = (int16_t *)calloc((size_t)n * 16, sizeof(int16_t));
        if (!ep || !xqd || !wn) { free(ep); free(xqd); free(wn); return EB_ErrorInsufficientResources; }
        for (int u = 0; u < n; u++) {
            const RestorationUnitInfo *rui = &rsi->unit_info[u];
            ep[u] = rui->restoration_type == RESTORE_SGRPROJ ? (uint8_t)rui->sgrproj_info.ep : (rui->restoration_type == RESTORE_WIENER ? 254 : 255);
            xqd[2 * u] = rui->sgrproj_info.xqd[0]; xqd[2 * u + 1] = rui->sgrproj_info.xqd[1];
            memcpy(wn + 16 * u, rui->wiener_info.vfilter, 8 * sizeof(int16_t)); memcpy(wn + 16 * u + 8, rui->wiener_info.hfilter, 8 * sizeof(int16_t));
        }
        int rc = svt_hip_memcpy_h2d(hip, p->d_unit_ep[pl], ep, n) | svt_hip_memcpy_h2d(hip, p->d_unit_xqd[pl], xqd, sizeof(int32_t) * 2 * n) |
                 svt_hip_memcpy_h2d(hip, p->d_unit_wiener[pl], wn, sizeof(int16_t) * 16 * n);
        free(ep); free(xqd); free(wn);
        if (rc != SVT_HIP_OK) return EB_ErrorUndefined;
        HIP_TRY(svt_hip_lr_apply_plane_dev(hip, p->pix_bytes, p->bd, plane_origin(p, p->d_cdef[pl], pl), p->stride[pl], plane_origin(p, p->d_rest[pl], pl), p->stride[pl],
                                           pw, ph, rsi->restoration_unit_size, pl > 0, plane_origin(p, p->d_recon[pl], pl), p->stride[pl], p->d_unit_ep[pl],
                                           p->d_unit_xqd[pl], p->d_unit_wiener[pl]));
        mask |= 1 << pl;
    }
    if (s->defer) { s->flags |= ST_REST | ST_HOST_STALE; s->rest_mask = mask; return EB_ErrorNone; }   /* comes back with svt_hip_hook_picture_done */
    return download(hip, p, p->d_rest, recon_of(pcs, p->pix_bytes == 2), mask, 1);
}
EbErrorType svt_hip_hook_rest_apply(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_REST_APPLY)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 0, &hip);
    EbErrorType rc = s ? rest_apply(hip, s) : EB_ErrorUndefined;
    if (s && rc == EB_ErrorNone && lf_fault("rest_apply")) { rc = EB_ErrorUndefined; s->flags &= ~ST_REST; s->rest_mask = 0; }
    if (s && rc != EB_ErrorNone) lf_recover(hip, s, 1);   /* svt_av1_loop_restoration_filter_frame reads the CDEF output and both sets of boundary lines */
    if (s) lf_leave(s);
    svt_hip_hooks_count(SVT_HIP_HOOK_REST_APPLY, rc == EB_ErrorNone);
    svt_hip_hooks_time(SVT_HIP_HOOK_REST_APPLY, t0);
    return rc;
}

/* every search_sgrproj_seg call of the picture (EbRestorationPick.c:1277-1317, per unit: search_selfguided_restoration :583 +
 * try_restoration_unit_seg :137).  One svt_hip_sgr_search_units_picture call gives the (ep, xqd) of every unit of the three planes; the units are
 * then filtered with exactly those parameters (stripe rules as in svt_av1_loop_restoration_filter_unit) and their SSE against the source is what
 * try_restoration_unit_seg -> sse_restoration_unit (:58) returns.  Results land in rusi_picture[plane][unit] (sgrproj, sse[RESTORE_SGRPROJ]) and
 * cm->sg_frame_ep_cnt, i.e. where search_sgrproj_finish (:1319) and rest_finish_search read them. */
static EbErrorType sgr_search(SvtHipCtx *hip, LfState *s) {
    SvtHipLfPicture *p = &s->pic;
    PictureControlSet *pcs = s->pcs;
    Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
    if (!(s->flags & ST_DBL) || !rest_geometry_ok(p, cm) || ensure_src(hip, s) != EB_ErrorNone || ensure_cdef_padded(hip, s) != EB_ErrorNone) return EB_ErrorUndefined;
    /* the set window of search_selfguided_restoration (:596-607) */
    const int8_t step = get_sg_step(cm->sg_filter_mode);
    const int8_t *re = cm->sg_ref_frame_ep;
    const int none = re[0] < 0 && re[1] < 0;
    const int mid = none ? 0 : (re[1] < 0 ? re[0] : (re[0] < 0 ? re[1] : (re[0] + re[1]) / 2));
    const int start_ep = none ? 0 : (mid - step > 0 ? mid - step : 0), end_ep = none ? 16 : (mid + step < 16 ? mid + step : 16);
    uint32_t mask = 0;
    for (int ep = start_ep; ep < end_ep; ep++) mask |= 1u << ep;

    SvtHipSgrSearchPlane job[3];
    int32_t *xqd[3] = {0}; int64_t *err[3] = {0}; uint8_t *best[3] = {0};
    uint8_t *c_ep[3] = {0}; int32_t *c_uq[3] = {0}; uint64_t *c_sse[3] = {0};   /* per plane: the chosen set / taps / SSE of every unit, committed at the end */
    uint64_t *c_none[3] = {0};   /* search_norestore_seg (:1476): sse_restoration_unit of the unfiltered unit -- the same rectangles, the CDEF output against the source */
    EbErrorType ret = EB_ErrorNone;
    for (int pl = 0; pl < 3; pl++) {
        const RestorationInfo *rsi = &cm->rst_info[pl];
        const int pw = p->cw >> (pl > 0), ph = p->ch >> (pl > 0), n = rsi->units_per_tile;
        xqd[pl] = (int32_t *)calloc((size_t)32 * n, sizeof(int32_t)); err[pl] = (int64_t *)calloc((size_t)16 * n, sizeof(int64_t)); best[pl] = (uint8_t *)calloc(n, 1);
        if (!xqd[pl] || !err[pl] || !best[pl]) { ret = EB_ErrorInsufficientResources; goto done; }
        job[pl].d_dgd = plane_origin(p, p->d_cdef[pl], pl); job[pl].stride = p->stride[pl];
        job[pl].d_src = p->src[pl]; job[pl].src_stride = p->src_st[pl];
        job[pl].pw = pw; job[pl].ph = ph; job[pl].unit_size = rsi->restoration_unit_size; job[pl].ss_y = pl > 0;
        job[pl].ep_mask = mask; job[pl].xqd_out = xqd[pl]; job[pl].err_out = err[pl]; job[pl].best_ep = best[pl];
    }
    /* an empty window (step 0 with reference sets: the loop :609 never runs) leaves ep 0, xqd {0, 0} — the calloc'ed values */
    if (mask && svt_hip_sgr_search_units_picture(hip, p->pix_bytes, p->bd, 3, job, NULL) != SVT_HIP_OK) { ret = EB_ErrorUndefined; goto done; }

    for (int pl = 0; pl < 3; pl++) {
        const RestorationInfo *rsi = &cm->rst_info[pl];
        const int pw = p->cw >> (pl > 0), ph = p->ch >> (pl > 0), n = rsi->units_per_tile, us = rsi->restoration_unit_size;
        uint8_t *ep = (uint8_t *)malloc(n); int32_t *uq = (int32_t *)malloc(sizeof(int32_t) * 2 * n);
        SvtHipBlkPair *rect = (SvtHipBlkPair *)malloc(sizeof(SvtHipBlkPair) * n); uint64_t *sse = (uint64_t *)malloc(sizeof(uint64_t) * 2 * n);
        void *d_rect = NULL, *d_sse = NULL;
        int ok = ep && uq && rect && sse;
        if (ok) {
            /* unit rectangles of foreach_rest_unit_in_tile (EbRestoration.c:1369-1411) */
            const int ext = us * 3 / 2, voff = 8 >> (pl > 0), hunits = rsi->horz_units_per_tile;
            int y0 = 0, i = 0;
            while (y0 < ph) {
                const int rem_h = ph - y0, h = rem_h < ext ? rem_h : us;
                int v0 = y0 - voff > 0 ? y0 - voff : 0, v1 = y0 + h;
                if (v1 < ph) v1 -= voff;
                int x0 = 0, j = 0;
                while (x0 < pw) {
                    const int rem_w = pw - x0, w = rem_w < ext ? rem_w : us, u = i * hunits + j;
                    if (u >= n) { ok = 0; break; }
                    ep[u] = best[pl][u];
                    uq[2 * u] = xqd[pl][(u * 16 + ep[u]) * 2]; uq[2 * u + 1] = xqd[pl][(u * 16 + ep[u]) * 2 + 1];
                    rect[u].a_x = rect[u].b_x = x0; rect[u].a_y = rect[u].b_y = v0; rect[u].w = (uint16_t)w; rect[u].h = (uint16_t)(v1 - v0);
                    x0 += w; j++;
                }
                y0 += h; i++;
            }
            ok = ok && svt_hip_hooks_malloc(hip, &d_rect, sizeof(SvtHipBlkPair) * n) == SVT_HIP_OK && svt_hip_hooks_malloc(hip, &d_sse, sizeof(uint64_t) * 2 * n) == SVT_HIP_OK &&
                 svt_hip_memcpy_h2d(hip, p->d_unit_ep[pl], ep, n) == SVT_HIP_OK && svt_hip_memcpy_h2d(hip, p->d_unit_xqd[pl], uq, sizeof(int32_t) * 2 * n) == SVT_HIP_OK &&
                 svt_hip_memcpy_h2d(hip, d_rect, rect, sizeof(SvtHipBlkPair) * n) == SVT_HIP_OK &&
                 /* try_restoration_unit_seg: the unit filtered for real (stripe context from the deblocked picture), then its SSE */
                 svt_hip_sgr_apply_plane_dev(hip, p->pix_bytes, p->bd, plane_origin(p, p->d_cdef[pl], pl), p->stride[pl], plane_origin(p, p->d_rest[pl], pl), p->stride[pl],
                                             pw, ph, us, pl > 0, plane_origin(p, p->d_recon[pl], pl), p->stride[pl], p->d_unit_ep[pl], p->d_unit_xqd[pl]) == SVT_HIP_OK &&
                 svt_hip_block_sse_batch_dev(hip, p->pix_bytes, p->src[pl], p->src_st[pl], plane_origin(p, p->d_rest[pl], pl), p->stride[pl],
                                             (const SvtHipBlkPair *)d_rect, n, (uint64_t *)d_sse) == SVT_HIP_OK &&
                 svt_hip_block_sse_batch_dev(hip, p->pix_bytes, p->src[pl], p->src_st[pl], plane_origin(p, p->d_cdef[pl], pl), p->stride[pl],
                                             (const SvtHipBlkPair *)d_rect, n, (uint64_t *)d_sse + n) == SVT_HIP_OK &&
                 svt_hip_memcpy_d2h(hip, sse, d_sse, sizeof(uint64_t) * 2 * n) == SVT_HIP_OK;
        }
        if (d_rect) svt_hip_hooks_free(hip, d_rect);
        if (d_sse) svt_hip_hooks_free(hip, d_sse);
        free(rect);
        c_ep[pl] = ep; c_uq[pl] = uq; c_sse[pl] = sse; c_none[pl] = sse ? sse + n : NULL;   /* committed below, once every plane has succeeded */
        if (!ok) { ret = EB_ErrorUndefined; goto done; }
    }
    /* every device step of the hook has succeeded: only now do the reference's objects change (a failure above leaves rusi and cm->sg_frame_ep_cnt as they
     * were, so the per-unit C search that then runs counts every unit exactly once) */
    for (int pl = 0; pl < 3; pl++) {
        RestUnitSearchInfo *rusi = pcs->parent_pcs_ptr->rusi_picture[pl];
        const int           n = cm->rst_info[pl].units_per_tile;
        for (int u = 0; u < n; u++) {
            rusi[u].sgrproj.ep = c_ep[pl][u]; rusi[u].sgrproj.xqd[0] = c_uq[pl][2 * u]; rusi[u].sgrproj.xqd[1] = c_uq[pl][2 * u + 1];
            rusi[u].sse[RESTORE_SGRPROJ] = (int64_t)c_sse[pl][u];
            rusi[u].sse[RESTORE_NONE] = (int64_t)c_none[pl][u];
            cm->sg_frame_ep_cnt[c_ep[pl][u]]++;
        }
    }
done:
    for (int pl = 0; pl < 3; pl++) { free(xqd[pl]); free(err[pl]); free(best[pl]); free(c_ep[pl]); free(c_uq[pl]); free(c_sse[pl]); }
    return ret;
}
/* called at the top of every restoration_seg_search of the picture: the first segment to arrive searches all units */
EbErrorType svt_hip_hook_sgr_search(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_SGR_SEARCH)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 0, &hip);
    EbErrorType rc = EB_ErrorUndefined;
    if (s && (s->flags & ST_SGR_DONE)) rc = EB_ErrorNone;
    else if (s && !(s->flags & ST_SGR_FAILED)) {
        rc = lf_fault("sgr_search") ? EB_ErrorUndefined : sgr_search(hip, s);
        s->flags |= rc == EB_ErrorNone ? ST_SGR_DONE : ST_SGR_FAILED;
        if (rc != EB_ErrorNone) lf_recover(hip, s, 1);   /* the per-unit C search reads the CDEF output, its copy and the boundary lines on the host */
        svt_hip_hooks_count(SVT_HIP_HOOK_SGR_SEARCH, rc == EB_ErrorNone);
    }
    if (s) lf_leave(s);
    svt_hip_hooks_time(SVT_HIP_HOOK_SGR_SEARCH, t0);
    return rc;
}

/* svt_av1_compute_stats[_highbd] of search_wiener_seg (EbRestorationPick.c:1347): the first call of a picture runs one statistics pass per
 * plane over all units; every call copies its unit's M / H. */
/* svt_av1_compute_stats of every unit of one plane into fresh device blocks: *dM [unit][win^2], *dH [unit][win^4]; the caller frees them once the context is drained */
static EbErrorType wiener_stats_plane(SvtHipCtx *hip, LfState *s, int pl, void **dM, void **dH) {
    SvtHipLfPicture *p = &s->pic;
    const Av1Common *cm = s->pcs->parent_pcs_ptr->av1_cm;
    const RestorationInfo *rsi = &cm->rst_info[pl];
    const int wn_luma = cm->wn_filter_mode == 1 ? WIENER_WIN_3TAP : cm->wn_filter_mode == 2 ? WIENER_WIN_CHROMA : WIENER_WIN;
    const int win = cm->wn_filter_mode == 1 ? WIENER_WIN_3TAP : (pl == 0 ? wn_luma : WIENER_WIN_CHROMA);   /* search_wiener_seg :1352-1358 */
    const int n = rsi->units_per_tile, w2 = win * win;
    p->wiener_win[pl] = win;
    *dM = *dH = NULL;
    if (svt_hip_hooks_malloc(hip, dM, sizeof(int64_t) * n * w2) != SVT_HIP_OK || svt_hip_hooks_malloc(hip, dH, sizeof(int64_t) * n * w2 * w2) != SVT_HIP_OK) return EB_ErrorInsufficientResources;
    HIP_TRY(svt_hip_wiener_stats_plane_dev(hip, p->pix_bytes, p->bd, win, plane_origin(p, p->d_cdef[pl], pl), p->stride[pl], p->src[pl], p->src_st[pl],
                                           p->cw >> (pl > 0), p->ch >> (pl > 0), rsi->restoration_unit_size, pl > 0, (int64_t *)*dM, (int64_t *)*dH));
    return EB_ErrorNone;
}
static EbErrorType wiener_stats_all(SvtHipCtx *hip, LfState *s) {
    SvtHipLfPicture *p = &s->pic;
    const Av1Common *cm = s->pcs->parent_pcs_ptr->av1_cm;
    if (!rest_geometry_ok(p, cm) || ensure_src(hip, s) != EB_ErrorNone || ensure_cdef_padded(hip, s) != EB_ErrorNone) return EB_ErrorUndefined;
    for (int pl = 0; pl < 3; pl++) {
        void *dM = NULL, *dH = NULL;
        int ok = wiener_stats_plane(hip, s, pl, &dM, &dH) == EB_ErrorNone;
        const int n = cm->rst_info[pl].units_per_tile, w2 = p->wiener_win[pl] * p->wiener_win[pl];
        free(p->h_wiener_M[pl]); free(p->h_wiener_H[pl]);
        p->h_wiener_M[pl] = (int64_t *)malloc(sizeof(int64_t) * n * w2); p->h_wiener_H[pl] = (int64_t *)malloc(sizeof(int64_t) * n * w2 * w2);
        ok = ok && p->h_wiener_M[pl] && p->h_wiener_H[pl] && svt_hip_memcpy_d2h(hip, p->h_wiener_M[pl], dM, sizeof(int64_t) * n * w2) == SVT_HIP_OK &&
             svt_hip_memcpy_d2h(hip, p->h_wiener_H[pl], dH, sizeof(int64_t) * n * w2 * w2) == SVT_HIP_OK;
        if (!ok) (void)svt_hip_sync(hip);   /* a launch may still be writing the blocks */
        if (dM) svt_hip_hooks_free(hip, dM);
        if (dH) svt_hip_hooks_free(hip, dH);
        if (!ok) return EB_ErrorUndefined;
    }
    return EB_ErrorNone;
}
EbErrorType svt_hip_hook_wiener_stats(PictureControlSet *pcs, int plane, int wiener_win, int unit_idx, int64_t *M, int64_t *H) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_WIENER_STATS)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 0, &hip);
    EbErrorType rc = EB_ErrorUndefined;
    if (s && !(s->flags & (ST_WIENER_DONE | ST_WIENER_FAILED))) {
        rc = wiener_stats_all(hip, s);
        s->flags |= rc == EB_ErrorNone ? ST_WIENER_DONE : ST_WIENER_FAILED;
        svt_hip_hooks_count(SVT_HIP_HOOK_WIENER_STATS, rc == EB_ErrorNone);
    }
    if (s && (s->flags & ST_WIENER_DONE) && s->pic.wiener_win[plane] == wiener_win && unit_idx >= 0 &&
        unit_idx < pcs->parent_pcs_ptr->av1_cm->rst_info[plane].units_per_tile) {
        const int w2 = wiener_win * wiener_win;
        memcpy(M, s->pic.h_wiener_M[plane] + (size_t)unit_idx * w2, sizeof(int64_t) * w2);
        memcpy(H, s->pic.h_wiener_H[plane] + (size_t)unit_idx * w2 * w2, sizeof(int64_t) * w2 * w2);
        rc = EB_ErrorNone;
    } else
        rc = EB_ErrorUndefined;
    if (s) lf_leave(s);
    svt_hip_hooks_time(SVT_HIP_HOOK_WIENER_STATS, t0);
    return rc;
}

/* try_restoration_unit_seg for a RESTORE_WIENER candidate (EbRestorationPick.c:137; the probe of finer_tile_search_wiener_seg, :1092): the unit is filtered
 * on the device with the probed taps (stripe context rows from the deblocked picture kept there) and only its SSE comes back. */
static EbErrorType wiener_try(SvtHipCtx *hip, LfState *s, int pl, int h_start, int h_end, int v_start, int v_end, const WienerInfo *wi, int64_t *err) {
    SvtHipLfPicture *p = &s->pic;
    const Av1Common *cm = s->pcs->parent_pcs_ptr->av1_cm;
    if (!(s->flags & ST_DBL) || !rest_geometry_ok(p, cm) || ensure_src(hip, s) != EB_ErrorNone || ensure_cdef_padded(hip, s) != EB_ErrorNone) return EB_ErrorUndefined;
    const RestorationInfo *rsi = &cm->rst_info[pl];
    const int pw = p->cw >> (pl > 0), ph = p->ch >> (pl > 0), us = rsi->restoration_unit_size, n = rsi->units_per_tile, voff = 8 >> (pl > 0);
    /* unit index from its rectangle: columns start at j * us, rows at i * us - voff (0 for the first row), EbRestoration.c:1369-1411 */
    const int j = h_start / us, i = v_start > 0 ? (v_start + voff) / us : 0, u = i * rsi->horz_units_per_tile + j;
    if (u < 0 || u >= n || h_end <= h_start || v_end <= v_start) return EB_ErrorUndefined;
    uint8_t ep = 254;   /* RESTORE_WIENER */
    int16_t wn[16];
    memcpy(wn, wi->vfilter, 8 * sizeof(int16_t)); memcpy(wn + 8, wi->hfilter, 8 * sizeof(int16_t));
    uint64_t sse = 0;
    HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_unit_ep[pl] + u, &ep, 1));
    HIP_TRY(svt_hip_memcpy_h2d(hip, p->d_unit_wiener[pl] + 16 * u, wn, sizeof(wn)));
    HIP_TRY(svt_hip_lr_try_unit_dev(hip, p->pix_bytes, p->bd, plane_origin(p, p->d_cdef[pl], pl), p->stride[pl], plane_origin(p, p->d_rest[pl], pl), p->stride[pl], pw, ph, us,
                                    pl > 0, plane_origin(p, p->d_recon[pl], pl), p->stride[pl], p->d_unit_ep[pl], p->d_unit_xqd[pl], p->d_unit_wiener[pl], p->src[pl],
                                    p->src_st[pl], u, p->d_sse));
    HIP_TRY(svt_hip_memcpy_d2h(hip, &sse, p->d_sse, sizeof(sse)));
    *err = (int64_t)sse;
    return EB_ErrorNone;
}
EbErrorType svt_hip_hook_wiener_try(PictureControlSet *pcs, int plane, int h_start, int h_end, int v_start, int v_end, const WienerInfo *wi, int64_t *err) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_WIENER_TRY)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 0, &hip);
    EbErrorType rc = s ? wiener_try(hip, s, plane, h_start, h_end, v_start, v_end, wi, err) : EB_ErrorUndefined;
    if (s) lf_leave(s);
    svt_hip_hooks_count(SVT_HIP_HOOK_WIENER_TRY, rc == EB_ErrorNone);
    svt_hip_hooks_time(SVT_HIP_HOOK_WIENER_TRY, t0);
    return rc;
}

/* ---- every search_wiener_seg of the picture at once (EbRestorationPick.c:1347-1423) -------------------------------------------------
 * Statistics of all units in one launch per plane, initial filters by the reference's own decomposition (svt_hip_wiener_unit_init, in the patched
 * file), then finer_tile_search_wiener_seg (:1092) of every unit ON THE DEVICE: svt_hip_wiener_walk_units_dev runs each unit's coordinate descent and
 * all of its probes (the unit filtered with the probed taps, squared error against the source) in one launch per plane — no host round trip per probe
 * (round 3: 20 - 40 lockstep rounds per picture, each an upload, a launch and a download per plane). */
static int wiener_init_on_host(void) {
    static int v = -1;
    if (v < 0) { const char *e = getenv("SVT_HIP_WIENER_INIT"); v = e && !strcmp(e, "host"); }
    return v;
}
static EbErrorType wiener_search(SvtHipCtx *hip, LfState *s) {
    SvtHipLfPicture *p = &s->pic;
    PictureControlSet *pcs = s->pcs;
    const Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
    if (!(s->flags & ST_DBL)) return EB_ErrorUndefined;
    const long long tw0 = svt_hip_hooks_now_ns();
    /* the initial filters on the device (svt_hip_wiener_init_units_dev): the statistics never leave it — statistics, initial filters and walks are queued back to back and
     * the host waits once, for the results.  With the statistics on the host already (the per-unit hook ran for this picture) or SVT_HIP_WIENER_INIT=host: the reference's
     * own decomposition on downloaded statistics (svt_hip_wiener_unit_init in the patched file; 2.9 ms per 3840 x 2160 picture on one thread, plus the 4 MB download). */
    const int dev_init = !(s->flags & ST_WIENER_DONE) && !wiener_init_on_host();
    if (!dev_init && !(s->flags & ST_WIENER_DONE)) {
        if ((s->flags & ST_WIENER_FAILED) || wiener_stats_all(hip, s) != EB_ErrorNone) { s->flags |= ST_WIENER_FAILED; return EB_ErrorUndefined; }
        s->flags |= ST_WIENER_DONE;
    }
    if (dev_init && (!rest_geometry_ok(p, cm) || ensure_src(hip, s) != EB_ErrorNone || ensure_cdef_padded(hip, s) != EB_ErrorNone)) return EB_ErrorUndefined;
    const long long tw1 = svt_hip_hooks_now_ns();
    long long t_init = 0, t_walk = 0;
    EbErrorType ret = EB_ErrorNone;
    uint8_t *act[3] = {0}; int16_t *wn[3] = {0}; int64_t *err[3] = {0}; uint32_t *probes[3] = {0}; int8_t *init[3] = {0};
    void *d_act[3] = {0}, *d_err[3] = {0}, *d_probes[3] = {0}, *d_init[3] = {0}, *dM[3] = {0}, *dH[3] = {0};
    long n_probes = 0, n_walks = 0;
    SvtHipWienerWalkPlane walk[3];
    int walk_pl[3], n_walk_planes = 0;
    for (int pl = 0; pl < 3 && ret == EB_ErrorNone; pl++) {
        const RestorationInfo *rsi = &cm->rst_info[pl];
        const int pw = p->cw >> (pl > 0), ph = p->ch >> (pl > 0), n = rsi->units_per_tile;
        act[pl] = (uint8_t *)calloc(n, 1); wn[pl] = (int16_t *)calloc((size_t)n * 16, sizeof(int16_t)); err[pl] = (int64_t *)calloc(n, sizeof(int64_t));
        probes[pl] = (uint32_t *)calloc(n, sizeof(uint32_t)); init[pl] = (int8_t *)calloc(n, 1);
        if (!act[pl] || !wn[pl] || !err[pl] || !probes[pl] || !init[pl] || svt_hip_hooks_malloc(hip, &d_act[pl], n) != SVT_HIP_OK || svt_hip_hooks_malloc(hip, &d_init[pl], n) != SVT_HIP_OK ||
            svt_hip_hooks_malloc(hip, &d_err[pl], sizeof(int64_t) * n) != SVT_HIP_OK || svt_hip_hooks_malloc(hip, &d_probes[pl], sizeof(uint32_t) * n) != SVT_HIP_OK) { ret = EB_ErrorInsufficientResources; break; }
        int any = 0;
        const long long ti0 = svt_hip_hooks_now_ns();
        if (dev_init) {
            if (wiener_stats_plane(hip, s, pl, &dM[pl], &dH[pl]) != EB_ErrorNone ||
                svt_hip_wiener_init_units_dev(hip, p->wiener_win[pl], n, (const int64_t *)dM[pl], (const int64_t *)dH[pl], p->d_unit_wiener[pl], (uint8_t *)d_act[pl], (int8_t *)d_init[pl]) != SVT_HIP_OK) {
                ret = EB_ErrorUndefined; break;
            }
            any = n > 0;   /* which units walk is known on the device only: units whose initial filter loses to the identity are skipped by the walk kernel (active = 0) */
        } else {
            /* search_wiener_seg up to the refinement: decomposition, tap quantisation, score against the identity filter (the reference's own code) */
            const int win = p->wiener_win[pl], w2 = win * win;
            for (int u = 0; u < n; u++) {
                WienerInfo wi;
                memset(&wi, 0, sizeof(wi));
                const int r = svt_hip_wiener_unit_init(win, p->h_wiener_M[pl] + (size_t)u * w2, p->h_wiener_H[pl] + (size_t)u * w2 * w2, &wi);
                init[pl][u] = (int8_t)r;
                if (r == 1) {
                    act[pl][u] = 1; any = 1;
                    memcpy(wn[pl] + 16 * u, wi.vfilter, 8 * sizeof(int16_t)); memcpy(wn[pl] + 16 * u + 8, wi.hfilter, 8 * sizeof(int16_t));
                }
            }
            if (any && (svt_hip_memcpy_h2d(hip, d_act[pl], act[pl], n) != SVT_HIP_OK || svt_hip_memcpy_h2d(hip, p->d_unit_wiener[pl], wn[pl], sizeof(int16_t) * 16 * n) != SVT_HIP_OK)) { ret = EB_ErrorUndefined; break; }
        }
        t_init += svt_hip_hooks_now_ns() - ti0;
        if (!any) continue;
        walk[n_walk_planes] = (SvtHipWienerWalkPlane){plane_origin(p, p->d_cdef[pl], pl), p->stride[pl], pw, ph, rsi->restoration_unit_size, pl > 0, plane_origin(p, p->d_recon[pl], pl), p->stride[pl],
                                                      p->src[pl], p->src_st[pl], p->d_unit_wiener[pl], (const uint8_t *)d_act[pl], p->wiener_win[pl], (int64_t *)d_err[pl], (uint32_t *)d_probes[pl]};
        walk_pl[n_walk_planes++] = pl;
    }
    /* the walks of all planes in ONE launch: a unit's walk is a serial chain of ~30 probes, so three launches in a row cost three times the longest walk */
    const long long tk0 = svt_hip_hooks_now_ns();
    if (ret == EB_ErrorNone && n_walk_planes && svt_hip_wiener_walk_units_picture_dev(hip, p->pix_bytes, p->bd, n_walk_planes, walk) != SVT_HIP_OK) ret = EB_ErrorUndefined;
    for (int k = 0; k < n_walk_planes && ret == EB_ErrorNone; k++) {
        const int pl = walk_pl[k], n = cm->rst_info[pl].units_per_tile;
        if (svt_hip_memcpy_d2h(hip, wn[pl], p->d_unit_wiener[pl], sizeof(int16_t) * 16 * n) != SVT_HIP_OK || svt_hip_memcpy_d2h(hip, err[pl], d_err[pl], sizeof(int64_t) * n) != SVT_HIP_OK ||
            svt_hip_memcpy_d2h(hip, probes[pl], d_probes[pl], sizeof(uint32_t) * n) != SVT_HIP_OK ||
            (dev_init && (svt_hip_memcpy_d2h(hip, act[pl], d_act[pl], n) != SVT_HIP_OK || svt_hip_memcpy_d2h(hip, init[pl], d_init[pl], n) != SVT_HIP_OK))) { ret = EB_ErrorUndefined; break; }
        for (int u = 0; u < n; u++) { n_probes += act[pl][u] ? probes[pl][u] : 0; n_walks += act[pl][u] != 0; }
    }
    if (ret != EB_ErrorNone && dev_init) (void)svt_hip_sync(hip);   /* launches may still be using the blocks freed below */
    t_walk = svt_hip_hooks_now_ns() - tk0;
    if (ret == EB_ErrorNone) {   /* every plane has succeeded: only now do the reference's objects change */
        for (int pl = 0; pl < 3; pl++) {
            RestUnitSearchInfo *rusi = pcs->parent_pcs_ptr->rusi_picture[pl];
            for (int u = 0; u < cm->rst_info[pl].units_per_tile; u++) {
                if (act[pl][u]) {
                    rusi[u].sse[RESTORE_WIENER] = err[pl][u];
                    memset(&rusi[u].wiener, 0, sizeof(rusi[u].wiener));
                    memcpy(rusi[u].wiener.vfilter, wn[pl] + 16 * u, 8 * sizeof(int16_t)); memcpy(rusi[u].wiener.hfilter, wn[pl] + 16 * u + 8, 8 * sizeof(int16_t));
                } else {
                    rusi[u].sse[RESTORE_WIENER] = INT64_MAX;
                    if (init[pl][u] == 0) rusi[u].best_rtype[RESTORE_WIENER - 1] = RESTORE_NONE;
                }
            }
        }
        if (dev_init)
            svt_hip_hooks_log("wiener_search: 1 launch, %ld walks, %ld probes on the device; statistics + initial filters (device) queued in %.2f ms, walks + results %.2f ms", n_walks, n_probes,
                              t_init / 1e6, t_walk / 1e6);
        else
            svt_hip_hooks_log("wiener_search: 1 launch, %ld walks, %ld probes on the device; statistics %.2f ms, initial filters (host) %.2f ms, walks %.2f ms", n_walks, n_probes,
                              (tw1 - tw0) / 1e6, t_init / 1e6, t_walk / 1e6);
    }
    for (int pl = 0; pl < 3; pl++) {
        free(act[pl]); free(wn[pl]); free(err[pl]); free(probes[pl]); free(init[pl]);
        if (d_act[pl]) svt_hip_hooks_free(hip, d_act[pl]);
        if (d_err[pl]) svt_hip_hooks_free(hip, d_err[pl]);
        if (d_probes[pl]) svt_hip_hooks_free(hip, d_probes[pl]);
        if (d_init[pl]) svt_hip_hooks_free(hip, d_init[pl]);
        if (dM[pl]) svt_hip_hooks_free(hip, dM[pl]);
        if (dH[pl]) svt_hip_hooks_free(hip, dH[pl]);
    }
    return ret;
}
EbErrorType svt_hip_hook_wiener_search(PictureControlSet *pcs) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_WIENER_SEARCH)) return EB_ErrorUndefined;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 0, &hip);
    EbErrorType rc = EB_ErrorUndefined;
    if (s && (s->flags & ST_WNSEARCH_DONE)) rc = EB_ErrorNone;
    else if (s && !(s->flags & ST_WNSEARCH_FAILED)) {
        rc = !lf_fault("wiener_search") && rest_geometry_ok(&s->pic, pcs->parent_pcs_ptr->av1_cm) && ensure_src(hip, s) == EB_ErrorNone && ensure_cdef_padded(hip, s) == EB_ErrorNone
                 ? wiener_search(hip, s) : EB_ErrorUndefined;
        s->flags |= rc == EB_ErrorNone ? ST_WNSEARCH_DONE : ST_WNSEARCH_FAILED;
        if (rc != EB_ErrorNone) lf_recover(hip, s, 1);
        svt_hip_hooks_count(SVT_HIP_HOOK_WIENER_SEARCH, rc == EB_ErrorNone);
    }
    if (s) lf_leave(s);
    svt_hip_hooks_time(SVT_HIP_HOOK_WIENER_SEARCH, t0);
    return rc;
}

/* rest_kernel, at the top of every restoration segment (EbRestProcess.c:505, before get_own_recon): with the picture deferred the picture-level searches run HERE,
 * so that a failure can still bring the host up to date before the segment copies the picture; 1 = the copy (and with it every host read of the picture in this
 * segment's search) can be skipped */
int svt_hip_hook_rest_begin(PictureControlSet *pcs) {
    if (!svt_hip_hook_skip_host_prep(pcs, 2)) return 0;
    if (svt_hip_hook_sgr_search(pcs) != EB_ErrorNone) return 0;   /* recovered: the C search of this segment follows */
    if (pcs->parent_pcs_ptr->av1_cm->wn_filter_mode && svt_hip_hook_wiener_search(pcs) != EB_ErrorNone) return 0;
    return svt_hip_hook_skip_host_prep(pcs, 2);
}

/* the picture's final reconstruction comes back: restored planes from d_rest, the others (and, inside a padded picture, the samples outside the cropped frame)
 * from the CDEF output with the border svt_extend_frame gives it, or the deblocked picture when CDEF was off */
static EbErrorType lf_final_attempt(SvtHipCtx *hip, LfState *s, int attempt) {
    SvtHipLfPicture *p = &s->pic;
    EbPictureBufferDesc *rec = recon_of(s->pcs, p->pix_bytes == 2);
    if (attempt == 0 && lf_fault("final_download")) return EB_ErrorUndefined;             /* tests: the device is gone before a byte has moved */
    if ((s->flags & ST_SKIP1) && ensure_cdef_padded(hip, s) != EB_ErrorNone) return EB_ErrorUndefined;
    void *const *base = (s->flags & ST_CDEF) ? p->d_cdef : p->d_recon;
    const int rest = (s->flags & ST_REST) ? s->rest_mask : 0;
    const int whole = p->cw == p->w && p->ch == p->h ? rest : 0;   /* planes whose restored version covers the coded picture */
    EbErrorType rc = EB_ErrorNone;
    if (7 & ~whole) rc = download(hip, p, base, rec, 7 & ~whole, 0);
    if (rc == EB_ErrorNone && rest) rc = download(hip, p, p->d_rest, rec, rest, 1);
    if (rc == EB_ErrorNone && attempt == 0 && lf_fault("final_download_late")) return EB_ErrorUndefined;   /* tests: a failure reported after copies were queued */
    return rc;
}
/* The reference's own filter chain on the host picture, with the decisions the (hooked or C) searches left in the reference's structures — what dlf_kernel
 * (EbDlfProcess.c:203-251), cdef_kernel (EbCdefProcess.c:524-572) and rest_kernel (EbRestProcess.c:540-548) would have done to it after their searches.  Only valid
 * while the host picture is still the coded (unfiltered) one, i.e. for a deferred picture of which nothing has come back. */
void svt_av1_cdef_frame(struct EncDecContext *context_ptr, SequenceControlSet *scs_ptr, PictureControlSet *pCs);
void av1_cdef_frame16bit(struct EncDecContext *context_ptr, SequenceControlSet *scs_ptr, PictureControlSet *pCs);
/* (no header declares it: EbRestProcess.c:51 and EbEncDecProcess.c:42 carry the same local prototype) */
void svt_av1_loop_restoration_filter_frame(Yv12BufferConfig *frame, Av1Common *cm, int32_t optimized_lr);
static void lf_host_chain(LfState *s) {
    PictureControlSet *pcs = s->pcs;
    SequenceControlSet *scs = (SequenceControlSet *)pcs->scs_wrapper_ptr->object_ptr;
    Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
    const int is16 = s->pic.pix_bytes == 2;
    EbPictureBufferDesc *rec = recon_of(pcs, is16);
    /* Every stage is gated by the REFERENCE's own condition: what runs here is what dlf_kernel / cdef_kernel / rest_kernel would run, not what the bridge's
     * state flags remember.  The flags only say which stages the DEVICE had applied to its copy; where the two disagree the bitstream follows the reference's conditions,
     * so the host chain does too, and the disagreement is logged (it would be a bridge bug). */
    const PictureParentControlSet *ppcs = pcs->parent_pcs_ptr;
    const FrameHeader *fh = &ppcs->frm_hdr;
    const int tiles = cm->tiles_info.tile_cols * cm->tiles_info.tile_rows;
    const int ref_dlf = ppcs->loop_filter_mode >= 2 || (ppcs->loop_filter_mode == 1 && tiles > 1);                                   /* EbDlfProcess.c:175-177 */
    const int ref_cdef = scs->seq_header.cdef_level && ppcs->cdef_level &&
                         (scs->seq_header.enable_restoration != 0 || ppcs->is_used_as_reference_flag || scs->static_config.recon_enabled);   /* EbCdefProcess.c:524-534 */
    const int ref_rest = scs->seq_header.enable_restoration && fh->allow_intrabc == 0 &&
                         (cm->rst_info[0].frame_restoration_type != RESTORE_NONE || cm->rst_info[1].frame_restoration_type != RESTORE_NONE ||
                          cm->rst_info[2].frame_restoration_type != RESTORE_NONE);                                                     /* EbRestProcess.c:540-548 */
    if (!!(s->flags & ST_CDEF) != !!ref_cdef || !!((s->flags & ST_REST) && s->rest_mask) != !!ref_rest)
        SVT_LOG("svt_hip: host filter chain: the bridge's stage flags (cdef %d, restoration %d) differ from the reference's conditions (%d, %d) - following the reference\n",
                !!(s->flags & ST_CDEF), !!((s->flags & ST_REST) && s->rest_mask), ref_cdef, ref_rest);
    /* deblocking: only when the device held the deblocked copy (ST_DBL; otherwise the host picture IS deblocked already: the C filter ran on it before the upload) */
    if ((s->flags & ST_DBL) && ref_dlf) svt_av1_loop_filter_frame(rec, pcs, 0, 3);   /* the levels are in the frame header (picked by the hook or by the C search) */
    if (scs->seq_header.enable_restoration) svt_av1_loop_restoration_save_boundary_lines(cm->frame_to_show, cm, 0);
    if (ref_cdef) {
        if (is16) av1_cdef_frame16bit(0, scs, pcs); else svt_av1_cdef_frame(0, scs, pcs);
    }
    if (scs->seq_header.enable_restoration) {
        svt_av1_loop_restoration_save_boundary_lines(cm->frame_to_show, cm, 1);
        for (int pl = 0; pl < 3; pl++)
            svt_extend_frame(cm->frame_to_show->buffers[pl], cm->frame_to_show->crop_widths[pl > 0], cm->frame_to_show->crop_heights[pl > 0], cm->frame_to_show->strides[pl > 0],
                             RESTORATION_BORDER, RESTORATION_BORDER, is16);
    }
    if (ref_rest) svt_av1_loop_restoration_filter_frame(cm->frame_to_show, cm, 0);
}
/* A deferred picture's only copy is the device's: if it cannot be brought back the encoder must not go on with the unfiltered host picture (the bitstream signals the
 * filters; the picture is a reference and the reconstruction output).  In order: the download; once more when copies had been queued (a second complete pass makes the
 * host whole whatever the first one left); the reference's own filter chain on the host when nothing had been written yet (the coded picture is still there, every
 * decision is in the reference's structures); otherwise the encoder's fatal-error exit (EbCallback::error_handler = lib_svt_encoder_send_error_exit). */
static void lf_final_download(SvtHipCtx *hip, LfState *s) {
    tls_download_started = 0;
    EbErrorType rc = lf_final_attempt(hip, s, 0);
    if (rc != EB_ErrorNone && tls_download_started) {
        SVT_LOG("svt_hip: download of a filtered picture failed (%s) - trying again\n", svt_hip_last_error(hip));
        (void)svt_hip_sync(hip);
        __sync_fetch_and_add(&g_lf_final_retries, 1);
        rc = lf_final_attempt(hip, s, 1);
    }
    if (rc != EB_ErrorNone && !tls_download_started) {
        SVT_LOG("svt_hip: the device lost a picture after its filter stages (%s) - filtering it on the host\n", svt_hip_last_error(hip));
        lf_host_chain(s);
        __sync_fetch_and_add(&g_lf_final_host_chains, 1);
        rc = EB_ErrorNone;
    }
    if (rc != EB_ErrorNone) {
        SVT_LOG("svt_hip: a filtered picture could not be brought back from the device (%s) - stopping the encoder\n", svt_hip_last_error(hip));
        __sync_fetch_and_add(&g_lf_final_fatal, 1);
        const SequenceControlSet *scs = (const SequenceControlSet *)s->pcs->scs_wrapper_ptr->object_ptr;
        if (scs->encode_context_ptr && scs->encode_context_ptr->app_callback_ptr && scs->encode_context_ptr->app_callback_ptr->error_handler)
            scs->encode_context_ptr->app_callback_ptr->error_handler(scs->encode_context_ptr->app_callback_ptr->handle, (uint32_t)EB_ErrorUndefined);
        return;   /* the host picture is NOT valid: ST_HOST_STALE stays set (the encoder is stopping) */
    }
    s->flags &= ~ST_HOST_STALE;
}
void svt_hip_hook_picture_done(PictureControlSet *pcs) {
    SvtHipCtx *hip = NULL;
    LfState   *s = lf_enter(pcs, 0, &hip);
    if (!s) return;
    const long long t0 = svt_hip_hooks_now_ns();
    if (s->flags & ST_HOST_STALE) lf_final_download(hip, s);
    else if (s->flags & ST_SKIP1) {   /* nothing was filtered on the device, but the border the reference gives the CDEF output was skipped: the host adds it now */
        const Av1Common *cm = pcs->parent_pcs_ptr->av1_cm;
        for (int pl = 0; pl < 3; pl++)
            svt_extend_frame(cm->frame_to_show->buffers[pl], cm->frame_to_show->crop_widths[pl > 0], cm->frame_to_show->crop_heights[pl > 0], cm->frame_to_show->strides[pl > 0],
                             RESTORATION_BORDER, RESTORATION_BORDER, s->pic.pix_bytes == 2);
    }
    for (int pl = 0; pl < 3; pl++)
        if (s->res_host[pl]) { svt_hip_resident_release(s->res_host[pl]); s->res_host[pl] = NULL; }
    svt_hip_hooks_time(SVT_HIP_HOOK_REST_APPLY, t0);
    svt_hip_hooks_unlock_any();
    pthread_mutex_lock(&g_tab_mu);
    s->pcs = NULL; s->flags = 0;   /* buffers stay allocated for the next picture of this size */
    pthread_mutex_unlock(&g_tab_mu);
    pthread_mutex_unlock(&s->mu);
}

/* ------------------------------------------------------------------ hook "cdef_finish": joint_strength_search_dual inside finish_cdef_search
 * (EbEncCdef.c:1258: four calls per picture, nb_strengths = 1, 2, 4, 8).  The first call of a picture sends the two distortion tables of the non-skipped
 * filter blocks and runs all four independent searches side by side (svt_hip_cdef_strength_select_dev: one launch per step index, no host round trip);
 * the three calls that follow read their result from the same download.  1 = handled. */
static __thread struct {
    const void            *key0, *key1; /* the tables the cached result belongs to */
    int32_t                sb_count, start_gi, end_gi;
    int                    valid;
    SvtHipCdefSelectResult res;
} tls_sel;

int svt_hip_hook_cdef_joint_search(int32_t *best_lev0, int32_t *best_lev1, int32_t nb_strengths, uint64_t (**mse)[64], int32_t sb_count, int32_t start_gi,
                                   int32_t end_gi, uint64_t *tot_mse) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_CDEF_FINISH) || sb_count < 0 || (nb_strengths != 1 && nb_strengths != 2 && nb_strengths != 4 && nb_strengths != 8)) return 0;
    const int ci = nb_strengths == 1 ? 0 : (nb_strengths == 2 ? 1 : (nb_strengths == 4 ? 2 : 3));
    if (nb_strengths == 1 || !tls_sel.valid || tls_sel.key0 != (const void *)mse[0] || tls_sel.key1 != (const void *)mse[1] || tls_sel.sb_count != sb_count ||
        tls_sel.start_gi != start_gi || tls_sel.end_gi != end_gi) {
        tls_sel.valid = 0;
        const long long t0 = svt_hip_hooks_now_ns();
        SvtHipCtx *hip = svt_hip_hooks_lock_any();
        if (!hip) return 0;
        const size_t mb = (size_t)sb_count * 64 * sizeof(uint64_t);
        void        *d_m0 = NULL, *d_m1 = NULL, *d_state = NULL;
        int          rc = svt_hip_hooks_malloc(hip, &d_m0, mb + 8);
        if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_m1, mb + 8);
        if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_state, SVT_HIP_CDEF_SELECT_STATE_BYTES);
        if (rc == SVT_HIP_OK && mb) rc = svt_hip_memcpy_h2d(hip, d_m0, mse[0], mb);
        if (rc == SVT_HIP_OK && mb) rc = svt_hip_memcpy_h2d(hip, d_m1, mse[1], mb);
        if (rc == SVT_HIP_OK)
            rc = svt_hip_cdef_strength_select_dev(hip, (const uint64_t *)d_m0, (const uint64_t *)d_m1, sb_count, start_gi, end_gi, d_state, SVT_HIP_CDEF_SELECT_STATE_BYTES);
        if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, &tls_sel.res, d_state, sizeof(tls_sel.res));
        if (rc == SVT_HIP_OK && tls_sel.res.status[0]) rc = SVT_HIP_ERR_RUNTIME; /* the one-launch form gave up waiting (SVT_HIP_CDEF_SELECT=resident only) */
        svt_hip_hooks_free(hip, d_m0); svt_hip_hooks_free(hip, d_m1); svt_hip_hooks_free(hip, d_state);
        if (rc != SVT_HIP_OK) SVT_LOG("CDEF strength selection on the device failed (%s): C search\n", svt_hip_last_error(hip));
        svt_hip_hooks_unlock_any();
        svt_hip_hooks_count(SVT_HIP_HOOK_CDEF_FINISH, rc == SVT_HIP_OK);
        svt_hip_hooks_time(SVT_HIP_HOOK_CDEF_FINISH, t0);
        if (rc != SVT_HIP_OK) return 0;
        tls_sel.key0 = mse[0]; tls_sel.key1 = mse[1]; tls_sel.sb_count = sb_count; tls_sel.start_gi = start_gi; tls_sel.end_gi = end_gi; tls_sel.valid = 1;
        svt_hip_hooks_log("cdef_finish: strength pairs for 1 / 2 / 4 / 8 over %d filter blocks in one pass, totals %llu %llu %llu %llu", sb_count,
                          (unsigned long long)tls_sel.res.tot_mse[0], (unsigned long long)tls_sel.res.tot_mse[1], (unsigned long long)tls_sel.res.tot_mse[2],
                          (unsigned long long)tls_sel.res.tot_mse[3]);
    }
    for (int i = 0; i < nb_strengths; i++) { best_lev0[i] = tls_sel.res.lev0[ci][i]; best_lev1[i] = tls_sel.res.lev1[ci][i]; }
    *tot_mse = tls_sel.res.tot_mse[ci];
    if (nb_strengths == 8) tls_sel.valid = 0;   /* the picture's last call */
    return 1;
}

/* ------------------------------------------------------------------ hook "cdef_finish", whole: everything finish_cdef_search does with the two distortion tables
 * (EbEncCdef.c:1258-1298) -- the four joint_strength_search_dual searches, the count of strength pairs by RDCOST, every filter block's pair -- in one
 * upload, 42 launches and one download.  1 = handled: *nb_strength_bits, y / uv strengths [8] and selected[sb_count] are filled in; the reference then only
 * copies them into the frame header and the mode-info grid.  0: the reference's own loops run (with the per-search hook above). */
int svt_hip_hook_cdef_finish(uint64_t (**mse)[64], int32_t sb_count, int32_t start_gi, int32_t end_gi, uint64_t lambda, int32_t *nb_strength_bits, int32_t *y_strength,
                             int32_t *uv_strength, int32_t *selected) {
    if (!svt_hip_hook_enabled(SVT_HIP_HOOK_CDEF_FINISH) || sb_count < 0) return 0;
    const long long t0 = svt_hip_hooks_now_ns();
    SvtHipCtx *hip = svt_hip_hooks_lock_any();
    if (!hip) return 0;
    const size_t mb = (size_t)sb_count * 64 * sizeof(uint64_t);
    void *d_m0 = NULL, *d_m1 = NULL, *d_state = NULL, *d_out = NULL, *d_sel = NULL;
    SvtHipCdefFinish fin;
    int rc = svt_hip_hooks_malloc(hip, &d_m0, mb + 8);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_m1, mb + 8);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_state, SVT_HIP_CDEF_SELECT_STATE_BYTES);
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_out, sizeof(fin));
    if (rc == SVT_HIP_OK) rc = svt_hip_hooks_malloc(hip, &d_sel, sizeof(int32_t) * (size_t)(sb_count + 1));
    if (rc == SVT_HIP_OK && mb) rc = svt_hip_memcpy_h2d(hip, d_m0, mse[0], mb);
    if (rc == SVT_HIP_OK && mb) rc = svt_hip_memcpy_h2d(hip, d_m1, mse[1], mb);
    if (rc == SVT_HIP_OK) rc = svt_hip_cdef_strength_select_dev(hip, (const uint64_t *)d_m0, (const uint64_t *)d_m1, sb_count, start_gi, end_gi, d_state, SVT_HIP_CDEF_SELECT_STATE_BYTES);
    if (rc == SVT_HIP_OK)
        rc = svt_hip_cdef_finish_dev(hip, (const uint64_t *)d_m0, (const uint64_t *)d_m1, sb_count, d_state, lambda, NULL, (SvtHipCdefFinish *)d_out, (int32_t *)d_sel, NULL, NULL);
    if (rc == SVT_HIP_OK) rc = svt_hip_memcpy_d2h(hip, &fin, d_out, sizeof(fin));
    if (rc == SVT_HIP_OK && fin.cdef_bits < 0) rc = SVT_HIP_ERR_RUNTIME; /* an incomplete selection (svt_hip.h: status[0]) */
    if (rc == SVT_HIP_OK && sb_count) rc = svt_hip_memcpy_d2h(hip, selected, d_sel, sizeof(int32_t) * (size_t)sb_count);
    svt_hip_hooks_free(hip, d_m0); svt_hip_hooks_free(hip, d_m1); svt_hip_hooks_free(hip, d_state); svt_hip_hooks_free(hip, d_out); svt_hip_hooks_free(hip, d_sel);
    if (rc != SVT_HIP_OK) SVT_LOG("CDEF strength decision on the device failed (%s): C loops\n", svt_hip_last_error(hip));
    svt_hip_hooks_unlock_any();
    svt_hip_hooks_count(SVT_HIP_HOOK_CDEF_FINISH, rc == SVT_HIP_OK);
    svt_hip_hooks_time(SVT_HIP_HOOK_CDEF_FINISH, t0);
    if (rc != SVT_HIP_OK) return 0;
    *nb_strength_bits = fin.cdef_bits;
    for (int j = 0; j < fin.nb_strengths; j++) { y_strength[j] = fin.y_strength[j]; uv_strength[j] = fin.uv_strength[j]; }
    svt_hip_hooks_log("cdef_finish: %d filter blocks, %d strength pair(s), first (%d, %d)", sb_count, fin.nb_strengths, fin.y_strength[0], fin.uv_strength[0]);
    return 1;
}
