/* svt_hip_hooks.c — see svt_hip_hooks.h: run-time hook selection, the shared device context and its lock, and the per-call dispatch
 * table entries of SVT_HIP_RTCD.  Reference-side glue (C, compiled into libSvtAv1Enc). */
#include <pthread.h>
#include <time.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svt_hip_hooks.h"
#include "svt_hip_rtcd.h"
#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"
#include "EbLog.h"

static const char *const k_hook_name[SVT_HIP_HOOK_COUNT] = {"me", "dlf", "dlf_search", "cdef_search", "cdef_apply",
                                                            "sgr_search", "wiener_stats", "rest_apply", "wiener_try", "wiener_search", "hme", "tf", "pa", "tf_me", "cdef_finish", "md_tx", "tf_subpel", "encdec_tx", "md_subpel", "encdec_sb", "md_pre"};
static const int k_hook_opt_in[SVT_HIP_HOOK_COUNT] = {[SVT_HIP_HOOK_MD_TX] = 1, [SVT_HIP_HOOK_ENCDEC_TX] = 1, [SVT_HIP_HOOK_MD_SUBPEL] = 1, [SVT_HIP_HOOK_ENCDEC_SB] = 1, [SVT_HIP_HOOK_MD_PRE] = 1};   /* not selected by "all": must be named */
static int             g_enabled[SVT_HIP_HOOK_COUNT];
static long            g_handled[SVT_HIP_HOOK_COUNT], g_fellback[SVT_HIP_HOOK_COUNT];
static int             g_verbose;
static SvtHipCtx      *g_ctx;
static SvtHipCtx      *g_rtcd_ctx;   /* the per-call wrappers run on a context of their own: they serialise on the library's wrapper mutex, the hooks on
                                      * g_lock, and a context (error string, library-owned scratch, stream) must not be driven from both at once */
static int             g_rtcd_installed;
static pthread_mutex_t g_lock   = PTHREAD_MUTEX_INITIALIZER;
static pthread_mutex_t g_cnt_mu = PTHREAD_MUTEX_INITIALIZER;
static int             g_inited;

static int in_list(const char *list, const char *name) {
    if (!list) return 0;
    const size_t n = strlen(name);
    for (const char *p = list; *p;) {
        const char *e = strchr(p, ',');
        const size_t l = e ? (size_t)(e - p) : strlen(p);
        if ((l == n && !strncmp(p, name, n)) || (l == 3 && !strncmp(p, "all", 3))) return 1;
        if (!e) break;
        p = e + 1;
    }
    return 0;
}

/* ME / TF segments (EbEncHandle.c:403-428, :477-478: 10 x 6 segments for every picture of at least 10 x 6 superblocks -- the reference's unit of CPU parallelism
 * inside a picture).  With the hooks on a segment is one batched launch per HME level and per integer search, i.e. four synchronous round trips: at 1280 x 720 the
 * reference's 60 segments hold FOUR superblocks each, and the hooked encoder spent more thread time on those round trips than the SIMD kernels they replace
 * (profiles/r04: 720p, 32 frames: 5580 + 1860 flushes).  Called from the patched load_default_buffer_configuration_settings (which runs before the encoder is
 * initialised, so the environment is read here directly): at least ~64 superblocks per segment.  A picture's results do not depend on how it is cut into segments.
 * SVT_HIP_SEGMENTS=0 keeps the reference's counts. */
void svt_hip_hooks_segments(uint32_t luma_width, uint32_t luma_height, uint32_t *me_cols, uint32_t *me_rows, uint32_t *tf_cols, uint32_t *tf_rows, uint32_t *cdef_cols,
                            uint32_t *cdef_rows, uint32_t *rest_cols, uint32_t *rest_rows) {
    const char *hooks = getenv("SVT_HIP_HOOKS"), *seg = getenv("SVT_HIP_SEGMENTS");
    if (!hooks || (seg && !atoi(seg))) return;
    const uint32_t sb_cols = (luma_width + 32) / 64, sb_rows = (luma_height + 32) / 64;
    const uint32_t cols = sb_cols / 8 ? sb_cols / 8 : 1, rows = sb_rows / 8 ? sb_rows / 8 : 1;
    if (in_list(hooks, "me") || in_list(hooks, "hme")) {
        if (cols < *me_cols) *me_cols = cols;
        if (rows < *me_rows) *me_rows = rows;
        /* SVT_HIP_ME_SEG=<columns>x<rows>: the ME segment grid outright (A/B runs: a segment is one launch per level and reference picture, tools/encoder_walltime.sh) */
        unsigned c = 0, r = 0;
        if (getenv("SVT_HIP_ME_SEG") && sscanf(getenv("SVT_HIP_ME_SEG"), "%ux%u", &c, &r) == 2 && c >= 1 && r >= 1 && c <= sb_cols && r <= sb_rows) { *me_cols = c; *me_rows = r; }
    }
    if (in_list(hooks, "tf") || in_list(hooks, "tf_me")) {
        if (cols < *tf_cols) *tf_cols = cols;
        if (rows < *tf_rows) *tf_rows = rows;
    }
    /* the CDEF search hook and the two restoration searches work per PICTURE (the first segment to arrive does it all, the others only wait for it): one segment */
    if (in_list(hooks, "cdef_search")) *cdef_cols = *cdef_rows = 1;
    if (in_list(hooks, "sgr_search") && in_list(hooks, "wiener_search")) *rest_cols = *rest_rows = 1;
}

static int in_list_exact(const char *list, const char *name) {   /* like in_list, but "all" does not match */
    if (!list) return 0;
    const size_t n = strlen(name);
    for (const char *p = list; *p;) {
        const char *e = strchr(p, ',');
        const size_t l = e ? (size_t)(e - p) : strlen(p);
        if (l == n && !strncmp(p, name, n)) return 1;
        if (!e) break;
        p = e + 1;
    }
    return 0;
}

static int g_device;
long svt_hip_hooks_early_unpins(void);
int svt_hip_hooks_device(void) { return g_device; }
int svt_hip_hook_enabled(int which) { return which >= 0 && which < SVT_HIP_HOOK_COUNT && g_ctx && g_enabled[which]; }
/* how long process threads waited for the context and how long they held it (nanoseconds; reported at exit: the serial share of the hooks) */
static long long g_lock_wait_ns, g_lock_held_ns, g_lock_t0, g_lock_n;
static long long now_ns(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (long long)t.tv_sec * 1000000000ll + t.tv_nsec; }
SvtHipCtx *svt_hip_hooks_lock(void) {
    if (!g_ctx) return NULL;
    const long long t0 = now_ns();
    pthread_mutex_lock(&g_lock);
    g_lock_t0 = now_ns();
    g_lock_wait_ns += g_lock_t0 - t0;
    g_lock_n++;
    return g_ctx;
}
void svt_hip_hooks_unlock(void) {
    g_lock_held_ns += now_ns() - g_lock_t0;
    pthread_mutex_unlock(&g_lock);
}
/* The source-side bridges (picture analysis, open-loop ME / HME, the temporal filter) keep no device state between two calls — every call allocates, uploads,
 * launches, downloads and frees under the lock — so they do not need THE context, only A context: a small pool (SVT_HIP_CONTEXTS, default 4), each with its own
 * stream, lets the reference's N motion-estimation / TF threads run their segments side by side instead of queueing behind one mutex.  The loop-filter bridge keeps
 * the main context and its lock: its per-picture state table and the "first segment searches the whole picture" rule rely on that mutual exclusion.  A context is
 * drained (svt_hip_sync) before it is handed back, so whichever context a later call gets sees the device memory complete. */
/* Device memory of the per-call bridges comes from a small cache: hipMalloc / hipFree cost tens of microseconds each and hipFree waits for the whole device, which
 * with several contexts at work turns every free into a global synchronisation point.  Blocks are kept by power-of-two size class (SVT_HIP_ALLOC_CACHE_MB in total,
 * default 1024; 0 = off); a block is only ever reused after the context that used it was drained (svt_hip_hooks_unlock_any / the synchronous copies), like a free. */
#define ALLOC_CLASSES 40
#define ALLOC_PER_CLASS 32
#define ALLOC_LIVE 2048
static pthread_mutex_t g_alloc_mu = PTHREAD_MUTEX_INITIALIZER;
static void           *g_alloc_free[ALLOC_CLASSES][ALLOC_PER_CLASS];
static int             g_alloc_n[ALLOC_CLASSES];
static struct { void *p; int c; } g_alloc_live[ALLOC_LIVE];   /* blocks handed out: pointer -> size class */
static size_t          g_alloc_cached, g_alloc_limit = (size_t)1024 << 20;
static long            g_alloc_hits, g_alloc_misses;
static int alloc_class(size_t bytes) { int c = 8; while (((size_t)1 << c) < bytes && c < ALLOC_CLASSES - 1) c++; return c; }
static int alloc_note(void *p, int c) {   /* g_alloc_mu held */
    for (int i = 0; i < ALLOC_LIVE; i++)
        if (!g_alloc_live[i].p) { g_alloc_live[i].p = p; g_alloc_live[i].c = c; return 1; }
    return 0;
}
static size_t alloc_block_size(size_t bytes) { const int c = alloc_class(bytes ? bytes : 1); return (!g_alloc_limit || ((size_t)1 << c) < bytes) ? bytes : (size_t)1 << c; }
int svt_hip_hooks_malloc(SvtHipCtx *hip, void **p, size_t bytes) {
    const int c = alloc_class(bytes ? bytes : 1);
    if (!g_alloc_limit || ((size_t)1 << c) < bytes) return svt_hip_malloc(hip, p, bytes);
    pthread_mutex_lock(&g_alloc_mu);
    if (g_alloc_n[c]) {
        void *q = g_alloc_free[c][g_alloc_n[c] - 1];
        if (alloc_note(q, c)) {
            g_alloc_n[c]--;
            g_alloc_cached -= (size_t)1 << c;
            g_alloc_hits++;
            pthread_mutex_unlock(&g_alloc_mu);
            *p = q;
            return SVT_HIP_OK;
        }
    }
    g_alloc_misses++;
    pthread_mutex_unlock(&g_alloc_mu);
    const int rc = svt_hip_malloc(hip, p, (size_t)1 << c);   /* the whole class: the block can serve any request of it later */
    if (rc == SVT_HIP_OK) {
        pthread_mutex_lock(&g_alloc_mu);
        (void)alloc_note(*p, c);   /* table full: the block is simply freed for real later */
        pthread_mutex_unlock(&g_alloc_mu);
    }
    return rc;
}
/* A block freed by a bridge that holds a pool context may still be written by a kernel of that context (failure paths free before the context is drained): it
 * goes on the thread's pending list and into the cache only once svt_hip_hooks_unlock_any has drained the context. */
#define PENDING_FREES 64
static __thread void *tls_pending[PENDING_FREES];
static __thread int   tls_pending_n;
static __thread int   tls_slot = -1, tls_pref = -1;
static void cache_put(SvtHipCtx *hip, void *p);
void svt_hip_hooks_free(SvtHipCtx *hip, void *p) {
    if (!p) return;
    if (tls_slot >= 0) {
        if (tls_pending_n == PENDING_FREES) {   /* full: drain now, then everything pending may be reused */
            (void)svt_hip_sync(hip);
            for (int i = 0; i < tls_pending_n; i++) cache_put(hip, tls_pending[i]);
            tls_pending_n = 0;
        }
        tls_pending[tls_pending_n++] = p;
        return;
    }
    cache_put(hip, p);
}
static void cache_put(SvtHipCtx *hip, void *p) {
    int c = -1;
    pthread_mutex_lock(&g_alloc_mu);
    for (int i = 0; i < ALLOC_LIVE; i++)
        if (g_alloc_live[i].p == p) { c = g_alloc_live[i].c; g_alloc_live[i].p = NULL; break; }
    if (c >= 0 && g_alloc_n[c] < ALLOC_PER_CLASS && g_alloc_cached + ((size_t)1 << c) <= g_alloc_limit) {
        g_alloc_free[c][g_alloc_n[c]++] = p;
        g_alloc_cached += (size_t)1 << c;
        pthread_mutex_unlock(&g_alloc_mu);
        return;
    }
    pthread_mutex_unlock(&g_alloc_mu);
    svt_hip_free(hip, p);
}

/* Resident planes (on by default since round 4, measured on the MI355X: profiles/r04/resident_first_call_summary.txt; SVT_HIP_RESIDENT=0 = off).  The luma planes of an EbPaReferenceObject — the padded
 * picture and its 1/4 and 1/16 versions — are written at exactly two places of the reference: picture analysis (picture_analysis_kernel, EbPictureAnalysisProcess.c
 * :3960-3994; the overlay twin in EbPictureDecisionProcess.c:3644-3680) and the end of the temporal filter (pad_and_decimate_filtered_pic, EbTemporalFiltering.c:2556),
 * and read by every ME / HME / TF-ME segment of the picture itself and of every picture that references it.  The patched reference announces each write
 * (svt_hip_hooks_resident_note_pa, AFTER the planes are complete); the first bridge call that needs a plane afterwards uploads it once, whole, and every later call of
 * any context reads that copy instead of uploading its own row band (the table: svt_hip_resident.c).  Only announced planes are ever resident: the buffers of an
 * EbPaReferenceObject and the input pictures are allocated once per encoder instance, whereas the resized references of the super-resolution / reference-scaling
 * modes are allocated and freed per picture (an address can come back with other content) — those are never announced and keep the upload path. */
static int g_res_on, g_pin_on;
int svt_hip_hooks_pin_enabled(void) { return g_pin_on; }
void svt_hip_hooks_resident_note_picture(const EbPictureBufferDesc *pic) {
    if (g_res_on && pic && pic->buffer_y) svt_hip_resident_note(pic->buffer_y, (size_t)pic->stride_y * (size_t)(pic->height + 2 * pic->origin_y));
}
/* downsample_decimation_input_picture / downsample_filtering_input_picture have just written quarter / sixteenth from padded (itself complete before the call).
 * With in-loop ME the "padded" picture IS the picture's own input buffer, which other stages write as well: never resident */
void svt_hip_hooks_resident_note_pa(const PictureParentControlSet *pcs, const EbPictureBufferDesc *padded, const EbPictureBufferDesc *quarter,
                                    const EbPictureBufferDesc *sixteenth, int with_padded) {
    if (!g_res_on || !pcs || padded == pcs->enhanced_picture_ptr) return;
    if (with_padded) {
        svt_hip_hooks_resident_note_picture(padded);
        /* The picture's own 8-bit planes (what the temporal filter reads of a window frame, svt_hip_tf_bridge.c) are complete at this point as well, at all three
         * call sites: padding and pre-processing come first in picture analysis (:3900-3903), generate_padding of the three planes first in
         * pad_and_decimate_filtered_pic (:2573-2592).  Not for a picture that has been replaced by its resized version (allocated per picture, EbResize.c:1602). */
        const EbPictureBufferDesc *in = pcs->enhanced_picture_ptr;
        if (in && in == pcs->enhanced_unscaled_picture_ptr && in->buffer_y && in->buffer_cb && in->buffer_cr && in->bit_depth == EB_8BIT) {
            const int ss_y = in->color_format >= EB_YUV422 ? 0 : 1;   /* chroma rows: halved for 4:2:0 only */
            svt_hip_resident_note(in->buffer_y, (size_t)in->stride_y * (size_t)(in->height + 2 * in->origin_y));
            svt_hip_resident_note(in->buffer_cb, (size_t)in->stride_cb * (size_t)((in->height >> ss_y) + 2 * (in->origin_y >> ss_y)));
            svt_hip_resident_note(in->buffer_cr, (size_t)in->stride_cr * (size_t)((in->height >> ss_y) + 2 * (in->origin_y >> ss_y)));
        }
    }
    svt_hip_hooks_resident_note_picture(quarter);
    svt_hip_hooks_resident_note_picture(sixteenth);
}

#define SVT_HIP_POOL_MAX 8
static SvtHipCtx      *g_pool[SVT_HIP_POOL_MAX];
static pthread_mutex_t g_pool_mu[SVT_HIP_POOL_MAX];
static int             g_pool_n, g_pool_next;
static long long       g_pool_wait_ns, g_pool_held_ns, g_pool_locks;
static __thread long long tls_t0;
SvtHipCtx *svt_hip_hooks_lock_any(void) {
    if (!g_ctx) return NULL;
    if (!g_pool_n) { tls_slot = -1; return svt_hip_hooks_lock(); }
    if (tls_pref < 0) tls_pref = __sync_fetch_and_add(&g_pool_next, 1) % g_pool_n;
    const long long t0 = now_ns();
    int got = -1;
    for (int i = 0; i < g_pool_n && got < 0; i++) {
        const int k = (tls_pref + i) % g_pool_n;
        if (pthread_mutex_trylock(&g_pool_mu[k]) == 0) got = k;
    }
    if (got < 0) { got = tls_pref; pthread_mutex_lock(&g_pool_mu[got]); }
    tls_slot = got;
    tls_t0 = now_ns();
    __sync_fetch_and_add(&g_pool_wait_ns, tls_t0 - t0);
    __sync_fetch_and_add(&g_pool_locks, 1);
    return g_pool[got];
}
void svt_hip_hooks_unlock_any(void) {
    if (tls_slot < 0) { svt_hip_hooks_unlock(); return; }
    (void)svt_hip_sync(g_pool[tls_slot]);
    for (int i = 0; i < tls_pending_n; i++) cache_put(g_pool[tls_slot], tls_pending[i]);   /* the context is drained: its blocks may serve anybody */
    tls_pending_n = 0;
    __sync_fetch_and_add(&g_pool_held_ns, now_ns() - tls_t0);
    const int k = tls_slot;
    tls_slot = -1;
    pthread_mutex_unlock(&g_pool_mu[k]);
}
void svt_hip_hooks_log(const char *fmt, ...) {
    if (!g_verbose) return;
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[svt_hip] ");
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}
/* wall time process threads spend inside each hook (lock waits included): what the hooked stage costs the encoder's pipeline */
static long long g_hook_ns[SVT_HIP_HOOK_COUNT];
static long      g_hook_calls[SVT_HIP_HOOK_COUNT];
long long svt_hip_hooks_now_ns(void) { return now_ns(); }
void svt_hip_hooks_time(int which, long long t0_ns) {
    __sync_fetch_and_add(&g_hook_ns[which], now_ns() - t0_ns);
    __sync_fetch_and_add(&g_hook_calls[which], 1);
}
void svt_hip_hooks_count(int which, int handled) {
    pthread_mutex_lock(&g_cnt_mu);
    if (handled) g_handled[which]++; else g_fellback[which]++;
    pthread_mutex_unlock(&g_cnt_mu);
}
/* one line per hook on stderr at exit: "svt_hip_hook me handled=12 fallback=0" (tests/test_encode_e2e*.py parse it) */
static int g_reported;
static void svt_hip_hooks_report_once(void) { if (!g_reported) { g_reported = 1; svt_hip_hooks_report(); } }
void svt_hip_hooks_report(void) {
    for (int i = 0; i < SVT_HIP_HOOK_COUNT; i++)
        if (g_enabled[i]) fprintf(stderr, "svt_hip_hook %s handled=%ld fallback=%ld\n", k_hook_name[i], g_handled[i], g_fellback[i]);
    for (int i = 0; i < SVT_HIP_HOOK_COUNT; i++)
        if (g_hook_calls[i]) fprintf(stderr, "svt_hip_hook_time %s calls=%ld wall_ms=%.1f\n", k_hook_name[i], g_hook_calls[i], g_hook_ns[i] / 1e6);
    fprintf(stderr, "svt_hip_context locks=%lld held_ms=%.1f waited_ms=%.1f\n", g_lock_n, g_lock_held_ns / 1e6, g_lock_wait_ns / 1e6);
    if (g_alloc_hits + g_alloc_misses) fprintf(stderr, "svt_hip_alloc_cache hits=%ld misses=%ld cached_mb=%.1f\n", g_alloc_hits, g_alloc_misses, g_alloc_cached / 1048576.0);
    if (g_res_on) {
        SvtHipResidentStats r;
        svt_hip_resident_stats(&r);
        fprintf(stderr, "svt_hip_resident notes=%ld uploads=%ld uploaded_mb=%.1f hits=%ld evictions=%ld resident_mb=%.1f refused=%ld\n", r.notes, r.uploads, r.uploaded_mb, r.hits,
                r.evictions, r.resident_mb, r.refused);
    }
    if (g_pool_n)
        fprintf(stderr, "svt_hip_context_pool contexts=%d locks=%lld held_ms=%.1f waited_ms=%.1f\n", g_pool_n, g_pool_locks, g_pool_held_ns / 1e6, g_pool_wait_ns / 1e6);
    if (svt_hip_hooks_early_unpins())   /* an encoder instance of the process ended while others went on: every page-locked range was released and the survivors' re-registered */
        fprintf(stderr, "svt_hip_pins released_while_other_instances_ran=%ld\n", svt_hip_hooks_early_unpins());
    if (g_enabled[SVT_HIP_HOOK_ENCDEC_TX]) {
        long blocks, calls;
        svt_hip_hook_encdec_tx_stats(&blocks, &calls);
        fprintf(stderr, "svt_hip_encdec_tx inter_blocks=%ld estimate_transform_calls_replaced=%ld\n", blocks, calls);
    }
    if (g_enabled[SVT_HIP_HOOK_ENCDEC_SB]) {
        long sbs, launches, blocks, calls;
        svt_hip_hook_encdec_sb_stats(&sbs, &launches, &blocks, &calls);
        fprintf(stderr, "svt_hip_encdec_sb superblocks=%ld launches=%ld inter_blocks_predicted_ahead=%ld estimate_transform_calls_replaced=%ld kernel_launches=%ld\n", sbs, launches, blocks, calls,
                svt_hip_hook_encdec_sb_kernels());
    }
    if (g_enabled[SVT_HIP_HOOK_MD_PRE]) {
        long pictures, launches, jobs, min_jobs, calls, inter, hits, late, declined;
        double ms;
        svt_hip_hook_md_pre_stats(&pictures, &launches, &jobs, &min_jobs, &calls, &inter, &hits, &late, &declined, &ms);
        fprintf(stderr, "svt_hip_md_pre pictures=%ld launches=%ld blocks=%ld min_blocks_per_launch=%ld declined=%ld config_thread_ms=%.1f fast_loop_calls=%ld inter=%ld served_from_table=%ld "
                        "predicted_late=%ld verify_mismatches=%ld\n", pictures, launches, jobs, min_jobs, declined, ms, calls, inter, hits, late, svt_hip_hook_md_pre_mismatches());
        long gp, probes, served;
        svt_hip_hook_md_pre_subpel_stats(&gp, &probes, &served);
        fprintf(stderr, "svt_hip_md_pre_subpel grid_pictures=%ld probes=%ld served_from_grid=%ld\n", gp, probes, served);
        long bp, bs;
        svt_hip_hook_md_pre_compound_stats(&bp, &bs);
        fprintf(stderr, "svt_hip_md_pre_compound pair_table_pictures=%ld served_from_pair_table=%ld\n", bp, bs);
        long miss[10];
        svt_hip_hook_md_pre_misses(miss, 10);
        fprintf(stderr, "svt_hip_md_pre_misses compound=%ld motion_mode=%ld hbd=%ld later_pass=%ld shape=%ld no_table_yet=%ld reference=%ld vector=%ld border=%ld mark=%ld device_ms=%.1f\n", miss[0], miss[1],
                miss[2], miss[3], miss[4], miss[5], miss[6], miss[7], miss[8], miss[9], svt_hip_hook_md_pre_device_ms());
    }
    if (g_rtcd_installed) svt_hip_rtcd_report();   /* "svt_hip_rtcd_calls ..." / "svt_hip_rtcd_delegated ..." per wrapper */
}

/* ---- per-call wrappers: SvtHipRtcd member <-> the reference's global pointer of the same name ---------------------------------------- */
#define RTCD_SIMPLE(X)                                                                                                                      \
    X(svt_sad_loop_kernel) X(svt_nxm_sad_kernel) X(svt_av1_selfguided_restoration) X(svt_apply_selfguided_restoration)                     \
    X(svt_av1_compute_stats) X(svt_av1_compute_stats_highbd)                                                                                \
    X(svt_ext_all_sad_calculation_8x8_16x16) X(svt_ext_eight_sad_calculation_32x32_64x64)                                                   \
    X(svt_aom_quantize_b) X(svt_aom_highbd_quantize_b) X(svt_av1_quantize_fp) X(svt_av1_quantize_fp_32x32) X(svt_av1_quantize_fp_64x64)    \
    X(svt_av1_highbd_quantize_fp) X(svt_cdef_find_dir) X(svt_cdef_filter_block) X(svt_residual_kernel8bit) X(svt_residual_kernel16bit)     \
    X(svt_aom_upsampled_pred) X(svt_compute_interm_var_four8x8) X(svt_av1_inv_txfm_add)                                                     \
    X(svt_av1_convolve_2d_sr) X(svt_av1_convolve_x_sr) X(svt_av1_convolve_y_sr) X(svt_av1_convolve_2d_copy_sr)                              \
    X(svt_av1_highbd_convolve_2d_sr) X(svt_av1_highbd_convolve_x_sr) X(svt_av1_highbd_convolve_y_sr) X(svt_av1_highbd_convolve_2d_copy_sr) \
    X(svt_aom_subtract_block) X(svt_aom_highbd_subtract_block) X(sad_16b_kernel) X(variance_highbd) X(svt_nxm_sad_kernel_sub_sampled)       \
    X(svt_ext_sad_calculation_8x8_16x16) X(svt_ext_sad_calculation_32x32_64x64) X(svt_copy_rect8_8bit_to_16bit)                             \
    X(svt_compute_cdef_dist_8bit) X(svt_compute_cdef_dist_16bit) X(svt_search_one_dual)                                                     \
    X(svt_full_distortion_kernel32_bits) X(svt_full_distortion_kernel_cbf_zero32_bits) X(svt_spatial_full_distortion_kernel)                \
    X(svt_full_distortion_kernel16_bits) X(svt_aom_sse) X(svt_aom_highbd_sse) X(svt_aom_satd) X(svt_av1_block_error)                        \
    X(svt_get_proj_subspace) X(svt_av1_lowbd_pixel_proj_error) X(svt_av1_highbd_pixel_proj_error)                                           \
    X(svt_compute_mean_square_values_8x8) X(svt_compute_sub_mean_8x8) X(svt_aom_convolve8_horiz) X(svt_aom_convolve8_vert)                  \
    X(svt_av1_wiener_convolve_add_src) X(svt_av1_highbd_wiener_convolve_add_src) X(svt_aom_mse16x16) X(svt_aom_highbd_8_mse16x16) \
    X(svt_convert_8bit_to_16bit) X(svt_convert_16bit_to_8bit) X(svt_c_pack) X(svt_compressed_packmsb) X(svt_pack2d_16_bit_src_mul4) X(svt_unpack_avg)   \
    X(svt_un_pack2d_16_bit_src_mul4) X(svt_un_pack8_bit_data)                                                                              \
    X(svt_av1_jnt_convolve_2d) X(svt_av1_jnt_convolve_x) X(svt_av1_jnt_convolve_y) X(svt_av1_jnt_convolve_2d_copy)                          \
    X(svt_av1_highbd_jnt_convolve_2d) X(svt_av1_highbd_jnt_convolve_x) X(svt_av1_highbd_jnt_convolve_y) X(svt_av1_highbd_jnt_convolve_2d_copy) \
    X(svt_av1_build_compound_diffwtd_mask) X(svt_av1_build_compound_diffwtd_mask_highbd) X(svt_av1_build_compound_diffwtd_mask_d16)          \
    X(svt_aom_lowbd_blend_a64_d16_mask) X(svt_aom_highbd_blend_a64_d16_mask)
/* array members <-> the reference's individually named pointers */
#define RTCD_INDEXED(X)                                                                                                                      \
    X(svt_aom_lpf_horizontal, 0, svt_aom_lpf_horizontal_4) X(svt_aom_lpf_horizontal, 1, svt_aom_lpf_horizontal_6)                           \
    X(svt_aom_lpf_horizontal, 2, svt_aom_lpf_horizontal_8) X(svt_aom_lpf_horizontal, 3, svt_aom_lpf_horizontal_14)                          \
    X(svt_aom_lpf_vertical, 0, svt_aom_lpf_vertical_4) X(svt_aom_lpf_vertical, 1, svt_aom_lpf_vertical_6)                                   \
    X(svt_aom_lpf_vertical, 2, svt_aom_lpf_vertical_8) X(svt_aom_lpf_vertical, 3, svt_aom_lpf_vertical_14)                                  \
    X(svt_aom_highbd_lpf_horizontal, 0, svt_aom_highbd_lpf_horizontal_4) X(svt_aom_highbd_lpf_horizontal, 1, svt_aom_highbd_lpf_horizontal_6) \
    X(svt_aom_highbd_lpf_horizontal, 2, svt_aom_highbd_lpf_horizontal_8) X(svt_aom_highbd_lpf_horizontal, 3, svt_aom_highbd_lpf_horizontal_14) \
    X(svt_aom_highbd_lpf_vertical, 0, svt_aom_highbd_lpf_vertical_4) X(svt_aom_highbd_lpf_vertical, 1, svt_aom_highbd_lpf_vertical_6)       \
    X(svt_aom_highbd_lpf_vertical, 2, svt_aom_highbd_lpf_vertical_8) X(svt_aom_highbd_lpf_vertical, 3, svt_aom_highbd_lpf_vertical_14)      \
    X(svt_handle_transform64, 0, svt_handle_transform16x64) X(svt_handle_transform64, 1, svt_handle_transform32x64)                         \
    X(svt_handle_transform64, 2, svt_handle_transform64x16) X(svt_handle_transform64, 3, svt_handle_transform64x32)                         \
    X(svt_handle_transform64, 4, svt_handle_transform64x64)                                                                                  \
    X(handle_transform64_N2_N4, 0, handle_transform16x64_N2_N4) X(handle_transform64_N2_N4, 1, handle_transform32x64_N2_N4)                 \
    X(handle_transform64_N2_N4, 2, handle_transform64x16_N2_N4) X(handle_transform64_N2_N4, 3, handle_transform64x32_N2_N4)                 \
    X(handle_transform64_N2_N4, 4, handle_transform64x64_N2_N4)

/* the families with one pointer per block size (22 sizes, SVT_HIP_RTCD_BLOCK_SIZES order): what the sub-pel searches of mode decision / the temporal filter
 * call through mefn_ptr[] (svt_aom_variance{W}x{H}, svt_aom_sad{W}x{H}[x4d], the OBMC costs) — mefn_ptr is built from these pointers after this init
 * (EbEncHandle.c:1147), so replacing them here reaches every caller */
#define RTCD_BY_SIZE(X)                                                                                                                 \
    SVT_HIP_RTCD_BLOCK_SIZES(X##_sad) SVT_HIP_RTCD_BLOCK_SIZES(X##_sadx4d) SVT_HIP_RTCD_BLOCK_SIZES(X##_var) SVT_HIP_RTCD_BLOCK_SIZES(X##_var10) \
    SVT_HIP_RTCD_BLOCK_SIZES(X##_osad) SVT_HIP_RTCD_BLOCK_SIZES(X##_ovar) SVT_HIP_RTCD_BLOCK_SIZES(X##_osvar)
#define SAVE_sad(I, W, H)    t.svt_aom_sad[I] = (void *)svt_aom_sad##W##x##H;
#define SAVE_sadx4d(I, W, H) t.svt_aom_sadx4d[I] = (void *)svt_aom_sad##W##x##H##x4d;
#define SAVE_var(I, W, H)    t.svt_aom_variance[I] = (void *)svt_aom_variance##W##x##H;
#define SAVE_var10(I, W, H)  t.svt_aom_highbd_10_variance[I] = (void *)svt_aom_highbd_10_variance##W##x##H;
#define SAVE_osad(I, W, H)   t.svt_aom_obmc_sad[I] = (void *)svt_aom_obmc_sad##W##x##H;
#define SAVE_ovar(I, W, H)   t.svt_aom_obmc_variance[I] = (void *)svt_aom_obmc_variance##W##x##H;
#define SAVE_osvar(I, W, H)  t.svt_aom_obmc_sub_pixel_variance[I] = (void *)svt_aom_obmc_sub_pixel_variance##W##x##H;
#define PUT(member, I, name)                                              \
    if (in_list(list, #name)) {                                           \
        name = (void *)t.member[I];                                       \
        fprintf(stderr, "svt_hip_rtcd %s -> hip wrapper\n", #name);       \
    }
#define PUT_sad(I, W, H)    PUT(svt_aom_sad, I, svt_aom_sad##W##x##H)
#define PUT_sadx4d(I, W, H) PUT(svt_aom_sadx4d, I, svt_aom_sad##W##x##H##x4d)
#define PUT_var(I, W, H)    PUT(svt_aom_variance, I, svt_aom_variance##W##x##H)
#define PUT_var10(I, W, H)  PUT(svt_aom_highbd_10_variance, I, svt_aom_highbd_10_variance##W##x##H)
#define PUT_osad(I, W, H)   PUT(svt_aom_obmc_sad, I, svt_aom_obmc_sad##W##x##H)
#define PUT_ovar(I, W, H)   PUT(svt_aom_obmc_variance, I, svt_aom_obmc_variance##W##x##H)
#define PUT_osvar(I, W, H)  PUT(svt_aom_obmc_sub_pixel_variance, I, svt_aom_obmc_sub_pixel_variance##W##x##H)

static SvtHipRtcd g_rtcd_saved;
static char       g_rtcd_list[8192];
#define BACK(member, I, name) if (in_list(g_rtcd_list, #name)) name = (void *)g_rtcd_saved.member[I];
#define BACK_sad(I, W, H)    BACK(svt_aom_sad, I, svt_aom_sad##W##x##H)
#define BACK_sadx4d(I, W, H) BACK(svt_aom_sadx4d, I, svt_aom_sad##W##x##H##x4d)
#define BACK_var(I, W, H)    BACK(svt_aom_variance, I, svt_aom_variance##W##x##H)
#define BACK_var10(I, W, H)  BACK(svt_aom_highbd_10_variance, I, svt_aom_highbd_10_variance##W##x##H)
#define BACK_osad(I, W, H)   BACK(svt_aom_obmc_sad, I, svt_aom_obmc_sad##W##x##H)
#define BACK_ovar(I, W, H)   BACK(svt_aom_obmc_variance, I, svt_aom_obmc_variance##W##x##H)
#define BACK_osvar(I, W, H)  BACK(svt_aom_obmc_sub_pixel_variance, I, svt_aom_obmc_sub_pixel_variance##W##x##H)
static void restore_rtcd(void) {
    if (!g_rtcd_installed) return;
#define X(n) if (in_list(g_rtcd_list, #n)) n = (void *)g_rtcd_saved.n;
    RTCD_SIMPLE(X)
#undef X
#define X(m, i, n) if (in_list(g_rtcd_list, #n)) n = (void *)g_rtcd_saved.m[i];
    RTCD_INDEXED(X)
#undef X
    RTCD_BY_SIZE(BACK)
    g_rtcd_installed = 0;
}

static void install_rtcd(const char *list) {
    SvtHipRtcd t;
    memset(&t, 0, sizeof(t));
    /* what is installed now (the C / SIMD kernels) becomes each wrapper's failure fallback */
#define X(n) t.n = (void *)n;
    RTCD_SIMPLE(X)
#undef X
#define X(m, i, n) t.m[i] = (void *)n;
    RTCD_INDEXED(X)
#undef X
    RTCD_BY_SIZE(SAVE)
    g_rtcd_saved = t;   /* what svt_hip_hooks_enc_deinit puts back */
    snprintf(g_rtcd_list, sizeof(g_rtcd_list), "%s", list);
    if (svt_hip_setup_rtcd(g_rtcd_ctx, &t) != SVT_HIP_OK) {
        SVT_LOG("svt_hip_setup_rtcd failed (%s) - keeping the C kernels\n", svt_hip_last_error(g_rtcd_ctx));
        return;
    }
    g_rtcd_installed = 1;
#define X(n)                                                      \
    if (in_list(list, #n)) {                                      \
        n = (void *)t.n;                                          \
        fprintf(stderr, "svt_hip_rtcd %s -> hip wrapper\n", #n);  \
    }
    RTCD_SIMPLE(X)
#undef X
#define X(m, i, n)                                                \
    if (in_list(list, #n)) {                                      \
        n = (void *)t.m[i];                                       \
        fprintf(stderr, "svt_hip_rtcd %s -> hip wrapper\n", #n);  \
    }
    RTCD_INDEXED(X)
#undef X
    RTCD_BY_SIZE(PUT)
}

/* svt_av1_enc_deinit_handle, after the component (and with it every process thread) is gone: the dispatch table gets its own pointers back, device memory and
 * contexts are released, and a later encoder instance of the process initialises from scratch.  The counters stay for the report at exit. */
static pthread_mutex_t g_init_mu = PTHREAD_MUTEX_INITIALIZER;
static int             g_instances;   /* encoder instances of the process between their init and deinit: they share every object below */
/* svt_av1_enc_deinit_handle, BEFORE svt_av1_enc_component_de_init frees the instance's pictures: host ranges that were page-locked in place are released
 * (the last instance only: the table is shared) */
static long g_early_unpins;
long svt_hip_hooks_early_unpins(void) { return g_early_unpins; }
void svt_hip_hooks_enc_predeinit(void) {
    pthread_mutex_lock(&g_init_mu);
    if (g_inited && g_ctx) {
        /* The tables of page-locked ranges are shared by the instances of the process and do not know whose buffers they hold, and this instance is about to free
         * its pictures: EVERY range is released and forgotten.  With other instances still encoding, the contexts are quiesced first (every bridge call holds g_lock or
         * a pool mutex for as long as its copies are in flight; svt_hip_hooks_unlock_any drains before it unlocks) and page-locking stays on: the survivors' buffers
         * are registered again by the next call that uses them.  (Their device copies in the resident table stay; the dead instance's are evicted by the budget.) */
        const int others = g_instances > 1;
        if (others) {
            pthread_mutex_lock(&g_lock);
            for (int i = 0; i < g_pool_n; i++) pthread_mutex_lock(&g_pool_mu[i]);
            (void)svt_hip_sync(g_ctx);
            for (int i = 0; i < g_pool_n; i++) (void)svt_hip_sync(g_pool[i]);
            svt_hip_md_bridge_quiesce();   /* hook "md_pre" issues on its own context */
            g_early_unpins++;
        }
        svt_hip_resident_unpin_all(g_ctx, others);
        svt_hip_lf_bridge_unpin(g_ctx, others);
        if (others) {
            svt_hip_md_bridge_resume();
            for (int i = g_pool_n - 1; i >= 0; i--) pthread_mutex_unlock(&g_pool_mu[i]);
            pthread_mutex_unlock(&g_lock);
        }
    }
    pthread_mutex_unlock(&g_init_mu);
}
void svt_hip_hooks_enc_deinit(void) {
    pthread_mutex_lock(&g_init_mu);
    if (!g_inited || --g_instances > 0) { pthread_mutex_unlock(&g_init_mu); return; }   /* another instance still runs on these contexts */
    svt_hip_hooks_report_once();
    restore_rtcd();
    if (g_ctx) {
        svt_hip_lf_bridge_release(g_ctx);
        svt_hip_md_bridge_release(g_ctx);
        svt_hip_resident_release_all(g_ctx);   /* before the block cache is emptied */
        pthread_mutex_lock(&g_alloc_mu);
        for (int c = 0; c < ALLOC_CLASSES; c++) {
            for (int i = 0; i < g_alloc_n[c]; i++) svt_hip_free(g_ctx, g_alloc_free[c][i]);
            g_alloc_n[c] = 0;
        }
        g_alloc_cached = 0;
        pthread_mutex_unlock(&g_alloc_mu);
    }
    if (g_rtcd_ctx) svt_hip_rtcd_release();   /* only a process that installed wrappers binds this symbol (the CPU test double does not have it) */
    for (int i = 0; i < g_pool_n; i++) { svt_hip_destroy(g_pool[i]); g_pool[i] = NULL; pthread_mutex_destroy(&g_pool_mu[i]); }
    g_pool_n = 0;
    if (g_rtcd_ctx) { svt_hip_destroy(g_rtcd_ctx); g_rtcd_ctx = NULL; }
    if (g_ctx) { svt_hip_destroy(g_ctx); g_ctx = NULL; }
    memset(g_enabled, 0, sizeof(g_enabled));
    g_inited = 0;
    g_instances = 0;
    pthread_mutex_unlock(&g_init_mu);
}

static void enc_init_locked(int target_socket);
void svt_hip_hooks_enc_init(int target_socket) {
    pthread_mutex_lock(&g_init_mu);
    g_instances++;
    if (!g_inited) { g_inited = 1; enc_init_locked(target_socket); }
    pthread_mutex_unlock(&g_init_mu);
}
static void enc_init_locked(int target_socket) {
    const char *hooks = getenv("SVT_HIP_HOOKS"), *rtcd = getenv("SVT_HIP_RTCD"), *dev = getenv("SVT_HIP_DEVICE");
    g_verbose = getenv("SVT_HIP_VERBOSE") && atoi(getenv("SVT_HIP_VERBOSE"));
    int any = rtcd && *rtcd;
    for (int i = 0; i < SVT_HIP_HOOK_COUNT; i++) {
        g_enabled[i] = k_hook_opt_in[i] ? in_list_exact(hooks, k_hook_name[i]) : in_list(hooks, k_hook_name[i]);
        any |= g_enabled[i];
    }
    if (!any) return;   /* the patched encoder is the reference encoder */
    /* target_socket is the reference's CPU-affinity knob: it doubles as the GPU ordinal only when it names a device (eight instances started with --socket 0..7 on
     * an 8-GPU node); on a dual-socket host with one GPU `--socket 1` keeps device 0.  SVT_HIP_DEVICE is taken literally. */
    int n_dev = 0;
    (void)svt_hip_device_count(&n_dev);
    const int device = dev ? atoi(dev) : (target_socket >= 0 && target_socket < n_dev ? target_socket : 0);
    fprintf(stderr, "svt_hip_device ordinal=%d of %d (%s)\n", device, n_dev, dev ? "SVT_HIP_DEVICE" : (target_socket >= 0 && target_socket < n_dev ? "target_socket" : "default"));
    g_device = device;
    if (svt_hip_init(device, &g_ctx) != SVT_HIP_OK) {
        /* error convention: never fail through the kernel surface — log, keep the C path */
        SVT_LOG("svt_hip_init failed - SVT_HIP_HOOKS / SVT_HIP_RTCD ignored, keeping the C kernels\n");
        if (!(getenv("SVT_HIP_SEGMENTS") && !atoi(getenv("SVT_HIP_SEGMENTS"))))   /* svt_hip_hooks_segments ran before the device was known (the buffer configuration comes first) */
            SVT_LOG("svt_hip: the ME / TF / CDEF / restoration segment counts were chosen for the hooks (fewer, larger segments): the C kernels now run with less "
                    "parallelism inside a picture; SVT_HIP_SEGMENTS=0 or an unset SVT_HIP_HOOKS keeps the reference's counts\n");
        g_ctx = NULL;
        return;
    }
    g_res_on = !(getenv("SVT_HIP_RESIDENT") && !atoi(getenv("SVT_HIP_RESIDENT")));   /* default on since round 4 (verified on the MI355X: profiles/r04/resident_first_call_summary.txt); SVT_HIP_RESIDENT=0: a band upload per segment again */
    /* SVT_HIP_RESIDENT_FAULT=1, for the tests only: a plane's later announcements are ignored (its copy goes stale) */
    svt_hip_resident_configure(g_res_on, getenv("SVT_HIP_RESIDENT_MB") ? (size_t)atol(getenv("SVT_HIP_RESIDENT_MB")) << 20 : (size_t)16384 << 20 /* ~27 MB per 4K picture in flight, of 288 GB */,
                               getenv("SVT_HIP_RESIDENT_FAULT") && atoi(getenv("SVT_HIP_RESIDENT_FAULT")), svt_hip_hooks_malloc, svt_hip_hooks_free);
    if (getenv("SVT_HIP_ALLOC_CACHE_MB")) g_alloc_limit = (size_t)atol(getenv("SVT_HIP_ALLOC_CACHE_MB")) << 20;
    /* SVT_HIP_PIN=0: the reference's picture buffers are not page-locked in place (A/B knob; default on) */
    g_pin_on = !(getenv("SVT_HIP_PIN") && !atoi(getenv("SVT_HIP_PIN")));
    svt_hip_resident_configure_blocks(alloc_block_size, g_pin_on);
    if (!(getenv("SVT_HIP_WARMUP") && !atoi(getenv("SVT_HIP_WARMUP")))) {   /* every kernel's code object is loaded here, not under the first pictures' clock */
        const long long t0 = now_ns();
        const int       rc = svt_hip_warmup(g_ctx);
        fprintf(stderr, "svt_hip_warmup rc=%d ms=%.1f\n", rc, (now_ns() - t0) / 1e6);
    }
    {   /* the pool of the source-side bridges; 0 = everything on the main context (the round-2 behaviour) */
        const char *pc = getenv("SVT_HIP_CONTEXTS");
        int         n = pc ? atoi(pc) : 4;
        n = n < 0 ? 0 : (n > SVT_HIP_POOL_MAX ? SVT_HIP_POOL_MAX : n);
        for (int i = 0; i < n; i++) {
            if (svt_hip_init(device, &g_pool[i]) != SVT_HIP_OK) { g_pool[i] = NULL; break; }
            pthread_mutex_init(&g_pool_mu[i], NULL);
            g_pool_n = i + 1;
        }
    }
    if (rtcd && *rtcd) {
        if (svt_hip_init(device, &g_rtcd_ctx) == SVT_HIP_OK) install_rtcd(rtcd);
        else SVT_LOG("svt_hip_init (per-call wrappers) failed - SVT_HIP_RTCD ignored, keeping the C kernels\n");
    }
    g_reported = 0;
    atexit(svt_hip_hooks_report_once);
}
