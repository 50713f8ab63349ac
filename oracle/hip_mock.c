/*
 * hip_mock.c — CPU test double of libsvtav1_hip.so.  TEST INFRASTRUCTURE ONLY (oracle/): it exists so that the reference-side glue in
 * integration/ (the patched process loops, svt_hip_hooks.c, the bridges) can be exercised end to end by a real encode on a box without
 * a GPU — tests/test_encode_e2e_cpu.py runs SvtAv1EncApp with every hook on against this library and requires the bitstream of the
 * unpatched reference.  "Device" pointers are host pointers; every kernel entry point is answered by the oracle restatement (oracle/
 * *_oracle.c, each pinned to the reference by tests/test_oracle_vs_ref.py).  The HOST-side logic of the product
 * (svt-av1_amd/csrc/svt_hip_host.cpp: edge builder = set_lpf_parameters, level-search control flow) is compiled in unchanged, so this
 * run pins exactly that code against the reference encoder.
 *
 * Never shipped, never loaded by the product: it is built into oracle/_ref/mock/libsvtav1_hip.so and only found when a test puts that
 * directory on LD_LIBRARY_PATH.  On the GPU box the same encoder binary loads the real svt-av1_amd/libsvtav1_hip.so.
 * Only the entry points the glue calls are provided.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "../include/svt_hip.h"
#include "../svt-av1_amd/csrc/svt_hip_host.h"
#include "svt_oracle.h"

struct SvtHipCtx { char err[256]; };

/* SVT_HIP_MOCK_PERTURB=<stage>: deliberately wrong output of one stage, so that the end-to-end test can prove that the bitstream / the
 * reconstruction really depend on what each hook returns (tests/test_encode_e2e.py::test_every_hook_matters) */
static int perturb(const char *stage) {
    const char *e = getenv("SVT_HIP_MOCK_PERTURB");
    return e && !strcmp(e, stage);
}

int svt_hip_device_count(int *n);
int svt_hip_init(int device_id, SvtHipCtx **ctx) {
    int n = 0;
    (void)svt_hip_device_count(&n);   /* SVT_HIP_MOCK_DEVICES (default 1): an ordinal that names no device is refused, like the product */
    *ctx = NULL;
    if (device_id < 0 || device_id >= n) return SVT_HIP_ERR_NO_DEVICE;
    *ctx = (SvtHipCtx *)calloc(1, sizeof(SvtHipCtx));
    fprintf(stderr, "svt_hip MOCK (oracle/hip_mock.c): CPU test double, not the product\n");
    return *ctx ? SVT_HIP_OK : SVT_HIP_ERR_RUNTIME;
}
void        svt_hip_destroy(SvtHipCtx *c) { free(c); }
const char *svt_hip_last_error(const SvtHipCtx *c) { return c ? c->err : "null context"; }
int svt_hip_set_stream(SvtHipCtx *c, void *s) { (void)c; (void)s; return SVT_HIP_OK; }
int svt_hip_sync(SvtHipCtx *c) { (void)c; return SVT_HIP_OK; }
int svt_hip_malloc(SvtHipCtx *c, void **p, size_t bytes) { (void)c; *p = malloc(bytes ? bytes : 1); return *p ? SVT_HIP_OK : SVT_HIP_ERR_RUNTIME; }
int svt_hip_free(SvtHipCtx *c, void *p) { (void)c; free(p); return SVT_HIP_OK; }
int svt_hip_memcpy_h2d(SvtHipCtx *c, void *d, const void *h, size_t n) { (void)c; memcpy(d, h, n); return SVT_HIP_OK; }
int svt_hip_memcpy_d2h(SvtHipCtx *c, void *h, const void *d, size_t n) { (void)c; memcpy(h, d, n); return SVT_HIP_OK; }
int svt_hip_memcpy_d2d(SvtHipCtx *c, void *d, const void *s, size_t n) { (void)c; memmove(d, s, n); return SVT_HIP_OK; }
int svt_hip_memcpy_h2d_async(SvtHipCtx *c, void *d, const void *h, size_t n) { return svt_hip_memcpy_h2d(c, d, h, n); }
int svt_hip_memcpy_d2h_async(SvtHipCtx *c, void *h, const void *d, size_t n) { return svt_hip_memcpy_d2h(c, h, d, n); }
int svt_hip_host_register(SvtHipCtx *c, void *h, size_t n) { (void)c; (void)h; (void)n; return SVT_HIP_OK; }
int svt_hip_host_unregister(SvtHipCtx *c, void *h) { (void)c; (void)h; return SVT_HIP_OK; }
int svt_hip_host_alloc(SvtHipCtx *c, void **h, size_t n) { (void)c; *h = malloc(n ? n : 1); return *h ? SVT_HIP_OK : SVT_HIP_ERR_RUNTIME; }
int svt_hip_host_free(SvtHipCtx *c, void *h) { (void)c; free(h); return SVT_HIP_OK; }
int svt_hip_device_count(int *n) { *n = getenv("SVT_HIP_MOCK_DEVICES") ? atoi(getenv("SVT_HIP_MOCK_DEVICES")) : 1; return SVT_HIP_OK; }
int svt_hip_warmup(SvtHipCtx *c) { (void)c; return SVT_HIP_OK; }
int svt_hip_memcpy2d_h2d(SvtHipCtx *c, void *d, size_t dpitch, const void *h, size_t hpitch, size_t wbytes, size_t rows) {
    (void)c;
    for (size_t y = 0; y < rows; y++) memcpy((uint8_t *)d + y * dpitch, (const uint8_t *)h + y * hpitch, wbytes);
    return SVT_HIP_OK;
}
int svt_hip_memcpy2d_d2h(SvtHipCtx *c, void *h, size_t hpitch, const void *d, size_t dpitch, size_t wbytes, size_t rows) {
    (void)c;
    for (size_t y = 0; y < rows; y++) memcpy((uint8_t *)h + y * hpitch, (const uint8_t *)d + y * dpitch, wbytes);
    return SVT_HIP_OK;
}
int svt_hip_memcpy2d_h2d_async(SvtHipCtx *c, void *d, size_t dpitch, const void *h, size_t hpitch, size_t wbytes, size_t rows) { return svt_hip_memcpy2d_h2d(c, d, dpitch, h, hpitch, wbytes, rows); }
int svt_hip_memcpy2d_d2h_async(SvtHipCtx *c, void *h, size_t hpitch, const void *d, size_t dpitch, size_t wbytes, size_t rows) { return svt_hip_memcpy2d_d2h(c, h, hpitch, d, dpitch, wbytes, rows); }

/* ------------------------------------------------------------------ ME */
int svt_hip_me_fullpel_frame_dev(SvtHipCtx *c, const uint8_t *src, const uint8_t *ref, int stride, int org_x, int org_y,
                                 const SvtHipSbSearch *sbs, int n_sb, int sub_sad, uint32_t *best_sad, uint32_t *best_mv) {
    (void)c;
    orc_me_fullpel_frame(src, ref, stride, org_x, org_y, (const OrcSbSearch *)sbs, n_sb, sub_sad, best_sad, best_mv, 0, n_sb);
    if (perturb("me") || perturb("tf_me"))   /* the temporal filter's search uses the same entry point */
        for (int i = 0; i < n_sb * 85; i++) best_mv[i] = (best_mv[i] & 0xffff0000u) | ((best_mv[i] + 8) & 0xffffu);   /* x_mv + 2 px */
    return SVT_HIP_OK;
}
/* mode decision, picture-level precompute: the oracle's restatement behind the product's entry point */
int svt_hip_md_fullpel_sad_picture_dev(SvtHipCtx *c, const uint8_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu *pus,
                                       int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *mv, uint32_t *sad) {
    (void)c;
    if (n_pus < 1 || n_pus > SVT_HIP_MD_MAX_PUS || n_refs < 1 || n_refs > SVT_HIP_MD_MAX_REFS) return SVT_HIP_ERR_BAD_ARG;
    uint8_t        pu4[SVT_HIP_MD_MAX_PUS][4];
    const uint8_t *planes[SVT_HIP_MD_MAX_REFS];
    int            strides[SVT_HIP_MD_MAX_REFS], box[SVT_HIP_MD_MAX_REFS][4];
    for (int i = 0; i < n_pus; i++) { pu4[i][0] = pus[i].x; pu4[i][1] = pus[i].y; pu4[i][2] = pus[i].w; pu4[i][3] = pus[i].h; }
    for (int r = 0; r < n_refs; r++) {
        planes[r] = refs[r].d_plane; strides[r] = refs[r].stride;
        box[r][0] = refs[r].x_min; box[r][1] = refs[r].y_min; box[r][2] = refs[r].x_max; box[r][3] = refs[r].y_max;
    }
    orc_md_fullpel_sad_picture(src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, (const uint8_t(*)[4])pu4, n_refs, planes, strides, (const int(*)[4])box, mv, sad);
    if (perturb("md_pre"))
        for (size_t i = 0; i < (size_t)n_sb * n_pus * n_refs; i++)
            if (sad[i] != 0xffffffffu) sad[i] = sad[i] * 3 + 1000;   /* reorders the candidates of stage 0 */
    return SVT_HIP_OK;
}
int svt_hip_md_fullpel_avg_sad_picture_dev(SvtHipCtx *c, const uint8_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu *pus,
                                           int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *mv, int n_pairs, const uint8_t (*pairs)[2], uint32_t *sad) {
    (void)c;
    if (n_pus < 1 || n_pus > SVT_HIP_MD_MAX_PUS || n_refs < 1 || n_refs > SVT_HIP_MD_MAX_REFS || n_pairs < 1 || n_pairs > SVT_HIP_MD_MAX_PAIRS) return SVT_HIP_ERR_BAD_ARG;
    uint8_t        pu4[SVT_HIP_MD_MAX_PUS][4];
    const uint8_t *planes[SVT_HIP_MD_MAX_REFS];
    int            strides[SVT_HIP_MD_MAX_REFS], box[SVT_HIP_MD_MAX_REFS][4];
    for (int i = 0; i < n_pus; i++) { pu4[i][0] = pus[i].x; pu4[i][1] = pus[i].y; pu4[i][2] = pus[i].w; pu4[i][3] = pus[i].h; }
    for (int r = 0; r < n_refs; r++) {
        planes[r] = refs[r].d_plane; strides[r] = refs[r].stride;
        box[r][0] = refs[r].x_min; box[r][1] = refs[r].y_min; box[r][2] = refs[r].x_max; box[r][3] = refs[r].y_max;
    }
    for (int i = 0; i < n_pairs; i++) if (pairs[i][0] >= n_refs || pairs[i][1] >= n_refs) return SVT_HIP_ERR_BAD_ARG;
    orc_md_fullpel_avg_sad_picture(src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, (const uint8_t(*)[4])pu4, n_refs, planes, strides, (const int(*)[4])box, mv, n_pairs, pairs, sad);
    if (perturb("md_pre_compound"))
        for (size_t i = 0; i < (size_t)n_sb * n_pus * n_pairs; i++)
            if (sad[i] != 0xffffffffu) sad[i] = sad[i] * 3 + 1000;   /* reorders the compound candidates of stage 0 */
    return SVT_HIP_OK;
}
/* the 16-bit forms: the bit depth of the clip is the context's (svt_hip_create ... the double keeps one global: 10, the only high bit depth the hooks are used with) */
int svt_hip_md_fullpel_sad_picture_hbd_dev(SvtHipCtx *c, const uint16_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu *pus,
                                           int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *mv, uint32_t *sad) {
    (void)c;
    if (n_pus < 1 || n_pus > SVT_HIP_MD_MAX_PUS || n_refs < 1 || n_refs > SVT_HIP_MD_MAX_REFS) return SVT_HIP_ERR_BAD_ARG;
    uint8_t         pu4[SVT_HIP_MD_MAX_PUS][4];
    const uint16_t *planes[SVT_HIP_MD_MAX_REFS];
    int             strides[SVT_HIP_MD_MAX_REFS], box[SVT_HIP_MD_MAX_REFS][4];
    for (int i = 0; i < n_pus; i++) { pu4[i][0] = pus[i].x; pu4[i][1] = pus[i].y; pu4[i][2] = pus[i].w; pu4[i][3] = pus[i].h; }
    for (int r = 0; r < n_refs; r++) {
        planes[r] = (const uint16_t *)refs[r].d_plane; strides[r] = refs[r].stride;
        box[r][0] = refs[r].x_min; box[r][1] = refs[r].y_min; box[r][2] = refs[r].x_max; box[r][3] = refs[r].y_max;
    }
    orc_md_fullpel_sad_picture16(src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, (const uint8_t(*)[4])pu4, n_refs, planes, strides, (const int(*)[4])box, mv, sad);
    if (perturb("md_pre"))
        for (size_t i = 0; i < (size_t)n_sb * n_pus * n_refs; i++)
            if (sad[i] != 0xffffffffu) sad[i] = sad[i] * 3 + 1000;
    return SVT_HIP_OK;
}
int svt_hip_md_fullpel_avg_sad_picture_hbd_dev(SvtHipCtx *c, const uint16_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu *pus,
                                               int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *mv, int n_pairs, const uint8_t (*pairs)[2], uint32_t *sad) {
    (void)c;
    if (n_pus < 1 || n_pus > SVT_HIP_MD_MAX_PUS || n_refs < 1 || n_refs > SVT_HIP_MD_MAX_REFS || n_pairs < 1 || n_pairs > SVT_HIP_MD_MAX_PAIRS) return SVT_HIP_ERR_BAD_ARG;
    uint8_t         pu4[SVT_HIP_MD_MAX_PUS][4];
    const uint16_t *planes[SVT_HIP_MD_MAX_REFS];
    int             strides[SVT_HIP_MD_MAX_REFS], box[SVT_HIP_MD_MAX_REFS][4];
    for (int i = 0; i < n_pus; i++) { pu4[i][0] = pus[i].x; pu4[i][1] = pus[i].y; pu4[i][2] = pus[i].w; pu4[i][3] = pus[i].h; }
    for (int r = 0; r < n_refs; r++) {
        planes[r] = (const uint16_t *)refs[r].d_plane; strides[r] = refs[r].stride;
        box[r][0] = refs[r].x_min; box[r][1] = refs[r].y_min; box[r][2] = refs[r].x_max; box[r][3] = refs[r].y_max;
    }
    for (int i = 0; i < n_pairs; i++) if (pairs[i][0] >= n_refs || pairs[i][1] >= n_refs) return SVT_HIP_ERR_BAD_ARG;
    orc_md_fullpel_avg_sad_picture16(src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, (const uint8_t(*)[4])pu4, n_refs, planes, strides, (const int(*)[4])box, mv, n_pairs, pairs, 10, sad);
    if (perturb("md_pre_compound"))
        for (size_t i = 0; i < (size_t)n_sb * n_pus * n_pairs; i++)
            if (sad[i] != 0xffffffffu) sad[i] = sad[i] * 3 + 1000;
    return SVT_HIP_OK;
}
int svt_hip_md_subpel_grid_picture_dev(SvtHipCtx *c, const uint8_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu *pus,
                                       int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *mv, int bank, uint32_t *out) {
    (void)c;
    if (n_pus < 1 || n_pus > SVT_HIP_MD_MAX_PUS || n_refs < 1 || n_refs > SVT_HIP_MD_MAX_REFS) return SVT_HIP_ERR_BAD_ARG;
    uint8_t        pu4[SVT_HIP_MD_MAX_PUS][4];
    const uint8_t *planes[SVT_HIP_MD_MAX_REFS];
    int            strides[SVT_HIP_MD_MAX_REFS], box[SVT_HIP_MD_MAX_REFS][4];
    for (int i = 0; i < n_pus; i++) { pu4[i][0] = pus[i].x; pu4[i][1] = pus[i].y; pu4[i][2] = pus[i].w; pu4[i][3] = pus[i].h; }
    for (int r = 0; r < n_refs; r++) {
        planes[r] = refs[r].d_plane; strides[r] = refs[r].stride;
        box[r][0] = refs[r].x_min; box[r][1] = refs[r].y_min; box[r][2] = refs[r].x_max; box[r][3] = refs[r].y_max;
    }
    orc_md_subpel_grid_picture(src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, (const uint8_t(*)[4])pu4, n_refs, planes, strides, (const int(*)[4])box, mv, bank, out);
    if (perturb("md_pre_subpel"))
        for (size_t i = 0; i < (size_t)n_sb * n_pus * n_refs * 98; i += 2)
            if (out[i] != 0xffffffffu) out[i] += (uint32_t)((i * 2654435761u) >> 22);   /* a different wrong variance per position: the tree's comparisons flip */
    return SVT_HIP_OK;
}
int svt_hip_md_halfpel_grid_picture_dev(SvtHipCtx *c, const uint8_t *src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu *pus,
                                        int n_refs, const SvtHipMdRefPlane *refs, const uint32_t *mv, int bank, uint32_t *out) {
    const size_t n = (size_t)n_sb * n_pus * n_refs;
    uint32_t    *full = (uint32_t *)malloc((n ? n : 1) * 98 * sizeof(uint32_t));
    if (!full) return SVT_HIP_ERR_BAD_ARG;
    const int rc = svt_hip_md_subpel_grid_picture_dev(c, src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, mv, bank, full);
    for (size_t i = 0; rc == SVT_HIP_OK && i < n; i++)
        for (int k = 0; k < 9; k++) {   /* positions (1, 3, 5) x (1, 3, 5) of the 7 x 7 table */
            const int g = 7 * (2 * (k / 3) + 1) + 2 * (k % 3) + 1;
            out[18 * i + 2 * k] = full[98 * i + 2 * g]; out[18 * i + 2 * k + 1] = full[98 * i + 2 * g + 1];
        }
    free(full);
    return rc;
}
int svt_hip_me_set_big_windows(SvtHipCtx *c, int enable) { (void)c; (void)enable; return SVT_HIP_OK; }   /* the double's search has one instance */
int svt_hip_me_get_big_windows(SvtHipCtx *c, int *enabled) { (void)c; *enabled = 1; return SVT_HIP_OK; }
int svt_hip_me_fullpel_frame(SvtHipCtx *c, const uint8_t *src, const uint8_t *ref, int stride, int plane_rows, int org_x, int org_y,
                             const SvtHipSbSearch *sbs, int n_sb, int sub_sad, uint32_t *best_sad, uint32_t *best_mv) {
    (void)plane_rows;   /* search areas above 65 536 candidates: the product takes its strip kernel, the same results */
    return svt_hip_me_fullpel_frame_dev(c, src, ref, stride, org_x, org_y, sbs, n_sb, sub_sad, best_sad, best_mv);
}
int svt_hip_sad_loop_batch_dev(SvtHipCtx *c, const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride,
                               const SvtHipSadLoop *searches, int n, uint32_t *best_sad, int16_t *best_xy) {
    (void)c;
    orc_sad_loop_batch(src, src_stride, ref, ref_stride, searches, 0, n, best_sad, best_xy);
    if (perturb("hme"))
        for (int i = 0; i < n; i++) { best_xy[2 * i] = (int16_t)(best_xy[2 * i] + 24); best_xy[2 * i + 1] = (int16_t)(best_xy[2 * i + 1] - 16); }   /* every search lands far off */
    return SVT_HIP_OK;
}

/* joint_strength_search_dual (Encoder/Codec/EbEncCdef.c:1140-1164) on top of svt_search_one_dual_c (:1070-1118), restated */
static uint64_t mock_search_one_dual(int *lev0, int *lev1, int nb, const uint64_t *m0, const uint64_t *m1, int sb_count, int start_gi, int end_gi) {
    static uint64_t tot[64][64];
    uint64_t        best_tot = (uint64_t)1 << 63;
    int             b0 = 0, b1 = 0;
    memset(tot, 0, sizeof(tot));
    for (int i = 0; i < sb_count; i++) {
        uint64_t best = (uint64_t)1 << 63;
        for (int g = 0; g < nb; g++) { const uint64_t c = m0[(size_t)i * 64 + lev0[g]] + m1[(size_t)i * 64 + lev1[g]]; if (c < best) best = c; }
        for (int j = start_gi; j < end_gi; j++)
            for (int k = start_gi; k < end_gi; k++) { const uint64_t c = m0[(size_t)i * 64 + j] + m1[(size_t)i * 64 + k]; tot[j][k] += c < best ? c : best; }
    }
    for (int j = start_gi; j < end_gi; j++)
        for (int k = start_gi; k < end_gi; k++)
            if (tot[j][k] < best_tot) { best_tot = tot[j][k]; b0 = j; b1 = k; }
    lev0[nb] = b0; lev1[nb] = b1;
    return best_tot;
}
int svt_hip_cdef_joint_strength_search_dev(SvtHipCtx *c, const uint64_t *m0, const uint64_t *m1, int sb_count, int *lev0, int *lev1, int nb, int start_gi, int end_gi,
                                           uint64_t *work) {
    (void)c;
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;   /* the static table above */
    pthread_mutex_lock(&mu);
    uint64_t tot = (uint64_t)1 << 63;
    for (int i = 0; i < nb; i++) tot = mock_search_one_dual(lev0, lev1, i, m0, m1, sb_count, start_gi, end_gi);
    for (int i = 0; i < 4 * nb; i++) {
        for (int j = 0; j < nb - 1; j++) { lev0[j] = lev0[j + 1]; lev1[j] = lev1[j + 1]; }
        tot = mock_search_one_dual(lev0, lev1, nb - 1, m0, m1, sb_count, start_gi, end_gi);
    }
    pthread_mutex_unlock(&mu);
    if (perturb("cdef_finish")) { lev0[0] = (lev0[0] + 5) % (end_gi > 0 ? end_gi : 1); tot += 12345; }
    work[0] = tot;
    return SVT_HIP_OK;
}

int svt_hip_cdef_strength_select_dev(SvtHipCtx *c, const uint64_t *m0, const uint64_t *m1, int sb_count, int start_gi, int end_gi, void *state, size_t state_bytes) {
    if (state_bytes < sizeof(SvtHipCdefSelectResult)) return SVT_HIP_ERR_BAD_ARG;
    SvtHipCdefSelectResult *r = (SvtHipCdefSelectResult *)state;
    memset(r, 0, sizeof(*r));
    for (int ci = 0; ci < 4; ci++) {
        int      lev0[8] = {0}, lev1[8] = {0};
        uint64_t work[1];
        svt_hip_cdef_joint_strength_search_dev(c, m0, m1, sb_count, lev0, lev1, 1 << ci, start_gi, end_gi, work);
        for (int i = 0; i < 8; i++) { r->lev0[ci][i] = lev0[i]; r->lev1[ci][i] = lev1[i]; }
        r->tot_mse[ci] = work[0];
    }
    return SVT_HIP_OK;
}
int svt_hip_set_cdef_select_form(SvtHipCtx *c, int form) { (void)c; return form < -1 || form > 1 ? SVT_HIP_ERR_BAD_ARG : SVT_HIP_OK; }
int svt_hip_cdef_strength_select_multi_dev(SvtHipCtx *c, int n_pictures, const uint64_t *const *m0, const uint64_t *const *m1, int sb_count, int start_gi, int end_gi,
                                           void *const *states, size_t state_bytes) {
    for (int i = 0; i < n_pictures; i++) {
        const int rc = svt_hip_cdef_strength_select_dev(c, m0[i], m1[i], sb_count, start_gi, end_gi, states[i], state_bytes);
        if (rc != SVT_HIP_OK) return rc;
    }
    return SVT_HIP_OK;
}

int svt_hip_cdef_finish_dev(SvtHipCtx *c, const uint64_t *m0, const uint64_t *m1, int sb_count, const void *state, uint64_t lambda, const int32_t *sb_fb, SvtHipCdefFinish *out,
                            int32_t *sel_gi, uint8_t *fb_y, uint8_t *fb_uv) {
    (void)c;
    const SvtHipCdefSelectResult *r = (const SvtHipCdefSelectResult *)state;
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (sb_count > 0 ? sb_count : 1));
    out->cdef_bits = orc_cdef_finish(m0, m1, sb_count, r->lev0, r->lev1, r->tot_mse, lambda, out->y_strength, out->uv_strength, sel, &out->best_cost);
    out->nb_strengths = 1 << out->cdef_bits;
    for (int i = 0; i < sb_count; i++) {
        if (sel_gi) sel_gi[i] = sel[i];
        const int fb = sb_fb ? sb_fb[i] : i;
        if (fb_y) fb_y[fb] = (uint8_t)out->y_strength[sel[i]];
        if (fb_uv) fb_uv[fb] = (uint8_t)out->uv_strength[sel[i]];
    }
    free(sel);
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------ mode decision (hook "md_tx"): coefficients only, 16-bit planes, blocks up to 32 x 32 */
int svt_hip_fwd_txfm_quant_batch_dev(SvtHipCtx *c, int tx_size, int pix_bytes, const void *src, int src_stride, const void *pred, int pred_stride, const uint32_t *descs, int nblk,
                                     const SvtHipQuantParams *qp, const SvtHipScanTables *scans, int32_t *coeff, int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob, int32_t *cul,
                                     uint64_t *energy) {
    (void)scans;
    static const uint8_t tw[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64}, th[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};
    if (pix_bytes != 2 || !coeff || qcoeff || dqcoeff || eob || cul || energy || tx_size < 0 || tx_size > 18 || tw[tx_size] > 32 || th[tx_size] > 32) {
        snprintf(c->err, sizeof(c->err), "mock: svt_hip_fwd_txfm_quant_batch_dev is only built for the coefficient-only form of the md_tx hook");
        return SVT_HIP_ERR_UNSUPPORTED;
    }
    const int w = tw[tx_size], h = th[tx_size];
    for (int b = 0; b < nblk; b++) {
        const int x = (int)(descs[b] & 0x3FFF), y = (int)((descs[b] >> 14) & 0x3FFF), tt = (int)(descs[b] >> 28);
        int16_t r[32 * 32];
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++)
                r[i * w + j] = (int16_t)((int)((const uint16_t *)src)[(size_t)(y + i) * src_stride + x + j] - (int)((const uint16_t *)pred)[(size_t)(y + i) * pred_stride + x + j]);
        orc_estimate_transform(r, (uint32_t)w, coeff + (size_t)b * w * h, tt, tx_size, 10, qp ? qp->coeff_shape : 0);
        if (perturb("md_tx") && b == 1) coeff[(size_t)b * w * h] += 64;
    }
    return SVT_HIP_OK;
}

/* the mixed-size form the encode pass's hook "encdec_tx" uses: the job lists one after the other (coefficient-only jobs) */
int svt_hip_fwd_txfm_quant_multi_dev(SvtHipCtx *c, int pix_bytes, const SvtHipFwdTxJob *jobs, int njobs) {
    for (int j = 0; j < njobs; j++) {
        const SvtHipFwdTxJob *J = &jobs[j];
        const int rc = svt_hip_fwd_txfm_quant_batch_dev(c, J->tx_size, pix_bytes, J->d_src, J->src_stride, J->d_pred, J->pred_stride, J->d_descs, J->nblk, &J->qp, NULL, J->d_coeff,
                                                        J->d_qcoeff, J->d_dqcoeff, J->d_eob, J->d_cul_level, J->d_energy);
        if (rc != SVT_HIP_OK) return rc;
        if (perturb("encdec_tx") && j == 0) J->d_coeff[0] += 96;
    }
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------ mode decision (hook "md_subpel"): the candidates of a sub-pel round */
int svt_hip_upsampled_pred_batch_dev(SvtHipCtx *c, const uint8_t *ref, int ref_stride, uint8_t *dst, const SvtHipUpsampledBlk *blks, int n) {
    (void)c;
    for (int i = 0; i < n; i++)
        orc_upsampled_pred(ref + blks[i].ref_off, ref_stride, dst + blks[i].dst_off, blks[i].w, blks[i].h, blks[i].subpel_x_q3, blks[i].subpel_y_q3, blks[i].bank);
    return SVT_HIP_OK;
}
int svt_hip_block_variance_batch_dev(SvtHipCtx *c, int pix_bytes, int bd, const void *a, int a_stride, const void *b, int b_stride, const SvtHipBlkPair *pairs, int n, uint32_t *var,
                                     uint32_t *sse) {
    if (pix_bytes != 1 || bd != 8) { snprintf(c->err, sizeof(c->err), "mock: svt_hip_block_variance_batch_dev is only built for 8-bit planes (hook md_subpel)"); return SVT_HIP_ERR_UNSUPPORTED; }
    for (int i = 0; i < n; i++) {
        uint32_t s2;
        var[i] = orc_variance((const uint8_t *)a + (size_t)pairs[i].a_y * a_stride + pairs[i].a_x, a_stride, (const uint8_t *)b + (size_t)pairs[i].b_y * b_stride + pairs[i].b_x, b_stride,
                              pairs[i].w, pairs[i].h, &s2);
        if (sse) sse[i] = s2;
        if (perturb("md_subpel") && i == 0) var[i] = var[i] > 5000 ? var[i] - 5000 : 0;
    }
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------ picture analysis */
int svt_hip_downsample_2d_dev(SvtHipCtx *c, const uint8_t *in, int in_stride, int w, int h, uint8_t *out, int out_stride, int step, int filtered) {
    (void)c;
    orc_downsample_2d(in, in_stride, w, h, out, out_stride, step, filtered);
    if (perturb("pa")) out[(size_t)5 * out_stride + 7] ^= 0x10;
    return SVT_HIP_OK;
}
int svt_hip_variance_pyramid_dev(SvtHipCtx *c, const uint8_t *plane, int stride, int sb_cols, int n_sb, int full_precision, uint8_t *mean, uint16_t *var) {
    (void)c;
    for (int i = 0; i < n_sb; i++)
        orc_variance_pyramid_sb(plane + (size_t)(i / sb_cols) * 64 * stride + (size_t)(i % sb_cols) * 64, stride, full_precision, mean + (size_t)i * 85, var + (size_t)i * 85);
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------ alt-ref temporal filter */
int svt_hip_tf_filter_frame_dev(SvtHipCtx *c, int pix_bytes, int bd, const void *const src[3], const int src_stride[3], void *const dst[3],
                                const int dst_stride[3], int w, int h, int ss_x, int ss_y, int tf_chroma, const SvtHipTfRef *refs, int n_refs,
                                const double noise_levels[3], int decay_control, int min_frame_size, uint64_t *sse) {
    (void)c;
    OrcTfRef r[SVT_HIP_TF_MAX_REFS];
    for (int f = 0; f < n_refs; f++) {
        for (int p = 0; p < 3; p++) { r[f].pred[p] = refs[f].pred[p]; r[f].pred_stride[p] = refs[f].pred_stride[p]; }
        r[f].blocks = (const OrcTfBlk64 *)refs[f].blocks;   /* the same layout (svt_oracle.h / svt_hip.h) */
    }
    orc_tf_filter_frame(pix_bytes, bd, src, src_stride, dst, dst_stride, w, h, ss_x, ss_y, tf_chroma, r, n_refs, noise_levels, decay_control, min_frame_size, sse);
    if (perturb("tf")) { ((uint8_t *)dst[0])[(size_t)9 * dst_stride[0] * pix_bytes + 9 * pix_bytes] ^= 4; ((uint8_t *)dst[0])[(size_t)30 * dst_stride[0] * pix_bytes + 41 * pix_bytes] ^= 8; }
    return SVT_HIP_OK;
}

int svt_hip_tf_subpel_frame_dev(SvtHipCtx *c, int pix_bytes, int bd, const void *const src[3], const int src_stride[3], const void *const ref[3],
                                const int ref_stride[3], void *const pred[3], const int pred_stride[3], int mi_cols, int mi_rows, uint64_t th16, int tf_hp,
                                int tf_chroma, const SvtHipTfSubpelBlk *jobs, int n_jobs, SvtHipTfBlk64 *blocks) {
    (void)c;
    orc_tf_subpel_frame(pix_bytes, bd, src, src_stride, ref, ref_stride, pred, pred_stride, mi_cols, mi_rows, th16, tf_hp, tf_chroma, jobs, n_jobs, (OrcTfBlk64 *)blocks);
    if (perturb("tf_subpel") && n_jobs > 0) {   /* a visibly wrong predictor block and error */
        blocks[jobs[0].blk_index].err32[0] += 4096;
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16 * pix_bytes; x++) ((uint8_t *)pred[0])[((size_t)(jobs[0].dst_y + 8 + y) * pred_stride[0] + jobs[0].dst_x + 8) * pix_bytes + x] ^= 0x10;
    }
    return SVT_HIP_OK;
}

int svt_hip_tf_estimate_noise_dev(SvtHipCtx *c, const void *src, int pix_bytes, int bd, int width, int height, int stride, int64_t *out) {
    (void)c;
    orc_tf_estimate_noise(src, pix_bytes, bd, width, height, stride, out);
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------ deblocking */
int svt_hip_deblock_plane_dev(SvtHipCtx *c, void *plane, int pix_bytes, int stride, int bd, const uint16_t *ev, const uint16_t *eh,
                              int units_w, int units_h, int sharpness) {
    (void)c;
    const size_t n = (size_t)units_w * units_h;
    uint16_t    *zero = (uint16_t *)calloc(n ? n : 1, sizeof(uint16_t));
    orc_deblock_plane(plane, pix_bytes, stride, bd, ev ? ev : zero, eh ? eh : zero, units_w, units_h, sharpness);
    free(zero);
    return SVT_HIP_OK;
}
int svt_hip_deblock_frame_dev(SvtHipCtx *c, void *const plane[3], int pix_bytes, const int stride[3], int bd, const uint16_t *const ev[3],
                              const uint16_t *const eh[3], const int units_w[3], const int units_h[3], int sharpness) {
    for (int p = 0; p < 3; p++)
        if (plane[p]) svt_hip_deblock_plane_dev(c, plane[p], pix_bytes, stride[p], bd, ev[p], eh[p], units_w[p], units_h[p], sharpness);
    if (perturb("dlf") && plane[0]) ((uint8_t *)plane[0])[(size_t)9 * stride[0] * pix_bytes + 9 * pix_bytes] ^= 1;
    return SVT_HIP_OK;
}
int svt_hip_deblock_frame_fused_dev(SvtHipCtx *c, const void *const src[3], void *const dst[3], int pix_bytes, const int stride[3], int bd, const int pw[3], const int ph[3],
                                    const uint16_t *const ev[3], const uint16_t *const eh[3], const int units_w[3], const int units_h[3], int sharpness) {
    for (int p = 0; p < 3; p++) {
        if (!src[p]) continue;
        for (int y = 0; y < ph[p]; y++) memcpy((uint8_t *)dst[p] + (size_t)y * stride[p] * pix_bytes, (const uint8_t *)src[p] + (size_t)y * stride[p] * pix_bytes, (size_t)pw[p] * pix_bytes);
        svt_hip_deblock_plane_dev(c, dst[p], pix_bytes, stride[p], bd, ev[p], eh[p], units_w[p], units_h[p], sharpness);
    }
    if (perturb("dlf") && src[0]) ((uint8_t *)dst[0])[(size_t)9 * stride[0] * pix_bytes + 9 * pix_bytes] ^= 1;
    return SVT_HIP_OK;
}
int svt_hip_dlf_build_edges_picture_dev(SvtHipCtx *c, const SvtHipDlfModeInfo *mi, int mi_cols, int mi_rows, int ss_x, int ss_y, const int pw[3], const int ph[3], const int fw[3],
                                        const int fh[3], const int (*level)[2], uint16_t *const ev[3], uint16_t *const eh[3]) {
    (void)c;
    const size_t       n = (size_t)mi_cols * mi_rows;
    SvtHipDlfModeInfo *t = (SvtHipDlfModeInfo *)malloc(n * sizeof(*t));   /* the host builder on a copy of the grid that carries the stand-in levels */
    if (!t) return SVT_HIP_ERR_RUNTIME;
    memcpy(t, mi, n * sizeof(*t));
    if (level)
        for (size_t i = 0; i < n; i++)
            for (int p = 0; p < 3; p++)
                for (int d = 0; d < 2; d++)
                    if (level[p][d] >= 0) t[i].level[p][d] = (uint8_t)level[p][d];
    int rc = SVT_HIP_OK;
    for (int p = 0; p < 3 && rc == SVT_HIP_OK; p++)
        if (ev[p] || eh[p]) rc = svt_hip_dlf_build_edges_crop(t, mi_cols, mi_rows, p, p ? ss_x : 0, p ? ss_y : 0, pw[p], ph[p], fw[p], fh[p], ev[p], eh[p]);
    free(t);
    return rc;
}
int svt_hip_plane_sse_dev(SvtHipCtx *c, int pix_bytes, const void *a, int a_stride, const void *b, int b_stride, int w, int h, uint64_t *sse) {
    (void)c;
    *sse = orc_plane_sse(pix_bytes, a, a_stride, b, b_stride, w, h);
    return SVT_HIP_OK;
}
typedef struct {
    const void *recon, *src; void *tmp;
    int pix_bytes, stride, bd, w, h, src_stride, uw, uh, sharpness;
    const uint16_t *ev, *eh;
} MockProbe;
static int64_t mock_try_level(void *user, int lv_v, int lv_h) {   /* the device's level override: every edge at the probed level */
    MockProbe   *m = (MockProbe *)user;
    const size_t n = (size_t)m->uw * m->uh;
    uint16_t    *v = (uint16_t *)malloc(n * 2), *h = (uint16_t *)malloc(n * 2);
    for (size_t i = 0; i < n; i++) {
        v[i] = (uint16_t)(((m->ev[i] & 0xff) && lv_v) ? ((lv_v << 8) | (m->ev[i] & 0xff)) : 0);   /* level 0 = no filtering, like the kernel (deblock.hip) */
        h[i] = (uint16_t)(((m->eh[i] & 0xff) && lv_h) ? ((lv_h << 8) | (m->eh[i] & 0xff)) : 0);
    }
    for (int y = 0; y < m->h; y++)
        memcpy((uint8_t *)m->tmp + (size_t)y * m->stride * m->pix_bytes, (const uint8_t *)m->recon + (size_t)y * m->stride * m->pix_bytes, (size_t)m->w * m->pix_bytes);
    orc_deblock_plane(m->tmp, m->pix_bytes, m->stride, m->bd, v, h, m->uw, m->uh, m->sharpness);
    free(v); free(h);
    return (int64_t)orc_plane_sse(m->pix_bytes, m->src, m->src_stride, m->tmp, m->stride, m->w, m->h);
}
int svt_hip_dlf_search_level_dev(SvtHipCtx *c, const SvtHipDlfSearch *p, const void *d_recon, void *d_tmp, int pix_bytes, int stride, int bd,
                                 int plane_w, int plane_h, const void *d_src, int src_stride, const uint16_t *ev, const uint16_t *eh, int units_w,
                                 int units_h, uint64_t *d_sse_scratch, int *best_level, int64_t *best_err) {
    (void)c; (void)d_sse_scratch;
    MockProbe m = {d_recon, d_src, d_tmp, pix_bytes, stride, bd, plane_w, plane_h, src_stride, units_w, units_h, p->sharpness, ev, eh};
    const int rc = svt_hip_dlf_search_levels_host(p, mock_try_level, &m, best_level, best_err);
    if (perturb("dlf_search")) *best_level = (*best_level + 3) & 63;
    return rc;
}

int svt_hip_dlf_search_levels_picture_dev(SvtHipCtx *c, int n_planes, const SvtHipDlfSearchPlane *planes, int pix_bytes, int bd, uint64_t *d_sse_scratch, int *best_level,
                                          int64_t *best_err) {   /* the planes' searches are independent: one after the other gives what the lockstep rounds give */
    if (n_planes < 1 || n_planes > 3) return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_planes; i++) {
        const SvtHipDlfSearchPlane *P = &planes[i];
        const int rc = svt_hip_dlf_search_level_dev(c, &P->q, P->d_recon, P->d_tmp[0], pix_bytes, P->stride, bd, P->plane_w, P->plane_h, P->d_src, P->src_stride, P->d_edges_v, P->d_edges_h,
                                                    P->units_w, P->units_h, d_sse_scratch, &best_level[i], best_err ? &best_err[i] : NULL);
        if (rc != SVT_HIP_OK) return rc;
    }
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------ CDEF */
int svt_hip_cdef_search_frame_dev(SvtHipCtx *c, int pix_bytes, const void *const rec[3], const int rec_stride[3], const void *const src[3],
                                  const int src_stride[3], int w, int h, const uint8_t *skip8, int pri_damping, int bd, uint64_t *mse, uint8_t *dir,
                                  int32_t *var) {
    (void)c; (void)dir; (void)var;
    const int nfb = ((w + 63) / 64) * ((h + 63) / 64);
    orc_cdef_search_frame(rec, rec_stride, src, src_stride, pix_bytes, w, h, skip8, pri_damping, bd, 0, mse, 0, nfb);
    if (perturb("cdef_search"))
        for (int i = 0; i < 2 * nfb * 64; i++) mse[i] = (uint64_t)(i % 64) << 24;   /* strength 0 always "wins" */
    return SVT_HIP_OK;
}
int svt_hip_cdef_apply_frame_dev(SvtHipCtx *c, int pix_bytes, const void *const in[3], void *const out[3], const int stride[3], int w, int h,
                                 const uint8_t *skip8, const uint8_t *ys, const uint8_t *uvs, int damping, int bd, uint8_t *dir, const int32_t *var) {
    (void)c; (void)dir; (void)var;
    /* the product writes EVERY sample of the picture (include/svt_hip.h: unfiltered blocks are passed through, d_out needs no initial copy); the oracle, like the
     * reference, only writes the filtered blocks of a picture that already holds the input */
    for (int p = 0; p < 3; p++)
        for (int y = 0; y < (h >> (p > 0)); y++)
            memcpy((uint8_t *)out[p] + (size_t)y * stride[p] * pix_bytes, (const uint8_t *)in[p] + (size_t)y * stride[p] * pix_bytes, (size_t)(w >> (p > 0)) * pix_bytes);
    orc_cdef_apply_frame(in, out, stride, pix_bytes, w, h, skip8, ys, uvs, damping, bd);
    if (perturb("cdef_apply")) ((uint8_t *)out[0])[(size_t)9 * stride[0] * pix_bytes + 9 * pix_bytes] ^= 1;
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------ restoration */
int svt_hip_picture_format_dev(SvtHipCtx *c, int mode, const void *in0, int in0_stride, const void *in1, int in1_stride, void *out0, int out0_stride, void *out1, int out1_stride,
                               int w, int h) {
    (void)c;
    if (mode < 0 || mode > 6 || w <= 0 || h <= 0) return SVT_HIP_ERR_BAD_ARG;
    orc_picture_format(mode, in0, in0_stride, in1, in1_stride, out0, out0_stride, out1, out1_stride, w, h);
    return SVT_HIP_OK;
}
int svt_hip_generate_padding_dev(SvtHipCtx *c, void *plane, int pix_bytes, int stride, int w, int h, int pad_w, int pad_h) {
    (void)c;
    orc_generate_padding(plane, pix_bytes, stride, w, h, pad_w, pad_h);
    return SVT_HIP_OK;
}
int svt_hip_lr_apply_plane_dev(SvtHipCtx *c, int pix_bytes, int bd, const void *dgd, int stride, void *dst, int dst_stride, int pw, int ph,
                               int unit_size, int ss_y, const void *dbl, int dbl_stride, const uint8_t *unit_ep, const int32_t *unit_xqd,
                               const int16_t *unit_wiener) {
    (void)c;
    if (!dbl) { snprintf(c->err, sizeof(c->err), "mock: lr_apply needs the deblocked plane"); return SVT_HIP_ERR_UNSUPPORTED; }
    /* the oracle swaps the stripe context rows into the CDEF plane and back (like the reference): work on a copy of the extended plane */
    const size_t rows = (size_t)ph + 6, bytes = rows * stride * pix_bytes;
    uint8_t     *copy = (uint8_t *)malloc(bytes);
    if (stride < pw + 6) { free(copy); snprintf(c->err, sizeof(c->err), "mock: lr_apply needs stride >= width + 6"); return SVT_HIP_ERR_UNSUPPORTED; }
    for (int y = -3; y < ph + 3; y++)   /* the plane with its 3-sample border (what the interface promises to be readable), sample (0,0) at (3,3) of the copy */
        memcpy(copy + (size_t)(y + 3) * stride * pix_bytes, (const uint8_t *)dgd + ((ptrdiff_t)y * stride - 3) * pix_bytes, (size_t)(pw + 6) * pix_bytes);
    orc_lr_apply_plane(dbl, dbl_stride, copy + ((size_t)3 * stride + 3) * pix_bytes, stride, pix_bytes, pw, ph, ss_y, ss_y, unit_size, bd, unit_ep, unit_xqd,
                       unit_wiener, dst, dst_stride);
    free(copy);
    if (perturb("rest_apply")) ((uint8_t *)dst)[(size_t)9 * dst_stride * pix_bytes + 9 * pix_bytes] ^= 1;
    return SVT_HIP_OK;
}
int svt_hip_lr_try_unit_dev(SvtHipCtx *c, int pix_bytes, int bd, const void *dgd, int stride, void *dst, int dst_stride, int pw, int ph, int unit_size, int ss_y,
                            const void *dbl, int dbl_stride, const uint8_t *unit_ep, const int32_t *unit_xqd, const int16_t *unit_wiener, const void *src, int src_stride,
                            int unit, uint64_t *sse) {
    /* the double filters the whole plane (the units other than `unit` are whatever their entries say) and measures the unit's rectangle */
    const int ux = orc_rest_units(pw, unit_size), uy = orc_rest_units(ph, unit_size), voff = 8 >> ss_y;
    if (unit < 0 || unit >= ux * uy) return SVT_HIP_ERR_BAD_ARG;
    uint8_t *ep1 = (uint8_t *)malloc((size_t)ux * uy);   /* only `unit` is filtered: every other unit is RESTORE_NONE here (their entries may be unset) */
    if (!ep1) return SVT_HIP_ERR_RUNTIME;
    memset(ep1, 255, (size_t)ux * uy);
    ep1[unit] = unit_ep[unit];
    const int rc = svt_hip_lr_apply_plane_dev(c, pix_bytes, bd, dgd, stride, dst, dst_stride, pw, ph, unit_size, ss_y, dbl, dbl_stride, ep1, unit_xqd, unit_wiener);
    free(ep1);
    if (rc != SVT_HIP_OK) return rc;
    const int uj = unit % ux, ui = unit / ux, x0 = uj * unit_size, w = uj == ux - 1 ? pw - x0 : unit_size, y0 = ui * unit_size, h = ui == uy - 1 ? ph - y0 : unit_size;
    const int v0 = y0 - voff > 0 ? y0 - voff : 0, v1 = (y0 + h < ph) ? y0 + h - voff : y0 + h;
    *sse = orc_plane_sse(pix_bytes, (const uint8_t *)src + ((size_t)v0 * src_stride + x0) * pix_bytes, src_stride,
                         (const uint8_t *)dst + ((size_t)v0 * dst_stride + x0) * pix_bytes, dst_stride, w, v1 - v0);
    if (perturb("wiener_try")) { static unsigned n_call; *sse += (uint64_t)((n_call++ * 2654435761u) >> 20); }   /* a different wrong answer per probe: comparisons flip */
    return SVT_HIP_OK;
}
int svt_hip_lr_try_units_dev(SvtHipCtx *c, int pix_bytes, int bd, const void *dgd, int stride, void *dst, int dst_stride, int pw, int ph, int unit_size, int ss_y,
                             const void *dbl, int dbl_stride, const uint8_t *unit_ep, const int32_t *unit_xqd, const int16_t *unit_wiener, const void *src, int src_stride,
                             const SvtHipBlkPair *rects, int n_rects, uint64_t *sse) {
    const int rc = svt_hip_lr_apply_plane_dev(c, pix_bytes, bd, dgd, stride, dst, dst_stride, pw, ph, unit_size, ss_y, dbl, dbl_stride, unit_ep, unit_xqd, unit_wiener);
    if (rc != SVT_HIP_OK) return rc;
    for (int i = 0; i < n_rects; i++) {
        sse[i] = orc_plane_sse(pix_bytes, (const uint8_t *)src + ((size_t)rects[i].a_y * src_stride + rects[i].a_x) * pix_bytes, src_stride,
                               (const uint8_t *)dst + ((size_t)rects[i].b_y * dst_stride + rects[i].b_x) * pix_bytes, dst_stride, rects[i].w, rects[i].h);
        if (perturb("wiener_search")) { static unsigned n_call; sse[i] += (uint64_t)((n_call++ * 2654435761u) >> 20); }
    }
    return SVT_HIP_OK;
}
/* finer_tile_search_wiener_seg (Encoder/Codec/EbRestorationPick.c:1092-1200) restated as a state machine per unit (the reference's coordinate descent: step sizes
 * 4, 2, 1; horizontal taps then vertical taps; downward probe, then upward; at step 4 an accepted probe repeats), every probe = the unit filtered by the oracle's
 * restoration apply with the probed taps + its SSE against the source.  The product runs the same walk on the device (wiener_walk_kernel); the two are compared
 * directly by tests/test_sgr_gpu.py::test_wiener_walk_units and through the end-to-end encodes. */
typedef struct { int state, s, ph, p, up, skip; int64_t err; int16_t v[8], h[8]; } WnWalk;
static void wn_apply(int16_t *f, int p, int d) { f[p] += (int16_t)d; f[6 - p] += (int16_t)d; f[3] -= (int16_t)(2 * d); }
static int wn_issue(WnWalk *w, int off) {   /* moves to the next probe: 1 = one is outstanding, 0 = the walk is over */
    static const int tmin[3] = {-5, -23, -17}, tmax[3] = {10, 8, 46};   /* WIENER_FILT_TAP{0,1,2}_{MINV,MAXV}, Common/Codec/EbRestoration.h:130-150 */
    for (;;) {
        if (w->s < 1) { w->state = 0; return 0; }
        if (w->p >= 3) {
            if (w->ph == 0) w->ph = 1; else { w->ph = 0; w->s >>= 1; }
            w->p = off; w->up = 0; w->skip = 0;
            continue;
        }
        int16_t *f = w->ph ? w->v : w->h;
        if (!w->up) {
            if (f[w->p] - w->s >= tmin[w->p]) { wn_apply(f, w->p, -w->s); w->state = 2; return 1; }
            if (w->skip) w->p = 3; else w->up = 1;
        } else {
            if (f[w->p] + w->s <= tmax[w->p]) { wn_apply(f, w->p, w->s); w->state = 2; return 1; }
            w->p++; w->up = 0; w->skip = 0;
        }
    }
}
static void wn_result(WnWalk *w, int64_t err2, int off) {
    if (w->state == 1) { w->err = err2; w->s = 4; w->ph = 0; w->p = off; w->up = 0; w->skip = 0; wn_issue(w, off); return; }
    int16_t  *f = w->ph ? w->v : w->h;
    const int d = w->up ? w->s : -w->s, accepted = !(err2 > w->err);
    if (!accepted) wn_apply(f, w->p, -d);
    else { w->err = err2; if (!w->up) w->skip = 1; }
    if (!(accepted && w->s == 4)) {
        if (!w->up) { if (w->skip) w->p = 3; else w->up = 1; }
        else { w->p++; w->up = 0; w->skip = 0; }
    }
    wn_issue(w, off);
}
int svt_hip_wiener_walk_units_dev(SvtHipCtx *c, int pix_bytes, int bd, const void *dgd, int stride, int pw, int ph, int unit_size, int ss_y, const void *dbl, int dbl_stride,
                                  const void *src, int src_stride, int16_t *unit_wiener, const uint8_t *active, int wiener_win, int64_t *err, uint32_t *probes) {
    const int n = orc_rest_units(pw, unit_size) * orc_rest_units(ph, unit_size), off = (7 - wiener_win) >> 1;
    WnWalk *w = (WnWalk *)calloc(n, sizeof(WnWalk));
    int32_t *lim = (int32_t *)malloc(sizeof(int32_t) * 4 * n), *xqd = (int32_t *)calloc(2 * n, sizeof(int32_t));
    SvtHipBlkPair *rect = (SvtHipBlkPair *)calloc(n, sizeof(SvtHipBlkPair));
    uint8_t *ep = (uint8_t *)malloc(n);
    int16_t *wn = (int16_t *)calloc((size_t)16 * n, sizeof(int16_t));
    uint64_t *sse = (uint64_t *)calloc(n, sizeof(uint64_t));
    void *dst = malloc((size_t)stride * (ph + 8) * pix_bytes);
    orc_rest_unit_limits(pw, ph, ss_y, unit_size, lim);
    int n_active = 0, rc = SVT_HIP_OK;
    for (int u = 0; u < n; u++) {
        rect[u].a_x = rect[u].b_x = lim[4 * u]; rect[u].a_y = rect[u].b_y = lim[4 * u + 2]; rect[u].w = (uint16_t)(lim[4 * u + 1] - lim[4 * u]); rect[u].h = (uint16_t)(lim[4 * u + 3] - lim[4 * u + 2]);
        if (probes) probes[u] = 0;
        if (!active[u]) continue;
        memcpy(w[u].v, unit_wiener + 16 * u, 16); memcpy(w[u].h, unit_wiener + 16 * u + 8, 16);
        w[u].state = 1; n_active++;
    }
    while (n_active > 0 && rc == SVT_HIP_OK) {
        for (int u = 0; u < n; u++) {
            ep[u] = w[u].state ? 254 : 255;
            memcpy(wn + 16 * u, w[u].v, 16); memcpy(wn + 16 * u + 8, w[u].h, 16);
        }
        rc = svt_hip_lr_try_units_dev(c, pix_bytes, bd, dgd, stride, dst, stride, pw, ph, unit_size, ss_y, dbl, dbl_stride, ep, xqd, wn, src, src_stride, rect, n, sse);
        for (int u = 0; u < n && rc == SVT_HIP_OK; u++) {
            if (!w[u].state) continue;
            if (probes) probes[u]++;
            wn_result(&w[u], (int64_t)sse[u], off);
            if (!w[u].state) n_active--;
        }
    }
    for (int u = 0; u < n && rc == SVT_HIP_OK; u++)
        if (active[u]) { memcpy(unit_wiener + 16 * u, w[u].v, 16); memcpy(unit_wiener + 16 * u + 8, w[u].h, 16); err[u] = w[u].err; }
    free(w); free(lim); free(xqd); free(rect); free(ep); free(wn); free(sse); free(dst);
    return rc;
}
int svt_hip_wiener_init_units_dev(SvtHipCtx *c, int win, int n_units, const int64_t *M, const int64_t *H, int16_t *unit_wiener, uint8_t *active, int8_t *status) {
    (void)c;
    const int w2 = win * win;
    for (int u = 0; u < n_units; u++) {
        status[u] = (int8_t)orc_wiener_unit_init(win, M + (size_t)u * w2, H + (size_t)u * w2 * w2, unit_wiener + 16 * (size_t)u, unit_wiener + 16 * (size_t)u + 8);
        active[u] = status[u] == 1;
    }
    if (perturb("wiener_init") && n_units) unit_wiener[2] ^= 1;
    return SVT_HIP_OK;
}

int svt_hip_wiener_walk_units_picture_dev(SvtHipCtx *c, int pix_bytes, int bd, int n_planes, const SvtHipWienerWalkPlane *pl) {
    if (!pl || n_planes < 1 || n_planes > 3) return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_planes; i++) {
        const int rc = svt_hip_wiener_walk_units_dev(c, pix_bytes, bd, pl[i].d_dgd, pl[i].stride, pl[i].pw, pl[i].ph, pl[i].unit_size, pl[i].ss_y, pl[i].d_dbl, pl[i].dbl_stride, pl[i].d_src,
                                                     pl[i].src_stride, pl[i].d_unit_wiener, pl[i].d_active, pl[i].wiener_win, pl[i].d_err, pl[i].d_probes);
        if (rc != SVT_HIP_OK) return rc;
    }
    return SVT_HIP_OK;
}
int svt_hip_sgr_apply_plane_dev(SvtHipCtx *c, int pix_bytes, int bd, const void *dgd, int stride, void *dst, int dst_stride, int pw, int ph,
                                int unit_size, int ss_y, const void *dbl, int dbl_stride, const uint8_t *unit_ep, const int32_t *unit_xqd) {
    return svt_hip_lr_apply_plane_dev(c, pix_bytes, bd, dgd, stride, dst, dst_stride, pw, ph, unit_size, ss_y, dbl, dbl_stride, unit_ep, unit_xqd, NULL);
}
int svt_hip_sgr_search_units_plane(SvtHipCtx *c, int pix_bytes, int bd, const void *dgd, int stride, const void *src, int src_stride, int pw, int ph,
                                   int unit_size, int ss_y, uint32_t ep_mask, int32_t *xqd_out, int64_t *err_out, uint8_t *best_ep, int *rounds) {
    (void)c;
    orc_sgr_search_units_plane(dgd, pix_bytes, stride, src, src_stride, pw, ph, ss_y, ss_y, unit_size, bd, ep_mask, xqd_out, err_out, best_ep);
    if (perturb("sgr_search")) {
        const int nu = orc_rest_units(pw, unit_size) * orc_rest_units(ph, unit_size);
        for (int i = 0; i < nu * 32; i += 2) xqd_out[i] = xqd_out[i] > -90 ? xqd_out[i] - 5 : xqd_out[i] + 5;
    }
    if (rounds) *rounds = 0;
    return SVT_HIP_OK;
}
int svt_hip_sgr_search_units_picture(SvtHipCtx *c, int pix_bytes, int bd, int n_planes, const SvtHipSgrSearchPlane *pl, int *rounds) {
    for (int i = 0; i < n_planes; i++)
        svt_hip_sgr_search_units_plane(c, pix_bytes, bd, pl[i].d_dgd, pl[i].stride, pl[i].d_src, pl[i].src_stride, pl[i].pw, pl[i].ph, pl[i].unit_size,
                                       pl[i].ss_y, pl[i].ep_mask, pl[i].xqd_out, pl[i].err_out, pl[i].best_ep, rounds);
    return SVT_HIP_OK;
}
int svt_hip_block_sse_batch_dev(SvtHipCtx *c, int pix_bytes, const void *a, int a_stride, const void *b, int b_stride, const SvtHipBlkPair *pairs,
                                int n, uint64_t *sse) {
    (void)c;
    for (int i = 0; i < n; i++)
        sse[i] = orc_plane_sse(pix_bytes, (const uint8_t *)a + ((size_t)pairs[i].a_y * a_stride + pairs[i].a_x) * pix_bytes, a_stride,
                               (const uint8_t *)b + ((size_t)pairs[i].b_y * b_stride + pairs[i].b_x) * pix_bytes, b_stride, pairs[i].w, pairs[i].h);
    return SVT_HIP_OK;
}
int svt_hip_wiener_stats_plane_dev(SvtHipCtx *c, int pix_bytes, int bd, int win, const void *dgd, int stride, const void *src, int src_stride, int pw,
                                   int ph, int unit_size, int ss_y, int64_t *M, int64_t *H) {
    (void)c;
    orc_wiener_stats_plane(win, dgd, stride, src, src_stride, pix_bytes, bd, pw, ph, ss_y, unit_size, M, H);
    if (perturb("wiener_stats")) {
        const int nu = orc_rest_units(pw, unit_size) * orc_rest_units(ph, unit_size);
        for (int i = 0; i < nu * win * win; i++) M[i] = M[i] / 2;
    }
    return SVT_HIP_OK;
}
