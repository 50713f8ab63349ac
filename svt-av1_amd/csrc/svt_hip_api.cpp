// svt_hip_api.cpp — the C-ABI layer of libsvtav1_hip.so (include/svt_hip.h): context, memory,
// host-pointer convenience wrappers around the batched kernel launchers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <limits.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "svt_hip_internal.h"
#include "svt_hip_host.h"

struct SvtHipCtx {
    int         device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
    int         me_waves = 4;   // 256 threads per SB: measured best on MI355X (tools/me_time.py)
    int         me_big = 1;     // also launch the strip-walking instance for search areas above 65 536 candidates
    void*       scratch = nullptr;   // library-owned device scratch (16-bit Wiener statistics, self-guided unit search), grown on demand
    size_t      scratch_bytes = 0;
    void*       host_scratch = nullptr;   // pinned host staging of the self-guided unit search
    size_t      host_scratch_bytes = 0;
    std::string err;
};

static int fail(SvtHipCtx* c, hipError_t e, const char* what) {
    if (c) c->err = std::string(what) + ": " + hipGetErrorString(e);
    return SVT_HIP_ERR_RUNTIME;
}
#define HIPCHK(c, call)                                   \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return fail((c), e_, #call); \
    } while (0)

// Every entry point makes the context's device current first: one encoder process may drive several GPUs, each from its own host thread
// ("host worker i owns GPU i"); hipSetDevice is a thread-local switch.
#define SVT_HIP_ENTER(c)                                                              \
    do {                                                                              \
        if ((c) && hipSetDevice((c)->device) != hipSuccess) return SVT_HIP_ERR_RUNTIME; \
    } while (0)

extern "C" {

int svt_hip_init(int device_id, SvtHipCtx** out) {
    if (!out) return SVT_HIP_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return SVT_HIP_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return SVT_HIP_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "svt_hip_init: device %d is %s; this library is built for gfx950 only\n", device_id,
                     prop.gcnArchName);
        return SVT_HIP_ERR_NO_DEVICE;
    }
    SvtHipCtx* c = new SvtHipCtx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->own_stream) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return SVT_HIP_ERR_RUNTIME;
    }
    c->stream = c->own_stream;
    *out = c;
    return SVT_HIP_OK;
}

void svt_hip_destroy(SvtHipCtx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->host_scratch) (void)hipHostFree(c->host_scratch);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* svt_hip_last_error(const SvtHipCtx* c) { return c ? c->err.c_str() : "null context"; }

int svt_hip_set_stream(SvtHipCtx* c, void* s) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return SVT_HIP_OK;
}
void* svt_hip_ctx_stream(SvtHipCtx* c) { return c ? (void*)c->stream : nullptr; }
int   svt_hip_ctx_device(SvtHipCtx* c) { return c ? c->device : 0; }
int svt_hip_sync(SvtHipCtx* c) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SVT_HIP_OK;
}
int svt_hip_malloc(SvtHipCtx* c, void** p, size_t bytes) {
    SVT_HIP_ENTER(c);
    if (!c || !p) return SVT_HIP_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(p, bytes ? bytes : 4));
    return SVT_HIP_OK;
}
int svt_hip_free(SvtHipCtx* c, void* p) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    HIPCHK(c, hipFree(p));
    return SVT_HIP_OK;
}
int svt_hip_memcpy_h2d(SvtHipCtx* c, void* d, const void* h, size_t bytes) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    HIPCHK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SVT_HIP_OK;
}
int svt_hip_memcpy_d2h(SvtHipCtx* c, void* h, const void* d, size_t bytes) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    HIPCHK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SVT_HIP_OK;
}
int svt_hip_memcpy_d2d(SvtHipCtx* c, void* dst, const void* src, size_t bytes) {   // asynchronous, ordered on the context's stream
    if (!c || (!dst && bytes) || (!src && bytes)) return SVT_HIP_ERR_BAD_ARG;
    if (!bytes) return SVT_HIP_OK;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return SVT_HIP_OK;
}
int svt_hip_memcpy2d_h2d(SvtHipCtx* c, void* d, size_t dpitch, const void* h, size_t hpitch, size_t wbytes, size_t rows) {
    SVT_HIP_ENTER(c);
    SVT_HIP_ENTER(c);
    if (!c || !d || !h || dpitch < wbytes || hpitch < wbytes) return SVT_HIP_ERR_BAD_ARG;
    if (!wbytes || !rows) return SVT_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy2DAsync(d, dpitch, h, hpitch, wbytes, rows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SVT_HIP_OK;
}
int svt_hip_memcpy2d_d2h(SvtHipCtx* c, void* h, size_t hpitch, const void* d, size_t dpitch, size_t wbytes, size_t rows) {
    SVT_HIP_ENTER(c);
    if (!c || !d || !h || dpitch < wbytes || hpitch < wbytes) return SVT_HIP_ERR_BAD_ARG;
    if (!wbytes || !rows) return SVT_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy2DAsync(h, hpitch, d, dpitch, wbytes, rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SVT_HIP_OK;
}
int svt_hip_timer_start(SvtHipCtx* c) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return SVT_HIP_OK;
}
int svt_hip_timer_stop_ms(SvtHipCtx* c, float* ms) {
    SVT_HIP_ENTER(c);
    if (!c || !ms) return SVT_HIP_ERR_BAD_ARG;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------------------------------- ME */
int svt_hip_me_set_waves_per_sb(SvtHipCtx* c, int waves) {
    SVT_HIP_ENTER(c);
    const int w = waves & 15;   // bits 4.. = KiB of LDS padding (experimental single-workgroup-per-CU mode, see me_fullpel.hip)
    if (!c || (w != 1 && w != 2 && w != 4) || (waves >> 4) > 120) return SVT_HIP_ERR_BAD_ARG;
    c->me_waves = waves;
    return SVT_HIP_OK;
}

int svt_hip_me_set_big_windows(SvtHipCtx* c, int enable) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    c->me_big = enable != 0;
    return SVT_HIP_OK;
}

int svt_hip_me_fullpel_frame_dev(SvtHipCtx* c, const uint8_t* d_src, const uint8_t* d_ref, int stride, int org_x,
                                 int org_y, const SvtHipSbSearch* d_sbs, int n_sb, int sub_sad, uint32_t* d_best_sad,
                                 uint32_t* d_best_mv) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_ref || !d_sbs || !d_best_sad || !d_best_mv || n_sb < 0 || (stride & 3)) {
        if (c) c->err = "svt_hip_me_fullpel_frame_dev: bad argument (stride must be a multiple of 4)";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_me_fullpel(c->stream, d_src, d_ref, stride, org_x, org_y, d_sbs, n_sb,
                                                        sub_sad, d_best_sad, d_best_mv, c->me_waves, c->me_big);
    if (e != hipSuccess) return fail(c, e, "me_fullpel launch");
    return SVT_HIP_OK;
}

int svt_hip_me_fullpel_frame(SvtHipCtx* c, const uint8_t* src, const uint8_t* ref, int stride, int plane_rows, int org_x,
                             int org_y, const SvtHipSbSearch* sbs, int n_sb, int sub_sad, uint32_t* best_sad,
                             uint32_t* best_mv) {
    SVT_HIP_ENTER(c);
    if (!c || !src || !ref || !sbs || !best_sad || !best_mv || n_sb < 0 || plane_rows <= 0) return SVT_HIP_ERR_BAD_ARG;
    int big = 0;
    for (int i = 0; i < n_sb; i++) {
        if (sbs[i].width < 0 || sbs[i].height < 0) {
            c->err = "svt_hip_me_fullpel_frame: negative search area";
            return SVT_HIP_ERR_BAD_ARG;
        }
        big |= (int)sbs[i].width * (int)sbs[i].height > 65536;
    }
    const int saved_big = c->me_big;
    c->me_big = big;   // the windows are known here: launch the strip-walking instance only when one needs it
    const size_t plane = (size_t)stride * plane_rows, nres = (size_t)n_sb * SVT_HIP_SQUARE_PU_COUNT * 4;
    uint8_t *d_src = nullptr, *d_ref = nullptr;
    SvtHipSbSearch* d_sbs = nullptr;
    uint32_t *d_sad = nullptr, *d_mv = nullptr;
    int rc = SVT_HIP_OK;
    if ((rc = svt_hip_malloc(c, (void**)&d_src, plane)) || (rc = svt_hip_malloc(c, (void**)&d_ref, plane)) ||
        (rc = svt_hip_malloc(c, (void**)&d_sbs, sizeof(SvtHipSbSearch) * (size_t)(n_sb ? n_sb : 1))) ||
        (rc = svt_hip_malloc(c, (void**)&d_sad, nres)) || (rc = svt_hip_malloc(c, (void**)&d_mv, nres)))
        goto done;
    if ((rc = svt_hip_memcpy_h2d(c, d_src, src, plane)) || (rc = svt_hip_memcpy_h2d(c, d_ref, ref, plane)) ||
        (rc = svt_hip_memcpy_h2d(c, d_sbs, sbs, sizeof(SvtHipSbSearch) * (size_t)n_sb)))
        goto done;
    if ((rc = svt_hip_me_fullpel_frame_dev(c, d_src, d_ref, stride, org_x, org_y, d_sbs, n_sb, sub_sad, d_sad, d_mv))) goto done;
    if ((rc = svt_hip_memcpy_d2h(c, best_sad, d_sad, nres)) || (rc = svt_hip_memcpy_d2h(c, best_mv, d_mv, nres))) goto done;
done:
    c->me_big = saved_big;
    if (d_src) (void)hipFree(d_src);
    if (d_ref) (void)hipFree(d_ref);
    if (d_sbs) (void)hipFree(d_sbs);
    if (d_sad) (void)hipFree(d_sad);
    if (d_mv) (void)hipFree(d_mv);
    return rc;
}

/* ------------------------------------------------------------------------- transform / quant */
int svt_hip_fwd_txfm_quant_batch_dev(SvtHipCtx* c, int tx_size, int pix_bytes, const void* d_src, int src_stride,
                                     const void* d_pred, int pred_stride, const uint32_t* d_descs, int nblk,
                                     const SvtHipQuantParams* qp, const SvtHipScanTables* scans, int32_t* d_coeff,
                                     int32_t* d_qcoeff, int32_t* d_dqcoeff, uint16_t* d_eob, int32_t* d_cul_level,
                                     uint64_t* d_energy) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_pred || !d_descs || nblk < 0 || tx_size < 0 || tx_size > 18 || (pix_bytes != 1 && pix_bytes != 2) ||
        ((d_qcoeff != nullptr) != (d_dqcoeff != nullptr)) || (d_qcoeff && (!qp || !scans || !scans->iscan[0])) ||
        (qp && (qp->variant < 0 || qp->variant > 3 || qp->log_scale < 0 || qp->log_scale > 2 || qp->coeff_shape < 0 || qp->coeff_shape > 3))) {
        if (c) c->err = "svt_hip_fwd_txfm_quant_batch_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_fwd_txfm_quant(c->stream, tx_size, pix_bytes, d_src, src_stride, d_pred, pred_stride,
                                                            d_descs, nblk, qp, scans, d_coeff, d_qcoeff, d_dqcoeff, d_eob,
                                                            d_cul_level, d_energy);
    if (e != hipSuccess) return fail(c, e, "fwd_txfm_quant launch");
    return SVT_HIP_OK;
}

int svt_hip_inv_txfm_add_batch_dev(SvtHipCtx* c, int tx_size, int pix_bytes, int bd, const int32_t* d_dqcoeff, const void* d_pred,
                                   int pred_stride, void* d_recon, int recon_stride, const uint32_t* d_descs, int nblk) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dqcoeff || !d_pred || !d_recon || !d_descs || nblk < 0 || tx_size < 0 || tx_size > 18 ||
        (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8)) {
        if (c) c->err = "svt_hip_inv_txfm_add_batch_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_inv_txfm_add(c->stream, tx_size, pix_bytes, bd, d_dqcoeff, d_pred, pred_stride, d_recon,
                                                          recon_stride, d_descs, nblk);
    if (e != hipSuccess) return fail(c, e, "inv_txfm_add launch");
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------------------- deblocking */
int svt_hip_deblock_plane_dev(SvtHipCtx* c, void* d_plane, int pix_bytes, int stride, int bd, const uint16_t* d_edges_v,
                              const uint16_t* d_edges_h, int units_w, int units_h, int sharpness) {
    SVT_HIP_ENTER(c);
    if (!c || !d_plane || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8) || units_w < 0 ||
        units_h < 0 || sharpness < 0 || sharpness > 7 || (!d_edges_v && !d_edges_h)) {
        if (c) c->err = "svt_hip_deblock_plane_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_deblock_plane(c->stream, d_plane, pix_bytes, stride, bd, d_edges_v, d_edges_h, units_w,
                                                           units_h, sharpness, -1, -1);
    if (e != hipSuccess) return fail(c, e, "deblock launch");
    return SVT_HIP_OK;
}

int svt_hip_deblock_frame_dev(SvtHipCtx* c, void* const d_plane[3], int pix_bytes, const int stride[3], int bd, const uint16_t* const d_edges_v[3],
                              const uint16_t* const d_edges_h[3], const int units_w[3], const int units_h[3], int sharpness) {
    SVT_HIP_ENTER(c);
    if (!c || !d_plane || !stride || !d_edges_v || !d_edges_h || !units_w || !units_h || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) ||
        (pix_bytes == 1 && bd != 8) || sharpness < 0 || sharpness > 7) {
        if (c) c->err = "svt_hip_deblock_frame_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    for (int p = 0; p < 3; p++)
        if (d_plane[p] && (units_w[p] < 0 || units_h[p] < 0 || !d_edges_v[p] || !d_edges_h[p])) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_deblock_frame(c->stream, d_plane, pix_bytes, stride, bd, d_edges_v, d_edges_h, units_w, units_h, sharpness);
    if (e != hipSuccess) return fail(c, e, "deblock frame launch");
    return SVT_HIP_OK;
}

int svt_hip_plane_sse_dev(SvtHipCtx* c, int pix_bytes, const void* d_a, int a_stride, const void* d_b, int b_stride, int w, int h,
                          uint64_t* d_sse) {
    SVT_HIP_ENTER(c);
    if (!c || !d_a || !d_b || !d_sse || (pix_bytes != 1 && pix_bytes != 2) || w <= 0 || h <= 0) {
        if (c) c->err = "svt_hip_plane_sse_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    HIPCHK(c, hipMemsetAsync(d_sse, 0, sizeof(uint64_t), c->stream));
    hipError_t e = (hipError_t)svt_hip_launch_plane_sse(c->stream, pix_bytes, d_a, a_stride, d_b, b_stride, w, h, d_sse);
    if (e != hipSuccess) return fail(c, e, "plane sse launch");
    return SVT_HIP_OK;
}

// search_filter_level (EbDeblockingFilter.c:1026-1187); every try_filter_frame (:966-1024) runs on the device.
int svt_hip_dlf_search_level_dev(SvtHipCtx* c, const SvtHipDlfSearch* p, const void* d_recon, void* d_tmp, int pix_bytes, int stride,
                                 int bd, int plane_w, int plane_h, const void* d_src, int src_stride, const uint16_t* d_edges_v,
                                 const uint16_t* d_edges_h, int units_w, int units_h, uint64_t* d_sse_scratch, int* best_level,
                                 int64_t* best_err_out) {
    SVT_HIP_ENTER(c);
    if (!c || !p || !d_recon || !d_tmp || !d_src || !d_edges_v || !d_edges_h || !d_sse_scratch || !best_level || p->plane < 0 ||
        p->plane > 2 || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8) || plane_w <= 0 ||
        plane_h <= 0 || units_w != (plane_w + 3) / 4 || units_h != (plane_h + 3) / 4 || p->sharpness < 0 || p->sharpness > 7) {
        if (c) c->err = "svt_hip_dlf_search_level_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    int rc = SVT_HIP_OK;
    auto try_level = [&](int lv_v, int lv_h) -> int64_t {   // try_filter_frame (:966-1024) on the device
        uint64_t sse = 0;
        hipError_t e = hipMemcpy2DAsync(d_tmp, (size_t)stride * pix_bytes, d_recon, (size_t)stride * pix_bytes, (size_t)plane_w * pix_bytes,
                                        plane_h, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = (hipError_t)svt_hip_launch_deblock_plane(c->stream, d_tmp, pix_bytes, stride, bd, d_edges_v, d_edges_h, units_w, units_h, p->sharpness, lv_v, lv_h);
        if (e == hipSuccess) e = hipMemsetAsync(d_sse_scratch, 0, sizeof(uint64_t), c->stream);
        if (e == hipSuccess) e = (hipError_t)svt_hip_launch_plane_sse(c->stream, pix_bytes, d_src, src_stride, d_tmp, stride, plane_w, plane_h, d_sse_scratch);
        if (e == hipSuccess) e = hipMemcpyAsync(&sse, d_sse_scratch, sizeof(sse), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(c, e, "dlf level probe"); return -1; }
        return (int64_t)sse;
    };
    struct Thunk { decltype(try_level)* f; } th{&try_level};
    const int hrc = svt_hip_dlf_search_levels_host(p, [](void* u, int lv_v, int lv_h) -> int64_t { return (*((Thunk*)u)->f)(lv_v, lv_h); }, &th,
                                                   best_level, best_err_out);
    return rc != SVT_HIP_OK ? rc : hrc;
}

int svt_hip_fwd_txfm_quant_multi_dev(SvtHipCtx* c, int pix_bytes, const SvtHipFwdTxJob* jobs, int njobs) {
    SVT_HIP_ENTER(c);
    if (!c || (!jobs && njobs) || njobs < 0 || (pix_bytes != 1 && pix_bytes != 2)) return SVT_HIP_ERR_BAD_ARG;
    for (int j = 0; j < njobs; j++) {
        const SvtHipFwdTxJob& J = jobs[j];
        if (J.nblk < 0 || J.tx_size < 0 || J.tx_size > 18 || (J.nblk && (!J.d_src || !J.d_pred || !J.d_descs)) ||
            ((J.d_qcoeff != nullptr) != (J.d_dqcoeff != nullptr)) || (J.d_qcoeff && !J.scans.iscan[0]) || J.qp.variant < 0 || J.qp.variant > 3 ||
            J.qp.log_scale < 0 || J.qp.log_scale > 2 || J.qp.coeff_shape < 0 || J.qp.coeff_shape > 3) {
            c->err = "svt_hip_fwd_txfm_quant_multi_dev: bad job";
            return SVT_HIP_ERR_BAD_ARG;
        }
    }
    hipError_t e = (hipError_t)svt_hip_launch_fwd_txfm_quant_multi(c->stream, pix_bytes, jobs, njobs);
    if (e != hipSuccess) return fail(c, e, "fwd_txfm_quant multi launch");
    return SVT_HIP_OK;
}
int svt_hip_inv_txfm_add_multi_dev(SvtHipCtx* c, int pix_bytes, int bd, const SvtHipInvTxJob* jobs, int njobs) {
    SVT_HIP_ENTER(c);
    if (!c || (!jobs && njobs) || njobs < 0 || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8))
        return SVT_HIP_ERR_BAD_ARG;
    for (int j = 0; j < njobs; j++) {
        const SvtHipInvTxJob& J = jobs[j];
        if (J.nblk < 0 || J.tx_size < 0 || J.tx_size > 18 || (J.nblk && (!J.d_dqcoeff || !J.d_pred || !J.d_recon || !J.d_descs))) {
            c->err = "svt_hip_inv_txfm_add_multi_dev: bad job";
            return SVT_HIP_ERR_BAD_ARG;
        }
    }
    hipError_t e = (hipError_t)svt_hip_launch_inv_txfm_add_multi(c->stream, pix_bytes, bd, jobs, njobs);
    if (e != hipSuccess) return fail(c, e, "inv_txfm_add multi launch");
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------------------------- CDEF */
int svt_hip_cdef_search_frame_dev(SvtHipCtx* c, int pix_bytes, const void* const d_rec[3], const int rec_stride[3],
                                  const void* const d_src[3], const int src_stride[3], int w, int h, const uint8_t* d_skip8,
                                  int pri_damping, int bd, uint64_t* d_mse, uint8_t* d_dir, int32_t* d_var) {
    SVT_HIP_ENTER(c);
    if (!c || !d_rec || !d_src || !rec_stride || !src_stride || !d_skip8 || !d_mse || !d_dir || !d_var || (pix_bytes != 1 && pix_bytes != 2) ||
        (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8) || w <= 0 || h <= 0 || (w & 7) || (h & 7) || ((w & 63) && (w & 63) < 16) ||
        ((h & 63) && (h & 63) < 16)) {
        if (c) c->err = "svt_hip_cdef_search_frame_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_cdef_search(c->stream, pix_bytes, d_rec, rec_stride, d_src, src_stride, w, h, d_skip8,
                                                         pri_damping, bd, d_mse, d_dir, d_var);
    if (e != hipSuccess) return fail(c, e, "cdef search launch");
    return SVT_HIP_OK;
}
int svt_hip_cdef_apply_frame_dev(SvtHipCtx* c, int pix_bytes, const void* const d_in[3], void* const d_out[3], const int stride[3], int w,
                                 int h, const uint8_t* d_skip8, const uint8_t* d_y_strength, const uint8_t* d_uv_strength, int damping,
                                 int bd, uint8_t* d_dir, const int32_t* d_var) {
    SVT_HIP_ENTER(c);
    if (!c || !d_in || !d_out || !stride || !d_skip8 || !d_y_strength || !d_uv_strength || !d_dir || (pix_bytes != 1 && pix_bytes != 2) ||
        (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8) || w <= 0 || h <= 0 || (w & 7) || (h & 7)) {
        if (c) c->err = "svt_hip_cdef_apply_frame_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_cdef_apply(c->stream, pix_bytes, d_in, d_out, stride, w, h, d_skip8, d_y_strength,
                                                        d_uv_strength, damping, bd, d_dir, d_var);
    if (e != hipSuccess) return fail(c, e, "cdef apply launch");
    return SVT_HIP_OK;
}

/* -------------------------------------------------------------- sub-pel predict / SAD / variance */
int svt_hip_subpel_predict_batch_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_ref, int ref_stride, void* d_dst, int dst_stride,
                                     const SvtHipConvBlk* d_blks, int nblk) {
    SVT_HIP_ENTER(c);
    if (!c || !d_ref || !d_dst || !d_blks || nblk < 0 || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8)) {
        if (c) c->err = "svt_hip_subpel_predict_batch_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_subpel_predict(c->stream, pix_bytes, bd, d_ref, ref_stride, d_dst, dst_stride, d_blks, nblk);
    if (e != hipSuccess) return fail(c, e, "subpel predict launch");
    return SVT_HIP_OK;
}
int svt_hip_block_sad_batch_dev(SvtHipCtx* c, int pix_bytes, const void* d_a, int a_stride, const void* d_b, int b_stride,
                                const SvtHipBlkPair* d_pairs, int n, uint32_t* d_sad) {
    SVT_HIP_ENTER(c);
    if (!c || !d_a || !d_b || !d_pairs || !d_sad || n < 0 || (pix_bytes != 1 && pix_bytes != 2)) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_block_sad(c->stream, pix_bytes, d_a, a_stride, d_b, b_stride, d_pairs, n, d_sad);
    if (e != hipSuccess) return fail(c, e, "block sad launch");
    return SVT_HIP_OK;
}
int svt_hip_coeff_distortion_batch_dev(SvtHipCtx* c, const int32_t* d_coeff, const int32_t* d_recon_coeff, int n_per_block, int nblk, uint64_t* d_out) {
    SVT_HIP_ENTER(c);
    if (!c || !d_coeff || !d_out || n_per_block <= 0 || nblk < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_coeff_distortion(c->stream, d_coeff, d_recon_coeff, n_per_block, nblk, d_out);
    if (e != hipSuccess) return fail(c, e, "coeff distortion launch");
    return SVT_HIP_OK;
}
int svt_hip_block_sse_batch_dev(SvtHipCtx* c, int pix_bytes, const void* d_a, int a_stride, const void* d_b, int b_stride, const SvtHipBlkPair* d_pairs,
                                int n, uint64_t* d_sse) {
    SVT_HIP_ENTER(c);
    if (!c || !d_a || !d_b || !d_pairs || !d_sse || n < 0 || (pix_bytes != 1 && pix_bytes != 2)) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_block_sse(c->stream, pix_bytes, d_a, a_stride, d_b, b_stride, d_pairs, n, d_sse);
    if (e != hipSuccess) return fail(c, e, "block sse launch");
    return SVT_HIP_OK;
}
int svt_hip_block_variance_batch_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_a, int a_stride, const void* d_b, int b_stride,
                                     const SvtHipBlkPair* d_pairs, int n, uint32_t* d_var, uint32_t* d_sse) {
    SVT_HIP_ENTER(c);
    if (!c || !d_a || !d_b || !d_pairs || !d_var || n < 0 || !((pix_bytes == 1 && bd == 8) || (pix_bytes == 2 && bd == 10))) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_block_variance(c->stream, pix_bytes, bd, d_a, a_stride, d_b, b_stride, d_pairs, n, d_var, d_sse);
    if (e != hipSuccess) return fail(c, e, "block variance launch");
    return SVT_HIP_OK;
}

/* ------------------------------------------------------------------- pyramids / HME search */
int svt_hip_downsample_2d_dev(SvtHipCtx* c, const uint8_t* d_in, int in_stride, int w, int h, uint8_t* d_out, int out_stride, int step,
                              int filtered) {
    SVT_HIP_ENTER(c);
    if (!c || !d_in || !d_out || (step != 2 && step != 4) || w < step || h < step) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_downsample(c->stream, d_in, in_stride, w, h, d_out, out_stride, step, filtered);
    if (e != hipSuccess) return fail(c, e, "downsample launch");
    return SVT_HIP_OK;
}
int svt_hip_variance_pyramid_dev(SvtHipCtx* c, const uint8_t* d_plane, int stride, int sb_cols, int n_sb, int full_precision,
                                 uint8_t* d_mean, uint16_t* d_var) {
    SVT_HIP_ENTER(c);
    if (!c || !d_plane || !d_mean || !d_var || sb_cols <= 0 || n_sb < 0 || (stride & 7) || ((uintptr_t)d_plane & 7)) {
        if (c) c->err = "svt_hip_variance_pyramid_dev: bad argument (plane and stride must be 8-byte aligned)";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_variance_pyramid(c->stream, d_plane, stride, sb_cols, n_sb, full_precision, d_mean, d_var);
    if (e != hipSuccess) return fail(c, e, "variance pyramid launch");
    return SVT_HIP_OK;
}
int svt_hip_sad_loop_batch_dev(SvtHipCtx* c, const uint8_t* d_src, int src_stride, const uint8_t* d_ref, int ref_stride,
                               const SvtHipSadLoop* d_searches, int n, uint32_t* d_best_sad, int16_t* d_best_xy) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_ref || !d_searches || !d_best_sad || !d_best_xy || n < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_sad_loop(c->stream, d_src, src_stride, d_ref, ref_stride, d_searches, n, d_best_sad, d_best_xy);
    if (e != hipSuccess) return fail(c, e, "sad loop launch");
    return SVT_HIP_OK;
}

int svt_hip_sad_loop16_batch_dev(SvtHipCtx* c, const uint16_t* d_src, int src_stride, const uint16_t* d_ref, int ref_stride, const SvtHipSadLoop* d_searches, int n,
                                 uint32_t* d_best_sad, int16_t* d_best_xy) {
    SVT_HIP_ENTER(c);
    if (!c || n < 0) return SVT_HIP_ERR_BAD_ARG;
    if (n == 0) return SVT_HIP_OK;
    if (!d_src || !d_ref || !d_searches || !d_best_sad || !d_best_xy) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_sad_loop16(c->stream, d_src, src_stride, d_ref, ref_stride, d_searches, n, d_best_sad, d_best_xy);
    if (e != hipSuccess) return fail(c, e, "sad loop (16-bit) launch");
    return SVT_HIP_OK;
}

/* ---------------------------------------------------------------- self-guided restoration */
static bool sgr_args_ok(int pix_bytes, int bd, int pw, int ph) {
    return (pix_bytes == 1 || pix_bytes == 2) && (bd == 8 || bd == 10) && !(pix_bytes == 1 && bd != 8) && pw > 0 && ph > 0;
}
static int sgr_units(int size, int unit) { const int n = (size + unit / 2) / unit; return n > 0 ? n : 1; }

int svt_hip_sgr_filter_plane_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_plane, int stride, int pw, int ph, int ep,
                                 int32_t* d_flt0, int32_t* d_flt1, int flt_stride) {
    SVT_HIP_ENTER(c);
    if (!c || !d_plane || !d_flt0 || !d_flt1 || ep < 0 || ep > 15 || !sgr_args_ok(pix_bytes, bd, pw, ph)) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_sgr_filter(c->stream, pix_bytes, bd, d_plane, stride, pw, ph, ep, d_flt0, d_flt1, flt_stride);
    if (e != hipSuccess) return fail(c, e, "sgr filter launch");
    return SVT_HIP_OK;
}
int svt_hip_sgr_search_plane_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, const void* d_src, int src_stride,
                                 int pw, int ph, int unit_size, int ss_y, uint32_t ep_mask, int64_t* d_sums) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dgd || !d_src || !d_sums || unit_size < 64 || (unit_size & 63) || (ss_y != 0 && ss_y != 1) || !sgr_args_ok(pix_bytes, bd, pw, ph))
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_sgr_search(c->stream, pix_bytes, bd, d_dgd, stride, d_src, src_stride, pw, ph, unit_size,
                                                        sgr_units(pw, unit_size), sgr_units(ph, unit_size), ss_y, ep_mask & 0xFFFFu, d_sums);
    if (e != hipSuccess) return fail(c, e, "sgr search launch");
    return SVT_HIP_OK;
}
int svt_hip_sgr_apply_plane_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, void* d_dst, int dst_stride, int pw,
                                int ph, int unit_size, int ss_y, const void* d_dbl, int dbl_stride, const uint8_t* d_unit_ep,
                                const int32_t* d_unit_xqd) {
    SVT_HIP_ENTER(c);
    return svt_hip_lr_apply_plane_dev(c, pix_bytes, bd, d_dgd, stride, d_dst, dst_stride, pw, ph, unit_size, ss_y, d_dbl, dbl_stride, d_unit_ep,
                                      d_unit_xqd, nullptr);
}
int svt_hip_lr_apply_plane_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, void* d_dst, int dst_stride, int pw, int ph,
                               int unit_size, int ss_y, const void* d_dbl, int dbl_stride, const uint8_t* d_unit_ep, const int32_t* d_unit_xqd,
                               const int16_t* d_unit_wiener) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dgd || !d_dst || !d_unit_ep || !d_unit_xqd || unit_size < 64 || (unit_size & 63) || (ss_y != 0 && ss_y != 1) ||
        !sgr_args_ok(pix_bytes, bd, pw, ph))
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_sgr_apply(c->stream, pix_bytes, bd, d_dgd, stride, d_dst, dst_stride, pw, ph, unit_size,
                                                       sgr_units(pw, unit_size), sgr_units(ph, unit_size), ss_y, d_dbl, dbl_stride, d_unit_ep,
                                                       d_unit_xqd, d_unit_wiener);
    if (e != hipSuccess) return fail(c, e, "sgr apply launch");
    return SVT_HIP_OK;
}

int svt_hip_sgr_proj_error_plane_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, const void* d_src, int src_stride,
                                     int pw, int ph, int unit_size, int ss_y, uint32_t ep_mask, int ncand, const int32_t* d_xqd, int64_t* d_err) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dgd || !d_src || !d_xqd || !d_err || unit_size < 64 || (unit_size & 63) || (ss_y != 0 && ss_y != 1) || ncand < 1 ||
        ncand > SVT_HIP_SGR_MAX_CAND || !sgr_args_ok(pix_bytes, bd, pw, ph))
        return SVT_HIP_ERR_BAD_ARG;
    const int ux = sgr_units(pw, unit_size), uy = sgr_units(ph, unit_size);
    HIPCHK(c, hipMemsetAsync(d_err, 0, sizeof(int64_t) * (size_t)ux * uy * 16 * ncand, c->stream));
    hipError_t e = (hipError_t)svt_hip_launch_sgr_proj_error(c->stream, pix_bytes, bd, d_dgd, stride, d_src, src_stride, pw, ph, unit_size, ux, uy, ss_y,
                                                            ep_mask & 0xFFFFu, ncand, d_xqd, d_err);
    if (e != hipSuccess) return fail(c, e, "sgr proj error launch");
    return SVT_HIP_OK;
}

/* ---- host side of search_selfguided_restoration (Encoder/Codec/EbRestorationPick.c:583-671) on top of the two plane kernels ---- */
namespace {
const int kSgrR[16][2] = {{2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {0, 1}, {0, 1}, {0, 1}, {0, 1}, {2, 0}, {2, 0}};
const int kTapMin[2] = {-96, -32}, kTapMax[2] = {31, 95};   // SGRPROJ_PRJ_MIN0/MAX0, MIN1/MAX1 (EbRestoration.h:100-103)
inline bool kSgr1(int ep) { return kSgrR[ep][1] > 0; }

struct SgrPoint { int x, y; int64_t err; };
struct SgrItem {            // one (restoration unit, parameter set)
    int xqd[2] = {0, 0};    // start point (encode_xq), then the result
    int64_t err = 0;
    bool done = false;
    std::vector<SgrPoint> cache;
    std::vector<std::pair<int, int>> want;
    bool lookup(int x, int y, int64_t& e) const {
        for (const SgrPoint& p : cache) if (p.x == x && p.y == y) { e = p.err; return true; }
        return false;
    }
};

// svt_get_proj_subspace_c's solve (EbRestorationPick.c:497-538) on the exact integer sums, operation order of the reference
void sgr_solve(const int64_t* sums, int size, int ep, int xq[2]) {
    double H00 = (double)sums[0], H01 = (double)sums[1], H11 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    H00 /= size; H01 /= size; H11 /= size; C0 /= size; C1 /= size;
    const double H10 = H01;
    xq[0] = xq[1] = 0;
    if (kSgrR[ep][0] == 0) { if (H11 < 1e-8) return; xq[1] = (int)rint((C1 / H11) * 128); }
    else if (kSgrR[ep][1] == 0) { if (H00 < 1e-8) return; xq[0] = (int)rint((C0 / H00) * 128); }
    else {
        const double det = H00 * H11 - H01 * H10;
        if (det < 1e-8) return;
        const double x0 = (H11 * C0 - H01 * C1) / det, x1 = (H00 * C1 - H10 * C0) / det;
        xq[0] = (int)rint(x0 * 128); xq[1] = (int)rint(x1 * 128);
    }
}
int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
void sgr_encode_xq(const int xq[2], int xqd[2], int ep) {   // encode_xq, EbRestorationPick.c:539-552
    if (kSgrR[ep][0] == 0) { xqd[0] = 0; xqd[1] = clampi(128 - xq[1], kTapMin[1], kTapMax[1]); }
    else if (kSgrR[ep][1] == 0) { xqd[0] = clampi(xq[0], kTapMin[0], kTapMax[0]); xqd[1] = clampi(128 - xqd[0], kTapMin[1], kTapMax[1]); }
    else { xqd[0] = clampi(xq[0], kTapMin[0], kTapMax[0]); xqd[1] = clampi(128 - xqd[0] - xq[1], kTapMin[1], kTapMax[1]); }
}

// finer_search_pixel_proj_error (EbRestorationPick.c:353-446) replayed on the cache of evaluated points.  Returns true when the walk
// finished on exact errors only (it.xqd / it.err hold the result).  At the first point whose error is not known yet the replay turns
// speculative: it keeps walking, but decides with the quadratic model of the error that the five projection sums give
// (sum e^2 = const + (xq' H xq - 256 C.xq) / 2048^2 up to the per-pixel rounding), and every point it visits goes to it.want (at most
// max_want).  The next round evaluates those points exactly; wherever the model took the same decisions as the exact errors do, the whole
// walk is then in the cache.  Mispredictions only cost another round, never exactness.
bool sgr_replay(SgrItem& it, int ep, int start_step, int max_want, const int64_t* sums) {
    const bool has0 = kSgrR[ep][0] > 0, has1 = kSgr1(ep);
    const double H00 = (double)sums[0], H01 = (double)sums[1], H11 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    auto model = [&](int x, int y) {
        const double xq0 = has0 ? x : 0, xq1 = !has1 ? 0 : (has0 ? 128 - x - y : 128 - y);   // svt_decode_xq
        return xq0 * xq0 * H00 + 2 * xq0 * xq1 * H01 + xq1 * xq1 * H11 - 256.0 * (xq0 * C0 + xq1 * C1);
    };
    int q[2] = {it.xqd[0], it.xqd[1]};
    bool spec = false;
    it.want.clear();
    // error of point (x, y): exact while everything so far was cached, the model afterwards (cur = the walk's current point, for the switch)
    auto value = [&](int x, int y, const int cur[2], double& cur_err) {
        int64_t e;
        if (!spec && it.lookup(x, y, e)) return (double)e;
        if (!spec) { spec = true; cur_err = model(cur[0], cur[1]); }
        if (!it.lookup(x, y, e)) {
            bool dup = false;
            for (const auto& w : it.want) dup = dup || (w.first == x && w.second == y);
            if (!dup && (int)it.want.size() < max_want) it.want.emplace_back(x, y);
        }
        return model(x, y);
    };
    double err = 0, err2;
    err = value(q[0], q[1], q, err);
    for (int s = start_step; s >= 1 && (int)it.want.size() < max_want; s >>= 1) {
        for (int p = 0; p < 2 && (int)it.want.size() < max_want; p++) {
            if (kSgrR[ep][p] == 0) continue;
            bool skip = false;
            for (;;) {
                if (q[p] - s >= kTapMin[p] && (int)it.want.size() < max_want) {
                    int c[2] = {q[0], q[1]}; c[p] -= s;
                    err2 = value(c[0], c[1], q, err);
                    if (!(err2 > err)) { q[p] -= s; err = err2; skip = true; if (s == start_step) continue; }
                }
                break;
            }
            if (skip) break;
            for (;;) {
                if (q[p] + s <= kTapMax[p] && (int)it.want.size() < max_want) {
                    int c[2] = {q[0], q[1]}; c[p] += s;
                    err2 = value(c[0], c[1], q, err);
                    if (!(err2 > err)) { q[p] += s; err = err2; if (s == start_step) continue; }
                }
                break;
            }
        }
    }
    if (spec) return false;
    it.xqd[0] = q[0]; it.xqd[1] = q[1]; it.err = (int64_t)err; it.done = true;
    return true;
}
}  // namespace

int svt_hip_sgr_search_units_picture(SvtHipCtx* c, int pix_bytes, int bd, int n_planes, const SvtHipSgrSearchPlane* planes, int* rounds_out) {
    SVT_HIP_ENTER(c);
    if (!c || !planes || n_planes < 1 || n_planes > 3) return SVT_HIP_ERR_BAD_ARG;
    const int NC = SVT_HIP_SGR_MAX_CAND;
    struct Job { int nu; size_t sums_o, xqd_o, err_o; std::vector<SgrItem> items; uint32_t mask; };
    Job job[3];
    size_t need = 0;
    for (int k = 0; k < n_planes; k++) {
        const SvtHipSgrSearchPlane& P = planes[k];
        if (!P.d_dgd || !P.d_src || !P.xqd_out || !P.err_out || P.unit_size < 64 || (P.unit_size & 63) || (P.ss_y != 0 && P.ss_y != 1) ||
            !sgr_args_ok(pix_bytes, bd, P.pw, P.ph) || !(P.ep_mask & 0xFFFFu))
            return SVT_HIP_ERR_BAD_ARG;
        Job& J = job[k];
        J.nu = sgr_units(P.pw, P.unit_size) * sgr_units(P.ph, P.unit_size);
        J.mask = P.ep_mask & 0xFFFFu;
        J.sums_o = need; need += sizeof(int64_t) * J.nu * 16 * 5;
        J.xqd_o = need;  need += sizeof(int32_t) * J.nu * 16 * NC * 2;
        J.err_o = need;  need += sizeof(int64_t) * J.nu * 16 * NC;
    }
    if (need > c->scratch_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->scratch) HIPCHK(c, hipFree(c->scratch));
        c->scratch = nullptr; c->scratch_bytes = 0;
        HIPCHK(c, hipMalloc(&c->scratch, need));
        c->scratch_bytes = need;
    }
    if (need > c->host_scratch_bytes) {   // pinned mirror of the device scratch: the per-round copies stay asynchronous and cheap
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->host_scratch) HIPCHK(c, hipHostFree(c->host_scratch));
        c->host_scratch = nullptr; c->host_scratch_bytes = 0;
        HIPCHK(c, hipHostMalloc(&c->host_scratch, need, hipHostMallocDefault));
        c->host_scratch_bytes = need;
    }
    char* dev = (char*)c->scratch; char* host = (char*)c->host_scratch;
    // 1. the projection sums of every (unit, set), all planes, one synchronisation
    for (int k = 0; k < n_planes; k++) {
        const SvtHipSgrSearchPlane& P = planes[k]; const Job& J = job[k];
        const size_t sums_b = sizeof(int64_t) * J.nu * 16 * 5;
        HIPCHK(c, hipMemsetAsync(dev + J.sums_o, 0, sums_b, c->stream));
        const int rc = svt_hip_sgr_search_plane_dev(c, pix_bytes, bd, P.d_dgd, P.stride, P.d_src, P.src_stride, P.pw, P.ph, P.unit_size, P.ss_y, J.mask,
                                                    (int64_t*)(dev + J.sums_o));
        if (rc != SVT_HIP_OK) return rc;
        HIPCHK(c, hipMemcpyAsync(host + J.sums_o, dev + J.sums_o, sums_b, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // 2. solve + encode_xq per (unit, set); unit sizes as foreach_rest_unit_in_tile (EbRestoration.c:1369-1411) cuts them
    for (int k = 0; k < n_planes; k++) {
        const SvtHipSgrSearchPlane& P = planes[k]; Job& J = job[k];
        const int64_t* sums = (const int64_t*)(host + J.sums_o);
        J.items.assign((size_t)J.nu * 16, SgrItem());
        const int ux = sgr_units(P.pw, P.unit_size), ext = P.unit_size * 3 / 2, voff = 8 >> P.ss_y;
        int y0 = 0, i = 0;
        while (y0 < P.ph) {
            const int rem_h = P.ph - y0, h = rem_h < ext ? rem_h : P.unit_size;
            int v_start = y0 - voff > 0 ? y0 - voff : 0, v_end = y0 + h;
            if (v_end < P.ph) v_end -= voff;
            int x0 = 0, j = 0;
            while (x0 < P.pw) {
                const int rem_w = P.pw - x0, w = rem_w < ext ? rem_w : P.unit_size;
                const int u = i * ux + j, size = w * (v_end - v_start);
                for (int ep = 0; ep < 16; ep++) {
                    SgrItem& it = J.items[(size_t)u * 16 + ep];
                    if (!((J.mask >> ep) & 1)) { it.done = true; continue; }
                    int xq[2];
                    sgr_solve(&sums[((size_t)u * 16 + ep) * 5], size, ep, xq);
                    sgr_encode_xq(xq, it.xqd, ep);
                }
                x0 += w; j++;
            }
            y0 += h; i++;
        }
    }
    // 3. the finer search in rounds: every round evaluates up to NC new points per unfinished (unit, set), one launch per plane, one
    //    synchronisation per round for the whole picture
    int rounds = 0;
    for (;; rounds++) {
        uint32_t round_mask[3] = {0, 0, 0};
        bool any = false;
        for (int k = 0; k < n_planes; k++) {
            Job& J = job[k];
            for (int u = 0; u < J.nu; u++)
                for (int ep = 0; ep < 16; ep++) {
                    SgrItem& it = J.items[(size_t)u * 16 + ep];
                    if (!it.done && !sgr_replay(it, ep, 2, NC, (const int64_t*)(host + J.sums_o) + ((size_t)u * 16 + ep) * 5)) round_mask[k] |= 1u << ep;
                }
            any = any || round_mask[k];
        }
        if (!any) break;
        if (rounds >= 256) { c->err = "svt_hip_sgr_search_units: finer search did not converge"; return SVT_HIP_ERR_RUNTIME; }
        for (int k = 0; k < n_planes; k++) {
            if (!round_mask[k]) continue;
            const SvtHipSgrSearchPlane& P = planes[k]; const Job& J = job[k];
            int32_t* h_xqd = (int32_t*)(host + J.xqd_o);
            for (int u = 0; u < J.nu; u++)
                for (int ep = 0; ep < 16; ep++) {
                    const SgrItem& it = J.items[(size_t)u * 16 + ep];
                    int32_t* q = &h_xqd[((size_t)u * 16 + ep) * NC * 2];
                    for (int n = 0; n < NC; n++) {
                        const bool live = !it.done && n < (int)it.want.size();
                        q[2 * n] = live ? it.want[n].first : INT32_MIN;   // INT32_MIN ends the list (first slot: the pair is skipped)
                        q[2 * n + 1] = live ? it.want[n].second : 0;
                    }
                }
            HIPCHK(c, hipMemcpyAsync(dev + J.xqd_o, h_xqd, sizeof(int32_t) * J.nu * 16 * NC * 2, hipMemcpyHostToDevice, c->stream));
            const int rc = svt_hip_sgr_proj_error_plane_dev(c, pix_bytes, bd, P.d_dgd, P.stride, P.d_src, P.src_stride, P.pw, P.ph, P.unit_size, P.ss_y,
                                                            round_mask[k], NC, (const int32_t*)(dev + J.xqd_o), (int64_t*)(dev + J.err_o));
            if (rc != SVT_HIP_OK) return rc;
            HIPCHK(c, hipMemcpyAsync(host + J.err_o, dev + J.err_o, sizeof(int64_t) * J.nu * 16 * NC, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < n_planes; k++) {
            if (!round_mask[k]) continue;
            Job& J = job[k];
            const int64_t* h_err = (const int64_t*)(host + J.err_o);
            for (int u = 0; u < J.nu; u++)
                for (int ep = 0; ep < 16; ep++) {
                    SgrItem& it = J.items[(size_t)u * 16 + ep];
                    if (it.done) continue;
                    for (int n = 0; n < (int)it.want.size(); n++)
                        it.cache.push_back({it.want[n].first, it.want[n].second, h_err[((size_t)u * 16 + ep) * NC + n]});
                }
        }
    }

    for (int k = 0; k < n_planes; k++) {
        const SvtHipSgrSearchPlane& P = planes[k]; const Job& J = job[k];
        for (int u = 0; u < J.nu; u++) {
            int64_t besterr = -1;
            for (int ep = 0; ep < 16; ep++) {
                if (!((J.mask >> ep) & 1)) continue;
                const SgrItem& it = J.items[(size_t)u * 16 + ep];
                P.xqd_out[((size_t)u * 16 + ep) * 2] = it.xqd[0]; P.xqd_out[((size_t)u * 16 + ep) * 2 + 1] = it.xqd[1];
                P.err_out[(size_t)u * 16 + ep] = it.err;
                if (besterr == -1 || it.err < besterr) { besterr = it.err; if (P.best_ep) P.best_ep[u] = (uint8_t)ep; }   // strict <, :659
            }
        }
    }
    if (rounds_out) *rounds_out = rounds;
    return SVT_HIP_OK;
}
int svt_hip_sgr_search_units_plane(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, const void* d_src, int src_stride, int pw,
                                   int ph, int unit_size, int ss_y, uint32_t ep_mask, int32_t* xqd_out, int64_t* err_out, uint8_t* best_ep,
                                   int* rounds_out) {
    SVT_HIP_ENTER(c);
    const SvtHipSgrSearchPlane P = {d_dgd, stride, d_src, src_stride, pw, ph, unit_size, ss_y, ep_mask, xqd_out, err_out, best_ep};
    return svt_hip_sgr_search_units_picture(c, pix_bytes, bd, 1, &P, rounds_out);
}

int svt_hip_wiener_stats_plane_dev(SvtHipCtx* c, int pix_bytes, int bd, int win, const void* d_dgd, int stride, const void* d_src, int src_stride,
                                   int pw, int ph, int unit_size, int ss_y, int64_t* d_M, int64_t* d_H) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dgd || !d_src || !d_M || !d_H || (win != 7 && win != 5 && win != 3) || unit_size < 64 || (unit_size & 63) || unit_size > 256 ||
        (ss_y != 0 && ss_y != 1) || pw <= 0 || ph <= 0)
        return SVT_HIP_ERR_BAD_ARG;
    if ((pix_bytes == 1 && bd != 8) || (pix_bytes == 2 && bd != 8 && bd != 10 && bd != 12) || (pix_bytes != 1 && pix_bytes != 2)) {
        c->err = "svt_hip_wiener_stats_plane_dev: bad sample format";
        return SVT_HIP_ERR_BAD_ARG;
    }
    if (pix_bytes == 2) {
        const int n_units = sgr_units(pw, unit_size) * sgr_units(ph, unit_size);
        const size_t need = svt_hip_wiener_stats16_scratch(win, pw, ph, n_units);
        if (need > c->scratch_bytes) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->scratch) HIPCHK(c, hipFree(c->scratch));
            c->scratch = nullptr; c->scratch_bytes = 0;
            HIPCHK(c, hipMalloc(&c->scratch, need));
            c->scratch_bytes = need;
        }
        hipError_t e16 = (hipError_t)svt_hip_launch_wiener_stats16(c->stream, win, bd, (const uint16_t*)d_dgd, stride, (const uint16_t*)d_src, src_stride, pw, ph,
                                                                  unit_size, sgr_units(pw, unit_size), sgr_units(ph, unit_size), ss_y, d_M, d_H, (uint8_t*)c->scratch);
        if (e16 != hipSuccess) return fail(c, e16, "wiener stats (16-bit) launch");
        return SVT_HIP_OK;
    }
    hipError_t e = (hipError_t)svt_hip_launch_wiener_stats8(c->stream, win, (const uint8_t*)d_dgd, stride, (const uint8_t*)d_src, src_stride, pw, ph,
                                                           unit_size, sgr_units(pw, unit_size), sgr_units(ph, unit_size), ss_y, d_M, d_H);
    if (e != hipSuccess) return fail(c, e, "wiener stats launch");
    return SVT_HIP_OK;
}

int svt_hip_tf_filter_frame_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* const d_src[3], const int src_stride[3], void* const d_dst[3],
                                const int dst_stride[3], int w, int h, int ss_x, int ss_y, int tf_chroma, const SvtHipTfRef* refs, int n_refs,
                                const double noise_levels[3], int decay_control, int min_frame_size, uint64_t* d_sse) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !src_stride || !d_dst || !dst_stride || !refs || !noise_levels || !d_sse || (pix_bytes != 1 && pix_bytes != 2) ||
        (pix_bytes == 1 && bd != 8) || (pix_bytes == 2 && (bd < 8 || bd > 12)) || w <= 0 || h <= 0 || (w & 63) || (h & 63) || n_refs < 1 ||
        n_refs > SVT_HIP_TF_MAX_REFS || (ss_x != 0 && ss_x != 1) || (ss_y != 0 && ss_y != 1) || (ss_y == 1 && ss_x == 0) || decay_control <= 0) {
        if (c) c->err = "svt_hip_tf_filter_frame_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    for (int p = 0; p < (tf_chroma ? 3 : 1); p++) {
        if (!d_src[p] || !d_dst[p]) return SVT_HIP_ERR_BAD_ARG;
        for (int f = 0; f < n_refs; f++)
            if (refs[f].blocks && !refs[f].pred[p]) return SVT_HIP_ERR_BAD_ARG;
    }
    // the per-call scalars of EbTemporalFiltering.c:706 / :731-733, in the reference's own double arithmetic (host libm log1p)
    double den[3];
    for (int p = 0; p < 3; p++) {
        const double n_decay = (double)decay_control * (0.7 + log1p(noise_levels[p]));
        den[p] = 2 * n_decay * n_decay;
    }
    const double thr = min_frame_size * 0.1;
    const double dist_thr = thr > 1 ? thr : 1;
    hipError_t e = hipMemsetAsync(d_sse, 0, 2 * sizeof(uint64_t), c->stream);
    if (e != hipSuccess) return fail(c, e, "tf sse memset");
    e = (hipError_t)svt_hip_launch_tf_filter(c->stream, pix_bytes, bd, d_src, src_stride, d_dst, dst_stride, w, h, ss_x, ss_y, tf_chroma, refs, n_refs,
                                             den, dist_thr, d_sse);
    if (e != hipSuccess) return fail(c, e, "tf filter launch");
    return SVT_HIP_OK;
}

int svt_hip_tf_estimate_noise_dev(SvtHipCtx* c, const void* d_src, int pix_bytes, int bd, int width, int height, int stride, int64_t* d_out) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_out || (pix_bytes != 1 && pix_bytes != 2) || (pix_bytes == 1 && bd != 8) || (pix_bytes == 2 && (bd < 8 || bd > 12)) ||
        width <= 0 || height <= 0 || stride < width)
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = hipMemsetAsync(d_out, 0, 2 * sizeof(int64_t), c->stream);
    if (e != hipSuccess) return fail(c, e, "tf noise memset");
    e = (hipError_t)svt_hip_launch_tf_noise(c->stream, d_src, pix_bytes, bd, width, height, stride, (uint64_t*)d_out);
    if (e != hipSuccess) return fail(c, e, "tf noise launch");
    return SVT_HIP_OK;
}

int svt_hip_compound_predict_batch_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_ref0, int ref0_stride, const void* d_ref1, int ref1_stride,
                                       void* d_dst, int dst_stride, uint8_t* d_masks, const SvtHipCompBlk* d_blks, int nblk) {
    SVT_HIP_ENTER(c);
    if (!c || nblk < 0 || (pix_bytes != 1 && pix_bytes != 2) || (pix_bytes == 1 && bd != 8) || (pix_bytes == 2 && bd != 8 && bd != 10 && bd != 12)) {
        if (c) c->err = "svt_hip_compound_predict_batch_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    if (nblk == 0) return SVT_HIP_OK;
    if (!d_ref0 || !d_ref1 || !d_dst || !d_blks) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_compound_predict(c->stream, pix_bytes, bd, d_ref0, ref0_stride, d_ref1, ref1_stride, d_dst, dst_stride, d_masks, d_blks, nblk);
    if (e != hipSuccess) return fail(c, e, "compound predict launch");
    return SVT_HIP_OK;
}

int svt_hip_obmc_cost_batch_dev(SvtHipCtx* c, const uint8_t* d_pre, int pre_stride, const int32_t* d_wsrc, const int32_t* d_mask, const SvtHipObmcBlk* d_blks,
                                int nblk, uint32_t* d_out) {
    SVT_HIP_ENTER(c);
    if (!c || nblk < 0) return SVT_HIP_ERR_BAD_ARG;
    if (nblk == 0) return SVT_HIP_OK;
    if (!d_pre || !d_wsrc || !d_mask || !d_blks || !d_out) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_obmc_cost(c->stream, d_pre, pre_stride, d_wsrc, d_mask, d_blks, nblk, d_out);
    if (e != hipSuccess) return fail(c, e, "obmc cost launch");
    return SVT_HIP_OK;
}

int svt_hip_warp_predict_batch_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_ref, int width, int height, int stride, void* d_dst, int dst_stride,
                                   int ss_x, int ss_y, const SvtHipWarpBlk* d_blks, int nblk) {
    SVT_HIP_ENTER(c);
    if (!c || nblk < 0 || (pix_bytes != 1 && pix_bytes != 2) || (pix_bytes == 1 && bd != 8) || (pix_bytes == 2 && bd != 8 && bd != 10 && bd != 12) ||
        (ss_x != 0 && ss_x != 1) || (ss_y != 0 && ss_y != 1)) {
        if (c) c->err = "svt_hip_warp_predict_batch_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    if (nblk == 0) return SVT_HIP_OK;
    if (!d_ref || !d_dst || !d_blks || width <= 0 || height <= 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_warp_predict(c->stream, pix_bytes, bd, d_ref, width, height, stride, d_dst, dst_stride, ss_x, ss_y, d_blks, nblk);
    if (e != hipSuccess) return fail(c, e, "warp predict launch");
    return SVT_HIP_OK;
}

int svt_hip_blend_a64_batch_dev(SvtHipCtx* c, int pix_bytes, const void* d_src0, int src0_stride, const void* d_src1, int src1_stride, void* d_dst, int dst_stride,
                                const uint8_t* d_masks, const SvtHipBlendBlk* d_blks, int nblk) {
    SVT_HIP_ENTER(c);
    if (!c || nblk < 0 || (pix_bytes != 1 && pix_bytes != 2)) return SVT_HIP_ERR_BAD_ARG;
    if (nblk == 0) return SVT_HIP_OK;
    if (!d_src0 || !d_src1 || !d_dst || !d_masks || !d_blks) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_blend_a64(c->stream, pix_bytes, d_src0, src0_stride, d_src1, src1_stride, d_dst, dst_stride, d_masks, d_blks, nblk);
    if (e != hipSuccess) return fail(c, e, "blend_a64 launch");
    return SVT_HIP_OK;
}

int svt_hip_picture_format_dev(SvtHipCtx* c, int mode, const void* d_in0, int in0_stride, const void* d_in1, int in1_stride, void* d_out0, int out0_stride,
                               void* d_out1, int out1_stride, int w, int h) {
    SVT_HIP_ENTER(c);
    const bool two_in = mode == 0 || mode == 1 || mode == 6;
    if (!c || mode < 0 || mode > 6 || w < 0 || h < 0 || ((mode == 1 || mode == 5) && (w & 3))) {
        if (c) c->err = "svt_hip_picture_format_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    if (w == 0 || h == 0) return SVT_HIP_OK;
    if (!d_in0 || !d_out0 || (two_in && !d_in1)) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_picture_format(c->stream, mode, d_in0, in0_stride, d_in1, in1_stride, d_out0, out0_stride, d_out1, out1_stride, w, h);
    if (e != hipSuccess) return fail(c, e, "picture format launch");
    return SVT_HIP_OK;
}

int svt_hip_generate_padding_dev(SvtHipCtx* c, void* d_plane, int pix_bytes, int stride, int w, int h, int pad_w, int pad_h) {
    SVT_HIP_ENTER(c);
    if (!c || (pix_bytes != 1 && pix_bytes != 2) || w < 0 || h < 0 || pad_w < 0 || pad_h < 0) return SVT_HIP_ERR_BAD_ARG;
    if (w == 0 || h == 0 || (pad_w == 0 && pad_h == 0)) return SVT_HIP_OK;
    if (!d_plane || stride < w + pad_w) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_generate_padding(c->stream, d_plane, pix_bytes, stride, w, h, pad_w, pad_h);
    if (e != hipSuccess) return fail(c, e, "generate padding launch");
    return SVT_HIP_OK;
}

}  // extern "C"
