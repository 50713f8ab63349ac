// svt_hip_api.cpp — the C-ABI layer of libsvtav1_hip.so (include/svt_hip.h): context, memory,
// host-pointer convenience wrappers around the batched kernel launchers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <limits.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "svt_hip_internal.h"
#include "svt_hip_host.h"
#include <mutex>

struct SvtHipCtx {
    int         device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
    int         select_form = -1;   // svt_hip_set_cdef_select_form
    hipEvent_t  ev_sel_in = nullptr, ev_sel_out = nullptr;   // hand-over to and from the device's selection stream (svt_hip_cdef_strength_select_dev)
    int         me_waves = 4;   // 256 threads per SB: measured best on MI355X (tools/me_time.py)
    int         me_big = 1;     // also launch the strip-walking instance for search areas above 65 536 candidates
    void*       scratch = nullptr;   // library-owned device scratch (16-bit Wiener statistics, self-guided unit search), grown on demand
    size_t      scratch_bytes = 0;
    void*       host_scratch = nullptr;   // pinned host staging of the self-guided unit search
    size_t      host_scratch_bytes = 0;
    void*       me_buf = nullptr;         // device staging of svt_hip_me_fullpel_frame (host-pointer form), grown on demand
    size_t      me_buf_bytes = 0;
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

// The one-launch CDEF strength selection keeps 256 workgroups resident that wait for each other; two such launches on different streams could each hold a
// part of the compute units and wait for the rest forever.  Every context of a device therefore issues them on ONE stream of that device (created with the
// first context, kept for the life of the process), ordered against the caller's stream with a pair of events; kernels that do not wait for other workgroups
// run beside it as before.  Works under stream capture too (the event wait pulls the selection stream into the capture, the second event joins it back).
// While the caller's stream is being captured the order is expressed inside the capture instead: each captured selection waits for the event the previous
// captured selection of the same capture recorded (a fresh event per call; they are kept until the process ends, a graph may be instantiated from them later),
// so the selection nodes of a graph's parallel branches form one chain.
static std::mutex  g_sel_mutex;
static hipStream_t g_sel_stream[64];
static hipEvent_t  g_sel_cap_event[64];
static unsigned long long g_sel_cap_id[64];
extern "C" int svt_hip_strength_select_is_resident(int form, int sb_count);

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
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_sel_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_sel_out, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return SVT_HIP_ERR_RUNTIME;
    }
    if (device_id < 64) {
        std::lock_guard<std::mutex> lk(g_sel_mutex);
        if (!g_sel_stream[device_id] && hipStreamCreateWithFlags(&g_sel_stream[device_id], hipStreamNonBlocking) != hipSuccess) g_sel_stream[device_id] = nullptr;
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
    if (c->ev_sel_in) (void)hipEventDestroy(c->ev_sel_in);
    if (c->ev_sel_out) (void)hipEventDestroy(c->ev_sel_out);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->host_scratch) (void)hipHostFree(c->host_scratch);
    if (c->me_buf) (void)hipFree(c->me_buf);
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
void  svt_hip_ctx_clear_error(SvtHipCtx* c) { if (c) c->err.clear(); }
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
int svt_hip_memcpy2d_h2d_async(SvtHipCtx* c, void* d, size_t dpitch, const void* h, size_t hpitch, size_t wbytes, size_t rows) {
    SVT_HIP_ENTER(c);
    if (!c || !d || !h || dpitch < wbytes || hpitch < wbytes) return SVT_HIP_ERR_BAD_ARG;
    if (!wbytes || !rows) return SVT_HIP_OK;
    HIPCHK(c, hipMemcpy2DAsync(d, dpitch, h, hpitch, wbytes, rows, hipMemcpyHostToDevice, c->stream));
    return SVT_HIP_OK;
}
int svt_hip_memcpy2d_d2h_async(SvtHipCtx* c, void* h, size_t hpitch, const void* d, size_t dpitch, size_t wbytes, size_t rows) {
    SVT_HIP_ENTER(c);
    if (!c || !d || !h || dpitch < wbytes || hpitch < wbytes) return SVT_HIP_ERR_BAD_ARG;
    if (!wbytes || !rows) return SVT_HIP_OK;
    HIPCHK(c, hipMemcpy2DAsync(h, hpitch, d, dpitch, wbytes, rows, hipMemcpyDeviceToHost, c->stream));
    return SVT_HIP_OK;
}
int svt_hip_memcpy_h2d_async(SvtHipCtx* c, void* d, const void* h, size_t bytes) {
    SVT_HIP_ENTER(c);
    if (!c || (bytes && (!d || !h))) return SVT_HIP_ERR_BAD_ARG;
    if (!bytes) return SVT_HIP_OK;
    HIPCHK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    return SVT_HIP_OK;
}
int svt_hip_memcpy_d2h_async(SvtHipCtx* c, void* h, const void* d, size_t bytes) {
    SVT_HIP_ENTER(c);
    if (!c || (bytes && (!d || !h))) return SVT_HIP_ERR_BAD_ARG;
    if (!bytes) return SVT_HIP_OK;
    HIPCHK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    return SVT_HIP_OK;
}
int svt_hip_device_count(int* count) {
    if (!count) return SVT_HIP_ERR_BAD_ARG;
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 0) return SVT_HIP_ERR_NO_DEVICE;
    *count = n;
    return SVT_HIP_OK;
}
// Page-locks a buffer of the caller in place (the reference's picture buffers are allocated once per encoder instance): copies to and from it are then direct
// DMA at the link's rate instead of going through the runtime's pageable staging.
int svt_hip_host_register(SvtHipCtx* c, void* host, size_t bytes) {
    SVT_HIP_ENTER(c);
    if (!c || !host || !bytes) return SVT_HIP_ERR_BAD_ARG;
    const hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return SVT_HIP_OK; }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, e, "hipHostRegister"); }
    return SVT_HIP_OK;
}
int svt_hip_host_unregister(SvtHipCtx* c, void* host) {
    SVT_HIP_ENTER(c);
    if (!c || !host) return SVT_HIP_ERR_BAD_ARG;
    const hipError_t e = hipHostUnregister(host);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, e, "hipHostUnregister"); }
    return SVT_HIP_OK;
}
int svt_hip_host_alloc(SvtHipCtx* c, void** host, size_t bytes) {
    SVT_HIP_ENTER(c);
    if (!c || !host) return SVT_HIP_ERR_BAD_ARG;
    HIPCHK(c, hipHostMalloc(host, bytes ? bytes : 4, hipHostMallocDefault));
    return SVT_HIP_OK;
}
int svt_hip_host_free(SvtHipCtx* c, void* host) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    if (host) HIPCHK(c, hipHostFree(host));
    return SVT_HIP_OK;
}
#define SVT_HIP_TUS(X) X(cdef) X(compound) X(conv) X(deblock) X(distortion) X(format) X(md_pre) X(me_fullpel) X(percall) X(percall2) X(pyramid) X(sgr) X(sgr_walk) X(tf_subpel) \
    X(tfilter) X(txfm2d) X(warp) X(wiener)
#define X(n) int svt_hip_tu_probe_##n();
SVT_HIP_TUS(X)
#undef X
int svt_hip_warmup(SvtHipCtx* c) {
    SVT_HIP_ENTER(c);
    if (!c) return SVT_HIP_ERR_BAD_ARG;
    int bad = 0;
#define X(n) bad |= svt_hip_tu_probe_##n();
    SVT_HIP_TUS(X)
#undef X
    if (bad) { c->err = "svt_hip_warmup: a translation unit's code object did not load"; (void)hipGetLastError(); return SVT_HIP_ERR_RUNTIME; }
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

int svt_hip_me_get_big_windows(SvtHipCtx* c, int* enabled) {
    if (!c || !enabled) return SVT_HIP_ERR_BAD_ARG;
    *enabled = c->me_big;
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
    if (!n_sb) return SVT_HIP_OK;
    // Only the rows the windows touch travel (an ME segment is a band of superblock rows), into one staging buffer the context keeps:
    // [source rows | reference rows | windows | SADs | MVs].  Both planes are uploaded over the same row range so that one origin serves both.
    int lo = plane_rows, hi = 0;
    for (int i = 0; i < n_sb; i++) {
        const int s0 = org_y + sbs[i].sb_y, r0 = s0 + sbs[i].y_origin;
        lo = std::min(lo, std::min(s0, r0));
        hi = std::max(hi, std::max(s0 + 64, r0 + (int)sbs[i].height + 63));
    }
    lo = std::max(lo, 0);
    hi = std::min(hi, plane_rows);
    if (lo >= hi) { lo = 0; hi = plane_rows; }
    const size_t band = (size_t)stride * (size_t)(hi - lo), nres = (size_t)n_sb * SVT_HIP_SQUARE_PU_COUNT * 4;
    const size_t off_ref = (band + 256 + 255) & ~(size_t)255 /* slack: dword-aligned window loads may run a few bytes past a row */, off_sbs = 2 * off_ref, off_sad = off_sbs + ((sizeof(SvtHipSbSearch) * (size_t)n_sb + 255) & ~(size_t)255);
    const size_t off_mv = off_sad + ((nres + 255) & ~(size_t)255), total = off_mv + nres + 256;
    if (c->me_buf_bytes < total) {
        if (c->me_buf) (void)hipFree(c->me_buf);
        c->me_buf = nullptr;
        c->me_buf_bytes = 0;
        const hipError_t e = hipMalloc(&c->me_buf, total + total / 4);
        if (e != hipSuccess) return fail(c, e, "svt_hip_me_fullpel_frame: staging buffer");
        c->me_buf_bytes = total + total / 4;
    }
    uint8_t* const base = (uint8_t*)c->me_buf;
    const int saved_big = c->me_big;
    c->me_big = big;   // the windows are known here: launch the strip-walking instance only when one needs it
    int rc = SVT_HIP_OK;
    if ((rc = svt_hip_memcpy_h2d(c, base, src + (size_t)lo * stride, band)) || (rc = svt_hip_memcpy_h2d(c, base + off_ref, ref + (size_t)lo * stride, band)) ||
        (rc = svt_hip_memcpy_h2d(c, base + off_sbs, sbs, sizeof(SvtHipSbSearch) * (size_t)n_sb)))
        goto done;
    if ((rc = svt_hip_me_fullpel_frame_dev(c, base, base + off_ref, stride, org_x, org_y - lo, (const SvtHipSbSearch*)(base + off_sbs), n_sb, sub_sad,
                                           (uint32_t*)(base + off_sad), (uint32_t*)(base + off_mv))))
        goto done;
    if ((rc = svt_hip_memcpy_d2h(c, best_sad, base + off_sad, nres)) || (rc = svt_hip_memcpy_d2h(c, best_mv, base + off_mv, nres))) goto done;
done:
    c->me_big = saved_big;
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

int svt_hip_iwht4x4_add_batch_dev(SvtHipCtx* c, int pix_bytes, int bd, const int32_t* d_dqcoeff, const uint16_t* d_eob, const void* d_pred, int pred_stride,
                                  void* d_recon, int recon_stride, const uint32_t* d_descs, int nblk) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dqcoeff || !d_pred || !d_recon || !d_descs || nblk < 0 || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8) ||
        ((uintptr_t)d_dqcoeff & 15)) {
        if (c) c->err = "svt_hip_iwht4x4_add_batch_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_iwht4x4_add(c->stream, pix_bytes, bd, d_dqcoeff, d_eob, d_pred, pred_stride, d_recon, recon_stride, d_descs, nblk);
    if (e != hipSuccess) return fail(c, e, "iwht4x4_add launch");
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

int svt_hip_deblock_frame_fused_dev(SvtHipCtx* c, const void* const d_src[3], void* const d_dst[3], int pix_bytes, const int stride[3], int bd, const int plane_w[3],
                                    const int plane_h[3], const uint16_t* const d_edges_v[3], const uint16_t* const d_edges_h[3], const int units_w[3],
                                    const int units_h[3], int sharpness) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_dst || !stride || !plane_w || !plane_h || !d_edges_v || !d_edges_h || !units_w || !units_h || (pix_bytes != 1 && pix_bytes != 2) ||
        (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8) || sharpness < 0 || sharpness > 7) {
        if (c) c->err = "svt_hip_deblock_frame_fused_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    for (int p = 0; p < 3; p++)
        if (d_src[p] && (!d_dst[p] || d_dst[p] == d_src[p] || plane_w[p] <= 0 || plane_h[p] <= 0 || units_w[p] < (plane_w[p] + 3) / 4 || units_h[p] < (plane_h[p] + 3) / 4 ||
                         !d_edges_v[p] || !d_edges_h[p])) {
            c->err = "svt_hip_deblock_frame_fused_dev: bad plane argument (the fused form is out of place)";
            return SVT_HIP_ERR_BAD_ARG;
        }
    hipError_t e = (hipError_t)svt_hip_launch_deblock_fused(c->stream, d_src, d_dst, pix_bytes, stride, bd, plane_w, plane_h, d_edges_v, d_edges_h, units_w, units_h, sharpness);
    if (e != hipSuccess) return fail(c, e, "fused deblock launch");
    return SVT_HIP_OK;
}

int svt_hip_dlf_build_edges_picture_dev(SvtHipCtx* c, const SvtHipDlfModeInfo* d_mi, int mi_cols, int mi_rows, int ss_x, int ss_y, const int plane_w[3], const int plane_h[3],
                                        const int filt_units_w[3], const int filt_units_h[3], const int (*level)[2], uint16_t* const d_edges_v[3], uint16_t* const d_edges_h[3]) {
    SVT_HIP_ENTER(c);
    if (!c || !d_mi || mi_cols <= 0 || mi_rows <= 0 || ss_x < 0 || ss_x > 1 || ss_y < 0 || ss_y > 1 || !plane_w || !plane_h || !filt_units_w || !filt_units_h || !d_edges_v ||
        !d_edges_h) {
        if (c) c->err = "svt_hip_dlf_build_edges_picture_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    for (int p = 0; p < 3; p++) {
        if (!d_edges_v[p] && !d_edges_h[p]) continue;
        if (!d_edges_v[p] || !d_edges_h[p] || plane_w[p] <= 0 || plane_h[p] <= 0 || filt_units_w[p] < 0 || filt_units_h[p] < 0 ||
            (level && (level[p][0] > 63 || level[p][1] > 63))) {
            c->err = "svt_hip_dlf_build_edges_picture_dev: bad plane argument";
            return SVT_HIP_ERR_BAD_ARG;
        }
    }
    hipError_t e = (hipError_t)svt_hip_launch_dlf_build_edges(c->stream, d_mi, mi_cols, mi_rows, ss_x, ss_y, plane_w, plane_h, filt_units_w, filt_units_h, level, d_edges_v, d_edges_h);
    if (e != hipSuccess) return fail(c, e, "edge builder launch");
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

int svt_hip_dlf_search_levels_picture_dev(SvtHipCtx* c, int n_planes, const SvtHipDlfSearchPlane* planes, int pix_bytes, int bd, uint64_t* d_sse_scratch, int* best_level,
                                          int64_t* best_err) {
    SVT_HIP_ENTER(c);
    if (!c || !planes || n_planes < 1 || n_planes > 3 || !d_sse_scratch || !best_level || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8))
        return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_planes; i++) {
        const SvtHipDlfSearchPlane& P = planes[i];
        if (!P.d_recon || !P.d_tmp[0] || !P.d_tmp[1] || !P.d_src || !P.d_edges_v || !P.d_edges_h || P.q.plane < 0 || P.q.plane > 2 || P.plane_w <= 0 || P.plane_h <= 0 ||
            P.units_w != (P.plane_w + 3) / 4 || P.units_h != (P.plane_h + 3) / 4 || P.q.sharpness < 0 || P.q.sharpness > 7) {
            c->err = "svt_hip_dlf_search_levels_picture_dev: bad plane";
            return SVT_HIP_ERR_BAD_ARG;
        }
    }
    int64_t ss_err[3][64];
    bool    done[3] = {false, false, false};
    for (int i = 0; i < 3; i++) for (int k = 0; k < 64; k++) ss_err[i][k] = -1;
    for (;;) {
        int need[3][2], n_need[3] = {0, 0, 0}, total = 0;
        for (int i = 0; i < n_planes; i++) {
            if (done[i]) continue;
            const int n = svt_hip_dlf_search_plan(&planes[i].q, ss_err[i], need[i], &best_level[i], best_err ? &best_err[i] : nullptr);
            if (n < 0) return n;
            if (n == 0) done[i] = true;
            n_need[i] = n; total += n;
        }
        if (!total) break;
        hipError_t e = hipMemsetAsync(d_sse_scratch, 0, sizeof(uint64_t) * 2 * n_planes, c->stream);
        for (int i = 0; i < n_planes && e == hipSuccess; i++)
            for (int k = 0; k < n_need[i] && e == hipSuccess; k++) {   // try_filter_frame (:966-1024) on the device: a copy of the plane as coded, deblocked at the probed level, against the source
                const SvtHipDlfSearchPlane& P = planes[i];
                int lv_v, lv_h;
                svt_hip_dlf_search_probe_levels(&P.q, need[i][k], &lv_v, &lv_h);
                e = hipMemcpy2DAsync(P.d_tmp[k], (size_t)P.stride * pix_bytes, P.d_recon, (size_t)P.stride * pix_bytes, (size_t)P.plane_w * pix_bytes, P.plane_h, hipMemcpyDeviceToDevice, c->stream);
                if (e == hipSuccess) e = (hipError_t)svt_hip_launch_deblock_plane(c->stream, P.d_tmp[k], pix_bytes, P.stride, bd, P.d_edges_v, P.d_edges_h, P.units_w, P.units_h, P.q.sharpness, lv_v, lv_h);
                if (e == hipSuccess) e = (hipError_t)svt_hip_launch_plane_sse(c->stream, pix_bytes, P.d_src, P.src_stride, P.d_tmp[k], P.stride, P.plane_w, P.plane_h, d_sse_scratch + 2 * i + k);
            }
        uint64_t sse[6] = {0, 0, 0, 0, 0, 0};
        if (e == hipSuccess) e = hipMemcpyAsync(sse, d_sse_scratch, sizeof(uint64_t) * 2 * n_planes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(c, e, "dlf level probes of a picture");
        for (int i = 0; i < n_planes; i++)
            for (int k = 0; k < n_need[i]; k++) ss_err[i][need[i][k]] = (int64_t)sse[2 * i + k];
    }
    return SVT_HIP_OK;
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
int svt_hip_enc_txfm_multi_dev(SvtHipCtx* c, int pix_bytes, int bd, const SvtHipEncTxJob* jobs, int njobs) {
    SVT_HIP_ENTER(c);
    if (!c || (!jobs && njobs) || njobs < 0 || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8)) return SVT_HIP_ERR_BAD_ARG;
    for (int j = 0; j < njobs; j++) {
        const SvtHipFwdTxJob& J = jobs[j].fwd;
        if (J.nblk < 0 || J.tx_size < 0 || J.tx_size > 18 || (J.nblk && (!J.d_src || !J.d_pred || !J.d_descs || !J.d_qcoeff || !jobs[j].d_recon || !J.scans.iscan[0])) ||
            J.qp.variant < 0 || J.qp.variant > 3 || J.qp.log_scale < 0 || J.qp.log_scale > 2 || J.qp.coeff_shape < 0 || J.qp.coeff_shape > 3) {
            c->err = "svt_hip_enc_txfm_multi_dev: bad job";
            return SVT_HIP_ERR_BAD_ARG;
        }
    }
    hipError_t e = (hipError_t)svt_hip_launch_enc_txfm_multi(c->stream, pix_bytes, bd, jobs, njobs);
    if (e != hipSuccess) return fail(c, e, "encode transform multi launch");
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
        (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8) || w <= 0 || h <= 0 || (w & 7) || (h & 7)) {
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
int svt_hip_subpel_jobs_from_me_dev(SvtHipCtx* c, const uint32_t* d_best_mv, int sb_cols, int w, int h, const uint8_t* d_frac_q4, SvtHipConvBlk* d_blks) {
    SVT_HIP_ENTER(c);
    if (!c || !d_best_mv || !d_blks || sb_cols < 1 || w < 16 || h < 16 || (w + 63) / 64 > sb_cols) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_subpel_jobs_from_me(c->stream, d_best_mv, sb_cols, w, h, d_frac_q4, d_blks);
    if (e != hipSuccess) return fail(c, e, "subpel jobs launch");
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
static int md_sad_picture(int pix_bytes, SvtHipCtx* c, const void* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                       int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, uint32_t* d_sad) {
    if (!c || !d_src || !pus || !refs || !d_mv || !d_sad || n_sb < 0 || sb_cols < 1 || pic_w < 1 || pic_h < 1 || n_pus < 1 || n_pus > SVT_HIP_MD_MAX_PUS || n_refs < 1 ||
        n_refs > SVT_HIP_MD_MAX_REFS)
        return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_pus; i++)
        if (pus[i].w < 4 || pus[i].w > 64 || (pus[i].w & 3) || pus[i].h < 1 || pus[i].h > 64 || pus[i].x + pus[i].w > 64 || pus[i].y + pus[i].h > 64) return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_refs; i++)
        if (!refs[i].d_plane || refs[i].stride < 1) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_md_fullpel_sad(c->stream, pix_bytes, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, d_sad);
    if (e != hipSuccess) return fail(c, e, "md full-pel sad launch");
    return SVT_HIP_OK;
}
int svt_hip_md_fullpel_sad_picture_dev(SvtHipCtx* c, const uint8_t* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                       int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, uint32_t* d_sad) {
    SVT_HIP_ENTER(c);
    return md_sad_picture(1, c, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, d_sad);
}
int svt_hip_md_fullpel_sad_picture_hbd_dev(SvtHipCtx* c, const uint16_t* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                           int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, uint32_t* d_sad) {
    SVT_HIP_ENTER(c);
    return md_sad_picture(2, c, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, d_sad);
}
static int md_avg_sad_picture(int pix_bytes, SvtHipCtx* c, const void* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                           int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, int n_pairs, const uint8_t (*pairs)[2], uint32_t* d_sad) {
    if (!c || !d_src || !pus || !refs || !d_mv || !d_sad || !pairs || n_sb < 0 || sb_cols < 1 || pic_w < 1 || pic_h < 1 || n_pus < 1 || n_pus > SVT_HIP_MD_MAX_PUS || n_refs < 1 ||
        n_refs > SVT_HIP_MD_MAX_REFS || n_pairs < 1 || n_pairs > SVT_HIP_MD_MAX_PAIRS)
        return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_pus; i++)
        if (pus[i].w < 4 || pus[i].w > 64 || (pus[i].w & 3) || pus[i].h < 1 || pus[i].h > 64 || pus[i].x + pus[i].w > 64 || pus[i].y + pus[i].h > 64) return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_refs; i++)
        if (!refs[i].d_plane || refs[i].stride < 1) return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_pairs; i++)
        if (pairs[i][0] >= n_refs || pairs[i][1] >= n_refs) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_md_fullpel_avg_sad(c->stream, pix_bytes, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, n_pairs, pairs, d_sad);
    if (e != hipSuccess) return fail(c, e, "md compound-average sad launch");
    return SVT_HIP_OK;
}
int svt_hip_md_fullpel_avg_sad_picture_dev(SvtHipCtx* c, const uint8_t* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                           int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, int n_pairs, const uint8_t (*pairs)[2], uint32_t* d_sad) {
    SVT_HIP_ENTER(c);
    return md_avg_sad_picture(1, c, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, n_pairs, pairs, d_sad);
}
int svt_hip_md_fullpel_avg_sad_picture_hbd_dev(SvtHipCtx* c, const uint16_t* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                               int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, int n_pairs, const uint8_t (*pairs)[2], uint32_t* d_sad) {
    SVT_HIP_ENTER(c);
    return md_avg_sad_picture(2, c, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, n_pairs, pairs, d_sad);
}
static int md_grid_picture(int grid, SvtHipCtx* c, const uint8_t* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                       int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, int bank, uint32_t* d_out) {
    if (!c || !d_src || !pus || !refs || !d_mv || !d_out || n_sb < 0 || sb_cols < 1 || pic_w < 1 || pic_h < 1 || n_pus < 1 || n_pus > SVT_HIP_MD_MAX_PUS || n_refs < 1 ||
        n_refs > SVT_HIP_MD_MAX_REFS || bank < 0 || bank > 5)
        return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_pus; i++)
        if (pus[i].x + pus[i].w > 64 || pus[i].y + pus[i].h > 64) return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_refs; i++)
        if (!refs[i].d_plane || refs[i].stride < 1) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_md_subpel_grid(c->stream, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, bank, grid, d_out);
    if (e != hipSuccess) return fail(c, e, "md sub-pel grid launch");
    return SVT_HIP_OK;
}
int svt_hip_md_subpel_grid_picture_dev(SvtHipCtx* c, const uint8_t* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                       int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, int bank, uint32_t* d_out) {
    SVT_HIP_ENTER(c);
    return md_grid_picture(7, c, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, bank, d_out);
}
int svt_hip_md_halfpel_grid_picture_dev(SvtHipCtx* c, const uint8_t* d_src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus, const SvtHipMdPu* pus,
                                        int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* d_mv, int bank, uint32_t* d_out) {
    SVT_HIP_ENTER(c);
    return md_grid_picture(3, c, d_src, src_stride, pic_w, pic_h, sb_cols, n_sb, n_pus, pus, n_refs, refs, d_mv, bank, d_out);
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
    if (!c || !d_a || !d_b || !d_pairs || !d_var || n < 0 || !((pix_bytes == 1 && bd == 8) || (pix_bytes == 2 && (bd == 10 || bd == 16)))) return SVT_HIP_ERR_BAD_ARG;
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
int svt_hip_lr_try_unit_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, void* d_dst, int dst_stride, int pw, int ph, int unit_size, int ss_y,
                            const void* d_dbl, int dbl_stride, const uint8_t* d_unit_ep, const int32_t* d_unit_xqd, const int16_t* d_unit_wiener, const void* d_src,
                            int src_stride, int unit, uint64_t* d_sse) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dgd || !d_dst || !d_unit_ep || !d_unit_xqd || !d_src || !d_sse || unit_size < 64 || (unit_size & 63) || (ss_y != 0 && ss_y != 1) ||
        !sgr_args_ok(pix_bytes, bd, pw, ph))
        return SVT_HIP_ERR_BAD_ARG;
    const int ux = sgr_units(pw, unit_size), uy = sgr_units(ph, unit_size);
    if (unit < 0 || unit >= ux * uy) return SVT_HIP_ERR_BAD_ARG;
    // the unit's rectangle: foreach_rest_unit_in_tile (Common/Codec/EbRestoration.c:1369-1411) — the last unit of a row / column takes the remainder
    const int uj = unit % ux, ui = unit / ux, voff = 8 >> ss_y;
    const int x0 = uj * unit_size, w = uj == ux - 1 ? pw - x0 : unit_size;
    const int y0 = ui * unit_size, h = ui == uy - 1 ? ph - y0 : unit_size;
    const int v0 = y0 - voff > 0 ? y0 - voff : 0, v1 = (y0 + h < ph) ? y0 + h - voff : y0 + h;
    // tiles are 64 x 32 starting at (0, -voff): unit boundaries fall on tile boundaries
    const int tx0 = x0 / 64, tx1 = (x0 + w + 63) / 64, ty0 = (v0 + voff) / 32, ty1 = (v1 + voff + 31) / 32;
    hipError_t e = (hipError_t)svt_hip_launch_sgr_apply_tiles(c->stream, pix_bytes, bd, d_dgd, stride, d_dst, dst_stride, pw, ph, unit_size, ux, uy, ss_y, d_dbl, dbl_stride,
                                                             d_unit_ep, d_unit_xqd, d_unit_wiener, tx0, ty0, tx1 - tx0, ty1 - ty0);
    if (e != hipSuccess) return fail(c, e, "restoration unit launch");
    HIPCHK(c, hipMemsetAsync(d_sse, 0, sizeof(uint64_t), c->stream));
    const uint8_t* a = (const uint8_t*)d_src + ((size_t)v0 * src_stride + x0) * pix_bytes;
    const uint8_t* b = (const uint8_t*)d_dst + ((size_t)v0 * dst_stride + x0) * pix_bytes;
    e = (hipError_t)svt_hip_launch_plane_sse(c->stream, pix_bytes, a, src_stride, b, dst_stride, w, v1 - v0, d_sse);
    if (e != hipSuccess) return fail(c, e, "restoration unit sse launch");
    return SVT_HIP_OK;
}
int svt_hip_lr_try_units_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, void* d_dst, int dst_stride, int pw, int ph, int unit_size, int ss_y,
                             const void* d_dbl, int dbl_stride, const uint8_t* d_unit_ep, const int32_t* d_unit_xqd, const int16_t* d_unit_wiener, const void* d_src,
                             int src_stride, const SvtHipBlkPair* d_rects, int n_rects, uint64_t* d_sse) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_rects || !d_sse || n_rects < 0) return SVT_HIP_ERR_BAD_ARG;
    int rc = svt_hip_lr_apply_plane_dev(c, pix_bytes, bd, d_dgd, stride, d_dst, dst_stride, pw, ph, unit_size, ss_y, d_dbl, dbl_stride, d_unit_ep, d_unit_xqd, d_unit_wiener);
    if (rc != SVT_HIP_OK) return rc;
    return svt_hip_block_sse_batch_dev(c, pix_bytes, d_src, src_stride, d_dst, dst_stride, d_rects, n_rects, d_sse);
}

int svt_hip_wiener_walk_units_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, int pw, int ph, int unit_size, int ss_y, const void* d_dbl, int dbl_stride,
                                  const void* d_src, int src_stride, int16_t* d_unit_wiener, const uint8_t* d_active, int wiener_win, int64_t* d_err, uint32_t* d_probes) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dgd || !d_src || !d_unit_wiener || !d_active || !d_err || unit_size < 64 || (unit_size & 63) || (ss_y != 0 && ss_y != 1) ||
        (wiener_win != 7 && wiener_win != 5 && wiener_win != 3) || !sgr_args_ok(pix_bytes, bd, pw, ph)) {
        if (c) c->err = "svt_hip_wiener_walk_units_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    const SvtHipWienerWalkPlane P = {d_dgd, stride, pw, ph, unit_size, ss_y, d_dbl, dbl_stride, d_src, src_stride, d_unit_wiener, d_active, wiener_win, d_err, d_probes};
    hipError_t e = (hipError_t)svt_hip_launch_wiener_walk_multi(c->stream, pix_bytes, bd, 1, &P);
    if (e != hipSuccess) return fail(c, e, "wiener walk launch");
    return SVT_HIP_OK;
}

int svt_hip_wiener_walk_units_picture_dev(SvtHipCtx* c, int pix_bytes, int bd, int n_planes, const SvtHipWienerWalkPlane* planes) {
    SVT_HIP_ENTER(c);
    if (!c || !planes || n_planes < 1 || n_planes > 3) return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_planes; i++) {
        const SvtHipWienerWalkPlane& P = planes[i];
        if (!P.d_dgd || !P.d_src || !P.d_unit_wiener || !P.d_active || !P.d_err || P.unit_size < 64 || (P.unit_size & 63) || (P.ss_y != 0 && P.ss_y != 1) ||
            (P.wiener_win != 7 && P.wiener_win != 5 && P.wiener_win != 3) || !sgr_args_ok(pix_bytes, bd, P.pw, P.ph)) {
            c->err = "svt_hip_wiener_walk_units_picture_dev: bad plane";
            return SVT_HIP_ERR_BAD_ARG;
        }
    }
    hipError_t e = (hipError_t)svt_hip_launch_wiener_walk_multi(c->stream, pix_bytes, bd, n_planes, planes);
    if (e != hipSuccess) return fail(c, e, "wiener walk launch");
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

/* ---- search_selfguided_restoration (Encoder/Codec/EbRestorationPick.c:583-671) for every unit of a plane, entirely on the device ----
 * launch 1: sgr_search8_kernel<STORE>: the five projection sums of every (unit, set) + the int16 planes flt0 - u, flt1 - u, dat - src
 * then sgr_walk.hip: replay (one wave per (unit, set): solve, encode_xq, the walk on exact errors, best-first speculation) and evaluate launches alternate a
 * fixed number of times, a last launch writes the results and every unit's best set
 * No host synchronisation in between; the scratch (sums, arrival counters, difference planes) is the caller's. */
namespace {
struct SgrScratch { size_t stats, sums, d2, states, esc_cnt, sd, pairs, esc, total, total_packed, dplane; int dstride, nu; };
// SVT_HIP_SGR_PACKED=1 (bit depth 8 only) runs the unit search on PACKED difference words (one 32-bit word per sample and set in `pairs`, no dat - src plane; sgr.hip
// STORE == 2, sgr_walk_packed_kernel).  A measured negative result, kept as the experiment it is (profiles/r06/sgr_packed_ab.txt): the form cuts the walk's memory traffic by a quarter (1.85 -> 1.41 GB per 4K frame) and
// raises its resident share from 38 % to 55-66 %, and the walk takes the same time -- its evaluation is bound by v_dot2 issue and by the one memory round trip per streamed chunk,
// not by bytes -- while the filter kernel pays 0.1 ms per 4K frame for the packing.  The default stays the 6-byte form.
static bool sgr_packed(int bd) {
    const char* env = getenv("SVT_HIP_SGR_PACKED");   // read per call: tests run both forms in one process
    return bd == 8 && env && env[0] == '1';
}
SgrScratch sgr_scratch_layout(int pw, int ph, int unit_size) {
    SgrScratch L;
    L.nu = sgr_units(pw, unit_size) * sgr_units(ph, unit_size);
    L.dstride = (pw + 63) & ~63;
    L.dplane = (size_t)L.dstride * (size_t)ph;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    L.stats = o;    o = al(o + 128);   // diagnostics over the plane: [0] evaluation passes, [1] evaluated points, [2] unfinished walks, [24..30] phase clocks of the walks (sgr_walk.hip)
    L.sums = o;     o = al(o + sizeof(int64_t) * (size_t)L.nu * 16 * 5);
    L.d2 = o;       o = al(o + sizeof(int64_t) * (size_t)L.nu);   // sum (dat - src)^2 per unit
    L.states = o;   o = al(o + svt_hip_sgr_walk_state_bytes(L.nu));   // per (unit, set): cache of evaluated points, points wanted next, result
    L.esc_cnt = o;  o = al(o + sizeof(uint32_t) * (size_t)L.nu * 16);   // packed form: listed samples per (unit, set)
    L.sd = o;       o = al(o + sizeof(int16_t) * L.dplane);            // everything before this is cleared per call
    L.pairs = o;    o = al(o + sizeof(uint32_t) * L.dplane * 16);
    // packed form: the escape lists, 13 filter pairs x one 8-byte entry per sample (the worst case -- every sample of a binary test picture -- is what the lists are
    // sized for, so that there is no second code path for "too many"; coded pictures leave them empty and untouched)
    L.esc = o;      L.total = o;   // the 6-byte form ends here
    L.total_packed = al(o + sizeof(uint64_t) * L.dplane * 13);
    return L;
}
}  // namespace

size_t svt_hip_sgr_search_units_scratch_bytes(int pw, int ph, int unit_size) {
    if (pw <= 0 || ph <= 0 || unit_size < 64 || (unit_size & 63)) return 0;
    const SgrScratch L = sgr_scratch_layout(pw, ph, unit_size);
    return sgr_packed(8) ? L.total_packed : L.total;   // the packed experiment's lists count only while it is switched on
}

int svt_hip_sgr_search_units_plane_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, const void* d_src, int src_stride, int pw,
                                       int ph, int unit_size, int ss_y, uint32_t ep_mask, int32_t* d_xqd, int64_t* d_err, uint8_t* d_best_ep,
                                       int32_t* d_best_xqd, void* d_scratch, size_t scratch_bytes) {
    SVT_HIP_ENTER(c);
    ep_mask &= 0xFFFFu;
    if (!c || !d_dgd || !d_src || !d_xqd || !d_err || !d_scratch || unit_size < 64 || (unit_size & 63) || (ss_y != 0 && ss_y != 1) ||
        !sgr_args_ok(pix_bytes, bd, pw, ph) || !ep_mask || ((uintptr_t)d_scratch & 15)) {
        if (c) c->err = "svt_hip_sgr_search_units_plane_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    const SgrScratch L = sgr_scratch_layout(pw, ph, unit_size);
    if (scratch_bytes < (sgr_packed(bd) ? L.total_packed : L.total)) {
        c->err = "svt_hip_sgr_search_units_plane_dev: scratch smaller than svt_hip_sgr_search_units_scratch_bytes()";
        return SVT_HIP_ERR_BAD_ARG;
    }
    char* base = (char*)d_scratch;
    HIPCHK(c, hipMemsetAsync(base, 0, L.sd, c->stream));   // statistics, sums, per-unit squared differences, the walk's arrival counters (one fill for all of them)
    const int ux = sgr_units(pw, unit_size), uy = sgr_units(ph, unit_size);
    const bool packed = sgr_packed(bd);
    hipError_t e = (hipError_t)svt_hip_launch_sgr_search_store(c->stream, pix_bytes, bd, d_dgd, stride, d_src, src_stride, pw, ph, unit_size, ux, uy, ss_y, ep_mask,
                                                              (int64_t*)(base + L.sums), (uint32_t*)(base + L.pairs), (int16_t*)(base + L.sd), L.dstride, L.dplane,
                                                              (int64_t*)(base + L.d2), packed ? base + L.esc : nullptr, (uint32_t*)(base + L.esc_cnt));
    if (e != hipSuccess) return fail(c, e, "sgr search (store) launch");
    const SvtHipSgrWalkPlane wp = {(const uint32_t*)(base + L.pairs), (const int16_t*)(base + L.sd), (const int64_t*)(base + L.sums), base + L.states, L.dplane, L.dstride,
                                   pw, ph, unit_size, ux, uy, ss_y, ep_mask, d_xqd, d_err, d_best_ep, d_best_xqd, (uint32_t*)(base + L.stats),
                                   packed ? base + L.esc : nullptr, (const uint32_t*)(base + L.esc_cnt)};
    e = (hipError_t)svt_hip_launch_sgr_walk_multi(c->stream, bd, 1, &wp);
    if (e != hipSuccess) return fail(c, e, "sgr walk launch");
    return SVT_HIP_OK;
}

// All planes of a picture: ONE launch of the sums / difference-plane kernel, then ONE walk launch for every (plane, unit, set) — one tail each instead of three.
int svt_hip_sgr_search_units_picture_dev(SvtHipCtx* c, int pix_bytes, int bd, int n_planes, const SvtHipSgrUnitsPlaneDev* pl) {
    SVT_HIP_ENTER(c);
    if (!c || !pl || n_planes < 1 || n_planes > SVT_HIP_SGR_MAX_PLANES) return SVT_HIP_ERR_BAD_ARG;
    SvtHipSgrWalkPlane wp[SVT_HIP_SGR_MAX_PLANES];
    SvtHipSgrSearchStorePlane sp[SVT_HIP_SGR_MAX_PLANES];
    const bool packed = sgr_packed(bd);
    for (int i = 0; i < n_planes; i++) {
        const SvtHipSgrUnitsPlaneDev& P = pl[i];
        const uint32_t ep_mask = P.ep_mask & 0xFFFFu;
        if (!P.d_dgd || !P.d_src || !P.d_xqd || !P.d_err || !P.d_scratch || P.unit_size < 64 || (P.unit_size & 63) || (P.ss_y != 0 && P.ss_y != 1) ||
            !sgr_args_ok(pix_bytes, bd, P.pw, P.ph) || !ep_mask || ((uintptr_t)P.d_scratch & 15)) {
            c->err = "svt_hip_sgr_search_units_picture_dev: bad plane";
            return SVT_HIP_ERR_BAD_ARG;
        }
        const SgrScratch L = sgr_scratch_layout(P.pw, P.ph, P.unit_size);
        if (P.scratch_bytes < (packed ? L.total_packed : L.total)) {
            c->err = "svt_hip_sgr_search_units_picture_dev: scratch smaller than svt_hip_sgr_search_units_scratch_bytes()";
            return SVT_HIP_ERR_BAD_ARG;
        }
        char* base = (char*)P.d_scratch;
        HIPCHK(c, hipMemsetAsync(base, 0, L.sd, c->stream));   // ... and the walk's arrival counters
        const int ux = sgr_units(P.pw, P.unit_size), uy = sgr_units(P.ph, P.unit_size);
        sp[i] = SvtHipSgrSearchStorePlane{P.d_dgd, P.d_src, (int64_t*)(base + L.sums), (uint32_t*)(base + L.pairs), (int16_t*)(base + L.sd), (int64_t*)(base + L.d2),
                                          packed ? base + L.esc : nullptr, (uint32_t*)(base + L.esc_cnt), L.dplane, P.stride, P.src_stride, P.pw, P.ph, P.unit_size, ux, uy, P.ss_y,
                                          L.dstride, ep_mask};
        wp[i] = SvtHipSgrWalkPlane{(const uint32_t*)(base + L.pairs), (const int16_t*)(base + L.sd), (const int64_t*)(base + L.sums), base + L.states, L.dplane, L.dstride,
                                   P.pw, P.ph, P.unit_size, ux, uy, P.ss_y, ep_mask, P.d_xqd, P.d_err, P.d_best_ep, P.d_best_xqd, (uint32_t*)(base + L.stats),
                                   packed ? base + L.esc : nullptr, (const uint32_t*)(base + L.esc_cnt)};
    }
    // one launch of the sums / difference-plane kernel for every plane (a chroma plane alone is one workgroup round: its launch lasts a workgroup's whole latency) ...
    hipError_t e = (hipError_t)svt_hip_launch_sgr_search_store_multi(c->stream, pix_bytes, bd, n_planes, sp);
    if (e != hipSuccess) return fail(c, e, "sgr search (store) launch");
    // ... and one walk launch for every (plane, unit, set)
    e = (hipError_t)svt_hip_launch_sgr_walk_multi(c->stream, bd, n_planes, wp);
    if (e != hipSuccess) return fail(c, e, "sgr walk launch");
    return SVT_HIP_OK;
}

// HOST-output convenience forms: the library's own scratch, one synchronisation at the very end (to hand the results over).
int svt_hip_sgr_search_units_picture(SvtHipCtx* c, int pix_bytes, int bd, int n_planes, const SvtHipSgrSearchPlane* planes, int* rounds_out) {
    SVT_HIP_ENTER(c);
    if (!c || !planes || n_planes < 1 || n_planes > 3) return SVT_HIP_ERR_BAD_ARG;
    struct Off { size_t scratch, xqd, err, best; int nu; } off[3];
    size_t need = 0;
    for (int k = 0; k < n_planes; k++) {
        const SvtHipSgrSearchPlane& P = planes[k];
        if (!P.d_dgd || !P.d_src || !P.xqd_out || !P.err_out || P.unit_size < 64 || (P.unit_size & 63) || (P.ss_y != 0 && P.ss_y != 1) ||
            !sgr_args_ok(pix_bytes, bd, P.pw, P.ph) || !(P.ep_mask & 0xFFFFu))
            return SVT_HIP_ERR_BAD_ARG;
        const SgrScratch L = sgr_scratch_layout(P.pw, P.ph, P.unit_size);
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        off[k].nu = L.nu;
        off[k].scratch = need; need = al(need + (sgr_packed(bd) ? L.total_packed : L.total));
        off[k].xqd = need;     need = al(need + sizeof(int32_t) * (size_t)L.nu * 32);
        off[k].err = need;     need = al(need + sizeof(int64_t) * (size_t)L.nu * 16);
        off[k].best = need;    need = al(need + (size_t)L.nu);
    }
    if (need > c->scratch_bytes) {   // the scratch may still be in use by work queued on ANY stream this context was pointed at
        HIPCHK(c, hipDeviceSynchronize());
        if (c->scratch) HIPCHK(c, hipFree(c->scratch));
        c->scratch = nullptr; c->scratch_bytes = 0;
        HIPCHK(c, hipMalloc(&c->scratch, need));
        c->scratch_bytes = need;
    }
    char* dev = (char*)c->scratch;
    for (int k = 0; k < n_planes; k++) {
        const SvtHipSgrSearchPlane& P = planes[k];
        const int rc = svt_hip_sgr_search_units_plane_dev(c, pix_bytes, bd, P.d_dgd, P.stride, P.d_src, P.src_stride, P.pw, P.ph, P.unit_size, P.ss_y, P.ep_mask,
                                                          (int32_t*)(dev + off[k].xqd), (int64_t*)(dev + off[k].err), (uint8_t*)(dev + off[k].best), nullptr,
                                                          dev + off[k].scratch, sgr_packed(bd) ? sgr_scratch_layout(P.pw, P.ph, P.unit_size).total_packed : sgr_scratch_layout(P.pw, P.ph, P.unit_size).total);
        if (rc != SVT_HIP_OK) return rc;
    }
    for (int k = 0; k < n_planes; k++) {
        const SvtHipSgrSearchPlane& P = planes[k];
        const size_t nu = (size_t)off[k].nu;
        std::vector<int32_t> xqd(nu * 32);
        std::vector<int64_t> err(nu * 16);
        HIPCHK(c, hipMemcpyAsync(xqd.data(), dev + off[k].xqd, sizeof(int32_t) * nu * 32, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(err.data(), dev + off[k].err, sizeof(int64_t) * nu * 16, hipMemcpyDeviceToHost, c->stream));
        if (P.best_ep) HIPCHK(c, hipMemcpyAsync(P.best_ep, dev + off[k].best, nu, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (size_t u = 0; u < nu; u++)
            for (int ep = 0; ep < 16; ep++) {
                if (!((P.ep_mask >> ep) & 1)) continue;   // sets outside the mask stay untouched
                if (err[u * 16 + ep] < 0) { c->err = "svt_hip_sgr_search_units: a walk did not finish within the pass budget"; return SVT_HIP_ERR_RUNTIME; }
                P.xqd_out[(u * 16 + ep) * 2] = xqd[(u * 16 + ep) * 2]; P.xqd_out[(u * 16 + ep) * 2 + 1] = xqd[(u * 16 + ep) * 2 + 1];
                P.err_out[u * 16 + ep] = err[u * 16 + ep];
            }
    }
    if (rounds_out) *rounds_out = 0;   // kept for source compatibility: there are no host rounds any more
    return SVT_HIP_OK;
}

int svt_hip_sgr_search_units_plane(SvtHipCtx* c, int pix_bytes, int bd, const void* d_dgd, int stride, const void* d_src, int src_stride, int pw,
                                   int ph, int unit_size, int ss_y, uint32_t ep_mask, int32_t* xqd_out, int64_t* err_out, uint8_t* best_ep, int* rounds_out) {
    SVT_HIP_ENTER(c);
    SvtHipSgrSearchPlane P = {d_dgd, stride, d_src, src_stride, pw, ph, unit_size, ss_y, ep_mask, xqd_out, err_out, best_ep};
    return svt_hip_sgr_search_units_picture(c, pix_bytes, bd, 1, &P, rounds_out);
}

int svt_hip_wiener_init_units_dev(SvtHipCtx* c, int win, int n_units, const int64_t* d_M, const int64_t* d_H, int16_t* d_unit_wiener, uint8_t* d_active, int8_t* d_status) {
    SVT_HIP_ENTER(c);
    if (!c || (win != 7 && win != 5 && win != 3) || n_units < 0 || (n_units && (!d_M || !d_H || !d_unit_wiener || !d_active || !d_status))) {
        if (c) c->err = "svt_hip_wiener_init_units_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    hipError_t e = (hipError_t)svt_hip_launch_wiener_init(c->stream, win, n_units, d_M, d_H, d_unit_wiener, d_active, d_status);
    if (e != hipSuccess) return fail(c, e, "wiener init launch");
    return SVT_HIP_OK;
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
        if (need > c->scratch_bytes) {   // the scratch may still be in use by work queued on ANY stream this context was pointed at
            HIPCHK(c, hipDeviceSynchronize());
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

int svt_hip_tf_subpel_frame_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* const d_src[3], const int src_stride[3], const void* const d_ref[3],
                                const int ref_stride[3], void* const d_pred[3], const int pred_stride[3], int mi_cols, int mi_rows, uint64_t th16, int tf_hp,
                                int tf_chroma, const SvtHipTfSubpelBlk* d_jobs, int n_jobs, SvtHipTfBlk64* d_blocks) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !src_stride || !d_ref || !ref_stride || !d_pred || !pred_stride || !d_jobs || !d_blocks || n_jobs < 0 || mi_cols <= 0 || mi_rows <= 0 ||
        (pix_bytes != 1 && pix_bytes != 2) || (pix_bytes == 1 && bd != 8) || (pix_bytes == 2 && bd != 8 && bd != 10)) {
        if (c) c->err = "svt_hip_tf_subpel_frame_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    for (int p = 0; p < (tf_chroma ? 3 : 1); p++)
        if (!d_src[p] || !d_ref[p] || !d_pred[p]) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_tf_subpel(c->stream, pix_bytes, bd, d_src, src_stride, d_ref, ref_stride, d_pred, pred_stride, mi_cols, mi_rows, th16,
                                                        tf_hp != 0, tf_chroma != 0, d_jobs, n_jobs, d_blocks);
    if (e != hipSuccess) return fail(c, e, "tf sub-pel launch");
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
int svt_hip_warp_compound_batch_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_ref, int width, int height, int stride, void* d_dst, int dst_stride,
                                    int ss_x, int ss_y, uint16_t* d_convbuf, const SvtHipWarpCompBlk* d_blks, int nblk) {
    SVT_HIP_ENTER(c);
    if (!c || nblk < 0 || (pix_bytes != 1 && pix_bytes != 2) || (pix_bytes == 1 && bd != 8) || (pix_bytes == 2 && bd != 8 && bd != 10 && bd != 12) ||
        (ss_x != 0 && ss_x != 1) || (ss_y != 0 && ss_y != 1)) {
        if (c) c->err = "svt_hip_warp_compound_batch_dev: bad argument";
        return SVT_HIP_ERR_BAD_ARG;
    }
    if (nblk == 0) return SVT_HIP_OK;
    if (!d_ref || !d_convbuf || !d_blks || width <= 0 || height <= 0) return SVT_HIP_ERR_BAD_ARG;   // d_dst may be NULL when no block averages
    hipError_t e = (hipError_t)svt_hip_launch_warp_compound(c->stream, pix_bytes, bd, d_ref, width, height, stride, d_dst, dst_stride, ss_x, ss_y, d_convbuf, d_blks, nblk);
    if (e != hipSuccess) return fail(c, e, "warp compound launch");
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


/* ------------------------------------------------------------------ per-call forms (percall.hip, cdef.hip, deblock.hip) */
int svt_hip_quantize_batch_dev(SvtHipCtx* c, const int32_t* d_coeff, int n_coeffs, int nblk, const SvtHipQuantParams* qp, const int16_t* d_iscan,
                               int32_t* d_qcoeff, int32_t* d_dqcoeff, uint16_t* d_eob) {
    SVT_HIP_ENTER(c);
    if (!c || !d_coeff || !qp || !d_iscan || !d_qcoeff || !d_dqcoeff || !d_eob || n_coeffs <= 0 || n_coeffs > 4096 || nblk < 0 || qp->variant < 0 || qp->variant > 3 ||
        qp->log_scale < 0 || qp->log_scale > 2)
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_quantize_blocks(c->stream, d_coeff, n_coeffs, nblk, qp, d_iscan, d_qcoeff, d_dqcoeff, d_eob);
    if (e != hipSuccess) return fail(c, e, "quantize launch");
    return SVT_HIP_OK;
}
int svt_hip_residual_dev(SvtHipCtx* c, int pix_bytes, const void* d_src, int src_stride, const void* d_pred, int pred_stride, int16_t* d_residual,
                         int residual_stride, int w, int h) {
    SVT_HIP_ENTER(c);
    if (!c || (pix_bytes != 1 && pix_bytes != 2) || w < 0 || h < 0) return SVT_HIP_ERR_BAD_ARG;
    if (w == 0 || h == 0) return SVT_HIP_OK;
    if (!d_src || !d_pred || !d_residual) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_residual(c->stream, pix_bytes, d_src, src_stride, d_pred, pred_stride, d_residual, residual_stride, w, h);
    if (e != hipSuccess) return fail(c, e, "residual launch");
    return SVT_HIP_OK;
}
int svt_hip_ext_all_sad_8x8_16x16_batch_dev(SvtHipCtx* c, const uint8_t* d_src, int src_stride, const uint8_t* d_ref, int ref_stride,
                                            const SvtHipExtSadJob* d_jobs, int n, uint32_t* d_state) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_ref || !d_jobs || !d_state || n < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_ext_all_sad(c->stream, d_src, src_stride, d_ref, ref_stride, d_jobs, n, d_state);
    if (e != hipSuccess) return fail(c, e, "ext all sad launch");
    return SVT_HIP_OK;
}
int svt_hip_ext_eight_sad_32x32_64x64_batch_dev(SvtHipCtx* c, const uint32_t* d_mv, int n, uint32_t* d_state) {
    SVT_HIP_ENTER(c);
    if (!c || !d_mv || !d_state || n < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_ext_eight_sad_32_64(c->stream, d_mv, n, d_state);
    if (e != hipSuccess) return fail(c, e, "ext eight sad launch");
    return SVT_HIP_OK;
}
int svt_hip_interm_var_four8x8_batch_dev(SvtHipCtx* c, const uint8_t* d_plane, int stride, const int32_t* d_offs, int n, uint64_t* d_mean, uint64_t* d_mean_sq) {
    SVT_HIP_ENTER(c);
    if (!c || !d_plane || !d_offs || !d_mean || !d_mean_sq || n < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_interm_var(c->stream, d_plane, stride, d_offs, n, d_mean, d_mean_sq);
    if (e != hipSuccess) return fail(c, e, "interm var launch");
    return SVT_HIP_OK;
}
int svt_hip_handle_transform64_batch_dev(SvtHipCtx* c, int tx_size, int32_t* d_coeff, int nblk, uint64_t* d_energy) {
    SVT_HIP_ENTER(c);
    if (!c || !d_coeff || !d_energy || nblk < 0 || (tx_size != 4 && tx_size != 11 && tx_size != 12 && tx_size != 17 && tx_size != 18)) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_handle_transform64(c->stream, tx_size, d_coeff, nblk, d_energy);
    if (e != hipSuccess) return fail(c, e, "handle transform64 launch");
    return SVT_HIP_OK;
}
int svt_hip_upsampled_pred_batch_dev(SvtHipCtx* c, const uint8_t* d_ref, int ref_stride, uint8_t* d_dst, const SvtHipUpsampledBlk* d_blks, int n) {
    SVT_HIP_ENTER(c);
    if (!c || !d_ref || !d_dst || !d_blks || n < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_upsampled_pred(c->stream, d_ref, ref_stride, d_dst, d_blks, n);
    if (e != hipSuccess) return fail(c, e, "upsampled pred launch");
    return SVT_HIP_OK;
}
int svt_hip_handle_transform64_n2n4_batch_dev(SvtHipCtx* c, int tx_size, int32_t* d_coeff, int nblk) {
    SVT_HIP_ENTER(c);
    if (!c || !d_coeff || nblk < 0 || (tx_size != 4 && tx_size != 11 && tx_size != 12 && tx_size != 17 && tx_size != 18)) return SVT_HIP_ERR_BAD_ARG;
    if (tx_size == 11 || tx_size == 17) return SVT_HIP_OK;   // 32x64 / 16x64: the reference's functions do nothing
    const int rows = tx_size == 18 ? 16 : 32;
    hipError_t e = (hipError_t)svt_hip_launch_repack64(c->stream, d_coeff, rows, 64 * (tx_size == 4 ? 64 : rows), nblk);
    if (e != hipSuccess) return fail(c, e, "handle transform64 N2 / N4 launch");
    return SVT_HIP_OK;
}
int svt_hip_diffwtd_mask_dev(SvtHipCtx* c, int elem_bytes, uint8_t* d_mask, const void* d_src0, int src0_stride, const void* d_src1, int src1_stride, int w, int h, int inverse,
                             int round, int shift) {
    SVT_HIP_ENTER(c);
    if (!c || !d_mask || !d_src0 || !d_src1 || w < 1 || h < 1 || (elem_bytes != 1 && elem_bytes != 2) || round < 0 || round > 15 || shift < 0 || shift > 8) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_diffwtd_mask(c->stream, elem_bytes, d_mask, d_src0, src0_stride, d_src1, src1_stride, w, h, inverse, round, shift);
    if (e != hipSuccess) return fail(c, e, "diffwtd mask launch");
    return SVT_HIP_OK;
}
int svt_hip_blend_a64_d16_dev(SvtHipCtx* c, int pix_bytes, int bd, void* d_dst, int dst_stride, const uint16_t* d_src0, int src0_stride, const uint16_t* d_src1, int src1_stride,
                              const uint8_t* d_mask, int mask_stride, int w, int h, int subw, int subh, int round_0, int round_1) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dst || !d_src0 || !d_src1 || !d_mask || w < 1 || h < 1 || !((pix_bytes == 1 && bd == 8) || (pix_bytes == 2 && bd >= 8 && bd <= 12)) || round_0 < 3 || round_0 > 5 ||
        round_1 < 1 || 14 - round_0 - round_1 < 0)
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_blend_d16(c->stream, pix_bytes, bd, d_dst, dst_stride, d_src0, src0_stride, d_src1, src1_stride, d_mask, mask_stride, w, h, subw != 0,
                                                        subh != 0, round_0, round_1);
    if (e != hipSuccess) return fail(c, e, "blend a64 d16 launch");
    return SVT_HIP_OK;
}
int svt_hip_jnt_convolve_dev(SvtHipCtx* c, int pix_bytes, int bd, int variant, const void* d_src, int src_stride, void* d_dst, int dst_stride, uint16_t* d_convbuf,
                             int convbuf_stride, const int16_t* d_taps, int w, int h, int round_0, int round_1, int do_average, int use_jnt_comp_avg, int fwd_offset,
                             int bck_offset) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_convbuf || !d_taps || (do_average && !d_dst) || w < 1 || h < 1 || variant < 0 || variant > 3 ||
        !((pix_bytes == 1 && bd == 8) || (pix_bytes == 2 && bd >= 8 && bd <= 12)) || round_0 < 3 || round_0 > 5 || round_1 < 1 || 14 - round_0 - round_1 < 0)
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_jnt_convolve(c->stream, pix_bytes, bd, variant, d_src, src_stride, d_dst, dst_stride, d_convbuf, convbuf_stride, d_taps, w, h,
                                                           round_0, round_1, do_average, use_jnt_comp_avg, fwd_offset, bck_offset);
    if (e != hipSuccess) return fail(c, e, "jnt convolve launch");
    return SVT_HIP_OK;
}
int svt_hip_block_mean_batch_dev(SvtHipCtx* c, const uint8_t* d_plane, int stride, const int32_t* d_offs, int n, int mode, int w, int h, uint64_t* d_out) {
    SVT_HIP_ENTER(c);
    if (!c || !d_plane || !d_offs || !d_out || n < 0 || (mode != 0 && mode != 1) || (mode == 0 && (w < 1 || h < 1))) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_block_mean(c->stream, d_plane, stride, d_offs, n, mode, w, h, d_out);
    if (e != hipSuccess) return fail(c, e, "block mean launch");
    return SVT_HIP_OK;
}
int svt_hip_ext_sad_16x16_batch_dev(SvtHipCtx* c, const uint8_t* d_src, int src_stride, const uint8_t* d_ref, int ref_stride, const SvtHipExtSadJob* d_jobs, int n,
                                    uint32_t* d_state) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_ref || !d_jobs || !d_state || n < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_ext_sad_16(c->stream, d_src, src_stride, d_ref, ref_stride, d_jobs, n, d_state);
    if (e != hipSuccess) return fail(c, e, "ext sad 16x16 launch");
    return SVT_HIP_OK;
}
int svt_hip_ext_sad_32x32_64x64_batch_dev(SvtHipCtx* c, uint32_t* d_state, const uint32_t* d_mv, int n) {
    SVT_HIP_ENTER(c);
    if (!c || !d_state || !d_mv || n < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_ext_sad_32_64(c->stream, d_state, d_mv, n);
    if (e != hipSuccess) return fail(c, e, "ext sad 32x32 / 64x64 launch");
    return SVT_HIP_OK;
}
int svt_hip_cdef_dist_dev(SvtHipCtx* c, int pix_bytes, const void* d_dst, int dstride, const void* d_src, const uint8_t* d_list, int n, int bw_log2, int bh_log2,
                          int coeff_shift, int pli, uint64_t* d_out) {
    SVT_HIP_ENTER(c);
    if (!c || !d_dst || !d_src || !d_list || !d_out || n < 0 || (pix_bytes != 1 && pix_bytes != 2) || (bw_log2 != 2 && bw_log2 != 3) || (bh_log2 != 2 && bh_log2 != 3) ||
        coeff_shift < 0 || coeff_shift > 4)
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_cdef_dist(c->stream, pix_bytes, d_dst, dstride, d_src, d_list, n, bw_log2, bh_log2, coeff_shift, pli, d_out);
    if (e != hipSuccess) return fail(c, e, "cdef dist launch");
    return SVT_HIP_OK;
}
int svt_hip_cdef_search_one_dual_dev(SvtHipCtx* c, const uint64_t* d_mse0, const uint64_t* d_mse1, int sb_count, int* d_lev0, int* d_lev1, int nb_strengths, int start_gi,
                                     int end_gi, uint64_t* d_work) {
    SVT_HIP_ENTER(c);
    if (!c || !d_mse0 || !d_mse1 || !d_lev0 || !d_lev1 || !d_work || sb_count < 0 || nb_strengths < 0 || nb_strengths > 7 || start_gi < 0 || end_gi > 64 || start_gi > end_gi)
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_search_one_dual(c->stream, d_mse0, d_mse1, sb_count, d_lev0, d_lev1, nb_strengths, start_gi, end_gi, d_work + 1 + 4096, d_work + 1,
                                                              d_work);
    if (e != hipSuccess) return fail(c, e, "search one dual launch");
    return SVT_HIP_OK;
}
int svt_hip_cdef_joint_strength_search_dev(SvtHipCtx* c, const uint64_t* d_mse0, const uint64_t* d_mse1, int sb_count, int* d_lev0, int* d_lev1, int nb_strengths, int start_gi,
                                           int end_gi, uint64_t* d_work) {
    SVT_HIP_ENTER(c);
    if (!c || !d_mse0 || !d_mse1 || !d_lev0 || !d_lev1 || !d_work || sb_count < 0 || nb_strengths < 1 || nb_strengths > 8 || start_gi < 0 || end_gi > 64 || start_gi > end_gi)
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_joint_strength_search(c->stream, d_mse0, d_mse1, sb_count, d_lev0, d_lev1, nb_strengths, start_gi, end_gi, d_work + 1 + 4096,
                                                                    d_work + 1, d_work);
    if (e != hipSuccess) return fail(c, e, "joint strength search launch");
    return SVT_HIP_OK;
}
int svt_hip_set_cdef_select_form(SvtHipCtx* c, int form) {
    if (!c || form < -1 || form > 1) return SVT_HIP_ERR_BAD_ARG;
    c->select_form = form;
    return SVT_HIP_OK;
}
int svt_hip_cdef_strength_select_dev(SvtHipCtx* c, const uint64_t* d_mse0, const uint64_t* d_mse1, int sb_count, int start_gi, int end_gi, void* d_state, size_t state_bytes) {
    SVT_HIP_ENTER(c);
    if (!c || !d_mse0 || !d_mse1 || !d_state || sb_count < 0 || start_gi < 0 || end_gi > 64 || start_gi > end_gi || state_bytes < svt_hip_joint_state_bytes())
        return SVT_HIP_ERR_BAD_ARG;
    return svt_hip_cdef_strength_select_multi_dev(c, 1, &d_mse0, &d_mse1, sb_count, start_gi, end_gi, &d_state, state_bytes);
}
int svt_hip_cdef_strength_select_multi_dev(SvtHipCtx* c, int n_pictures, const uint64_t* const* d_mse0, const uint64_t* const* d_mse1, int sb_count, int start_gi, int end_gi,
                                           void* const* d_states, size_t state_bytes) {
    SVT_HIP_ENTER(c);
    if (!c || n_pictures < 0 || !d_mse0 || !d_mse1 || !d_states || sb_count < 0 || start_gi < 0 || end_gi > 64 || start_gi > end_gi || state_bytes < svt_hip_joint_state_bytes())
        return SVT_HIP_ERR_BAD_ARG;
    for (int i = 0; i < n_pictures; i++)
        if (!d_mse0[i] || !d_mse1[i] || !d_states[i]) return SVT_HIP_ERR_BAD_ARG;
    hipStream_t sel = c->device < 64 ? g_sel_stream[c->device] : nullptr;
    const int resident = svt_hip_strength_select_is_resident(c->select_form, sb_count) && sel;
    if (!resident) {
        hipError_t e = (hipError_t)svt_hip_launch_strength_select_multi(c->stream, n_pictures, d_mse0, d_mse1, sb_count, start_gi, end_gi, d_states, 0);
        if (e != hipSuccess) return fail(c, e, "strength select (multi) launch");
        return SVT_HIP_OK;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    unsigned long long     cap_id = 0;
    if (hipStreamGetCaptureInfo(c->stream, &cap, &cap_id) != hipSuccess) cap = hipStreamCaptureStatusNone;
    if (cap == hipStreamCaptureStatusActive) {
        std::lock_guard<std::mutex> lk(g_sel_mutex);
        const int d = c->device;
        if (g_sel_cap_event[d] && g_sel_cap_id[d] == cap_id) HIPCHK(c, hipStreamWaitEvent(c->stream, g_sel_cap_event[d], 0));
        hipError_t e = (hipError_t)svt_hip_launch_strength_select_multi(c->stream, n_pictures, d_mse0, d_mse1, sb_count, start_gi, end_gi, d_states, 1);
        if (e != hipSuccess) return fail(c, e, "strength select (multi) launch");
        hipEvent_t ev = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(ev, c->stream));
        g_sel_cap_event[d] = ev; g_sel_cap_id[d] = cap_id;
        return SVT_HIP_OK;
    }
    HIPCHK(c, hipEventRecord(c->ev_sel_in, c->stream));
    {
        std::lock_guard<std::mutex> lk(g_sel_mutex);   // wait / launches / record of one call stay together on the shared stream
        HIPCHK(c, hipStreamWaitEvent(sel, c->ev_sel_in, 0));
        hipError_t e = (hipError_t)svt_hip_launch_strength_select_multi(sel, n_pictures, d_mse0, d_mse1, sb_count, start_gi, end_gi, d_states, 1);
        if (e != hipSuccess) return fail(c, e, "strength select (multi) launch");
        HIPCHK(c, hipEventRecord(c->ev_sel_out, sel));
    }
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_sel_out, 0));
    return SVT_HIP_OK;
}
int svt_hip_cdef_finish_dev(SvtHipCtx* c, const uint64_t* d_mse0, const uint64_t* d_mse1, int sb_count, const void* d_state, uint64_t lambda, const int32_t* d_sb_fb,
                            SvtHipCdefFinish* d_out, int32_t* d_sel_gi, uint8_t* d_fb_y, uint8_t* d_fb_uv) {
    SVT_HIP_ENTER(c);
    if (!c || !d_mse0 || !d_mse1 || !d_state || !d_out || sb_count < 0) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_cdef_finish(c->stream, d_mse0, d_mse1, sb_count, d_state, lambda, d_sb_fb, d_out, d_sel_gi, d_fb_y, d_fb_uv);
    if (e != hipSuccess) return fail(c, e, "cdef finish launch");
    return SVT_HIP_OK;
}
int svt_hip_sgr_flt_proj_dev(SvtHipCtx* c, int pix_bytes, const void* d_src, int src_stride, const void* d_dat, int dat_stride, const int32_t* d_flt0, int flt0_stride,
                             const int32_t* d_flt1, int flt1_stride, int w, int h, int r0, int r1, int mode, const int32_t* xq, int64_t* d_acc, int32_t* d_xq) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_dat || !d_acc || w < 1 || h < 1 || (pix_bytes != 1 && pix_bytes != 2) || (mode != 0 && mode != 1) || (r0 > 0 && !d_flt0) || (r1 > 0 && !d_flt1) ||
        (mode == 0 && !d_xq) || (mode == 1 && !xq))
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = hipMemsetAsync(d_acc, 0, 5 * sizeof(int64_t), c->stream);
    if (e != hipSuccess) return fail(c, e, "sgr flt proj clear");
    e = (hipError_t)svt_hip_launch_sgr_flt_proj(c->stream, pix_bytes, d_src, src_stride, d_dat, dat_stride, d_flt0, flt0_stride, d_flt1, flt1_stride, w, h, r0, r1, mode,
                                                mode ? xq[0] : 0, mode ? xq[1] : 0, (long long*)d_acc, d_xq);
    if (e != hipSuccess) return fail(c, e, "sgr flt proj launch");
    return SVT_HIP_OK;
}
int svt_hip_convolve8_dev(SvtHipCtx* c, int vert, const uint8_t* d_src, int src_stride, uint8_t* d_dst, int dst_stride, const int16_t* d_filters, int q0, int step_q4, int w,
                          int h) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_dst || !d_filters || w < 1 || h < 1 || q0 < 0 || q0 > 15 || step_q4 < 1 || step_q4 > 64) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_convolve8(c->stream, vert, d_src, src_stride, d_dst, dst_stride, d_filters, q0, step_q4, w, h);
    if (e != hipSuccess) return fail(c, e, "convolve8 launch");
    return SVT_HIP_OK;
}
int svt_hip_wiener_convolve_add_src_dev(SvtHipCtx* c, int pix_bytes, int bd, const void* d_src, int src_stride, void* d_dst, int dst_stride, const int16_t* d_taps, int w, int h,
                                        int round_0, int round_1) {
    SVT_HIP_ENTER(c);
    if (!c || !d_src || !d_dst || !d_taps || w < 1 || h < 1 || !((pix_bytes == 1 && bd == 8) || (pix_bytes == 2 && bd >= 8 && bd <= 12)) || round_0 < 1 || round_0 > 7 ||
        round_1 < 1 || round_1 > 14)
        return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_wiener_convolve(c->stream, pix_bytes, bd, d_src, src_stride, d_dst, dst_stride, d_taps, w, h, round_0, round_1);
    if (e != hipSuccess) return fail(c, e, "wiener convolve launch");
    return SVT_HIP_OK;
}
int svt_hip_cdef_find_dir_batch_dev(SvtHipCtx* c, const uint16_t* d_img, int stride, const int32_t* d_offs, int n, int coeff_shift, int32_t* d_dir, int32_t* d_var) {
    SVT_HIP_ENTER(c);
    if (!c || !d_img || !d_offs || !d_dir || !d_var || n < 0 || coeff_shift < 0 || coeff_shift > 4) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_cdef_find_dir_list(c->stream, d_img, d_offs, n, stride, coeff_shift, d_dir, d_var);
    if (e != hipSuccess) return fail(c, e, "cdef find dir launch");
    return SVT_HIP_OK;
}
int svt_hip_cdef_filter_block_batch_dev(SvtHipCtx* c, const uint16_t* d_in, int in_stride, const SvtHipCdefBlk* d_blks, int n, uint8_t* d_dst8, uint16_t* d_dst16,
                                        int dst_stride) {
    SVT_HIP_ENTER(c);
    if (!c || !d_in || !d_blks || n < 0 || (!d_dst8) == (!d_dst16)) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_cdef_filter_block_list(c->stream, d_in, in_stride, d_blks, n, d_dst8, d_dst16, dst_stride);
    if (e != hipSuccess) return fail(c, e, "cdef filter block launch");
    return SVT_HIP_OK;
}
int svt_hip_lpf_edges_batch_dev(SvtHipCtx* c, int pix_bytes, int bd, void* d_plane, int stride, const SvtHipLpfEdge* d_edges, int n) {
    SVT_HIP_ENTER(c);
    if (!c || !d_plane || !d_edges || n < 0 || (pix_bytes != 1 && pix_bytes != 2) || (bd != 8 && bd != 10) || (pix_bytes == 1 && bd != 8)) return SVT_HIP_ERR_BAD_ARG;
    hipError_t e = (hipError_t)svt_hip_launch_lpf_edge_list(c->stream, d_plane, pix_bytes, stride, bd, d_edges, n);
    if (e != hipSuccess) return fail(c, e, "lpf edges launch");
    return SVT_HIP_OK;
}

}  // extern "C"
