// rtcd_hip.cpp — per-call wrappers with the reference's RTCD signatures (include/svt_hip_rtcd.h).
// Every wrapper stages its host operands into a small device arena, launches the batched entry point of
// include/svt_hip.h with a batch of ONE, and copies the result back.  No arithmetic happens on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "../../include/svt_hip_rtcd.h"
#include "svt_hip_internal.h"
#include "interp_kernels.h"

namespace {

std::mutex  g_mu;
SvtHipCtx*  g_ctx = nullptr;
SvtHipRtcd  g_c;                 // the pointers that were installed before us (failure fallbacks)
const int16_t h_interp[6][16][8] = SVT_HIP_INTERP_TABLE;

// per-wrapper bookkeeping for svt_hip_rtcd_report: calls by wrapper function, delegations by dispatch-table name (both touched with g_mu held)
struct Tally { const char* key; long n, device_failures; };
Tally g_calls[512], g_deleg[512];
int   g_ncalls = 0, g_ndeleg = 0;
Tally* tally(Tally* tab, int* n, const char* key) {
    for (int i = 0; i < *n; i++)
        if (tab[i].key == key || !std::strcmp(tab[i].key, key)) return &tab[i];
    if (*n >= 512) return &tab[511];
    tab[*n] = Tally{key, 0, 0};
    return &tab[(*n)++];
}

// every wrapper: serialise on the one mutex and make the context's device current (the reference calls these pointers from many threads)
struct Guard {
    std::lock_guard<std::mutex> lk;
    Guard(const char* fn = __builtin_FUNCTION()) : lk(g_mu) {
        tally(g_calls, &g_ncalls, fn)->n++;
        if (g_ctx) { (void)hipSetDevice(svt_hip_ctx_device(g_ctx)); svt_hip_ctx_clear_error(g_ctx); }
    }
};
struct Slot { void* p = nullptr; size_t cap = 0; };
Slot g_slot[8];

void* dev(int i, size_t bytes) {   // device scratch, grown on demand (called with g_mu held)
    Slot& s = g_slot[i];
    bytes = (bytes + 255) & ~(size_t)255;
    if (s.cap < bytes) {
        if (s.p) (void)hipFree(s.p);
        s.p = nullptr; s.cap = 0;
        if (hipMalloc(&s.p, bytes + 256) != hipSuccess) return nullptr;
        s.cap = bytes;
    }
    return s.p;
}
hipStream_t stream() { return (hipStream_t)svt_hip_ctx_stream(g_ctx); }

// host rectangle -> packed device plane of pitch `dpitch` bytes
bool up2d(void* d, size_t dpitch, const void* h, size_t hpitch, size_t wbytes, size_t rows) {
    return hipMemcpy2DAsync(d, dpitch, h, hpitch, wbytes, rows, hipMemcpyHostToDevice, stream()) == hipSuccess;
}
bool down2d(void* h, size_t hpitch, const void* d, size_t dpitch, size_t wbytes, size_t rows) {
    return hipMemcpy2DAsync(h, hpitch, d, dpitch, wbytes, rows, hipMemcpyDeviceToHost, stream()) == hipSuccess &&
           hipStreamSynchronize(stream()) == hipSuccess;
}
bool up(void* d, const void* h, size_t bytes) { return hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream()) == hipSuccess; }
bool down(void* h, const void* d, size_t bytes) {
    return hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream()) == hipSuccess && hipStreamSynchronize(stream()) == hipSuccess;
}
[[noreturn]] void die(const char* what) {
    std::fprintf(stderr, "libsvtav1_hip: %s failed on the device and no fallback pointer was installed (%s)\n", what,
                 g_ctx ? svt_hip_last_error(g_ctx) : "no context");
    std::abort();
}
// A call the wrapper does not cover (an argument outside the batched entry point's domain) or a device failure is delegated to the pointer
// that was installed before.  Every delegation is counted per table entry (svt_hip_rtcd_report); a domain delegation is logged on its first
// occurrence only (a flood of identical lines would only hide it), a delegation that follows a failed device call is logged every time with
// the context's error string -- the two must not look alike.  Called with g_mu held (the wrapper's Guard).
void note_delegation(const char* name) {
    Tally* t = tally(g_deleg, &g_ndeleg, name);
    const char* err = g_ctx ? svt_hip_last_error(g_ctx) : "no context";
    const bool device = !g_ctx || (err && *err);
    if (device) {
        t->device_failures++;
        std::fprintf(stderr, "libsvtav1_hip: %s DEVICE FAILURE (%s), delegated to the installed C pointer\n", name, err);
    } else if (!t->n)
        std::fprintf(stderr, "libsvtav1_hip: %s delegated to the installed C pointer (first occurrence)\n", name);
    t->n++;
}
#define FALLBACK(name, member, ...)                                                                              \
    do {                                                                                                         \
        note_delegation(name);                                                                                   \
        if (!g_c.member) die(name);                                                                              \
        return g_c.member(__VA_ARGS__);                                                                          \
    } while (0)
inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------- SAD
void sad_loop_hip(uint8_t* src, uint32_t src_stride, uint8_t* ref, uint32_t ref_stride, uint32_t bh, uint32_t bw, uint64_t* best_sad,
                  int16_t* xc, int16_t* yc, uint32_t src_stride_raw, int16_t saw, int16_t sah) {
    Guard lk;
    bool ok = g_ctx && src_stride_raw && bw && bh && saw > 0 && sah > 0 && (ref_stride % src_stride_raw) == 0;
    // sub-SAD convention: the caller doubles both strides and halves the height; candidate rows advance by the raw stride
    const int step = ok ? (int)(ref_stride / src_stride_raw) : 1;
    ok = ok && (step == 1 || step == 2);
    if (ok) {
        const int srows = ((int)bh - 1) * step + 1, rrows = (sah - 1) + srows, rw = saw + (int)bw - 1;
        const size_t sp = rup(bw, 4), rp = rup((size_t)rw + 8, 4);
        uint8_t *d_src = (uint8_t*)dev(0, sp * srows + 64), *d_ref = (uint8_t*)dev(1, rp * (rrows + 1) + 64);
        SvtHipSadLoop job = {0, 0, 0, 0, (int16_t)bw, (int16_t)(bh * step), saw, sah, (int16_t)step, 0};
        void* d_job = dev(2, sizeof(job)); uint32_t* d_out = (uint32_t*)dev(3, 16);
        const uint32_t init[4] = {0xffffffu, 0, 0, 0};
        struct { uint32_t sad; int16_t xy[2]; uint32_t pad[2]; } res;
        ok = d_src && d_ref && d_job && d_out && up2d(d_src, sp * step, src, src_stride, bw, bh) &&
             up2d(d_ref, rp, ref, src_stride_raw, rw, rrows) && up(d_job, &job, sizeof(job)) && up(d_out, init, sizeof(init)) &&
             svt_hip_sad_loop_batch_dev(g_ctx, d_src, (int)sp, d_ref, (int)rp, (const SvtHipSadLoop*)d_job, 1, d_out, (int16_t*)(d_out + 1)) == 0 &&
             down(&res, d_out, 8);
        if (ok) {
            *best_sad = res.sad;   // EbComputeSAD_C.c:73: 0xffffff when nothing beat it (centres untouched)
            if (res.sad != 0xffffffu) { *xc = res.xy[0]; *yc = res.xy[1]; }
            return;
        }
    }
    {
        static bool once = false;   // which call shape was not covered (diagnostic, printed once)
        if (!once) {
            once = true;
            std::fprintf(stderr, "libsvtav1_hip: svt_sad_loop_kernel(src_stride %u ref_stride %u h %u w %u raw %u sa %d x %d) not covered: %s\n", src_stride, ref_stride, bh, bw,
                         src_stride_raw, (int)saw, (int)sah, g_ctx ? svt_hip_last_error(g_ctx) : "no context");
        }
    }
    FALLBACK("svt_sad_loop_kernel", svt_sad_loop_kernel, src, src_stride, ref, ref_stride, bh, bw, best_sad, xc, yc, src_stride_raw, saw, sah);
}

bool pair_stage(int pix_bytes, const void* a, int a_stride, const void* b, int b_stride, int w, int h, void** da, void** db, void** dj, uint32_t** dout,
                size_t* pitch) {
    const size_t p = rup((size_t)w * pix_bytes, 4);
    *pitch = p;
    *da = dev(0, p * h + 64); *db = dev(1, p * h + 64); *dj = dev(2, sizeof(SvtHipBlkPair)); *dout = (uint32_t*)dev(3, 16);
    SvtHipBlkPair job = {0, 0, 0, 0, (uint16_t)w, (uint16_t)h};
    return *da && *db && *dj && *dout && up2d(*da, p, a, (size_t)a_stride * pix_bytes, (size_t)w * pix_bytes, h) &&
           up2d(*db, p, b, (size_t)b_stride * pix_bytes, (size_t)w * pix_bytes, h) && up(*dj, &job, sizeof(job));
}
bool sad_generic(const uint8_t* a, int a_stride, const uint8_t* b, int b_stride, int w, int h, uint32_t* out) {
    void *da, *db, *dj; uint32_t* dout; size_t p;
    return g_ctx && pair_stage(1, a, a_stride, b, b_stride, w, h, &da, &db, &dj, &dout, &p) &&
           svt_hip_block_sad_batch_dev(g_ctx, 1, da, (int)p, db, (int)p, (const SvtHipBlkPair*)dj, 1, dout) == 0 && down(out, dout, 4);
}
uint32_t nxm_sad_hip(const uint8_t* src, uint32_t src_stride, const uint8_t* ref, uint32_t ref_stride, uint32_t height, uint32_t width) {
    Guard lk;
    uint32_t r;
    if (sad_generic(src, (int)src_stride, ref, (int)ref_stride, (int)width, (int)height, &r)) return r;
    FALLBACK("svt_nxm_sad_kernel", svt_nxm_sad_kernel, src, src_stride, ref, ref_stride, height, width);
}
bool var_generic(int pix_bytes, int bd, const void* a, int a_stride, const void* b, int b_stride, int w, int h, unsigned* var, unsigned* sse) {
    void *da, *db, *dj; uint32_t* dout; size_t p; uint32_t res[2];
    if (!(g_ctx && pair_stage(pix_bytes, a, a_stride, b, b_stride, w, h, &da, &db, &dj, &dout, &p) &&
          svt_hip_block_variance_batch_dev(g_ctx, pix_bytes, bd, da, (int)(p / pix_bytes), db, (int)(p / pix_bytes), (const SvtHipBlkPair*)dj, 1, dout, dout + 1) == 0 &&
          down(res, dout, 8)))
        return false;
    *var = res[0]; *sse = res[1];
    return true;
}
template <int IDX, int W, int H> uint32_t sad_wxh_hip(const uint8_t* a, int as, const uint8_t* b, int bs) {
    Guard lk;
    uint32_t r;
    if (sad_generic(a, as, b, bs, W, H, &r)) return r;
    FALLBACK("svt_aom_sadWxH", svt_aom_sad[IDX], a, as, b, bs);
}
template <int IDX, int W, int H> unsigned var_wxh_hip(const uint8_t* a, int as, const uint8_t* b, int bs, unsigned* sse) {
    Guard lk;
    unsigned v;
    if (var_generic(1, 8, a, as, b, bs, W, H, &v, sse)) return v;
    FALLBACK("svt_aom_varianceWxH", svt_aom_variance[IDX], a, as, b, bs, sse);
}
template <int IDX, int W, int H> unsigned var10_wxh_hip(const uint8_t* a8, int as, const uint8_t* b8, int bs, unsigned* sse) {
    Guard lk;
    unsigned v;   // CONVERT_TO_SHORTPTR (Common/Codec/EbDefinitions.h): the byte pointer carries the uint16_t address >> 1
    if (var_generic(2, 10, (const void*)((uintptr_t)a8 << 1), as, (const void*)((uintptr_t)b8 << 1), bs, W, H, &v, sse)) return v;
    FALLBACK("svt_aom_highbd_10_varianceWxH", svt_aom_highbd_10_variance[IDX], a8, as, b8, bs, sse);
}

// ------------------------------------------------------------------------------------ convolve
int bank_of(const SvtHipInterpFilterParams* f) {   // the kernel indexes its own normative tables; find the one the caller passed
    if (!f || !f->filter_ptr) return -1;
    for (int b = 0; b < 6; b++)
        if (!std::memcmp(f->filter_ptr, h_interp[b], sizeof(h_interp[b]))) return b;
    return -1;
}
bool conv_generic(int pix_bytes, int bd, const void* src, int src_stride, void* dst, int dst_stride, int w, int h, const SvtHipInterpFilterParams* fx,
                  const SvtHipInterpFilterParams* fy, int sx, int sy, const SvtHipConvolveParams* cp) {
    const int bx = bank_of(fx), by = bank_of(fy);
    if (!g_ctx || bx < 0 || by < 0 || w < 2 || h < 2 || w > 128 || h > 128 || (cp && (cp->is_compound || cp->do_average))) return false;
    // reference window: 3 samples left / above, 4 right / below, plus the slack the batched kernel's tile loads may touch
    const int padl = 3, padr = 4 + 16, rw = w + padl + padr, rh = h + padl + padr;
    const size_t rp = rup((size_t)rw * pix_bytes, 4), dp = rup((size_t)w * pix_bytes, 4);
    uint8_t* d_ref = (uint8_t*)dev(0, rp * rh + 64); uint8_t* d_dst = (uint8_t*)dev(1, dp * h + 64); void* d_job = dev(2, sizeof(SvtHipConvBlk));
    if (!d_ref || !d_dst || !d_job || hipMemsetAsync(d_ref, 0, rp * rh, stream()) != hipSuccess) return false;
    // only the rows / columns the C kernel reads are copied (it reads [-3, +4] around the block: EbInterPrediction.c:349-393)
    const uint8_t* s0 = (const uint8_t*)src - ((size_t)padl * src_stride + padl) * pix_bytes;
    SvtHipConvBlk job = {padl, padl, 0, 0, (uint8_t)w, (uint8_t)h, (uint8_t)bx, (uint8_t)by, (uint8_t)(sx & 15), (uint8_t)(sy & 15), 0, 0};
    return up2d(d_ref, rp, s0, (size_t)src_stride * pix_bytes, (size_t)(w + 7) * pix_bytes, h + 7) && up(d_job, &job, sizeof(job)) &&
           svt_hip_subpel_predict_batch_dev(g_ctx, pix_bytes, bd, d_ref, (int)(rp / pix_bytes), d_dst, (int)(dp / pix_bytes), (const SvtHipConvBlk*)d_job, 1) == 0 &&
           down2d(dst, (size_t)dst_stride * pix_bytes, d_dst, dp, (size_t)w * pix_bytes, h);
}
#define CONV_WRAPPER(NAME, MEMBER, SXM, SYM)                                                                                                         \
    void NAME(const uint8_t* src, int32_t ss, uint8_t* dst, int32_t ds, int32_t w, int32_t h, SvtHipInterpFilterParams* fx, SvtHipInterpFilterParams* fy, \
              const int32_t sx, const int32_t sy, SvtHipConvolveParams* cp) {                                                                        \
        Guard lk;                                                                                                        \
        if (conv_generic(1, 8, src, ss, dst, ds, w, h, fx, fy, (SXM) ? sx : 0, (SYM) ? sy : 0, cp)) return;                                          \
        FALLBACK("svt_av1_" #MEMBER, svt_av1_##MEMBER, src, ss, dst, ds, w, h, fx, fy, sx, sy, cp);                                                                       \
    }                                                                                                                                                \
    void NAME##_hbd(const uint16_t* src, int32_t ss, uint16_t* dst, int32_t ds, int32_t w, int32_t h, const SvtHipInterpFilterParams* fx,             \
                    const SvtHipInterpFilterParams* fy, const int32_t sx, const int32_t sy, SvtHipConvolveParams* cp, int32_t bd) {                  \
        Guard lk;                                                                                                        \
        if ((bd == 8 || bd == 10) && conv_generic(2, bd, src, ss, dst, ds, w, h, fx, fy, (SXM) ? sx : 0, (SYM) ? sy : 0, cp)) return;                 \
        FALLBACK("highbd " #MEMBER, svt_av1_highbd_##MEMBER, src, ss, dst, ds, w, h, fx, fy, sx, sy, cp, bd);                                       \
    }
// each reference function ignores the phase of the direction it does not filter (EbInterPrediction.c:395-470)
CONV_WRAPPER(conv_2d_hip, convolve_2d_sr, 1, 1)
CONV_WRAPPER(conv_x_hip, convolve_x_sr, 1, 0)
CONV_WRAPPER(conv_y_hip, convolve_y_sr, 0, 1)
CONV_WRAPPER(conv_copy_hip, convolve_2d_copy_sr, 0, 0)

// ----------------------------------------------------------------------------------- transforms
constexpr int kTxW[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
constexpr int kTxH[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};

bool fwd_generic(int ts, const int16_t* in, int32_t* out, int stride, int tx_type, int shape = 0) {
    const int w = kTxW[ts], h = kTxH[ts];
    if (!g_ctx) return false;
    // the batched entry point forms the residual itself; any residual r is src - pred with src = max(r, 0), pred = max(-r, 0)
    static thread_local uint16_t hs[32 * 32], hp[32 * 32];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) { const int r = in[y * stride + x]; hs[y * w + x] = (uint16_t)(r > 0 ? r : 0); hp[y * w + x] = (uint16_t)(r < 0 ? -r : 0); }
    uint16_t *d_s = (uint16_t*)dev(0, sizeof(hs)), *d_p = (uint16_t*)dev(1, sizeof(hp)); uint32_t* d_desc = (uint32_t*)dev(2, 16); int32_t* d_c = (int32_t*)dev(3, 32 * 32 * 4);
    const uint32_t desc = SVT_HIP_TX_DESC(0, 0, tx_type);
    SvtHipQuantParams qp = {};
    qp.coeff_shape = shape;   // N2 / N4 transform families = the default transform restricted to the top-left corner
    return d_s && d_p && d_desc && d_c && up(d_s, hs, (size_t)w * h * 2) && up(d_p, hp, (size_t)w * h * 2) && up(d_desc, &desc, 4) &&
           svt_hip_fwd_txfm_quant_batch_dev(g_ctx, ts, 2, d_s, w, d_p, w, d_desc, 1, &qp, nullptr, d_c, nullptr, nullptr, nullptr, nullptr, nullptr) == 0 &&
           down(out, d_c, (size_t)w * h * 4);
}
template <int SLOT, int TS> void fwd_hip(int16_t* in, int32_t* out, uint32_t stride, uint8_t tt, uint8_t bd) {
    Guard lk;
    if (fwd_generic(TS, in, out, (int)stride, tt)) return;
    FALLBACK("svt_av1_fwd_txfm2d_WxH", svt_av1_fwd_txfm2d[SLOT], in, out, stride, tt, bd);
}
template <int SLOT, int TS> void fwd_n2_hip(int16_t* in, int32_t* out, uint32_t stride, uint8_t tt, uint8_t bd) {
    Guard lk;
    if (fwd_generic(TS, in, out, (int)stride, tt, 1)) return;
    FALLBACK("svt_av1_fwd_txfm2d_WxH_N2", svt_av1_fwd_txfm2d_N2[SLOT], in, out, stride, tt, bd);
}
template <int SLOT, int TS> void fwd_n4_hip(int16_t* in, int32_t* out, uint32_t stride, uint8_t tt, uint8_t bd) {
    Guard lk;
    if (fwd_generic(TS, in, out, (int)stride, tt, 2)) return;
    FALLBACK("svt_av1_fwd_txfm2d_WxH_N4", svt_av1_fwd_txfm2d_N4[SLOT], in, out, stride, tt, bd);
}
bool inv_generic(int ts, const int32_t* in, void* out_r, int stride_r, void* out_w, int stride_w, int tx_type, int bd, int pb = 2) {
    const int w = kTxW[ts], h = kTxH[ts], kw = w < 32 ? w : 32, kh = h < 32 ? h : 32;
    if (!g_ctx || (bd != 8 && bd != 10) || (pb == 1 && bd != 8)) return false;
    const size_t p = (size_t)w * pb, pitch = rup(p, 4);
    int32_t* d_c = (int32_t*)dev(0, (size_t)kw * kh * 4); uint8_t *d_r = (uint8_t*)dev(1, pitch * h), *d_w = (uint8_t*)dev(3, pitch * h); uint32_t* d_desc = (uint32_t*)dev(2, 16);
    const uint32_t desc = SVT_HIP_TX_DESC(0, 0, tx_type);
    return d_c && d_r && d_w && d_desc && up(d_c, in, (size_t)kw * kh * 4) && up2d(d_r, pitch, out_r, (size_t)stride_r * pb, p, h) && up(d_desc, &desc, 4) &&
           svt_hip_inv_txfm_add_batch_dev(g_ctx, ts, pb, bd, d_c, d_r, (int)(pitch / pb), d_w, (int)(pitch / pb), d_desc, 1) == 0 &&
           down2d(out_w, (size_t)stride_w * pb, d_w, pitch, p, h);
}
// the lossless branch of svt_av1_highbd_inv_txfm_add_4x4 (EbInvTransforms.c:2870-2882): TX_4X4 only, the Walsh-Hadamard pair chosen by eob
bool iwht_generic(const int32_t* in, const uint8_t* out_r, int stride_r, uint8_t* out_w, int stride_w, int eob, int bd) {
    if (!g_ctx || bd != 8) return false;
    int32_t* d_c = (int32_t*)dev(0, 64); uint8_t *d_r = (uint8_t*)dev(1, 16), *d_w = (uint8_t*)dev(3, 16); uint32_t* d_desc = (uint32_t*)dev(2, 16);
    const uint32_t desc[2] = {SVT_HIP_TX_DESC(0, 0, 0), (uint32_t)(eob > 1 ? 2 : 1)};   // the eob (uint16_t) rides behind the descriptor
    return d_c && d_r && d_w && d_desc && up(d_c, in, 64) && up2d(d_r, 4, out_r, (size_t)stride_r, 4, 4) && up(d_desc, desc, 8) &&
           svt_hip_iwht4x4_add_batch_dev(g_ctx, 1, bd, d_c, (const uint16_t*)(d_desc + 1), d_r, 4, d_w, 4, d_desc, 1) == 0 && down2d(out_w, (size_t)stride_w, d_w, 4, 4, 4);
}
void inv_txfm_add_hip(const int32_t* dq, uint8_t* dst_r, int32_t sr, uint8_t* dst_w, int32_t sw, const SvtHipTxfmParam* tp) {
    Guard lk;   // svt_av1_inv_txfm_add_c (EbInvTransforms.c:3302): 8-bit destination through the high bit-depth inverse at bd = tp->bd
    if (tp && tp->lossless && tp->tx_size == 0 && iwht_generic(dq, dst_r, sr, dst_w, sw, tp->eob, tp->bd)) return;
    if (tp && !tp->lossless && tp->tx_size < 19 && tp->tx_type < 16 && inv_generic(tp->tx_size, dq, dst_r, sr, dst_w, sw, tp->tx_type, tp->bd, 1)) return;
    FALLBACK("svt_av1_inv_txfm_add", svt_av1_inv_txfm_add, dq, dst_r, sr, dst_w, sw, tp);
}
template <int SLOT, int TS> void inv_sq_hip(const int32_t* in, uint16_t* r, int32_t sr, uint16_t* wv, int32_t sw, uint8_t tt, int32_t bd) {
    Guard lk;
    if (inv_generic(TS, in, r, sr, wv, sw, tt, bd)) return;
    FALLBACK("svt_av1_inv_txfm2d_add (square)", svt_av1_inv_txfm2d_add_sq[SLOT], in, r, sr, wv, sw, tt, bd);
}
void inv_rect_hip(const int32_t* in, uint16_t* r, int32_t sr, uint16_t* wv, int32_t sw, uint8_t tt, uint8_t ts, int32_t eob, int32_t bd) {
    Guard lk;
    if (ts < 19 && inv_generic(ts, in, r, sr, wv, sw, tt, bd)) return;
    FALLBACK("svt_av1_inv_txfm2d_add (rect)", svt_av1_inv_txfm2d_add_rect, in, r, sr, wv, sw, tt, ts, eob, bd);
}
void inv_rect4_hip(const int32_t* in, uint16_t* r, int32_t sr, uint16_t* wv, int32_t sw, uint8_t tt, uint8_t ts, int32_t bd) {
    Guard lk;
    if (ts < 19 && inv_generic(ts, in, r, sr, wv, sw, tt, bd)) return;
    FALLBACK("svt_av1_inv_txfm2d_add (4xN)", svt_av1_inv_txfm2d_add_rect4, in, r, sr, wv, sw, tt, ts, bd);
}

// --------------------------------------------------------------------------------- self-guided
bool sgr_stage(const uint8_t* dat8, int highbd, int w, int h, int stride, void** d_in, size_t* pitch_px) {
    const int pb = highbd ? 2 : 1;
    const uint8_t* base = highbd ? (const uint8_t*)((uintptr_t)dat8 << 1) : dat8;
    const size_t p = rup((size_t)(w + 6) * pb, 4);
    *pitch_px = p / pb;
    *d_in = dev(0, p * (h + 6) + 64);
    return *d_in && up2d(*d_in, p, base - ((size_t)3 * stride + 3) * pb, (size_t)stride * pb, (size_t)(w + 6) * pb, h + 6);
}
void sgr_filter_hip(const uint8_t* dgd8, int32_t w, int32_t h, int32_t stride, int32_t* flt0, int32_t* flt1, int32_t fs, int32_t ep, int32_t bd, int32_t highbd) {
    Guard lk;
    void* d_in; size_t pp; const int pb = highbd ? 2 : 1;
    bool ok = g_ctx && w > 0 && h > 0 && w <= 64 && h <= 64 && ep >= 0 && ep < 16 && (bd == 8 || bd == 10) && sgr_stage(dgd8, highbd, w, h, stride, &d_in, &pp);
    if (ok) {
        int32_t *d_f0 = (int32_t*)dev(1, (size_t)w * h * 4), *d_f1 = (int32_t*)dev(3, (size_t)w * h * 4);
        ok = d_f0 && d_f1 && svt_hip_sgr_filter_plane_dev(g_ctx, pb, bd, (uint8_t*)d_in + (3 * pp + 3) * pb, (int)pp, w, h, ep, d_f0, d_f1, w) == 0;
        // eb_sgr_params: sets 10-13 have r0 = 0 (flt0 untouched), 14-15 r1 = 0 (flt1 untouched), EbRestoration.c:136-153
        if (ok && !(ep >= 10 && ep <= 13)) ok = down2d(flt0, (size_t)fs * 4, d_f0, (size_t)w * 4, (size_t)w * 4, h);
        if (ok && ep < 14) ok = down2d(flt1, (size_t)fs * 4, d_f1, (size_t)w * 4, (size_t)w * 4, h);
        if (ok) return;
    }
    FALLBACK("svt_av1_selfguided_restoration", svt_av1_selfguided_restoration, dgd8, w, h, stride, flt0, flt1, fs, ep, bd, highbd);
}
void sgr_apply_hip(const uint8_t* dat8, int32_t w, int32_t h, int32_t stride, int32_t eps, const int32_t* xqd, uint8_t* dst8, int32_t dst_stride, int32_t* tmpbuf,
                   int32_t bd, int32_t highbd) {
    Guard lk;
    void* d_in; size_t pp; const int pb = highbd ? 2 : 1;
    // one call = one processing unit (<= 64 x 64) of one stripe: a single restoration unit, rows attributed from the unit's own origin
    bool ok = g_ctx && w > 0 && h > 0 && w <= 64 && h <= 64 && eps >= 0 && eps < 16 && (bd == 8 || bd == 10) && sgr_stage(dat8, highbd, w, h, stride, &d_in, &pp);
    if (ok) {
        const size_t dp = rup((size_t)w * pb, 4);
        uint8_t* d_dst = (uint8_t*)dev(1, dp * h + 64); uint8_t* d_ep = (uint8_t*)dev(2, 16); int32_t* d_xqd = (int32_t*)dev(3, 16);
        const uint8_t ep8 = (uint8_t)eps;
        ok = d_dst && d_ep && d_xqd && up(d_ep, &ep8, 1) && up(d_xqd, xqd, 8) &&
             svt_hip_sgr_apply_plane_dev(g_ctx, pb, bd, (uint8_t*)d_in + (3 * pp + 3) * pb, (int)pp, d_dst, (int)(dp / pb), w, h, 64, 0, nullptr, 0, d_ep, d_xqd) == 0 &&
             down2d(highbd ? (uint8_t*)((uintptr_t)dst8 << 1) : dst8, (size_t)dst_stride * pb, d_dst, dp, (size_t)w * pb, h);
        if (ok) return;
    }
    FALLBACK("svt_apply_selfguided_restoration", svt_apply_selfguided_restoration, dat8, w, h, stride, eps, xqd, dst8, dst_stride, tmpbuf, bd, highbd);
}

// ----------------------------------------------------------------------------------- OBMC costs
bool obmc_generic(const uint8_t* pre, int pre_stride, const int32_t* wsrc, const int32_t* mask, int w, int h, int xo, int yo, uint32_t res[3]) {
    if (!g_ctx) return false;
    const size_t pp = rup((size_t)w + 1, 4);
    uint8_t* d_pre = (uint8_t*)dev(0, pp * (h + 1) + 64); int32_t* d_w = (int32_t*)dev(1, (size_t)w * h * 4); int32_t* d_m = (int32_t*)dev(2, (size_t)w * h * 4);
    void* d_job = dev(3, sizeof(SvtHipObmcBlk)); uint32_t* d_out = (uint32_t*)dev(4, 16);
    if (!d_pre || !d_w || !d_m || !d_job || !d_out) return false;
    SvtHipObmcBlk job = {0, 0, (uint8_t)w, (uint8_t)h, (uint8_t)xo, (uint8_t)yo, 0};
    // the sub-pixel variance reads one extra column / row (the 2-tap bilinear passes); the plain functions must not touch them
    const int ew = (xo || yo) ? w + 1 : w, eh = (xo || yo) ? h + 1 : h;
    return hipMemsetAsync(d_pre, 0, pp * (h + 1), stream()) == hipSuccess && up2d(d_pre, pp, pre, (size_t)pre_stride, (size_t)ew, eh) && up(d_w, wsrc, (size_t)w * h * 4) &&
           up(d_m, mask, (size_t)w * h * 4) && up(d_job, &job, sizeof(job)) &&
           svt_hip_obmc_cost_batch_dev(g_ctx, d_pre, (int)pp, d_w, d_m, (const SvtHipObmcBlk*)d_job, 1, d_out) == 0 && down(res, d_out, 12);
}
template <int IDX, int W, int H> unsigned obmc_sad_hip(const uint8_t* pre, int ps, const int32_t* wsrc, const int32_t* mask) {
    Guard lk;
    uint32_t r[3];
    if (obmc_generic(pre, ps, wsrc, mask, W, H, 0, 0, r)) return r[0];
    FALLBACK("svt_aom_obmc_sadWxH", svt_aom_obmc_sad[IDX], pre, ps, wsrc, mask);
}
template <int IDX, int W, int H> unsigned obmc_var_hip(const uint8_t* pre, int ps, const int32_t* wsrc, const int32_t* mask, unsigned* sse) {
    Guard lk;
    uint32_t r[3];
    if (obmc_generic(pre, ps, wsrc, mask, W, H, 0, 0, r)) { *sse = r[1]; return r[2]; }
    FALLBACK("svt_aom_obmc_varianceWxH", svt_aom_obmc_variance[IDX], pre, ps, wsrc, mask, sse);
}
template <int IDX, int W, int H> unsigned obmc_subvar_hip(const uint8_t* pre, int ps, int xo, int yo, const int32_t* wsrc, const int32_t* mask, unsigned* sse) {
    Guard lk;
    uint32_t r[3];
    if (xo >= 0 && xo < 8 && yo >= 0 && yo < 8 && obmc_generic(pre, ps, wsrc, mask, W, H, xo, yo, r)) { *sse = r[1]; return r[2]; }
    FALLBACK("svt_aom_obmc_sub_pixel_varianceWxH", svt_aom_obmc_sub_pixel_variance[IDX], pre, ps, xo, yo, wsrc, mask, sse);
}

// ----------------------------------------------------------------------------------- pixel-domain blends
bool blend_generic(int pix_bytes, void* dst, int dst_stride, const void* s0, int s0_stride, const void* s1, int s1_stride, const uint8_t* mask, int mask_stride, int w, int h,
                   int mode, int subw, int subh) {
    if (!g_ctx || w < 1 || h < 1 || w > 128 || h > 128) return false;
    const size_t p = rup((size_t)w * pix_bytes, 4);
    const int mw = mode == 0 ? (w << subw) : (mode == 1 ? w : h), mh = mode == 0 ? (h << subh) : 1;
    const size_t mp = rup((size_t)mw, 4);
    uint8_t *d0 = (uint8_t*)dev(0, p * h), *d1 = (uint8_t*)dev(1, p * h), *dd = (uint8_t*)dev(2, p * h), *dm = (uint8_t*)dev(3, mp * mh);
    void* d_job = dev(4, sizeof(SvtHipBlendBlk));
    if (!d0 || !d1 || !dd || !dm || !d_job) return false;
    SvtHipBlendBlk job = {0, 0, 0, 0, 0, 0, (uint8_t)w, (uint8_t)h, (uint8_t)mode, (uint8_t)subw, (uint8_t)subh, {0, 0, 0}, 0, (int32_t)mp};
    return up2d(d0, p, s0, (size_t)s0_stride * pix_bytes, (size_t)w * pix_bytes, h) && up2d(d1, p, s1, (size_t)s1_stride * pix_bytes, (size_t)w * pix_bytes, h) &&
           up2d(dm, mp, mask, mode == 0 ? (size_t)mask_stride : (size_t)mw, (size_t)mw, mh) && up(d_job, &job, sizeof(job)) &&
           svt_hip_blend_a64_batch_dev(g_ctx, pix_bytes, d0, (int)(p / pix_bytes), d1, (int)(p / pix_bytes), dd, (int)(p / pix_bytes), dm, (const SvtHipBlendBlk*)d_job, 1) == 0 &&
           down2d(dst, (size_t)dst_stride * pix_bytes, dd, p, (size_t)w * pix_bytes, h);
}
void blend_mask_hip(uint8_t* dst, uint32_t ds, const uint8_t* s0, uint32_t s0s, const uint8_t* s1, uint32_t s1s, const uint8_t* mask, uint32_t ms, int w, int h, int subw, int subh) {
    Guard lk;
    if (blend_generic(1, dst, (int)ds, s0, (int)s0s, s1, (int)s1s, mask, (int)ms, w, h, 0, subw, subh)) return;
    FALLBACK("svt_aom_blend_a64_mask", svt_aom_blend_a64_mask, dst, ds, s0, s0s, s1, s1s, mask, ms, w, h, subw, subh);
}
void blend_hmask_hip(uint8_t* dst, uint32_t ds, const uint8_t* s0, uint32_t s0s, const uint8_t* s1, uint32_t s1s, const uint8_t* mask, int w, int h) {
    Guard lk;
    if (blend_generic(1, dst, (int)ds, s0, (int)s0s, s1, (int)s1s, mask, 0, w, h, 1, 0, 0)) return;
    FALLBACK("svt_aom_blend_a64_hmask", svt_aom_blend_a64_hmask, dst, ds, s0, s0s, s1, s1s, mask, w, h);
}
void blend_vmask_hip(uint8_t* dst, uint32_t ds, const uint8_t* s0, uint32_t s0s, const uint8_t* s1, uint32_t s1s, const uint8_t* mask, int w, int h) {
    Guard lk;
    if (blend_generic(1, dst, (int)ds, s0, (int)s0s, s1, (int)s1s, mask, 0, w, h, 2, 0, 0)) return;
    FALLBACK("svt_aom_blend_a64_vmask", svt_aom_blend_a64_vmask, dst, ds, s0, s0s, s1, s1s, mask, w, h);
}
void blend_mask_hbd_hip(uint8_t* dst, uint32_t ds, const uint8_t* s0, uint32_t s0s, const uint8_t* s1, uint32_t s1s, const uint8_t* mask, uint32_t ms, int w, int h, int subw, int subh, int bd) {
    Guard lk;
    if (blend_generic(2, dst, (int)ds, s0, (int)s0s, s1, (int)s1s, mask, (int)ms, w, h, 0, subw, subh)) return;
    FALLBACK("svt_aom_highbd_blend_a64_mask", svt_aom_highbd_blend_a64_mask, dst, ds, s0, s0s, s1, s1s, mask, ms, w, h, subw, subh, bd);
}
void blend_hmask_hbd_hip(uint8_t* dst, uint32_t ds, const uint8_t* s0, uint32_t s0s, const uint8_t* s1, uint32_t s1s, const uint8_t* mask, int w, int h, int bd) {
    Guard lk;
    if (blend_generic(2, dst, (int)ds, s0, (int)s0s, s1, (int)s1s, mask, 0, w, h, 1, 0, 0)) return;
    FALLBACK("svt_aom_highbd_blend_a64_hmask_8bit", svt_aom_highbd_blend_a64_hmask_8bit, dst, ds, s0, s0s, s1, s1s, mask, w, h, bd);
}
void blend_vmask_hbd_hip(uint8_t* dst, uint32_t ds, const uint8_t* s0, uint32_t s0s, const uint8_t* s1, uint32_t s1s, const uint8_t* mask, int w, int h, int bd) {
    Guard lk;
    if (blend_generic(2, dst, (int)ds, s0, (int)s0s, s1, (int)s1s, mask, 0, w, h, 2, 0, 0)) return;
    FALLBACK("svt_aom_highbd_blend_a64_vmask_8bit", svt_aom_highbd_blend_a64_vmask_8bit, dst, ds, s0, s0s, s1, s1s, mask, w, h, bd);
}

// ----------------------------------------------------------------------------------- warped prediction
bool warp_generic(int pix_bytes, int bd, const int32_t* mat, const void* ref, int width, int height, int stride, void* pred, int p_col, int p_row, int p_width, int p_height,
                  int p_stride, int ss_x, int ss_y, const SvtHipConvolveParams* cp, int alpha, int beta, int gamma, int delta) {
    const bool comp = cp && cp->is_compound;
    // the compound branches use av1's fixed rounding for compound prediction (round_0 3, 5 at 12 bits; round_1 = COMPOUND_ROUND1_BITS)
    if (!g_ctx || (cp && !comp && cp->do_average) || (comp && (!cp->dst || cp->round_1 != 7 || cp->round_0 != (bd == 12 ? 5 : 3) || cp->dst_stride < p_width)) || p_width < 8 ||
        p_height < 8 || p_width > 128 || p_height > 128 || (p_width & 7) || (p_height & 7) || ss_x != ss_y || width <= 0 || height <= 0)
        return false;
    // the whole reference plane is uploaded (the model decides which part is read); a production caller keeps it resident and uses the batched entry
    const size_t rp = rup((size_t)width * pix_bytes, 4), dp = rup((size_t)p_width * pix_bytes, 4);
    uint8_t* d_ref = (uint8_t*)dev(5, rp * height); uint8_t* d_dst = (uint8_t*)dev(1, dp * p_height); void* d_job = dev(2, sizeof(SvtHipWarpBlk));
    if (!d_ref || !d_dst || !d_job) return false;
    SvtHipWarpBlk job;
    for (int i = 0; i < 6; i++) job.mat[i] = mat[i];
    job.alpha = (int16_t)alpha; job.beta = (int16_t)beta; job.gamma = (int16_t)gamma; job.delta = (int16_t)delta;
    job.p_col = p_col; job.p_row = p_row; job.p_width = (uint8_t)p_width; job.p_height = (uint8_t)p_height; job.reserved[0] = job.reserved[1] = 0;
    // destination pointer is biased so that (p_col, p_row) of the "plane" is element 0 of the packed block buffer
    uint8_t* d_plane0 = d_dst - ((ptrdiff_t)p_row * (ptrdiff_t)(dp / pix_bytes) + p_col) * pix_bytes;
    if (comp) {   // first reference: the 16-bit compound buffer comes back; second reference: it goes up and the averaged pixels come back
        const size_t cbp = rup((size_t)p_width * 2, 4);
        uint16_t* d_cb = (uint16_t*)dev(3, cbp * p_height); void* d_cjob = dev(4, sizeof(SvtHipWarpCompBlk));
        SvtHipWarpCompBlk cj = {};
        cj.blk = job; cj.cb_off = 0; cj.cb_stride = (int32_t)(cbp / 2); cj.do_average = cp->do_average ? 1 : 0; cj.use_jnt_comp_avg = cp->use_jnt_comp_avg ? 1 : 0;
        cj.fwd_offset = (uint8_t)cp->fwd_offset; cj.bck_offset = (uint8_t)cp->bck_offset;
        if (!d_cb || !d_cjob || !up2d(d_ref, rp, ref, (size_t)stride * pix_bytes, (size_t)width * pix_bytes, height) || !up(d_cjob, &cj, sizeof(cj))) return false;
        if (cp->do_average && !up2d(d_cb, cbp, cp->dst, (size_t)cp->dst_stride * 2, (size_t)p_width * 2, p_height)) return false;
        if (svt_hip_warp_compound_batch_dev(g_ctx, pix_bytes, bd, d_ref, width, height, (int)(rp / pix_bytes), d_plane0, (int)(dp / pix_bytes), ss_x, ss_y, d_cb,
                                            (const SvtHipWarpCompBlk*)d_cjob, 1) != 0)
            return false;
        return cp->do_average ? down2d(pred, (size_t)p_stride * pix_bytes, d_dst, dp, (size_t)p_width * pix_bytes, p_height)
                              : down2d(cp->dst, (size_t)cp->dst_stride * 2, d_cb, cbp, (size_t)p_width * 2, p_height);
    }
    return up2d(d_ref, rp, ref, (size_t)stride * pix_bytes, (size_t)width * pix_bytes, height) && up(d_job, &job, sizeof(job)) &&
           svt_hip_warp_predict_batch_dev(g_ctx, pix_bytes, bd, d_ref, width, height, (int)(rp / pix_bytes), d_plane0, (int)(dp / pix_bytes), ss_x, ss_y, (const SvtHipWarpBlk*)d_job, 1) == 0 &&
           down2d(pred, (size_t)p_stride * pix_bytes, d_dst, dp, (size_t)p_width * pix_bytes, p_height);
}
void warp_hip(const int32_t* mat, const uint8_t* ref, int width, int height, int stride, uint8_t* pred, int p_col, int p_row, int p_width, int p_height, int p_stride, int ssx,
              int ssy, SvtHipConvolveParams* cp, int16_t alpha, int16_t beta, int16_t gamma, int16_t delta) {
    Guard lk;
    if (warp_generic(1, 8, mat, ref, width, height, stride, pred, p_col, p_row, p_width, p_height, p_stride, ssx, ssy, cp, alpha, beta, gamma, delta)) return;
    FALLBACK("svt_av1_warp_affine", svt_av1_warp_affine, mat, ref, width, height, stride, pred, p_col, p_row, p_width, p_height, p_stride, ssx, ssy, cp, alpha, beta, gamma, delta);
}
void warp_hbd_hip(const int32_t* mat, const uint16_t* ref, int width, int height, int stride, uint16_t* pred, int p_col, int p_row, int p_width, int p_height, int p_stride,
                  int ssx, int ssy, int bd, SvtHipConvolveParams* cp, int16_t alpha, int16_t beta, int16_t gamma, int16_t delta) {
    Guard lk;
    if ((bd == 8 || bd == 10 || bd == 12) && warp_generic(2, bd, mat, ref, width, height, stride, pred, p_col, p_row, p_width, p_height, p_stride, ssx, ssy, cp, alpha, beta, gamma, delta)) return;
    FALLBACK("svt_av1_highbd_warp_affine", svt_av1_highbd_warp_affine, mat, ref, width, height, stride, pred, p_col, p_row, p_width, p_height, p_stride, ssx, ssy, bd, cp, alpha, beta, gamma, delta);
}

// ----------------------------------------------------------------------------------- Wiener statistics of one unit rectangle
bool stats_generic(int pix_bytes, int bd, int win, const void* dgd, const void* src, int h0, int h1, int v0, int v1, int dgd_stride, int src_stride, int64_t* M, int64_t* H) {
    const int w = h1 - h0, h = v1 - v0;
    if (!g_ctx || (win != 7 && win != 5 && win != 3) || w < 1 || h < 1 || w > 383 || h > 383) return false;   // one restoration unit of size 256 spans at most 383 samples
    // the rectangle becomes a one-unit "plane" (unit size 256 covers up to 384 samples), extended by 3 samples read from the caller's picture
    const size_t dp = rup((size_t)(w + 6) * pix_bytes, 4), sp = rup((size_t)w * pix_bytes, 4);
    const int w2 = win * win;
    uint8_t* d_d = (uint8_t*)dev(5, dp * (h + 6) + 64); uint8_t* d_s = (uint8_t*)dev(1, sp * h + 64); int64_t* d_M = (int64_t*)dev(6, (size_t)w2 * 8); int64_t* d_H = (int64_t*)dev(7, (size_t)w2 * w2 * 8);
    if (!d_d || !d_s || !d_M || !d_H) return false;
    const uint8_t* dg = (const uint8_t*)dgd + ((ptrdiff_t)(v0 - 3) * dgd_stride + (h0 - 3)) * pix_bytes;
    const uint8_t* sr = (const uint8_t*)src + ((ptrdiff_t)v0 * src_stride + h0) * pix_bytes;
    // unit rows start 8 above a multiple of the unit size in a plane; a single unit covering the whole "plane" needs ss_y such that voff does not split it:
    // with ph <= 1.5 * unit the plane has exactly one unit row, whatever the offset
    return up2d(d_d, dp, dg, (size_t)dgd_stride * pix_bytes, (size_t)(w + 6) * pix_bytes, h + 6) && up2d(d_s, sp, sr, (size_t)src_stride * pix_bytes, (size_t)w * pix_bytes, h) &&
           svt_hip_wiener_stats_plane_dev(g_ctx, pix_bytes, bd, win, d_d + 3 * dp + 3 * pix_bytes, (int)(dp / pix_bytes), d_s, (int)(sp / pix_bytes), w, h, 256, 0, d_M, d_H) == 0 &&
           down(M, d_M, (size_t)w2 * 8) && down(H, d_H, (size_t)w2 * w2 * 8);
}
void stats_hip(int32_t win, const uint8_t* dgd, const uint8_t* src, int32_t h0, int32_t h1, int32_t v0, int32_t v1, int32_t ds, int32_t ss, int64_t* M, int64_t* H) {
    Guard lk;
    if (stats_generic(1, 8, win, dgd, src, h0, h1, v0, v1, ds, ss, M, H)) return;
    FALLBACK("svt_av1_compute_stats", svt_av1_compute_stats, win, dgd, src, h0, h1, v0, v1, ds, ss, M, H);
}
void stats_hbd_hip(int32_t win, const uint8_t* dgd8, const uint8_t* src8, int32_t h0, int32_t h1, int32_t v0, int32_t v1, int32_t ds, int32_t ss, int64_t* M, int64_t* H, int32_t bd) {
    Guard lk;
    if ((bd == 8 || bd == 10 || bd == 12) &&
        stats_generic(2, bd, win, (const void*)((uintptr_t)dgd8 << 1), (const void*)((uintptr_t)src8 << 1), h0, h1, v0, v1, ds, ss, M, H)) return;
    FALLBACK("svt_av1_compute_stats_highbd", svt_av1_compute_stats_highbd, win, dgd8, src8, h0, h1, v0, v1, ds, ss, M, H, bd);
}


// ----------------------------------------------------------------------------------- 8-candidate SAD ladders of the open-loop search
void ext_all_sad_hip(uint8_t* src, uint32_t ss, uint8_t* ref, uint32_t rs, uint32_t mv, uint32_t* bs8, uint32_t* bs16, uint32_t* bm8, uint32_t* bm16, uint32_t e16[16][8],
                     uint32_t e8[64][8], uint8_t sub_sad) {
    Guard lk;
    if (g_ctx) {
        uint8_t *d_src = (uint8_t*)dev(0, 64 * 64), *d_ref = (uint8_t*)dev(1, 72 * 64 + 64); void* d_job = dev(2, sizeof(SvtHipExtSadJob)); uint32_t* d_st = (uint32_t*)dev(3, 800 * 4);
        const SvtHipExtSadJob job = {0, 0, mv, sub_sad ? 1 : 0};
        static thread_local uint32_t st[800];
        std::memcpy(st, bs8, 256); std::memcpy(st + 64, bs16, 64); std::memcpy(st + 80, bm8, 256); std::memcpy(st + 144, bm16, 64);
        const int rows = sub_sad ? 63 : 64;   // the sub-sampled form never reads the last row
        if (d_src && d_ref && d_job && d_st && up2d(d_src, 64, src, ss, 64, rows) && up2d(d_ref, 72, ref, rs, 71, rows) && up(d_job, &job, sizeof(job)) && up(d_st, st, 640) &&
            svt_hip_ext_all_sad_8x8_16x16_batch_dev(g_ctx, d_src, 64, d_ref, 72, (const SvtHipExtSadJob*)d_job, 1, d_st) == 0 && down(st, d_st, sizeof(st))) {
            std::memcpy(bs8, st, 256); std::memcpy(bs16, st + 64, 64); std::memcpy(bm8, st + 80, 256); std::memcpy(bm16, st + 144, 64);
            std::memcpy(e16, st + 160, 512); std::memcpy(e8, st + 288, 2048);
            return;
        }
    }
    FALLBACK("svt_ext_all_sad_calculation_8x8_16x16", svt_ext_all_sad_calculation_8x8_16x16, src, ss, ref, rs, mv, bs8, bs16, bm8, bm16, e16, e8, sub_sad);
}
void ext_eight_sad_hip(uint32_t s16[16][8], uint32_t* bs32, uint32_t* bs64, uint32_t* bm32, uint32_t* bm64, uint32_t mv, uint32_t s32[4][8]) {
    Guard lk;
    if (g_ctx) {
        uint32_t* d_st = (uint32_t*)dev(0, 170 * 4); uint32_t* d_mv = (uint32_t*)dev(1, 16);
        uint32_t st[170];
        std::memcpy(st, s16, 512); std::memcpy(st + 128, bs32, 16); st[132] = *bs64; std::memcpy(st + 133, bm32, 16); st[137] = *bm64;
        if (d_st && d_mv && up(d_st, st, 138 * 4) && up(d_mv, &mv, 4) && svt_hip_ext_eight_sad_32x32_64x64_batch_dev(g_ctx, d_mv, 1, d_st) == 0 && down(st, d_st, sizeof(st))) {
            std::memcpy(bs32, st + 128, 16); *bs64 = st[132]; std::memcpy(bm32, st + 133, 16); *bm64 = st[137]; std::memcpy(s32, st + 138, 128);
            return;
        }
    }
    FALLBACK("svt_ext_eight_sad_calculation_32x32_64x64", svt_ext_eight_sad_calculation_32x32_64x64, s16, bs32, bs64, bm32, bm64, mv, s32);
}

// ----------------------------------------------------------------------------------- quantizers
bool flat_qm(const uint8_t* qm, intptr_t n) {
    if (!qm) return true;
    for (intptr_t i = 0; i < n; i++)
        if (qm[i] != 32) return false;   // 1 << AOM_QM_BITS: the only matrix the device quantizer covers
    return true;
}
bool quant_generic(int variant, const int32_t* coeff, intptr_t n, const int16_t* zbin, const int16_t* round, const int16_t* quant, const int16_t* shift, int32_t* q, int32_t* dq,
                   const int16_t* dequant, uint16_t* eob, const int16_t* iscan, const uint8_t* qm, const uint8_t* iqm, int log_scale) {
    if (!g_ctx || n <= 0 || n > 4096 || log_scale < 0 || log_scale > 2 || !iscan || !flat_qm(qm, n) || !flat_qm(iqm, n)) return false;
    SvtHipQuantParams qp = {};
    for (int i = 0; i < 2; i++) {
        qp.zbin[i] = zbin ? zbin[i] : 0; qp.round[i] = round[i]; qp.quant[i] = quant[i]; qp.quant_shift[i] = shift ? shift[i] : 0; qp.dequant[i] = dequant[i];
    }
    qp.log_scale = log_scale; qp.variant = variant;
    int32_t *d_c = (int32_t*)dev(0, (size_t)n * 4), *d_q = (int32_t*)dev(1, (size_t)n * 4), *d_dq = (int32_t*)dev(3, (size_t)n * 4); int16_t* d_is = (int16_t*)dev(2, (size_t)n * 2);
    uint16_t* d_eob = (uint16_t*)dev(4, 16);
    return d_c && d_q && d_dq && d_is && d_eob && up(d_c, coeff, (size_t)n * 4) && up(d_is, iscan, (size_t)n * 2) &&
           svt_hip_quantize_batch_dev(g_ctx, d_c, (int)n, 1, &qp, d_is, d_q, d_dq, d_eob) == 0 && down(q, d_q, (size_t)n * 4) && down(dq, d_dq, (size_t)n * 4) && down(eob, d_eob, 2);
}
void quantize_b_hip(const int32_t* c, intptr_t n, const int16_t* zb, const int16_t* rd, const int16_t* qt, const int16_t* qs, int32_t* q, int32_t* dq, const int16_t* deq,
                    uint16_t* eob, const int16_t* scan, const int16_t* iscan, const uint8_t* qm, const uint8_t* iqm, const int32_t ls) {
    Guard lk;
    if (zb && qs && quant_generic(0, c, n, zb, rd, qt, qs, q, dq, deq, eob, iscan, qm, iqm, ls)) return;
    FALLBACK("svt_aom_quantize_b", svt_aom_quantize_b, c, n, zb, rd, qt, qs, q, dq, deq, eob, scan, iscan, qm, iqm, ls);
}
void quantize_b_hbd_hip(const int32_t* c, intptr_t n, const int16_t* zb, const int16_t* rd, const int16_t* qt, const int16_t* qs, int32_t* q, int32_t* dq, const int16_t* deq,
                        uint16_t* eob, const int16_t* scan, const int16_t* iscan, const uint8_t* qm, const uint8_t* iqm, const int32_t ls) {
    Guard lk;
    if (zb && qs && quant_generic(1, c, n, zb, rd, qt, qs, q, dq, deq, eob, iscan, qm, iqm, ls)) return;
    FALLBACK("svt_aom_highbd_quantize_b", svt_aom_highbd_quantize_b, c, n, zb, rd, qt, qs, q, dq, deq, eob, scan, iscan, qm, iqm, ls);
}
template <int LS> void quantize_fp_hip(const int32_t* c, intptr_t n, const int16_t* zb, const int16_t* rd, const int16_t* qt, const int16_t* qs, int32_t* q, int32_t* dq,
                                       const int16_t* deq, uint16_t* eob, const int16_t* scan, const int16_t* iscan) {
    Guard lk;
    if (quant_generic(2, c, n, zb, rd, qt, qs, q, dq, deq, eob, iscan, nullptr, nullptr, LS)) return;
    if (LS == 0) FALLBACK("svt_av1_quantize_fp", svt_av1_quantize_fp, c, n, zb, rd, qt, qs, q, dq, deq, eob, scan, iscan);
    if (LS == 1) FALLBACK("svt_av1_quantize_fp_32x32", svt_av1_quantize_fp_32x32, c, n, zb, rd, qt, qs, q, dq, deq, eob, scan, iscan);
    FALLBACK("svt_av1_quantize_fp_64x64", svt_av1_quantize_fp_64x64, c, n, zb, rd, qt, qs, q, dq, deq, eob, scan, iscan);
}
void quantize_fp_hbd_hip(const int32_t* c, intptr_t n, const int16_t* zb, const int16_t* rd, const int16_t* qt, const int16_t* qs, int32_t* q, int32_t* dq, const int16_t* deq,
                         uint16_t* eob, const int16_t* scan, const int16_t* iscan, int16_t ls) {
    Guard lk;
    if (quant_generic(3, c, n, zb, rd, qt, qs, q, dq, deq, eob, iscan, nullptr, nullptr, ls)) return;
    FALLBACK("svt_av1_highbd_quantize_fp", svt_av1_highbd_quantize_fp, c, n, zb, rd, qt, qs, q, dq, deq, eob, scan, iscan, ls);
}

// ----------------------------------------------------------------------------------- deblocking: one 4-sample edge segment
constexpr int kLpfLen[4] = {4, 6, 8, 14};
bool lpf_generic(int pb, int bd, void* s, int pitch, bool vertical_edge, int len, const uint8_t* blimit, const uint8_t* limit, const uint8_t* thresh) {
    if (!g_ctx || !s || !blimit || !limit || !thresh || (bd != 8 && bd != 10)) return false;
    const int half = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
    const int rw = vertical_edge ? 2 * half : 4, rh = vertical_edge ? 4 : 2 * half;      // exactly the samples the C function reads
    const size_t p = rup((size_t)rw * pb, 4);
    uint8_t* d = (uint8_t*)dev(0, p * rh + 64); void* d_job = dev(2, sizeof(SvtHipLpfEdge));
    uint8_t* origin = (uint8_t*)s - (vertical_edge ? (size_t)half : (size_t)half * pitch) * pb;
    SvtHipLpfEdge e = {};
    e.off = vertical_edge ? half : half * (int)(p / pb);
    e.dir = vertical_edge ? 0 : 1; e.len = (uint8_t)len; e.blimit = *blimit; e.limit = *limit; e.thresh = *thresh;
    return d && d_job && up2d(d, p, origin, (size_t)pitch * pb, (size_t)rw * pb, rh) && up(d_job, &e, sizeof(e)) &&
           svt_hip_lpf_edges_batch_dev(g_ctx, pb, bd, d, (int)(p / pb), (const SvtHipLpfEdge*)d_job, 1) == 0 && down2d(origin, (size_t)pitch * pb, d, p, (size_t)rw * pb, rh);
}
template <int V, int LI> void lpf_hip(uint8_t* s, int32_t pitch, const uint8_t* bl, const uint8_t* l, const uint8_t* t) {
    Guard lk;
    if (lpf_generic(1, 8, s, pitch, V, kLpfLen[LI], bl, l, t)) return;
    if (V) FALLBACK("svt_aom_lpf_vertical_N", svt_aom_lpf_vertical[LI], s, pitch, bl, l, t);
    FALLBACK("svt_aom_lpf_horizontal_N", svt_aom_lpf_horizontal[LI], s, pitch, bl, l, t);
}
template <int V, int LI> void lpf_hbd_hip(uint16_t* s, int32_t pitch, const uint8_t* bl, const uint8_t* l, const uint8_t* t, int32_t bd) {
    Guard lk;
    if (lpf_generic(2, bd, s, pitch, V, kLpfLen[LI], bl, l, t)) return;
    if (V) FALLBACK("svt_aom_highbd_lpf_vertical_N", svt_aom_highbd_lpf_vertical[LI], s, pitch, bl, l, t, bd);
    FALLBACK("svt_aom_highbd_lpf_horizontal_N", svt_aom_highbd_lpf_horizontal[LI], s, pitch, bl, l, t, bd);
}

// ----------------------------------------------------------------------------------- CDEF: one block
int32_t cdef_find_dir_hip(const uint16_t* img, int32_t stride, int32_t* var, int32_t coeff_shift) {
    Guard lk;
    if (g_ctx && coeff_shift >= 0 && coeff_shift <= 4) {
        uint16_t* d_img = (uint16_t*)dev(0, 8 * 8 * 2); int32_t* d_off = (int32_t*)dev(2, 16); int32_t* d_out = (int32_t*)dev(3, 16);
        const int32_t zero = 0; int32_t res[2];
        if (d_img && d_off && d_out && up2d(d_img, 16, img, (size_t)stride * 2, 16, 8) && up(d_off, &zero, 4) &&
            svt_hip_cdef_find_dir_batch_dev(g_ctx, d_img, 8, d_off, 1, coeff_shift, d_out, d_out + 1) == 0 && down(res, d_out, 8)) {
            *var = res[1];
            return res[0];
        }
    }
    FALLBACK("svt_cdef_find_dir", svt_cdef_find_dir, img, stride, var, coeff_shift);
}
void cdef_filter_block_hip(uint8_t* dst8, uint16_t* dst16, int32_t dstride, const uint16_t* in, int32_t pri, int32_t sec, int32_t dir, int32_t pdamp, int32_t sdamp, int32_t bsize,
                           int32_t cs) {
    Guard lk;
    // BLOCK_4X4 0, BLOCK_4X8 1, BLOCK_8X4 2, BLOCK_8X8 3 (EbCdef.c:211-212); the staging image has stride CDEF_BSTRIDE = 144 (EbCdef.h:35) and the taps reach 2 samples out
    if (g_ctx && bsize >= 0 && bsize <= 3 && dir >= 0 && dir < 8 && (dst8 || dst16)) {
        const int bw = (bsize == 3 || bsize == 2) ? 8 : 4, bh = (bsize == 3 || bsize == 1) ? 8 : 4, pb = dst8 ? 1 : 2;
        const int iw = bw + 4, ih = bh + 4, ip = 12;
        uint16_t* d_in = (uint16_t*)dev(0, (size_t)ip * ih * 2); void* d_job = dev(2, sizeof(SvtHipCdefBlk)); uint8_t* d_dst = (uint8_t*)dev(1, 8 * 8 * 2);
        const SvtHipCdefBlk job = {2 * ip + 2, 0, pri, sec, dir, pdamp, sdamp, bw == 8 ? 3 : 2, bh == 8 ? 3 : 2, cs};
        if (d_in && d_job && d_dst && up2d(d_in, (size_t)ip * 2, in - 2 * 144 - 2, 144 * 2, (size_t)iw * 2, ih) && up(d_job, &job, sizeof(job)) &&
            svt_hip_cdef_filter_block_batch_dev(g_ctx, d_in, ip, (const SvtHipCdefBlk*)d_job, 1, dst8 ? d_dst : nullptr, dst8 ? nullptr : (uint16_t*)d_dst, 8) == 0 &&
            down2d(dst8 ? (void*)dst8 : (void*)dst16, (size_t)dstride * pb, d_dst, (size_t)8 * pb, (size_t)bw * pb, bh))
            return;
    }
    FALLBACK("svt_cdef_filter_block", svt_cdef_filter_block, dst8, dst16, dstride, in, pri, sec, dir, pdamp, sdamp, bsize, cs);
}

// ----------------------------------------------------------------------------------- residual, 4-reference SAD, OBMC sub-pel prediction, variance intermediates
bool residual_generic(int pb, const void* in, uint32_t is, const void* pred, uint32_t ps, int16_t* res, uint32_t rs, uint32_t w, uint32_t h) {
    if (!g_ctx || !w || !h || w > 128 || h > 128) return false;
    const size_t p = rup((size_t)w * pb, 4), rp = rup((size_t)w * 2, 4);
    uint8_t *d_a = (uint8_t*)dev(0, p * h), *d_b = (uint8_t*)dev(1, p * h); int16_t* d_r = (int16_t*)dev(3, rp * h);
    return d_a && d_b && d_r && up2d(d_a, p, in, (size_t)is * pb, (size_t)w * pb, h) && up2d(d_b, p, pred, (size_t)ps * pb, (size_t)w * pb, h) &&
           svt_hip_residual_dev(g_ctx, pb, d_a, (int)(p / pb), d_b, (int)(p / pb), d_r, (int)(rp / 2), (int)w, (int)h) == 0 && down2d(res, (size_t)rs * 2, d_r, rp, (size_t)w * 2, h);
}
void residual8_hip(uint8_t* in, uint32_t is, uint8_t* pred, uint32_t ps, int16_t* res, uint32_t rs, uint32_t w, uint32_t h) {
    Guard lk;
    if (residual_generic(1, in, is, pred, ps, res, rs, w, h)) return;
    FALLBACK("svt_residual_kernel8bit", svt_residual_kernel8bit, in, is, pred, ps, res, rs, w, h);
}
void residual16_hip(uint16_t* in, uint32_t is, uint16_t* pred, uint32_t ps, int16_t* res, uint32_t rs, uint32_t w, uint32_t h) {
    Guard lk;
    if (residual_generic(2, in, is, pred, ps, res, rs, w, h)) return;
    FALLBACK("svt_residual_kernel16bit", svt_residual_kernel16bit, in, is, pred, ps, res, rs, w, h);
}
template <int IDX, int W, int H> void sadx4d_hip(const uint8_t* src, int ss, const uint8_t* const ref[], int rs, uint32_t* out) {
    Guard lk;
    if (g_ctx) {
        const size_t p = rup((size_t)W, 4);
        uint8_t *d_a = (uint8_t*)dev(0, p * H + 64), *d_b = (uint8_t*)dev(1, p * H * 4 + 64); SvtHipBlkPair* d_j = (SvtHipBlkPair*)dev(2, 4 * sizeof(SvtHipBlkPair)); uint32_t* d_o = (uint32_t*)dev(3, 16);
        SvtHipBlkPair jobs[4];
        bool ok = d_a && d_b && d_j && d_o && up2d(d_a, p, src, (size_t)ss, W, H);
        for (int k = 0; k < 4 && ok; k++) {
            jobs[k] = SvtHipBlkPair{0, 0, 0, k * H, (uint16_t)W, (uint16_t)H};
            ok = up2d(d_b + (size_t)k * H * p, p, ref[k], (size_t)rs, W, H);
        }
        if (ok && up(d_j, jobs, sizeof(jobs)) && svt_hip_block_sad_batch_dev(g_ctx, 1, d_a, (int)p, d_b, (int)p, d_j, 4, d_o) == 0 && down(out, d_o, 16)) return;
    }
    FALLBACK("svt_aom_sadWxHx4d", svt_aom_sadx4d[IDX], src, ss, ref, rs, out);
}
void upsampled_pred_hip(void* xd, const void* cm, int mi_row, int mi_col, const void* mv, uint8_t* comp_pred, int w, int h, int sx, int sy, const uint8_t* ref, int rs, int search) {
    Guard lk;
    // USE_2_TAPS 1 -> bilinear, USE_4_TAPS 2 -> the 4-tap regular family, USE_8_TAPS 3 -> 8-tap regular (EbDefinitions.h:487-490, variance.c:200-209)
    const int bank = search == 1 ? 3 : (search == 2 ? 4 : (search == 3 ? 0 : -1));
    if (g_ctx && bank >= 0 && w > 0 && h > 0 && w <= 128 && h <= 128 && sx >= 0 && sx < 8 && sy >= 0 && sy < 8) {
        // only the window the C function reads is staged: 3 samples before and 4 after in each direction that is interpolated
        const int x0 = sx ? 3 : 0, x1 = sx ? 4 : 0, y0 = sy ? 3 : 0, y1 = sy ? 4 : 0, rw = w + x0 + x1, rh = h + y0 + y1;
        const size_t p = rup((size_t)rw, 4);
        uint8_t *d_ref = (uint8_t*)dev(0, p * rh + 64), *d_dst = (uint8_t*)dev(1, (size_t)w * h + 64); void* d_job = dev(2, sizeof(SvtHipUpsampledBlk));
        SvtHipUpsampledBlk job = {};
        job.ref_off = (int32_t)(y0 * p + x0); job.w = (uint8_t)w; job.h = (uint8_t)h; job.subpel_x_q3 = (uint8_t)sx; job.subpel_y_q3 = (uint8_t)sy; job.bank = (uint8_t)bank;
        if (d_ref && d_dst && d_job && up2d(d_ref, p, ref - (ptrdiff_t)y0 * rs - x0, (size_t)rs, rw, rh) && up(d_job, &job, sizeof(job)) &&
            svt_hip_upsampled_pred_batch_dev(g_ctx, d_ref, (int)p, d_dst, (const SvtHipUpsampledBlk*)d_job, 1) == 0 && down(comp_pred, d_dst, (size_t)w * h))
            return;
    }
    FALLBACK("svt_aom_upsampled_pred", svt_aom_upsampled_pred, xd, cm, mi_row, mi_col, mv, comp_pred, w, h, sx, sy, ref, rs, search);
}
void interm_var_hip(uint8_t* in, uint16_t stride, uint64_t* mean, uint64_t* mean_sq) {
    Guard lk;
    if (g_ctx) {
        uint8_t* d_in = (uint8_t*)dev(0, 32 * 8); int32_t* d_off = (int32_t*)dev(2, 16); uint64_t* d_o = (uint64_t*)dev(3, 64);
        const int32_t zero = 0;
        uint64_t res[8];
        if (d_in && d_off && d_o && up2d(d_in, 32, in, stride, 32, 7) && up(d_off, &zero, 4) && svt_hip_interm_var_four8x8_batch_dev(g_ctx, d_in, 32, d_off, 1, d_o, d_o + 4) == 0 &&
            down(res, d_o, 64)) {
            std::memcpy(mean, res, 32); std::memcpy(mean_sq, res + 4, 32);
            return;
        }
    }
    FALLBACK("svt_compute_interm_var_four8x8", svt_compute_interm_var_four8x8, in, stride, mean, mean_sq);
}
template <int SLOT, int TS> uint64_t handle_transform_hip(int32_t* output) {
    Guard lk;
    if (g_ctx) {
        const size_t n = (size_t)kTxW[TS] * kTxH[TS];
        int32_t* d_c = (int32_t*)dev(0, n * 4); uint64_t* d_e = (uint64_t*)dev(3, 16);
        uint64_t e;
        if (d_c && d_e && up(d_c, output, n * 4) && svt_hip_handle_transform64_batch_dev(g_ctx, TS, d_c, 1, d_e) == 0 && down(output, d_c, n * 4) && down(&e, d_e, 8)) return e;
    }
    FALLBACK("svt_handle_transform64xN", svt_handle_transform64[SLOT], output);
}

// ----------------------------------------------------------------------------------- the small helpers of the same kernel classes
void subtract_block_hip(int rows, int cols, int16_t* diff, ptrdiff_t ds, const uint8_t* src, ptrdiff_t ss, const uint8_t* pred, ptrdiff_t ps) {
    Guard lk;
    if (rows > 0 && cols > 0 && residual_generic(1, src, (uint32_t)ss, pred, (uint32_t)ps, diff, (uint32_t)ds, (uint32_t)cols, (uint32_t)rows)) return;
    FALLBACK("svt_aom_subtract_block", svt_aom_subtract_block, rows, cols, diff, ds, src, ss, pred, ps);
}
void subtract_block_hbd_hip(int rows, int cols, int16_t* diff, ptrdiff_t ds, const uint8_t* src8, ptrdiff_t ss, const uint8_t* pred8, ptrdiff_t ps, int bd) {
    Guard lk;   // the byte pointers are plain casts of uint16_t pointers here (EbInterPrediction.c:52-53), not CONVERT_TO_BYTEPTR values
    if (rows > 0 && cols > 0 && residual_generic(2, src8, (uint32_t)ss, pred8, (uint32_t)ps, diff, (uint32_t)ds, (uint32_t)cols, (uint32_t)rows)) return;
    FALLBACK("svt_aom_highbd_subtract_block", svt_aom_highbd_subtract_block, rows, cols, diff, ds, src8, ss, pred8, ps, bd);
}
uint32_t sad_16b_hip(uint16_t* src, uint32_t ss, uint16_t* ref, uint32_t rs, uint32_t h, uint32_t w) {
    Guard lk;
    void *da, *db, *dj; uint32_t* dout; size_t p; uint32_t r;
    if (g_ctx && w && h && pair_stage(2, src, (int)ss, ref, (int)rs, (int)w, (int)h, &da, &db, &dj, &dout, &p) &&
        svt_hip_block_sad_batch_dev(g_ctx, 2, da, (int)(p / 2), db, (int)(p / 2), (const SvtHipBlkPair*)dj, 1, dout) == 0 && down(&r, dout, 4))
        return r;
    FALLBACK("sad_16b_kernel", sad_16b_kernel, src, ss, ref, rs, h, w);
}
uint32_t variance_highbd_hip(const uint16_t* a, int as, const uint16_t* b, int bs, int w, int h, uint32_t* sse) {
    Guard lk;
    unsigned v;
    if (w > 0 && h > 0 && var_generic(2, 16, a, as, b, bs, w, h, &v, sse)) return v;
    FALLBACK("variance_highbd", variance_highbd, a, as, b, bs, w, h, sse);
}
uint32_t nxm_sad_sub_hip(const uint8_t* src, uint32_t ss, const uint8_t* ref, uint32_t rs, uint32_t h, uint32_t w) {
    Guard lk;
    uint32_t r;
    if (sad_generic(src, (int)ss, ref, (int)rs, (int)w, (int)h, &r)) return r;
    FALLBACK("svt_nxm_sad_kernel_sub_sampled", svt_nxm_sad_kernel_sub_sampled, src, ss, ref, rs, h, w);
}
void ext_sad_16_hip(uint8_t* src, uint32_t ss, uint8_t* ref, uint32_t rs, uint32_t* bs8, uint32_t* bs16, uint32_t* bm8, uint32_t* bm16, uint32_t mv, uint32_t* s16, uint32_t* s8,
                    uint8_t sub) {
    Guard lk;
    if (g_ctx) {
        uint8_t *d_s = (uint8_t*)dev(0, 16 * 16), *d_r = (uint8_t*)dev(1, 16 * 16); SvtHipExtSadJob* d_j = (SvtHipExtSadJob*)dev(2, sizeof(SvtHipExtSadJob));
        uint32_t* d_st = (uint32_t*)dev(3, 15 * 4);
        const SvtHipExtSadJob job = {0, 0, mv, sub ? 1 : 0};
        uint32_t st[15];
        std::memcpy(st, bs8, 16); st[4] = *bs16; std::memcpy(st + 5, bm8, 16); st[9] = *bm16;
        std::memset(st + 10, 0, 20);
        if (d_s && d_r && d_j && d_st && up2d(d_s, 16, src, ss, 16, 16) && up2d(d_r, 16, ref, rs, 16, 16) && up(d_j, &job, sizeof(job)) && up(d_st, st, sizeof(st)) &&
            svt_hip_ext_sad_16x16_batch_dev(g_ctx, d_s, 16, d_r, 16, d_j, 1, d_st) == 0 && down(st, d_st, sizeof(st))) {
            std::memcpy(bs8, st, 16); *bs16 = st[4]; std::memcpy(bm8, st + 5, 16); *bm16 = st[9]; *s16 = st[10]; std::memcpy(s8, st + 11, 16);
            return;
        }
    }
    FALLBACK("svt_ext_sad_calculation_8x8_16x16", svt_ext_sad_calculation_8x8_16x16, src, ss, ref, rs, bs8, bs16, bm8, bm16, mv, s16, s8, sub);
}
void ext_sad_32_64_hip(uint32_t* s16, uint32_t* bs32, uint32_t* bs64, uint32_t* bm32, uint32_t* bm64, uint32_t mv, uint32_t* s32) {
    Guard lk;
    if (g_ctx) {
        uint32_t* d_st = (uint32_t*)dev(3, 30 * 4); uint32_t* d_mv = (uint32_t*)dev(2, 4);
        uint32_t st[30];
        std::memcpy(st, s16, 64); std::memcpy(st + 16, bs32, 16); st[20] = *bs64; std::memcpy(st + 21, bm32, 16); st[25] = *bm64;
        std::memset(st + 26, 0, 16);
        if (d_st && d_mv && up(d_st, st, sizeof(st)) && up(d_mv, &mv, 4) && svt_hip_ext_sad_32x32_64x64_batch_dev(g_ctx, d_st, d_mv, 1) == 0 && down(st, d_st, sizeof(st))) {
            std::memcpy(bs32, st + 16, 16); *bs64 = st[20]; std::memcpy(bm32, st + 21, 16); *bm64 = st[25]; std::memcpy(s32, st + 26, 16);
            return;
        }
    }
    FALLBACK("svt_ext_sad_calculation_32x32_64x64", svt_ext_sad_calculation_32x32_64x64, s16, bs32, bs64, bm32, bm64, mv, s32);
}
void copy_rect8_hip(uint16_t* dst, int32_t ds, const uint8_t* src, int32_t ss, int32_t v, int32_t h) {
    Guard lk;   // v rows of h samples (the reference's argument order)
    if (g_ctx && v > 0 && h > 0) {
        const size_t ip = rup((size_t)h, 4), op = rup((size_t)h * 2, 4);
        uint8_t* d_i = (uint8_t*)dev(0, ip * v); uint16_t* d_o = (uint16_t*)dev(1, op * v);
        if (d_i && d_o && up2d(d_i, ip, src, (size_t)ss, (size_t)h, v) &&
            svt_hip_picture_format_dev(g_ctx, 3, d_i, (int)ip, nullptr, 0, d_o, (int)(op / 2), nullptr, 0, h, v) == 0 && down2d(dst, (size_t)ds * 2, d_o, op, (size_t)h * 2, v))
            return;
    }
    FALLBACK("svt_copy_rect8_8bit_to_16bit", svt_copy_rect8_8bit_to_16bit, dst, ds, src, ss, v, h);
}
// BlockSize -> log2 of the block's width / height; only the four sizes compute_cdef_dist handles
bool cdef_bsize(int bsize, int* bwl, int* bhl) {
    if (bsize < 0 || bsize > 3) return false;
    *bwl = (bsize == 2 || bsize == 3) ? 3 : 2;   // BLOCK_4X4 0, BLOCK_4X8 1, BLOCK_8X4 2, BLOCK_8X8 3
    *bhl = (bsize == 1 || bsize == 3) ? 3 : 2;
    return true;
}
bool cdef_dist_generic(int pb, const void* dst, int32_t dstride, const void* src, const SvtHipCdefList* dlist, int32_t n, int bsize, int32_t cs, int32_t pli, uint64_t* out) {
    int bwl, bhl;
    if (!g_ctx || n < 0 || n > 64 || !cdef_bsize(bsize, &bwl, &bhl)) return false;
    if (n == 0) { *out = 0; return true; }
    int max_by = 0, max_bx = 0;
    for (int i = 0; i < n; i++) { if (dlist[i].by > max_by) max_by = dlist[i].by; if (dlist[i].bx > max_bx) max_bx = dlist[i].bx; }
    const int    w = (max_bx + 1) << bwl, h = (max_by + 1) << bhl;   // the part of the plane the listed blocks cover
    const size_t p = rup((size_t)w * pb, 4), sb = ((size_t)n << (bwl + bhl)) * pb;
    uint8_t *d_d = (uint8_t*)dev(0, p * h), *d_s = (uint8_t*)dev(1, sb), *d_l = (uint8_t*)dev(2, 3 * (size_t)n); uint64_t* d_o = (uint64_t*)dev(3, 8);
    return d_d && d_s && d_l && d_o && up2d(d_d, p, dst, (size_t)dstride * pb, (size_t)w * pb, h) && up(d_s, src, sb) && up(d_l, dlist, 3 * (size_t)n) &&
           svt_hip_cdef_dist_dev(g_ctx, pb, d_d, (int)(p / pb), d_s, d_l, n, bwl, bhl, cs, pli, d_o) == 0 && down(out, d_o, 8);
}
uint64_t cdef_dist8_hip(const uint8_t* dst, int32_t dstride, const uint8_t* src, const SvtHipCdefList* dlist, int32_t n, uint8_t bsize, int32_t cs, int32_t pli) {
    Guard lk;
    uint64_t r;
    if (cdef_dist_generic(1, dst, dstride, src, dlist, n, bsize, cs, pli, &r)) return r;
    FALLBACK("svt_compute_cdef_dist_8bit", svt_compute_cdef_dist_8bit, dst, dstride, src, dlist, n, bsize, cs, pli);
}
uint64_t cdef_dist16_hip(const uint16_t* dst, int32_t dstride, const uint16_t* src, const SvtHipCdefList* dlist, int32_t n, uint8_t bsize, int32_t cs, int32_t pli) {
    Guard lk;
    uint64_t r;
    if (cdef_dist_generic(2, dst, dstride, src, dlist, n, bsize, cs, pli, &r)) return r;
    FALLBACK("svt_compute_cdef_dist_16bit", svt_compute_cdef_dist_16bit, dst, dstride, src, dlist, n, bsize, cs, pli);
}
uint64_t search_one_dual_hip(int* lev0, int* lev1, int nb, uint64_t (**mse)[64], int sb_count, int start_gi, int end_gi) {
    Guard lk;
    if (g_ctx && sb_count >= 0 && nb >= 0 && nb < 8 && start_gi >= 0 && end_gi <= 64 && start_gi <= end_gi) {
        const size_t mb = (size_t)sb_count * 64 * 8;
        uint64_t *d_m0 = (uint64_t*)dev(0, mb + 8), *d_m1 = (uint64_t*)dev(1, mb + 8), *d_w = (uint64_t*)dev(3, (4097 + (size_t)sb_count) * 8);
        int* d_lev = (int*)dev(2, 16 * sizeof(int));
        int lev[16];
        std::memcpy(lev, lev0, 8 * sizeof(int)); std::memcpy(lev + 8, lev1, 8 * sizeof(int));
        uint64_t r;
        if (d_m0 && d_m1 && d_w && d_lev && (!mb || (up(d_m0, mse[0], mb) && up(d_m1, mse[1], mb))) && up(d_lev, lev, sizeof(lev)) &&
            svt_hip_cdef_search_one_dual_dev(g_ctx, d_m0, d_m1, sb_count, d_lev, d_lev + 8, nb, start_gi, end_gi, d_w) == 0 && down(lev, d_lev, sizeof(lev)) && down(&r, d_w, 8)) {
            lev0[nb] = lev[nb]; lev1[nb] = lev[8 + nb];
            return r;
        }
    }
    FALLBACK("svt_search_one_dual", svt_search_one_dual, lev0, lev1, nb, mse, sb_count, start_gi, end_gi);
}
// coefficient-domain sums of one w x h block with strides: packed on the way up, then the batched entry point with one block
bool coeff_dist_generic(const int32_t* coeff, uint32_t cs, const int32_t* recon, uint32_t rs, uint32_t w, uint32_t h, uint64_t out[3]) {
    if (!g_ctx || !w || !h || (size_t)w * h > 128 * 128) return false;
    const size_t n = (size_t)w * h;
    int32_t *d_c = (int32_t*)dev(0, n * 4), *d_r = recon ? (int32_t*)dev(1, n * 4) : nullptr; uint64_t* d_o = (uint64_t*)dev(3, 24);
    return d_c && (!recon || d_r) && d_o && up2d(d_c, (size_t)w * 4, coeff, (size_t)cs * 4, (size_t)w * 4, h) &&
           (!recon || up2d(d_r, (size_t)w * 4, recon, (size_t)rs * 4, (size_t)w * 4, h)) && svt_hip_coeff_distortion_batch_dev(g_ctx, d_c, d_r, (int)n, 1, d_o) == 0 &&
           down(out, d_o, 24);
}
void full_dist32_hip(int32_t* coeff, uint32_t cs, int32_t* recon, uint32_t rs, uint64_t res[2], uint32_t w, uint32_t h) {
    Guard lk;
    uint64_t o[3];
    if (coeff_dist_generic(coeff, cs, recon, rs, w, h, o)) { res[0] = o[0]; res[1] = o[1]; return; }
    FALLBACK("svt_full_distortion_kernel32_bits", svt_full_distortion_kernel32_bits, coeff, cs, recon, rs, res, w, h);
}
void full_dist_cbf_zero32_hip(int32_t* coeff, uint32_t cs, uint64_t res[2], uint32_t w, uint32_t h) {
    Guard lk;
    uint64_t o[3];
    if (coeff_dist_generic(coeff, cs, nullptr, 0, w, h, o)) { res[0] = o[1]; res[1] = o[1]; return; }
    FALLBACK("svt_full_distortion_kernel_cbf_zero32_bits", svt_full_distortion_kernel_cbf_zero32_bits, coeff, cs, res, w, h);
}
bool sse_generic(int pb, const void* a, int as, const void* b, int bs, int w, int h, uint64_t* out) {
    void *da, *db, *dj; uint32_t* dout; size_t p;
    return g_ctx && w > 0 && h > 0 && pair_stage(pb, a, as, b, bs, w, h, &da, &db, &dj, &dout, &p) &&
           svt_hip_block_sse_batch_dev(g_ctx, pb, da, (int)(p / pb), db, (int)(p / pb), (const SvtHipBlkPair*)dj, 1, (uint64_t*)dout) == 0 && down(out, dout, 8);
}
uint64_t spatial_dist_hip(uint8_t* in, uint32_t io, uint32_t is, uint8_t* rec, int32_t ro, uint32_t rs, uint32_t w, uint32_t h) {
    Guard lk;
    uint64_t r;
    if (sse_generic(1, in + io, (int)is, rec + ro, (int)rs, (int)w, (int)h, &r)) return r;
    FALLBACK("svt_spatial_full_distortion_kernel", svt_spatial_full_distortion_kernel, in, io, is, rec, ro, rs, w, h);
}
uint64_t full_dist16_hip(uint8_t* in, uint32_t io, uint32_t is, uint8_t* rec, int32_t ro, uint32_t rs, uint32_t w, uint32_t h) {
    Guard lk;   // 16-bit planes behind uint8_t* (a plain cast in the reference, offsets in samples)
    uint64_t r;
    if (sse_generic(2, (const uint16_t*)in + io, (int)is, (const uint16_t*)rec + ro, (int)rs, (int)w, (int)h, &r)) return r;
    FALLBACK("svt_full_distortion_kernel16_bits", svt_full_distortion_kernel16_bits, in, io, is, rec, ro, rs, w, h);
}
int64_t sse_hip(const uint8_t* a, int as, const uint8_t* b, int bs, int w, int h) {
    Guard lk;
    uint64_t r;
    if (sse_generic(1, a, as, b, bs, w, h, &r)) return (int64_t)r;
    FALLBACK("svt_aom_sse", svt_aom_sse, a, as, b, bs, w, h);
}
int64_t sse_hbd_hip(const uint8_t* a8, int as, const uint8_t* b8, int bs, int w, int h) {
    Guard lk;
    uint64_t r;
    if (sse_generic(2, a8, as, b8, bs, w, h, &r)) return (int64_t)r;   // plain casts of uint16_t pointers (EbEncInterPrediction.c:789-790)
    FALLBACK("svt_aom_highbd_sse", svt_aom_highbd_sse, a8, as, b8, bs, w, h);
}
int satd_hip(const int32_t* coeff, int length) {
    Guard lk;
    uint64_t o[3];
    if (length > 0 && coeff_dist_generic(coeff, (uint32_t)length, nullptr, 0, (uint32_t)length, 1, o)) return (int)o[2];
    FALLBACK("svt_aom_satd", svt_aom_satd, coeff, length);
}
int64_t block_error_hip(const int32_t* coeff, const int32_t* dq, intptr_t n, int64_t* ssz) {
    Guard lk;
    uint64_t o[3];
    if (n > 0 && n <= 128 * 128 && coeff_dist_generic(coeff, (uint32_t)n, dq, (uint32_t)n, (uint32_t)n, 1, o)) { *ssz = (int64_t)o[1]; return (int64_t)o[0]; }
    FALLBACK("svt_av1_block_error", svt_av1_block_error, coeff, dq, n, ssz);
}
// materialised flt0 / flt1: stage the unit's four planes; mode 0 returns the solved pair, mode 1 the error
bool sgr_flt_generic(int pb, const void* src, int w, int h, int ss, const void* dat, int ds, const int32_t* f0, int f0s, const int32_t* f1, int f1s, const SvtHipSgrParamsType* prm,
                     int mode, const int32_t* xq_in, int32_t* xq_out, int64_t* err) {
    if (!g_ctx || w < 1 || h < 1 || !prm) return false;
    const int    r0 = prm->r[0], r1 = prm->r[1];
    const size_t p = rup((size_t)w * pb, 4), fp = (size_t)w * 4;
    uint8_t *d_s = (uint8_t*)dev(0, p * h), *d_d = (uint8_t*)dev(1, p * h); int64_t* d_acc = (int64_t*)dev(3, 48);
    int32_t *d_f0 = r0 > 0 ? (int32_t*)dev(4, fp * h) : nullptr, *d_f1 = r1 > 0 ? (int32_t*)dev(5, fp * h) : nullptr;
    int64_t acc[6];
    if (!(d_s && d_d && d_acc && (r0 <= 0 || d_f0) && (r1 <= 0 || d_f1) && up2d(d_s, p, src, (size_t)ss * pb, (size_t)w * pb, h) && up2d(d_d, p, dat, (size_t)ds * pb, (size_t)w * pb, h) &&
          (r0 <= 0 || up2d(d_f0, fp, f0, (size_t)f0s * 4, fp, h)) && (r1 <= 0 || up2d(d_f1, fp, f1, (size_t)f1s * 4, fp, h)) &&
          svt_hip_sgr_flt_proj_dev(g_ctx, pb, d_s, (int)(p / pb), d_d, (int)(p / pb), d_f0, w, d_f1, w, w, h, r0, r1, mode, xq_in, d_acc, (int32_t*)(d_acc + 5)) == 0 &&
          down(acc, d_acc, sizeof(acc))))
        return false;
    if (mode == 0) std::memcpy(xq_out, &acc[5], 8);
    else *err = acc[0];
    return true;
}
void get_proj_subspace_hip(const uint8_t* src8, int w, int h, int ss, const uint8_t* dat8, int ds, int hbd, int32_t* f0, int f0s, int32_t* f1, int f1s, int* xq,
                           const SvtHipSgrParamsType* prm) {
    Guard lk;
    const void *s = hbd ? (const void*)((uintptr_t)src8 << 1) : (const void*)src8, *d = hbd ? (const void*)((uintptr_t)dat8 << 1) : (const void*)dat8;
    int32_t q[2];
    if (sgr_flt_generic(hbd ? 2 : 1, s, w, h, ss, d, ds, f0, f0s, f1, f1s, prm, 0, nullptr, q, nullptr)) { xq[0] = q[0]; xq[1] = q[1]; return; }
    FALLBACK("svt_get_proj_subspace", svt_get_proj_subspace, src8, w, h, ss, dat8, ds, hbd, f0, f0s, f1, f1s, xq, prm);
}
int64_t pixel_proj_error_hip(const uint8_t* src8, int32_t w, int32_t h, int32_t ss, const uint8_t* dat8, int32_t ds, int32_t* f0, int32_t f0s, int32_t* f1, int32_t f1s, int32_t xq[2],
                             const SvtHipSgrParamsType* prm) {
    Guard lk;
    int64_t e;
    if (sgr_flt_generic(1, src8, w, h, ss, dat8, ds, f0, f0s, f1, f1s, prm, 1, xq, nullptr, &e)) return e;
    FALLBACK("svt_av1_lowbd_pixel_proj_error", svt_av1_lowbd_pixel_proj_error, src8, w, h, ss, dat8, ds, f0, f0s, f1, f1s, xq, prm);
}
int64_t pixel_proj_error_hbd_hip(const uint8_t* src8, int32_t w, int32_t h, int32_t ss, const uint8_t* dat8, int32_t ds, int32_t* f0, int32_t f0s, int32_t* f1, int32_t f1s,
                                 int32_t xq[2], const SvtHipSgrParamsType* prm) {
    Guard lk;
    int64_t e;
    if (sgr_flt_generic(2, (const void*)((uintptr_t)src8 << 1), w, h, ss, (const void*)((uintptr_t)dat8 << 1), ds, f0, f0s, f1, f1s, prm, 1, xq, nullptr, &e)) return e;
    FALLBACK("svt_av1_highbd_pixel_proj_error", svt_av1_highbd_pixel_proj_error, src8, w, h, ss, dat8, ds, f0, f0s, f1, f1s, xq, prm);
}
uint64_t mean_sq_8x8_hip(uint8_t* in, uint32_t stride, uint32_t w, uint32_t h) {
    Guard lk;
    if (g_ctx && w && h && w <= 64 && h <= 64) {
        const size_t p = rup(w, 4);
        uint8_t* d_in = (uint8_t*)dev(0, p * h); int32_t* d_off = (int32_t*)dev(2, 16); uint64_t* d_o = (uint64_t*)dev(3, 16);
        const int32_t zero = 0;
        uint64_t r;
        if (d_in && d_off && d_o && up2d(d_in, p, in, stride, w, h) && up(d_off, &zero, 4) && svt_hip_block_mean_batch_dev(g_ctx, d_in, (int)p, d_off, 1, 0, (int)w, (int)h, d_o) == 0 &&
            down(&r, d_o, 8))
            return r;
    }
    FALLBACK("svt_compute_mean_square_values_8x8", svt_compute_mean_square_values_8x8, in, stride, w, h);
}
uint64_t sub_mean_8x8_hip(uint8_t* in, uint16_t stride) {
    Guard lk;
    if (g_ctx) {
        uint8_t* d_in = (uint8_t*)dev(0, 8 * 8); int32_t* d_off = (int32_t*)dev(2, 16); uint64_t* d_o = (uint64_t*)dev(3, 16);
        const int32_t zero = 0;
        uint64_t r;
        if (d_in && d_off && d_o && up2d(d_in, 8, in, stride, 8, 7) && up(d_off, &zero, 4) && svt_hip_block_mean_batch_dev(g_ctx, d_in, 8, d_off, 1, 1, 8, 8, d_o) == 0 && down(&r, d_o, 8))
            return r;
    }
    FALLBACK("svt_compute_sub_mean_8x8", svt_compute_sub_mean_8x8, in, stride);
}
// the reference finds the 16-kernel table by aligning the filter pointer down to 256 bytes (convolve.c:49-57); the table travels as it is
bool convolve8_generic(bool vert, const uint8_t* src, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds, const int16_t* filter, int step, int w, int h) {
    if (!g_ctx || w < 1 || h < 1 || w > 128 || h > 128 || step < 1 || step > 64 || !filter) return false;
    const int16_t* base = (const int16_t*)((uintptr_t)filter & ~(uintptr_t)0xff);
    const int      q0 = (int)((filter - base) / 8);
    const int      span = (((vert ? h : w) - 1) * step + q0) / 16 + 8;   // samples touched along the filtered axis, starting 3 before the first output
    const int      cw = vert ? w : span, ch = vert ? span : h;
    const size_t   ip = rup((size_t)cw, 4), op = rup((size_t)w, 4);
    uint8_t *d_i = (uint8_t*)dev(0, ip * ch + 64), *d_o = (uint8_t*)dev(1, op * h); int16_t* d_f = (int16_t*)dev(2, 256);
    const uint8_t* s0 = vert ? src - 3 * ss : src - 3;
    return d_i && d_o && d_f && up2d(d_i, ip, s0, (size_t)ss, (size_t)cw, ch) && up(d_f, base, 256) &&
           svt_hip_convolve8_dev(g_ctx, vert, vert ? d_i + 3 * ip : d_i + 3, (int)ip, d_o, (int)op, d_f, q0, step, w, h) == 0 && down2d(dst, (size_t)ds, d_o, op, (size_t)w, h);
}
void convolve8_horiz_hip(const uint8_t* src, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds, const int16_t* fx, int xs, const int16_t* fy, int ys, int w, int h) {
    Guard lk;
    if (convolve8_generic(false, src, ss, dst, ds, fx, xs, w, h)) return;
    FALLBACK("svt_aom_convolve8_horiz", svt_aom_convolve8_horiz, src, ss, dst, ds, fx, xs, fy, ys, w, h);
}
void convolve8_vert_hip(const uint8_t* src, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds, const int16_t* fx, int xs, const int16_t* fy, int ys, int w, int h) {
    Guard lk;
    if (convolve8_generic(true, src, ss, dst, ds, fy, ys, w, h)) return;
    FALLBACK("svt_aom_convolve8_vert", svt_aom_convolve8_vert, src, ss, dst, ds, fx, xs, fy, ys, w, h);
}
bool wiener_generic(int pb, int bd, const void* src, ptrdiff_t ss, void* dst, ptrdiff_t ds, const int16_t* fx, const int16_t* fy, int w, int h, const SvtHipConvolveParams* cp) {
    if (!g_ctx || w < 1 || h < 1 || w > 128 || h > 128 || !cp || !fx || !fy) return false;
    const int    cw = w + 7, ch = h + 7;
    const size_t ip = rup((size_t)cw * pb, 4), op = rup((size_t)w * pb, 4);
    uint8_t *d_i = (uint8_t*)dev(0, ip * ch + 64), *d_o = (uint8_t*)dev(1, op * h); int16_t* d_t = (int16_t*)dev(2, 32);
    int16_t taps[16];
    std::memcpy(taps, fx, 16); std::memcpy(taps + 8, fy, 16);
    const uint8_t* s0 = (const uint8_t*)src - (3 * ss + 3) * pb;
    return d_i && d_o && d_t && up2d(d_i, ip, s0, (size_t)ss * pb, (size_t)cw * pb, ch) && up(d_t, taps, 32) &&
           svt_hip_wiener_convolve_add_src_dev(g_ctx, pb, bd, d_i + 3 * ip + 3 * pb, (int)(ip / pb), d_o, (int)(op / pb), d_t, w, h, cp->round_0, cp->round_1) == 0 &&
           down2d(dst, (size_t)ds * pb, d_o, op, (size_t)w * pb, h);
}
void wiener_convolve_hip(const uint8_t* src, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds, const int16_t* fx, const int16_t* fy, int32_t w, int32_t h, const SvtHipConvolveParams* cp) {
    Guard lk;
    if (wiener_generic(1, 8, src, ss, dst, ds, fx, fy, w, h, cp)) return;
    FALLBACK("svt_av1_wiener_convolve_add_src", svt_av1_wiener_convolve_add_src, src, ss, dst, ds, fx, fy, w, h, cp);
}
void wiener_convolve_hbd_hip(const uint8_t* src8, ptrdiff_t ss, uint8_t* dst8, ptrdiff_t ds, const int16_t* fx, const int16_t* fy, int32_t w, int32_t h, const SvtHipConvolveParams* cp,
                             int32_t bd) {
    Guard lk;
    if (wiener_generic(2, bd, (const void*)((uintptr_t)src8 << 1), ss, (void*)((uintptr_t)dst8 << 1), ds, fx, fy, w, h, cp)) return;
    FALLBACK("svt_av1_highbd_wiener_convolve_add_src", svt_av1_highbd_wiener_convolve_add_src, src8, ss, dst8, ds, fx, fy, w, h, cp, bd);
}

template <int SLOT, int TS> uint64_t handle_transform_n2n4_hip(int32_t* output) {
    Guard lk;
    if (g_ctx) {
        const size_t n = (size_t)kTxW[TS] * kTxH[TS];
        int32_t* d_c = (int32_t*)dev(0, n * 4);
        if (d_c && up(d_c, output, n * 4) && svt_hip_handle_transform64_n2n4_batch_dev(g_ctx, TS, d_c, 1) == 0 && down(output, d_c, n * 4)) return 0;
    }
    FALLBACK("handle_transform64xN_N2_N4", handle_transform64_N2_N4[SLOT], output);
}
unsigned mse16x16_hip(const uint8_t* a, int as, const uint8_t* b, int bs, unsigned* sse) {
    Guard lk;
    unsigned v;
    if (var_generic(1, 8, a, as, b, bs, 16, 16, &v, sse)) return v;
    FALLBACK("svt_aom_mse16x16", svt_aom_mse16x16, a, as, b, bs, sse);
}
void mse16x16_hbd8_hip(const uint8_t* a8, int32_t as, const uint8_t* b8, int32_t bs, uint32_t* sse) {
    Guard lk;
    uint64_t r;
    if (sse_generic(2, (const void*)((uintptr_t)a8 << 1), as, (const void*)((uintptr_t)b8 << 1), bs, 16, 16, &r)) { *sse = (uint32_t)r; return; }
    FALLBACK("svt_aom_highbd_8_mse16x16", svt_aom_highbd_8_mse16x16, a8, as, b8, bs, sse);
}

// ----------------------------------------------------------------------------------- picture formats (svt_hip_picture_format_dev, one rectangle)
// in / out planes: pointer, stride and width in BYTES per row as the host sees them, and the stride the device call takes (samples of that plane)
struct FmtPlane { const void* h; size_t hpitch, wbytes; int dev_stride_div; };
bool format_generic(int mode, const FmtPlane& i0, const FmtPlane& i1, void* o0, size_t o0pitch, size_t o0w, int o0div, void* o1, size_t o1pitch, size_t o1w, int w, int h) {
    if (!g_ctx || w < 1 || h < 1) return false;
    const size_t p0 = rup(i0.wbytes, 4), p1 = i1.h ? rup(i1.wbytes, 4) : 0, q0 = rup(o0w, 4), q1 = o1 ? rup(o1w, 4) : 0;
    uint8_t *d_i0 = (uint8_t*)dev(0, p0 * h), *d_i1 = i1.h ? (uint8_t*)dev(1, p1 * h) : nullptr, *d_o0 = (uint8_t*)dev(4, q0 * h), *d_o1 = o1 ? (uint8_t*)dev(5, q1 * h) : nullptr;
    return d_i0 && (!i1.h || d_i1) && d_o0 && (!o1 || d_o1) && up2d(d_i0, p0, i0.h, i0.hpitch, i0.wbytes, h) && (!i1.h || up2d(d_i1, p1, i1.h, i1.hpitch, i1.wbytes, h)) &&
           svt_hip_picture_format_dev(g_ctx, mode, d_i0, (int)(p0 / i0.dev_stride_div), d_i1, i1.h ? (int)(p1 / i1.dev_stride_div) : 0, d_o0, (int)(q0 / o0div), d_o1, (int)q1, w, h) == 0 &&
           down2d(o0, o0pitch, d_o0, q0, o0w, h) && (!o1 || down2d(o1, o1pitch, d_o1, q1, o1w, h));
}
void convert_8_to_16_hip(uint8_t* src, uint32_t ss, uint16_t* dst, uint32_t ds, uint32_t w, uint32_t h) {
    Guard lk;
    if (format_generic(3, {src, ss, w, 1}, {nullptr, 0, 0, 1}, dst, (size_t)ds * 2, (size_t)w * 2, 2, nullptr, 0, 0, (int)w, (int)h)) return;
    FALLBACK("svt_convert_8bit_to_16bit", svt_convert_8bit_to_16bit, src, ss, dst, ds, w, h);
}
void convert_16_to_8_hip(uint16_t* src, uint32_t ss, uint8_t* dst, uint32_t ds, uint32_t w, uint32_t h) {
    Guard lk;
    if (format_generic(4, {src, (size_t)ss * 2, (size_t)w * 2, 2}, {nullptr, 0, 0, 1}, dst, ds, w, 1, nullptr, 0, 0, (int)w, (int)h)) return;
    FALLBACK("svt_convert_16bit_to_8bit", svt_convert_16bit_to_8bit, src, ss, dst, ds, w, h);
}
void c_pack_hip(const uint8_t* inn, uint32_t is, uint8_t* out, uint32_t os, uint8_t* cache, uint32_t w, uint32_t h) {
    Guard lk;
    if (!(w & 3) && format_generic(5, {inn, is, w, 1}, {nullptr, 0, 0, 1}, out, os, w / 4, 1, nullptr, 0, 0, (int)w, (int)h)) return;
    FALLBACK("svt_c_pack", svt_c_pack, inn, is, out, os, cache, w, h);
}
void compressed_packmsb_hip(uint8_t* in8, uint32_t s8, uint8_t* inn, uint16_t* out, uint32_t sn, uint32_t os, uint32_t w, uint32_t h) {
    Guard lk;
    if (!(w & 3) && format_generic(1, {in8, s8, w, 1}, {inn, sn, w / 4, 1}, out, (size_t)os * 2, (size_t)w * 2, 2, nullptr, 0, 0, (int)w, (int)h)) return;
    FALLBACK("svt_compressed_packmsb", svt_compressed_packmsb, in8, s8, inn, out, sn, os, w, h);
}
void pack2d_hip(uint8_t* in8, uint32_t s8, uint8_t* inn, uint16_t* out, uint32_t sn, uint32_t os, uint32_t w, uint32_t h) {
    Guard lk;
    if (format_generic(0, {in8, s8, w, 1}, {inn, sn, w, 1}, out, (size_t)os * 2, (size_t)w * 2, 2, nullptr, 0, 0, (int)w, (int)h)) return;
    FALLBACK("svt_pack2d_16_bit_src_mul4", svt_pack2d_16_bit_src_mul4, in8, s8, inn, out, sn, os, w, h);
}
void unpack_avg_hip(uint16_t* l0, uint32_t s0, uint16_t* l1, uint32_t s1, uint8_t* dst, uint32_t ds, uint32_t w, uint32_t h) {
    Guard lk;
    if (format_generic(6, {l0, (size_t)s0 * 2, (size_t)w * 2, 2}, {l1, (size_t)s1 * 2, (size_t)w * 2, 2}, dst, ds, w, 1, nullptr, 0, 0, (int)w, (int)h)) return;
    FALLBACK("svt_unpack_avg", svt_unpack_avg, l0, s0, l1, s1, dst, ds, w, h);
}
void un_pack2d_hip(uint16_t* in, uint32_t is, uint8_t* o8, uint8_t* on, uint32_t s8, uint32_t sn, uint32_t w, uint32_t h) {
    Guard lk;
    if (format_generic(2, {in, (size_t)is * 2, (size_t)w * 2, 2}, {nullptr, 0, 0, 1}, o8, s8, w, 1, on, sn, w, (int)w, (int)h)) return;
    FALLBACK("svt_un_pack2d_16_bit_src_mul4", svt_un_pack2d_16_bit_src_mul4, in, is, o8, on, s8, sn, w, h);
}
void un_pack8_hip(uint16_t* in, uint32_t is, uint8_t* o8, uint32_t s8, uint32_t w, uint32_t h) {
    Guard lk;
    if (format_generic(2, {in, (size_t)is * 2, (size_t)w * 2, 2}, {nullptr, 0, 0, 1}, o8, s8, w, 1, nullptr, 0, 0, (int)w, (int)h)) return;
    FALLBACK("svt_un_pack8_bit_data", svt_un_pack8_bit_data, in, is, o8, s8, w, h);
}

// ----------------------------------------------------------------------------------- one reference of a compound prediction
// variant 0 = 2d, 1 = x, 2 = y, 3 = 2d_copy.  The compound buffer (ConvolveParams::dst) travels both ways: written when do_average = 0, read otherwise.
bool jnt_generic(int pb, int bd, int variant, const void* src, int ss, void* dst, int ds, int w, int h, const SvtHipInterpFilterParams* fx, const SvtHipInterpFilterParams* fy, int sx,
                 int sy, const SvtHipConvolveParams* cp) {
    if (!g_ctx || !cp || !cp->dst || w < 1 || h < 1 || w > 128 || h > 128 || !fx || !fy || !fx->filter_ptr || !fy->filter_ptr || fx->taps != 8 || fy->taps != 8) return false;
    const bool   fh = variant == 0 || variant == 1, fv = variant == 0 || variant == 2;
    const int    cw = w + (fh ? 7 : 0), ch = h + (fv ? 7 : 0), ox = fh ? 3 : 0, oy = fv ? 3 : 0;
    const size_t ip = rup((size_t)cw * pb, 4), op = rup((size_t)w * pb, 4), cp_pitch = (size_t)w * 2;
    uint8_t *d_i = (uint8_t*)dev(0, ip * ch + 64), *d_o = (uint8_t*)dev(1, op * h); uint16_t* d_cb = (uint16_t*)dev(4, cp_pitch * h); int16_t* d_t = (int16_t*)dev(2, 32);
    int16_t taps[16];
    std::memcpy(taps, fx->filter_ptr + 8 * (sx & 15), 16); std::memcpy(taps + 8, fy->filter_ptr + 8 * (sy & 15), 16);
    const uint8_t* s0 = (const uint8_t*)src - ((size_t)oy * ss + ox) * pb;
    if (!(d_i && d_o && d_cb && d_t && up2d(d_i, ip, s0, (size_t)ss * pb, (size_t)cw * pb, ch) && up(d_t, taps, 32))) return false;
    if (cp->do_average && !up2d(d_cb, cp_pitch, cp->dst, (size_t)cp->dst_stride * 2, (size_t)w * 2, h)) return false;
    if (svt_hip_jnt_convolve_dev(g_ctx, pb, bd, variant, d_i + oy * ip + (size_t)ox * pb, (int)(ip / pb), d_o, (int)(op / pb), d_cb, w, d_t, w, h, cp->round_0, cp->round_1,
                                 cp->do_average, cp->use_jnt_comp_avg, cp->fwd_offset, cp->bck_offset) != 0)
        return false;
    return cp->do_average ? down2d(dst, (size_t)ds * pb, d_o, op, (size_t)w * pb, h) : down2d(cp->dst, (size_t)cp->dst_stride * 2, d_cb, cp_pitch, (size_t)w * 2, h);
}
#define JNT_WRAPPER(NAME, MEMBER, VARIANT)                                                                                                            \
    void NAME(const uint8_t* src, int32_t ss, uint8_t* dst, int32_t ds, int32_t w, int32_t h, SvtHipInterpFilterParams* fx, SvtHipInterpFilterParams* fy, \
              const int32_t sx, const int32_t sy, SvtHipConvolveParams* cp) {                                                                         \
        Guard lk;                                                                                                                                     \
        if (jnt_generic(1, 8, VARIANT, src, ss, dst, ds, w, h, fx, fy, sx, sy, cp)) return;                                                          \
        FALLBACK("svt_av1_" #MEMBER, svt_av1_##MEMBER, src, ss, dst, ds, w, h, fx, fy, sx, sy, cp);                                                 \
    }                                                                                                                                                 \
    void NAME##_hbd(const uint16_t* src, int32_t ss, uint16_t* dst, int32_t ds, int32_t w, int32_t h, const SvtHipInterpFilterParams* fx,              \
                    const SvtHipInterpFilterParams* fy, const int32_t sx, const int32_t sy, SvtHipConvolveParams* cp, int32_t bd) {                   \
        Guard lk;                                                                                                                                     \
        if (bd >= 8 && bd <= 12 && jnt_generic(2, bd, VARIANT, src, ss, dst, ds, w, h, fx, fy, sx, sy, cp)) return;                                  \
        FALLBACK("highbd " #MEMBER, svt_av1_highbd_##MEMBER, src, ss, dst, ds, w, h, fx, fy, sx, sy, cp, bd);                                       \
    }
JNT_WRAPPER(jnt_2d_hip, jnt_convolve_2d, 0)
JNT_WRAPPER(jnt_x_hip, jnt_convolve_x, 1)
JNT_WRAPPER(jnt_y_hip, jnt_convolve_y, 2)
JNT_WRAPPER(jnt_copy_hip, jnt_convolve_2d_copy, 3)

// ----------------------------------------------------------------------------------- masked compound
bool diffwtd_generic(int eb, uint8_t* mask, int type, const void* s0, int s0s, const void* s1, int s1s, int h, int w, int round, int shift) {
    if (!g_ctx || w < 1 || h < 1 || w > 128 || h > 128 || type < 0 || type > 1) return false;
    const size_t p = rup((size_t)w * eb, 4);
    uint8_t *d_a = (uint8_t*)dev(0, p * h), *d_b = (uint8_t*)dev(1, p * h), *d_m = (uint8_t*)dev(3, (size_t)w * h);
    return d_a && d_b && d_m && up2d(d_a, p, s0, (size_t)s0s * eb, (size_t)w * eb, h) && up2d(d_b, p, s1, (size_t)s1s * eb, (size_t)w * eb, h) &&
           svt_hip_diffwtd_mask_dev(g_ctx, eb, d_m, d_a, (int)(p / eb), d_b, (int)(p / eb), w, h, type, round, shift) == 0 && down(mask, d_m, (size_t)w * h);
}
void diffwtd_mask_hip(uint8_t* mask, uint8_t type, const uint8_t* s0, int s0s, const uint8_t* s1, int s1s, int h, int w) {
    Guard lk;
    if (diffwtd_generic(1, mask, type, s0, s0s, s1, s1s, h, w, 0, 0)) return;
    FALLBACK("svt_av1_build_compound_diffwtd_mask", svt_av1_build_compound_diffwtd_mask, mask, type, s0, s0s, s1, s1s, h, w);
}
void diffwtd_mask_hbd_hip(uint8_t* mask, uint8_t type, const uint8_t* s0, int s0s, const uint8_t* s1, int s1s, int h, int w, int bd) {
    Guard lk;
    if (bd >= 8 && bd <= 12 && diffwtd_generic(2, mask, type, s0, s0s, s1, s1s, h, w, 0, bd - 8)) return;
    FALLBACK("svt_av1_build_compound_diffwtd_mask_highbd", svt_av1_build_compound_diffwtd_mask_highbd, mask, type, s0, s0s, s1, s1s, h, w, bd);
}
void diffwtd_mask_d16_hip(uint8_t* mask, uint8_t type, const uint16_t* s0, int s0s, const uint16_t* s1, int s1s, int h, int w, SvtHipConvolveParams* cp, int bd) {
    Guard lk;
    if (cp && bd >= 8 && bd <= 12 && diffwtd_generic(2, mask, type, s0, s0s, s1, s1s, h, w, 14 - cp->round_0 - cp->round_1 + bd - 8, 0)) return;
    FALLBACK("svt_av1_build_compound_diffwtd_mask_d16", svt_av1_build_compound_diffwtd_mask_d16, mask, type, s0, s0s, s1, s1s, h, w, cp, bd);
}
bool blend_d16_generic(int pb, int bd, void* dst, uint32_t ds, const uint16_t* s0, uint32_t s0s, const uint16_t* s1, uint32_t s1s, const uint8_t* mask, uint32_t ms, int w, int h, int subw,
                       int subh, const SvtHipConvolveParams* cp) {
    if (!g_ctx || !cp || w < 1 || h < 1 || w > 128 || h > 128) return false;
    const int    mw = w << (subw ? 1 : 0), mh = h << (subh ? 1 : 0);
    const size_t sp = (size_t)w * 2, mp = rup((size_t)mw, 4), dp = rup((size_t)w * pb, 4);
    uint16_t *d_0 = (uint16_t*)dev(0, sp * h), *d_1 = (uint16_t*)dev(1, sp * h); uint8_t *d_m = (uint8_t*)dev(4, mp * mh), *d_d = (uint8_t*)dev(5, dp * h);
    // src0 / src1 may alias dst in the reference's callers (it reads each sample before it writes it): both are uploaded before anything is written
    return d_0 && d_1 && d_m && d_d && up2d(d_0, sp, s0, (size_t)s0s * 2, sp, h) && up2d(d_1, sp, s1, (size_t)s1s * 2, sp, h) && up2d(d_m, mp, mask, ms, (size_t)mw, mh) &&
           svt_hip_blend_a64_d16_dev(g_ctx, pb, bd, d_d, (int)(dp / pb), d_0, w, d_1, w, d_m, (int)mp, w, h, subw, subh, cp->round_0, cp->round_1) == 0 &&
           down2d(dst, (size_t)ds * pb, d_d, dp, (size_t)w * pb, h);
}
void blend_d16_hip(uint8_t* dst, uint32_t ds, const uint16_t* s0, uint32_t s0s, const uint16_t* s1, uint32_t s1s, const uint8_t* mask, uint32_t ms, int w, int h, int subw, int subh,
                   SvtHipConvolveParams* cp) {
    Guard lk;
    if (blend_d16_generic(1, 8, dst, ds, s0, s0s, s1, s1s, mask, ms, w, h, subw, subh, cp)) return;
    FALLBACK("svt_aom_lowbd_blend_a64_d16_mask", svt_aom_lowbd_blend_a64_d16_mask, dst, ds, s0, s0s, s1, s1s, mask, ms, w, h, subw, subh, cp);
}
void blend_d16_hbd_hip(uint8_t* dst, uint32_t ds, const uint16_t* s0, uint32_t s0s, const uint16_t* s1, uint32_t s1s, const uint8_t* mask, uint32_t ms, int w, int h, int subw, int subh,
                       SvtHipConvolveParams* cp, int bd) {
    Guard lk;
    if ((bd == 8 || bd == 10 || bd == 12) && blend_d16_generic(2, bd, dst, ds, s0, s0s, s1, s1s, mask, ms, w, h, subw, subh, cp)) return;
    FALLBACK("svt_aom_highbd_blend_a64_d16_mask", svt_aom_highbd_blend_a64_d16_mask, dst, ds, s0, s0s, s1, s1s, mask, ms, w, h, subw, subh, cp, bd);
}

}  // namespace

// One line per wrapper that was called and per table entry that was delegated, e.g.
//   svt_hip_rtcd_calls quantize_b_hip calls=1234
//   svt_hip_rtcd_delegated svt_aom_quantize_b count=3 device_failures=0
// (the hooked encoder prints it at exit; tests/test_encode_e2e.py pins the delegated set).  Returns the number of delegations.
extern "C" int svt_hip_rtcd_report(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    long total = 0;
    for (int i = 0; i < g_ncalls; i++) std::fprintf(stderr, "svt_hip_rtcd_calls %s calls=%ld\n", g_calls[i].key, g_calls[i].n);
    for (int i = 0; i < g_ndeleg; i++) {
        std::fprintf(stderr, "svt_hip_rtcd_delegated %s count=%ld device_failures=%ld\n", g_deleg[i].key, g_deleg[i].n, g_deleg[i].device_failures);
        total += g_deleg[i].n;
    }
    return (int)(total > 0x7fffffff ? 0x7fffffff : total);
}

// The wrappers' device staging buffers and their context binding, released: call when the pointers of svt_hip_setup_rtcd are no longer installed anywhere
// (the encoder instance that installed them is gone); a later svt_hip_setup_rtcd starts afresh.
extern "C" void svt_hip_rtcd_release(void) {
    Guard lk;
    for (auto& s : g_slot) { if (s.p) (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
    g_ctx = nullptr;
}

extern "C" int svt_hip_setup_rtcd(SvtHipCtx* ctx, SvtHipRtcd* t) {
    if (!ctx || !t) return SVT_HIP_ERR_BAD_ARG;
    Guard lk;
    g_ctx = ctx;
    g_c = *t;
    t->svt_sad_loop_kernel = sad_loop_hip;
    t->svt_nxm_sad_kernel = nxm_sad_hip;
#define X(I, W, H) t->svt_aom_sad[I] = sad_wxh_hip<I, W, H>; t->svt_aom_variance[I] = var_wxh_hip<I, W, H>; t->svt_aom_highbd_10_variance[I] = var10_wxh_hip<I, W, H>;
    SVT_HIP_RTCD_BLOCK_SIZES(X)
#undef X
    t->svt_av1_convolve_2d_sr = conv_2d_hip; t->svt_av1_convolve_x_sr = conv_x_hip; t->svt_av1_convolve_y_sr = conv_y_hip; t->svt_av1_convolve_2d_copy_sr = conv_copy_hip;
    t->svt_av1_highbd_convolve_2d_sr = conv_2d_hip_hbd; t->svt_av1_highbd_convolve_x_sr = conv_x_hip_hbd; t->svt_av1_highbd_convolve_y_sr = conv_y_hip_hbd;
    t->svt_av1_highbd_convolve_2d_copy_sr = conv_copy_hip_hbd;
#define X(I, TS, W, H) t->svt_av1_fwd_txfm2d[I] = fwd_hip<I, TS>;
    SVT_HIP_RTCD_FWD_SIZES(X)
#undef X
    t->svt_av1_inv_txfm2d_add_sq[0] = inv_sq_hip<0, 0>; t->svt_av1_inv_txfm2d_add_sq[1] = inv_sq_hip<1, 1>; t->svt_av1_inv_txfm2d_add_sq[2] = inv_sq_hip<2, 2>;
    t->svt_av1_inv_txfm2d_add_sq[3] = inv_sq_hip<3, 3>; t->svt_av1_inv_txfm2d_add_sq[4] = inv_sq_hip<4, 4>;
    t->svt_av1_inv_txfm2d_add_rect = inv_rect_hip;
    t->svt_av1_inv_txfm2d_add_rect4 = inv_rect4_hip;
    t->svt_av1_selfguided_restoration = sgr_filter_hip;
    t->svt_apply_selfguided_restoration = sgr_apply_hip;
#define X(I, W, H) t->svt_aom_obmc_sad[I] = obmc_sad_hip<I, W, H>; t->svt_aom_obmc_variance[I] = obmc_var_hip<I, W, H>; t->svt_aom_obmc_sub_pixel_variance[I] = obmc_subvar_hip<I, W, H>;
    SVT_HIP_RTCD_BLOCK_SIZES(X)
#undef X
    t->svt_aom_blend_a64_mask = blend_mask_hip; t->svt_aom_blend_a64_hmask = blend_hmask_hip; t->svt_aom_blend_a64_vmask = blend_vmask_hip;
    t->svt_aom_highbd_blend_a64_mask = blend_mask_hbd_hip; t->svt_aom_highbd_blend_a64_hmask_8bit = blend_hmask_hbd_hip; t->svt_aom_highbd_blend_a64_vmask_8bit = blend_vmask_hbd_hip;
    t->svt_av1_warp_affine = warp_hip; t->svt_av1_highbd_warp_affine = warp_hbd_hip;
    t->svt_av1_compute_stats = stats_hip; t->svt_av1_compute_stats_highbd = stats_hbd_hip;
#define X(I, TS, W, H) t->svt_av1_fwd_txfm2d_N2[I] = fwd_n2_hip<I, TS>; t->svt_av1_fwd_txfm2d_N4[I] = fwd_n4_hip<I, TS>;
    SVT_HIP_RTCD_FWD_SIZES(X)
#undef X
    t->svt_ext_all_sad_calculation_8x8_16x16 = ext_all_sad_hip; t->svt_ext_eight_sad_calculation_32x32_64x64 = ext_eight_sad_hip;
    t->svt_aom_quantize_b = quantize_b_hip; t->svt_aom_highbd_quantize_b = quantize_b_hbd_hip;
    t->svt_av1_quantize_fp = quantize_fp_hip<0>; t->svt_av1_quantize_fp_32x32 = quantize_fp_hip<1>; t->svt_av1_quantize_fp_64x64 = quantize_fp_hip<2>;
    t->svt_av1_highbd_quantize_fp = quantize_fp_hbd_hip;
#define X(LI) t->svt_aom_lpf_horizontal[LI] = lpf_hip<0, LI>; t->svt_aom_lpf_vertical[LI] = lpf_hip<1, LI>; \
              t->svt_aom_highbd_lpf_horizontal[LI] = lpf_hbd_hip<0, LI>; t->svt_aom_highbd_lpf_vertical[LI] = lpf_hbd_hip<1, LI>;
    X(0) X(1) X(2) X(3)
#undef X
    t->svt_cdef_find_dir = cdef_find_dir_hip; t->svt_cdef_filter_block = cdef_filter_block_hip;
    t->svt_residual_kernel8bit = residual8_hip; t->svt_residual_kernel16bit = residual16_hip;
#define X(I, W, H) t->svt_aom_sadx4d[I] = sadx4d_hip<I, W, H>;
    SVT_HIP_RTCD_BLOCK_SIZES(X)
#undef X
    t->svt_aom_upsampled_pred = upsampled_pred_hip;
    t->svt_compute_interm_var_four8x8 = interm_var_hip;
    t->svt_handle_transform64[0] = handle_transform_hip<0, 17>; t->svt_handle_transform64[1] = handle_transform_hip<1, 11>; t->svt_handle_transform64[2] = handle_transform_hip<2, 18>;
    t->svt_handle_transform64[3] = handle_transform_hip<3, 12>; t->svt_handle_transform64[4] = handle_transform_hip<4, 4>;
    t->svt_av1_inv_txfm_add = inv_txfm_add_hip;
    t->svt_aom_subtract_block = subtract_block_hip; t->svt_aom_highbd_subtract_block = subtract_block_hbd_hip;
    t->sad_16b_kernel = sad_16b_hip; t->variance_highbd = variance_highbd_hip; t->svt_nxm_sad_kernel_sub_sampled = nxm_sad_sub_hip;
    t->svt_ext_sad_calculation_8x8_16x16 = ext_sad_16_hip; t->svt_ext_sad_calculation_32x32_64x64 = ext_sad_32_64_hip;
    t->svt_copy_rect8_8bit_to_16bit = copy_rect8_hip;
    t->svt_compute_cdef_dist_8bit = cdef_dist8_hip; t->svt_compute_cdef_dist_16bit = cdef_dist16_hip; t->svt_search_one_dual = search_one_dual_hip;
    t->svt_full_distortion_kernel32_bits = full_dist32_hip; t->svt_full_distortion_kernel_cbf_zero32_bits = full_dist_cbf_zero32_hip;
    t->svt_spatial_full_distortion_kernel = spatial_dist_hip; t->svt_full_distortion_kernel16_bits = full_dist16_hip;
    t->svt_aom_sse = sse_hip; t->svt_aom_highbd_sse = sse_hbd_hip; t->svt_aom_satd = satd_hip; t->svt_av1_block_error = block_error_hip;
    t->svt_get_proj_subspace = get_proj_subspace_hip; t->svt_av1_lowbd_pixel_proj_error = pixel_proj_error_hip; t->svt_av1_highbd_pixel_proj_error = pixel_proj_error_hbd_hip;
    t->svt_compute_mean_square_values_8x8 = mean_sq_8x8_hip; t->svt_compute_sub_mean_8x8 = sub_mean_8x8_hip;
    t->svt_aom_convolve8_horiz = convolve8_horiz_hip; t->svt_aom_convolve8_vert = convolve8_vert_hip;
    t->svt_av1_wiener_convolve_add_src = wiener_convolve_hip; t->svt_av1_highbd_wiener_convolve_add_src = wiener_convolve_hbd_hip;
    t->handle_transform64_N2_N4[0] = handle_transform_n2n4_hip<0, 17>; t->handle_transform64_N2_N4[1] = handle_transform_n2n4_hip<1, 11>;
    t->handle_transform64_N2_N4[2] = handle_transform_n2n4_hip<2, 18>; t->handle_transform64_N2_N4[3] = handle_transform_n2n4_hip<3, 12>;
    t->handle_transform64_N2_N4[4] = handle_transform_n2n4_hip<4, 4>;
    t->svt_aom_mse16x16 = mse16x16_hip; t->svt_aom_highbd_8_mse16x16 = mse16x16_hbd8_hip;
    t->svt_convert_8bit_to_16bit = convert_8_to_16_hip; t->svt_convert_16bit_to_8bit = convert_16_to_8_hip; t->svt_c_pack = c_pack_hip;
    t->svt_compressed_packmsb = compressed_packmsb_hip; t->svt_pack2d_16_bit_src_mul4 = pack2d_hip; t->svt_unpack_avg = unpack_avg_hip;
    t->svt_un_pack2d_16_bit_src_mul4 = un_pack2d_hip; t->svt_un_pack8_bit_data = un_pack8_hip;
    t->svt_av1_jnt_convolve_2d = jnt_2d_hip; t->svt_av1_jnt_convolve_x = jnt_x_hip; t->svt_av1_jnt_convolve_y = jnt_y_hip; t->svt_av1_jnt_convolve_2d_copy = jnt_copy_hip;
    t->svt_av1_highbd_jnt_convolve_2d = jnt_2d_hip_hbd; t->svt_av1_highbd_jnt_convolve_x = jnt_x_hip_hbd; t->svt_av1_highbd_jnt_convolve_y = jnt_y_hip_hbd;
    t->svt_av1_highbd_jnt_convolve_2d_copy = jnt_copy_hip_hbd;
    t->svt_av1_build_compound_diffwtd_mask = diffwtd_mask_hip; t->svt_av1_build_compound_diffwtd_mask_highbd = diffwtd_mask_hbd_hip;
    t->svt_av1_build_compound_diffwtd_mask_d16 = diffwtd_mask_d16_hip; t->svt_aom_lowbd_blend_a64_d16_mask = blend_d16_hip; t->svt_aom_highbd_blend_a64_d16_mask = blend_d16_hbd_hip;
    return SVT_HIP_OK;
}
