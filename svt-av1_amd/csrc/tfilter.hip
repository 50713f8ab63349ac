// tfilter.hip — alt-ref temporal filtering, plane-wise strategy, for a whole picture and all frames of the window in one launch; gfx950.
//
// Replaces (file:line under /root/reference/Source/Lib/Encoder/Codec):
//   EbTemporalFiltering.c:643  svt_av1_apply_temporal_filter_planewise_c      (8-bit, per 32x32 block and reference frame)
//   EbTemporalFiltering.c:829  svt_av1_apply_temporal_filter_planewise_hbd_c  (16-bit)
//   EbTemporalFiltering.c:557  apply_filtering_central(_highbd)               (central picture, weight 1000)
//   EbTemporalFiltering.c:1943 get_final_filtered_pixels                      (normalise + filtered SSE)
//   EbTemporalFiltering.c:2414 estimate_noise / :2451 estimate_noise_highbd   (Laplacian noise estimate: sum and count)
//
// Shape: the reference walks 64x64 blocks, and inside each keeps a 3 x 4096 accumulator / counter pair that every frame of the window
// adds into, 32x32 by 32x32 (the 5x5 error window is clamped to the 32x32 block, so 32x32 blocks are fully independent).  Here one
// workgroup owns one 32x32 luma block (+ its chroma) for the WHOLE window: accum / count live in registers, the central picture is read
// once, each predictor once, the filtered picture is written once; no accumulator ever touches HBM (the reference layout would move
// 6 B / pixel / frame of them).  Algorithmic bytes: (n_refs + 1) reads + 1 write per sample.
//
// Arithmetic: squared differences and window sums are integers (LDS, separable 5 + 5).  The weight is
//   (int)(expf(-(min((5 * sum / num + block_error) / 6 * d_factor / (2 n_decay^2), 7))) * 1000)
// in the reference's double / float mix, evaluated here operation by operation in IEEE double (no contraction: the file is built with
// -ffp-contract=off, and AMDGPU's f64 divide / sqrt and f32<->f64 converts are correctly rounded).  expf is glibc's (the reference
// calls libm): its table + cubic algorithm is reproduced below; tools/expf_pin.c checks that it returns glibc 2.35's value for every
// float in [-7, 0], the only range the filter can produce.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "svt_hip_internal.h"

namespace {

constexpr int kMaxRefs = SVT_HIP_TF_MAX_REFS;

struct TfArgs {
    const void* src[3]; int src_stride[3];
    void* dst[3]; int dst_stride[3];
    const void* pred[kMaxRefs][3]; int pred_stride[kMaxRefs][3];
    const SvtHipTfBlk64* blocks[kMaxRefs];
    int n_refs, bc64, tf_chroma, sq_shift, hbd;
    double den[3];            // 2 * n_decay^2 per plane
    double dist_thr;          // max(min_frame_size * 0.1, 1)
    double rden[3], rdist_thr; // their correctly rounded reciprocals (host: 1.0 / y), for quot3
    unsigned long long* sse;  // [2]
};

// 2^(i/32) as glibc's __exp2f_data.tab stores it: bits(2^(i/32)) - (i << 47)
__device__ const unsigned long long kExp2Tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// glibc expf for x in [-7, 0] (sysdeps/ieee754/flt-32/e_expf.c, glibc 2.27+): x * 32 / ln2 = k + r, exp(x) = 2^(k/32) * (1 + c2 r + c1 r^2 + c0 r^3)
__device__ __forceinline__ float glibc_expf(float x, const unsigned long long* tab) {
    const double inv_ln2_n = 0x1.71547652b82fep+0 * 32, shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, c1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, c2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double xd = (double)x;
    double z = inv_ln2_n * xd;
    double kd = z + shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= shift;
    const double r = z - kd;
    const unsigned long long t = tab[ki & 31] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    z = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

// The weight divides three times per sample: by the window's sample count (25 .. 29), by 6 and by 2 n_decay^2.  An IEEE double division is a dozen
// instructions on this chip (v_div_scale x 2, v_rcp_f64, a Newton ladder of v_fma_f64, v_div_fmas, v_div_fixup), and the kernel is bound by its FP64 issue.
// With the divisor's correctly rounded reciprocal at hand, q = x r; e = fma(-y, q, x); q' = fma(e, r, q) IS the correctly rounded quotient (Markstein's
// correction step) — tools/div_pin.c shows it on the CPU's identical arithmetic: every uint32 dividend for y = 25, 26, 27, 29, and the hardest dividends (a few
// ulps around y Q and y (Q +- ulp / 2)) for y = 6 and for arbitrary y.  The theorem leaves out divisors whose significand is all ones; those, and divisors
// outside 2^-500 .. 2^500, take the IEEE division: the divisors are launch constants, so the launcher picks the kernel instance (FAST) on the host.
struct Recip { double y, r; };
__device__ __forceinline__ double quot3(double x, double y, double r) {
    const double q = x * r;
    const double e = __builtin_fma(-y, q, x);
    return __builtin_fma(e, r, q);
}
template <bool FAST> __device__ __forceinline__ double divide(double x, const Recip& R) { return FAST ? quot3(x, R.y, R.r) : x / R.y; }
static bool recip_ok(double y) {   // host: may this launch constant go through quot3?
    unsigned long long b; memcpy(&b, &y, 8);
    const int ex = (int)((b >> 52) & 0x7ff);
    return (b >> 63) == 0 && ex > 523 && ex < 1523 && (b & 0xFFFFFFFFFFFFFull) != 0xFFFFFFFFFFFFFull;
}

// EbTemporalFiltering.c:718-741: window error -> integer weight.  NUM = samples of the error window (25 luma; 25 + the co-located luma samples for chroma)
template <int NUM, bool FAST>
__device__ __forceinline__ int tf_weight(uint32_t sum, double block_error, double d_factor, const Recip& den, const unsigned long long* tab) {
    static_assert(NUM == 25 || NUM == 26 || NUM == 27 || NUM == 29, "tools/div_pin.c covers these sample counts");
    const double window_error = FAST ? quot3((double)sum, (double)NUM, 1.0 / (double)NUM) : (double)sum / (double)NUM;
    const double combined = FAST ? quot3(5.0 * window_error + block_error, 6.0, 1.0 / 6.0) : (5.0 * window_error + block_error) / 6.0;
    double scaled = divide<FAST>(combined * d_factor, den);
    scaled = scaled < 7.0 ? scaled : 7.0;
    return (int)(glibc_expf((float)(-scaled), tab) * 1000.0f);
}

// block error and motion-distance factor of sub-block `sub` of 32x32 block idx32 (EbTemporalFiltering.c:707-734 / :886-924)
template <bool FAST>
__device__ __forceinline__ void block_terms(const SvtHipTfBlk64* __restrict__ b, int idx32, int sub, int hbd, const Recip& dist_thr, double& block_error, double& d_factor) {
    unsigned long long err; int mvx, mvy; double div;
    if (b->split[idx32]) { err = b->err16[idx32 * 4 + sub]; mvx = b->mv16_x[idx32 * 4 + sub]; mvy = b->mv16_y[idx32 * 4 + sub]; div = 1.0 / 256.0; }
    else { err = b->err32[idx32]; mvx = b->mv32_x[idx32]; mvy = b->mv32_y[idx32]; div = 1.0 / 1024.0; }
    if (hbd) err >>= 4;
    block_error = (double)err * div;   // / 256 or / 1024: exact either way
    const float fr = (float)mvy, fc = (float)mvx;
    const float s = fr * fr + fc * fc;                         // powf(v, 2) is exact for a short; one rounding in the sum
    const float distance = (float)sqrt((double)s);            // == sqrtf(s): a correctly rounded f64 root rounds to the correctly rounded f32 root
    const double dd = divide<FAST>((double)distance, dist_thr);
    d_factor = dd > 1.0 ? dd : 1.0;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += ((unsigned long long)(uint32_t)__shfl_xor((int)(v >> 32), m, 64) << 32) | (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One workgroup = one 32x32 luma block and its chroma, all frames.  Luma: thread t owns row t >> 3, columns 4 (t & 7) .. + 3.
// Chroma (CW x CH = (32 >> SSX) x (32 >> SSY)): thread t owns samples t + 256 k, k < NC.
template <typename PIX, int SSX, int SSY, bool FAST>
__global__ void __launch_bounds__(256)
tf_filter_kernel(const TfArgs a) {
    constexpr int CW = 32 >> SSX, CH = 32 >> SSY, NC = (CW * CH) / 256;
    __shared__ uint32_t yd[32 * 32], yh[32 * 32];
    __shared__ uint32_t ud[CW * CH], vd[CW * CH], uh[CW * CH], vh[CW * CH];
    __shared__ unsigned long long tab[32];
    __shared__ unsigned long long sse_part[2][4];
    const int t = threadIdx.x;
    if (t < 32) tab[t] = kExp2Tab[t];
    const int bx = blockIdx.x, by = blockIdx.y;
    const int idx32 = (bx & 1) + (by & 1) * 2;
    const size_t blk64 = (size_t)(by >> 1) * a.bc64 + (bx >> 1);
    const int row = t >> 3, col0 = (t & 7) * 4;
    const int ysub = (row >= 16) * 2 + (col0 >= 16);
    const bool chroma = a.tf_chroma != 0;
    const int n_refs = a.n_refs, hbd = a.hbd, sq_shift = a.sq_shift;
    const Recip dist_thr = {a.dist_thr, a.rdist_thr}, den0 = {a.den[0], a.rden[0]}, den1 = {a.den[1], a.rden[1]}, den2 = {a.den[2], a.rden[2]};

    // ---- central picture samples of this thread
    int ys[4], us[NC], vs[NC];
    {
        const PIX* p = (const PIX*)a.src[0] + (size_t)(by * 32 + row) * a.src_stride[0] + bx * 32 + col0;
#pragma unroll
        for (int i = 0; i < 4; i++) ys[i] = p[i];
    }
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int idx = t + 256 * k, r = idx / CW, c = idx % CW;
        us[k] = vs[k] = 0;
        if (chroma) {
            us[k] = ((const PIX*)a.src[1])[(size_t)(by * CH + r) * a.src_stride[1] + bx * CW + c];
            vs[k] = ((const PIX*)a.src[2])[(size_t)(by * CH + r) * a.src_stride[2] + bx * CW + c];
        }
    }
    uint32_t yacc[4] = {0, 0, 0, 0}, ycnt[4] = {0, 0, 0, 0}, uacc[NC], ucnt[NC], vacc[NC], vcnt[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) uacc[k] = ucnt[k] = vacc[k] = vcnt[k] = 0;

    for (int f = 0; f < n_refs; f++) {
        const SvtHipTfBlk64* blocks = a.blocks[f];
        if (!blocks) {   // apply_filtering_central: weight TF_PLANEWISE_FILTER_WEIGHT_SCALE on the picture itself
#pragma unroll
            for (int i = 0; i < 4; i++) { yacc[i] += 1000u * (uint32_t)ys[i]; ycnt[i] = (ycnt[i] + 1000u) & 0xffffu; }
            if (chroma) {
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    uacc[k] += 1000u * (uint32_t)us[k]; ucnt[k] = (ucnt[k] + 1000u) & 0xffffu;
                    vacc[k] += 1000u * (uint32_t)vs[k]; vcnt[k] = (vcnt[k] + 1000u) & 0xffffu;
                }
            }
            continue;
        }
        const PIX* py = (const PIX*)a.pred[f][0]; const int sy = a.pred_stride[f][0];
        const PIX* pu = (const PIX*)a.pred[f][1]; const int su = a.pred_stride[f][1];
        const PIX* pv = (const PIX*)a.pred[f][2]; const int sv = a.pred_stride[f][2];
        // ---- squared differences (calculate_squared_errors) and predictor samples
        int yp[4], up[NC], vp[NC];
        {
            const PIX* p = py + (size_t)(by * 32 + row) * sy + bx * 32 + col0;
#pragma unroll
            for (int i = 0; i < 4; i++) { yp[i] = p[i]; const int d = ys[i] - yp[i]; yd[row * 32 + col0 + i] = (uint32_t)(d * d); }
        }
        if (chroma) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int idx = t + 256 * k, r = idx / CW, c = idx % CW;
                up[k] = pu[(size_t)(by * CH + r) * su + bx * CW + c];
                vp[k] = pv[(size_t)(by * CH + r) * sv + bx * CW + c];
                const int du = us[k] - up[k], dv = vs[k] - vp[k];
                ud[idx] = (uint32_t)(du * du); vd[idx] = (uint32_t)(dv * dv);
            }
        }
        __syncthreads();
        // ---- horizontal 5-sums, columns clamped to the block
        {
            uint32_t w8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w8[i] = yd[row * 32 + clampi(col0 - 2 + i, 0, 31)];
#pragma unroll
            for (int i = 0; i < 4; i++) yh[row * 32 + col0 + i] = w8[i] + w8[i + 1] + w8[i + 2] + w8[i + 3] + w8[i + 4];
        }
        if (chroma) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int idx = t + 256 * k, r = idx / CW, c = idx % CW;
                uint32_t su5 = 0, sv5 = 0;
#pragma unroll
                for (int dx = -2; dx <= 2; dx++) { const int o = r * CW + clampi(c + dx, 0, CW - 1); su5 += ud[o]; sv5 += vd[o]; }
                uh[idx] = su5; vh[idx] = sv5;
            }
        }
        __syncthreads();
        // ---- vertical 5-sums, weights, accumulate
        const SvtHipTfBlk64* b = blocks + blk64;
        {
            double be, df;
            block_terms<FAST>(b, idx32, ysub, hbd, dist_thr, be, df);
            uint32_t s4[4] = {0, 0, 0, 0};
#pragma unroll
            for (int dy = -2; dy <= 2; dy++) {
                const int rr = clampi(row + dy, 0, 31);
#pragma unroll
                for (int i = 0; i < 4; i++) s4[i] += yh[rr * 32 + col0 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int w = tf_weight<25, FAST>(s4[i] >> sq_shift, be, df, den0, tab);
                ycnt[i] = (ycnt[i] + (uint32_t)w) & 0xffffu; yacc[i] += (uint32_t)(w * yp[i]);
            }
        }
        if (chroma) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int idx = t + 256 * k, r = idx / CW, c = idx % CW;
                const int li = r << SSY, lj = c << SSX;                      // the luma sample this chroma sample is filtered with (:746)
                double be, df;
                block_terms<FAST>(b, idx32, (li >= 16) * 2 + (lj >= 16), hbd, dist_thr, be, df);
                uint32_t ysum = 0;
#pragma unroll
                for (int dy = 0; dy < (1 << SSY); dy++)
#pragma unroll
                    for (int dx = 0; dx < (1 << SSX); dx++) ysum += yd[(li + dy) * 32 + lj + dx];
                uint32_t usum = ysum, vsum = ysum;
#pragma unroll
                for (int dy = -2; dy <= 2; dy++) { const int o = clampi(r + dy, 0, CH - 1) * CW + c; usum += uh[o]; vsum += vh[o]; }
                constexpr int num = 25 + (1 << SSX) * (1 << SSY);
                const int wu = tf_weight<num, FAST>(usum >> sq_shift, be, df, den1, tab);
                const int wv = tf_weight<num, FAST>(vsum >> sq_shift, be, df, den2, tab);
                ucnt[k] = (ucnt[k] + (uint32_t)wu) & 0xffffu; uacc[k] += (uint32_t)(wu * up[k]);
                vcnt[k] = (vcnt[k] + (uint32_t)wv) & 0xffffu; vacc[k] += (uint32_t)(wv * vp[k]);
            }
        }
        __syncthreads();
    }

    // ---- get_final_filtered_pixels: OD_DIVU(accum + (count >> 1), count) == plain division (count >= 1000 once the central picture is in)
    unsigned long long sse_y = 0, sse_c = 0;
    {
        PIX* o = (PIX*)a.dst[0] + (size_t)(by * 32 + row) * a.dst_stride[0] + bx * 32 + col0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t cnt = ycnt[i] ? ycnt[i] : 1u;
            const int v = (int)((yacc[i] + (cnt >> 1)) / cnt);
            const int d = ys[i] - v;
            sse_y += (unsigned long long)(d * d);
            o[i] = (PIX)v;
        }
    }
    if (chroma) {
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int idx = t + 256 * k, r = idx / CW, c = idx % CW;
            const uint32_t cu = ucnt[k] ? ucnt[k] : 1u, cv = vcnt[k] ? vcnt[k] : 1u;
            const int u = (int)((uacc[k] + (cu >> 1)) / cu), v = (int)((vacc[k] + (cv >> 1)) / cv);
            const int du = us[k] - u, dv = vs[k] - v;
            sse_c += (unsigned long long)(du * du) + (unsigned long long)(dv * dv);
            ((PIX*)a.dst[1])[(size_t)(by * CH + r) * a.dst_stride[1] + bx * CW + c] = (PIX)u;
            ((PIX*)a.dst[2])[(size_t)(by * CH + r) * a.dst_stride[2] + bx * CW + c] = (PIX)v;
        }
    }
    sse_y = wave_sum_u64(sse_y); sse_c = wave_sum_u64(sse_c);
    if ((t & 63) == 0) { sse_part[0][t >> 6] = sse_y; sse_part[1][t >> 6] = sse_c; }
    __syncthreads();
    if (t < 2 && a.sse) {
        const unsigned long long s = sse_part[t][0] + sse_part[t][1] + sse_part[t][2] + sse_part[t][3];
        if (s) atomicAdd(a.sse + t, s);
    }
}

// estimate_noise(_highbd): out[0] += sum |laplacian| over smooth pixels, out[1] += their count.  One row per workgroup pass.
template <typename PIX>
__global__ void __launch_bounds__(256)
tf_noise_kernel(const PIX* __restrict__ src, int width, int height, int stride, int sh, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[2][4];
    unsigned long long sum = 0, num = 0;
    const int rnd = sh ? (1 << sh) >> 1 : 0;
    for (int i = 1 + blockIdx.x; i < height - 1; i += gridDim.x)
        for (int j = 1 + threadIdx.x; j < width - 1; j += 256) {
            const PIX* p = src + (size_t)i * stride + j;
            const int a = p[-stride - 1], b = p[-stride], c = p[-stride + 1], d = p[-1], e = p[0], f = p[1], g = p[stride - 1], h = p[stride], k = p[stride + 1];
            const int gx = (a - c) + (g - k) + 2 * (d - f);
            const int gy = (a - g) + (c - k) + 2 * (b - h);
            const int ga = (abs(gx) + abs(gy) + rnd) >> sh;
            if (ga < 50) {
                const int v = 4 * e - 2 * (d + f + b + h) + (a + c + g + k);
                sum += (unsigned long long)((abs(v) + rnd) >> sh); num++;
            }
        }
    sum = wave_sum_u64(sum); num = wave_sum_u64(num);
    const int t = threadIdx.x;
    if ((t & 63) == 0) { part[0][t >> 6] = sum; part[1][t >> 6] = num; }
    __syncthreads();
    if (t < 2) {
        const unsigned long long s = part[t][0] + part[t][1] + part[t][2] + part[t][3];
        if (s) atomicAdd(out + t, s);
    }
}

template <typename PIX>
int launch_filter(hipStream_t st, const TfArgs& a, int w, int h, int ss_x, int ss_y) {
    const dim3 grid(w / 32, h / 32), block(256);
    static const char* div_env = getenv("SVT_HIP_TF_DIV");   // "ieee": the IEEE divisions everywhere (A/B runs, tools/tf_time.py)
    const bool fast = recip_ok(a.den[0]) && recip_ok(a.den[1]) && recip_ok(a.den[2]) && recip_ok(a.dist_thr) && !(div_env && !strcmp(div_env, "ieee"));
#define TF_LAUNCH(SX, SY) do { if (fast) hipLaunchKernelGGL((tf_filter_kernel<PIX, SX, SY, true>), grid, block, 0, st, a); \
                               else hipLaunchKernelGGL((tf_filter_kernel<PIX, SX, SY, false>), grid, block, 0, st, a); } while (0)
    if (ss_x == 1 && ss_y == 1) TF_LAUNCH(1, 1);
    else if (ss_x == 1 && ss_y == 0) TF_LAUNCH(1, 0);
    else TF_LAUNCH(0, 0);
#undef TF_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int svt_hip_launch_tf_filter(hipStream_t st, int pix_bytes, int bd, const void* const src[3], const int src_stride[3], void* const dst[3],
                                        const int dst_stride[3], int w, int h, int ss_x, int ss_y, int tf_chroma, const SvtHipTfRef* refs, int n_refs,
                                        const double den[3], double dist_thr, uint64_t* sse) {
    TfArgs a;
    for (int p = 0; p < 3; p++) { a.src[p] = src[p]; a.src_stride[p] = src_stride[p]; a.dst[p] = dst[p]; a.dst_stride[p] = dst_stride[p]; a.den[p] = den[p]; a.rden[p] = 1.0 / den[p]; }
    for (int f = 0; f < kMaxRefs; f++) {
        for (int p = 0; p < 3; p++) { a.pred[f][p] = f < n_refs ? refs[f].pred[p] : nullptr; a.pred_stride[f][p] = f < n_refs ? refs[f].pred_stride[p] : 0; }
        a.blocks[f] = f < n_refs ? refs[f].blocks : nullptr;
    }
    a.n_refs = n_refs; a.bc64 = w / 64; a.tf_chroma = tf_chroma; a.hbd = pix_bytes == 2; a.sq_shift = pix_bytes == 2 ? (bd - 8) * 2 : 0;
    a.dist_thr = dist_thr; a.rdist_thr = 1.0 / dist_thr; a.sse = (unsigned long long*)sse;
    return pix_bytes == 1 ? launch_filter<uint8_t>(st, a, w, h, ss_x, ss_y) : launch_filter<uint16_t>(st, a, w, h, ss_x, ss_y);
}

extern "C" int svt_hip_launch_tf_noise(hipStream_t st, const void* src, int pix_bytes, int bd, int width, int height, int stride, uint64_t* out) {
    const int rows = height - 2 > 0 ? height - 2 : 1;
    const int grid = rows < 2048 ? rows : 2048;
    if (pix_bytes == 1) hipLaunchKernelGGL(tf_noise_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, (const uint8_t*)src, width, height, stride, 0, (unsigned long long*)out);
    else hipLaunchKernelGGL(tf_noise_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)src, width, height, stride, bd - 8, (unsigned long long*)out);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(tfilter)
