/*
 * pyramid_oracle.c — CPU restatement of the picture-analysis kernels that feed open-loop ME:
 * 2-D decimation / 2x2 down-sampling for the HME pyramids and the per-SB mean / variance pyramid.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 */
#include "svt_oracle.h"

/* decimation_2d (Encoder/Codec/EbPictureAnalysisProcess.c:193-216) / downsample_2d (:223-255).
 * step = 2 or 4; output is (w/step) x (h/step). */
void orc_downsample_2d(const uint8_t *in, int in_stride, int w, int h, uint8_t *out, int out_stride, int step, int filtered) {
    if (!filtered) {
        for (int y = 0; y < h; y += step)
            for (int x = 0; x < w; x += step) out[(y / step) * out_stride + (x >> (step >> 1))] = in[(size_t)y * in_stride + x];
        return;
    }
    const int half = step >> 1;
    for (int y = half, oy = 0; y < h; y += step, oy++)
        for (int x = half, ox = 0; x < w; x += step, ox++) {
            const unsigned s = in[(size_t)(y - 1) * in_stride + x - 1] + in[(size_t)(y - 1) * in_stride + x] +
                               in[(size_t)y * in_stride + x - 1] + in[(size_t)y * in_stride + x];
            out[oy * out_stride + ox] = (uint8_t)((s + 2) >> 2);
        }
}

/* compute_block_mean_compute_variance (EbPictureAnalysisProcess.c:1005-2575) for one 64x64 SB.
 * full_precision 0 = BLOCK_MEAN_PREC_SUB (default, EbSequenceControlSet.c:192): rows 0,2,4,6 only
 * (svt_compute_sub_mean_8x8_c :310, compute_sub_mean_squared_values_c :329); 1 = BLOCK_MEAN_PREC_FULL
 * (compute_mean_8x8 / svt_compute_mean_squared_values_c :287).  Output layout [85]: [0] 64x64,
 * [1..4] 32x32, [5..20] 16x16, [21..84] 8x8, each level in RASTER order (as the reference writes
 * pcs->y_mean / pcs->variance). */
void orc_variance_pyramid_sb(const uint8_t *sb, int stride, int full_precision, uint8_t mean_out[85], uint16_t var_out[85]) {
    uint64_t m8[64], q8[64], m16[16], q16[16], m32[4], q32[4], m64, q64;
    for (int b = 0; b < 64; b++) {
        const uint8_t *p = sb + (size_t)(b >> 3) * 8 * stride + (b & 7) * 8;
        uint64_t s = 0, s2 = 0;
        for (int y = 0; y < 8; y += full_precision ? 1 : 2)
            for (int x = 0; x < 8; x++) { const unsigned v = p[(size_t)y * stride + x]; s += v; s2 += v * v; }
        if (full_precision) { m8[b] = (s << 8) / 64; q8[b] = (s2 << 16) / 64; }
        else                { m8[b] = s << 3;        q8[b] = s2 << 11; }
    }
    for (int Y = 0; Y < 4; Y++)
        for (int X = 0; X < 4; X++) {
            const int b = 16 * Y + 2 * X;
            m16[4 * Y + X] = (m8[b] + m8[b + 1] + m8[b + 8] + m8[b + 9]) >> 2;
            q16[4 * Y + X] = (q8[b] + q8[b + 1] + q8[b + 8] + q8[b + 9]) >> 2;
        }
    for (int Y = 0; Y < 2; Y++)
        for (int X = 0; X < 2; X++) {
            const int b = 8 * Y + 2 * X;
            m32[2 * Y + X] = (m16[b] + m16[b + 1] + m16[b + 4] + m16[b + 5]) >> 2;
            q32[2 * Y + X] = (q16[b] + q16[b + 1] + q16[b + 4] + q16[b + 5]) >> 2;
        }
    m64 = (m32[0] + m32[1] + m32[2] + m32[3]) >> 2;
    q64 = (q32[0] + q32[1] + q32[2] + q32[3]) >> 2;
#define PUT(i, m, q) do { mean_out[i] = (uint8_t)((m) >> 8); var_out[i] = (uint16_t)(((q) - (m) * (m)) >> 16); } while (0)
    PUT(0, m64, q64);
    for (int i = 0; i < 4; i++) PUT(1 + i, m32[i], q32[i]);
    for (int i = 0; i < 16; i++) PUT(5 + i, m16[i], q16[i]);
    for (int i = 0; i < 64; i++) PUT(21 + i, m8[i], q8[i]);
#undef PUT
}

/* The HME batch the GPU entry point svt_hip_sad_loop_batch_dev receives (include/svt_hip.h, SvtHipSadLoop), walked with the
 * single-search restatement above: jobs [begin, end).  Used by tests and by bench.py's CPU baseline (one C call per thread). */
typedef struct {
    int32_t src_x, src_y, ref_x, ref_y;
    int16_t bw, bh, sa_w, sa_h, row_step, reserved;
} OrcSadLoopJob;
void orc_sad_loop_batch(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, const void *jobs_, int begin, int end,
                        uint32_t *best_sad, int16_t *best_xy) {
    const OrcSadLoopJob *jobs = (const OrcSadLoopJob *)jobs_;
    for (int i = begin; i < end; i++) {
        const OrcSadLoopJob *j = &jobs[i];
        uint64_t bs = 0xffffff;
        int16_t xc = 0, yc = 0;
        if (j->sa_w > 0 && j->sa_h > 0)
            orc_sad_loop(src + (size_t)j->src_y * src_stride + j->src_x, (uint32_t)(src_stride * j->row_step), ref + (size_t)j->ref_y * ref_stride + j->ref_x,
                         (uint32_t)(ref_stride * j->row_step), (uint32_t)(j->bh / j->row_step), (uint32_t)j->bw, &bs, &xc, &yc, (uint32_t)ref_stride, j->sa_w, j->sa_h);
        best_sad[i] = (uint32_t)bs;
        if (bs != 0xffffff) { best_xy[2 * i] = xc; best_xy[2 * i + 1] = yc; }
    }
}

/* The 16-bit twin of orc_sad_loop_batch for the high-bit-depth path: svt_sad_loop_kernel_c's candidate loop (Encoder/C_DEFAULT/EbComputeSAD_C.c:58-92:
 * raster order, strict '<', initial best 0xffffff) around sad_16b_kernel_c (:39). */
void orc_sad_loop16_batch(const uint16_t *src, int src_stride, const uint16_t *ref, int ref_stride, const void *jobs_, int begin, int end, uint32_t *best_sad,
                          int16_t *best_xy) {
    const OrcSadLoopJob *jobs = (const OrcSadLoopJob *)jobs_;
    for (int i = begin; i < end; i++) {
        const OrcSadLoopJob *j = &jobs[i];
        uint32_t bs = 0xffffff;
        int16_t xc = 0, yc = 0;
        for (int cy = 0; cy < j->sa_h; cy++)
            for (int cx = 0; cx < j->sa_w; cx++) {
                const uint32_t sad = orc_sad_16b(src + (size_t)j->src_y * src_stride + j->src_x, (uint32_t)(src_stride * j->row_step),
                                                 ref + (size_t)(j->ref_y + cy) * ref_stride + j->ref_x + cx, (uint32_t)(ref_stride * j->row_step),
                                                 (uint32_t)(j->bh / j->row_step), (uint32_t)j->bw);
                if (sad < bs) { bs = sad; xc = (int16_t)cx; yc = (int16_t)cy; }
            }
        best_sad[i] = bs;
        if (bs != 0xffffff) { best_xy[2 * i] = xc; best_xy[2 * i + 1] = yc; }
    }
}
