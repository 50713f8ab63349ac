/*
 * warp_oracle.c — CPU restatement of AV1 warped (affine) prediction, single reference.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Pinned by tests/test_oracle_vs_ref.py against svt_av1_warp_affine_c /
 * svt_av1_highbd_warp_affine_c (Common/Codec/EbWarpedMotion.c:577-694, :733-842), non-compound path.
 */
#include <stdint.h>
#include <stddef.h>
#include "svt_oracle.h"
#include "warp_filter_table.h"

static const int16_t warped_filter[193][8] = SVT_WARPED_FILTER_TABLE;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int rp2(int v, int n) { return n == 0 ? v : ((v + ((1 << n) >> 1)) >> n); }
static inline int rd(const void *p, int pb, size_t i) { return pb == 1 ? ((const uint8_t *)p)[i] : ((const uint16_t *)p)[i]; }

/* one block (p_col, p_row, p_width, p_height) of a plane under the affine model mat[6] / shear alpha, beta, gamma, delta;
 * round_0 = 3 (5 at 12 bits), non-compound: reduce_bits_vert = 2 * FILTER_BITS - reduce_bits_horiz */
void orc_warp_affine(const int32_t *mat, const void *ref, int pix_bytes, int bd, int width, int height, int stride, void *pred, int p_col, int p_row, int p_width,
                     int p_height, int p_stride, int ss_x, int ss_y, int alpha, int beta, int gamma, int delta) {
    const int round_0 = bd == 12 ? 5 : 3;
    const int extra = bd + 7 - round_0 - 14;
    const int rbh = pix_bytes == 1 ? round_0 : round_0 + (extra > 0 ? extra : 0);      /* :584 / :739-740 */
    const int rbv = 14 - rbh, obh = bd + 6, obv = bd + 14 - rbh;
    for (int i = p_row; i < p_row + p_height; i += 8)
        for (int j = p_col; j < p_col + p_width; j += 8) {
            int32_t tmp[15 * 8];
            const int32_t src_x = (j + 4) << ss_x, src_y = (i + 4) << ss_y;
            const int32_t dst_x = mat[2] * src_x + mat[3] * src_y + mat[0], dst_y = mat[4] * src_x + mat[5] * src_y + mat[1];
            const int32_t x4 = dst_x >> ss_x, y4 = dst_y >> ss_y;
            const int32_t ix4 = x4 >> 16, iy4 = y4 >> 16;
            int32_t sx4 = x4 & 0xffff, sy4 = y4 & 0xffff;
            sx4 += alpha * (-4) + beta * (-4); sy4 += gamma * (-4) + delta * (-4);
            sx4 &= ~63; sy4 &= ~63;
            for (int k = -7; k < 8; k++) {
                const int iy = clampi(iy4 + k, 0, height - 1);
                int sx = sx4 + beta * (k + 4);
                for (int l = -4; l < 4; l++) {
                    const int ix = ix4 + l - 3;
                    const int16_t *c = warped_filter[rp2(sx, 10) + 64];
                    int32_t sum = 1 << obh;
                    for (int m = 0; m < 8; m++) sum += rd(ref, pix_bytes, (size_t)iy * stride + clampi(ix + m, 0, width - 1)) * c[m];
                    tmp[(k + 7) * 8 + (l + 4)] = rp2(sum, rbh);
                    sx += alpha;
                }
            }
            const int kmax = p_row + p_height - i - 4 < 4 ? p_row + p_height - i - 4 : 4, lmax = p_col + p_width - j - 4 < 4 ? p_col + p_width - j - 4 : 4;
            for (int k = -4; k < kmax; k++) {
                int sy = sy4 + delta * (k + 4);
                for (int l = -4; l < lmax; l++) {
                    const int16_t *c = warped_filter[rp2(sy, 10) + 64];
                    int32_t sum = 1 << obv;
                    for (int m = 0; m < 8; m++) sum += tmp[(k + m + 4) * 8 + (l + 4)] * c[m];
                    sum = rp2(sum, rbv) - (1 << (bd - 1)) - (1 << bd);
                    sum = clampi(sum, 0, (1 << bd) - 1);
                    const size_t o = (size_t)(i - p_row + k + 4) * p_stride + (j - p_col + l + 4);
                    if (pix_bytes == 1) ((uint8_t *)pred)[o] = (uint8_t)sum; else ((uint16_t *)pred)[o] = (uint16_t)sum;
                    sy += gamma;
                }
            }
        }
}

/* the block list of svt_hip_warp_predict_batch_dev (include/svt_hip.h, SvtHipWarpBlk): pred is the destination PLANE, block at (p_col, p_row) */
typedef struct { int32_t mat[6]; int16_t alpha, beta, gamma, delta; int32_t p_col, p_row; uint8_t p_width, p_height, reserved[2]; } OrcWarpBlk;
void orc_warp_predict_batch(int pix_bytes, int bd, const void *ref, int width, int height, int stride, void *dst, int dst_stride, int ss_x, int ss_y,
                            const void *blks_, int n) {
    const OrcWarpBlk *b = (const OrcWarpBlk *)blks_;
    for (int i = 0; i < n; i++)
        orc_warp_affine(b[i].mat, ref, pix_bytes, bd, width, height, stride, (uint8_t *)dst + ((size_t)b[i].p_row * dst_stride + b[i].p_col) * pix_bytes, b[i].p_col,
                        b[i].p_row, b[i].p_width, b[i].p_height, dst_stride, ss_x, ss_y, b[i].alpha, b[i].beta, b[i].gamma, b[i].delta);
}
