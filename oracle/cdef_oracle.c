/*
 * cdef_oracle.c — CPU restatement of SVT-AV1's CDEF: direction search, 8x8 / 4x4 block filter,
 * per-64x64 filter-block strength search (distortion table) and frame application.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 *
 * The reference stages every 64x64 filter block (fb) into a uint16 buffer with a 3-row / 8-column
 * halo in which everything outside the picture is CDEF_VERY_LARGE (Encoder/Codec/EbCdefProcess.c:
 * 210-226 for the search, Encoder/Codec/EbEncCdef.c:353-476 line/column buffers for the apply —
 * both amount to "neighbours are the pre-CDEF picture, outside the picture is VERY_LARGE").  Here
 * the same thing is expressed with a sample accessor instead of the staging buffer.
 */
#include "svt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define VERY_LARGE 16384 /* CDEF_VERY_LARGE, Common/Codec/EbCdef.h:37 */

static inline int msb(unsigned n) { int l = 0; while (n >>= 1) l++; return l; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Common/Codec/EbCdef.c:87-93 */
static int constrain(int diff, int threshold, int damping) {
    if (!threshold) return 0;
    const int shift = imax(0, damping - msb((unsigned)threshold));
    const int a = abs(diff);
    const int v = imin(a, imax(0, threshold - (a >> shift)));
    return diff < 0 ? -v : v;
}
/* Common/Codec/EbCdef.c:112-116 */
int orc_cdef_adjust_strength(int strength, int var) {
    const int i = (var >> 6) ? imin(msb((unsigned)(var >> 6)), 12) : 0;
    return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

/* Common/Codec/EbCdef.c:132-196 (svt_cdef_find_dir_c) */
int orc_cdef_find_dir(const uint16_t *img, int stride, int32_t *var, int coeff_shift) {
    int32_t cost[8] = {0}, partial[8][15];
    static const int div_table[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
    memset(partial, 0, sizeof(partial));
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const int x = (img[i * stride + j] >> coeff_shift) - 128;
            partial[0][i + j] += x;
            partial[1][i + j / 2] += x;
            partial[2][i] += x;
            partial[3][3 + i - j / 2] += x;
            partial[4][7 + i - j] += x;
            partial[5][3 - i / 2 + j] += x;
            partial[6][j] += x;
            partial[7][i / 2 + j] += x;
        }
    for (int i = 0; i < 8; i++) {
        cost[2] += partial[2][i] * partial[2][i];
        cost[6] += partial[6][i] * partial[6][i];
    }
    cost[2] *= div_table[8];
    cost[6] *= div_table[8];
    for (int i = 0; i < 7; i++) {
        cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * div_table[i + 1];
        cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * div_table[i + 1];
    }
    cost[0] += partial[0][7] * partial[0][7] * div_table[8];
    cost[4] += partial[4][7] * partial[4][7] * div_table[8];
    for (int i = 1; i < 8; i += 2) {
        for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
        cost[i] *= div_table[8];
        for (int j = 0; j < 3; j++)
            cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * div_table[2 * j + 2];
    }
    int best = 0, best_cost = 0;
    for (int i = 0; i < 8; i++)
        if (cost[i] > best_cost) { best_cost = cost[i]; best = i; }
    *var = (best_cost - cost[(best + 4) & 7]) >> 10;
    return best;
}

/* direction offsets as (dy, dx) for k = 0, 1: eb_cdef_directions, Common/Codec/EbCdef.c:96-104 */
static const int8_t dir_dy[8][2] = {{-1, -2}, {0, -1}, {0, 0}, {0, 1}, {1, 2}, {1, 2}, {1, 2}, {1, 2}};
static const int8_t dir_dx[8][2] = {{1, 2}, {1, 2}, {1, 2}, {1, 2}, {1, 2}, {0, 1}, {0, 0}, {0, -1}};

/* Common/Codec/EbCdef.c:202-257 (svt_cdef_filter_block_c); `in` has row stride `istride`
 * (CDEF_BSTRIDE in the reference), block is bw x bh samples. */
void orc_cdef_filter_block(uint8_t *dst8, uint16_t *dst16, int dstride, const uint16_t *in, int istride, int pri_strength,
                           int sec_strength, int dir, int pri_damping, int sec_damping, int bw, int bh, int coeff_shift) {
    static const int pri_taps_t[2][2] = {{4, 2}, {3, 3}}, sec_taps_t[2] = {2, 1};
    const int *pt = pri_taps_t[(pri_strength >> coeff_shift) & 1];
    for (int i = 0; i < bh; i++)
        for (int j = 0; j < bw; j++) {
            const int16_t x = (int16_t)in[i * istride + j];
            int16_t sum = 0;
            int mx = x, mn = x;
            for (int k = 0; k < 2; k++) {
                const int o0 = dir_dy[dir][k] * istride + dir_dx[dir][k];
                const int16_t p0 = (int16_t)in[i * istride + j + o0], p1 = (int16_t)in[i * istride + j - o0];
                sum = (int16_t)(sum + (int16_t)(pt[k] * constrain(p0 - x, pri_strength, pri_damping)));
                sum = (int16_t)(sum + (int16_t)(pt[k] * constrain(p1 - x, pri_strength, pri_damping)));
                if (p0 != VERY_LARGE) mx = imax(p0, mx);
                if (p1 != VERY_LARGE) mx = imax(p1, mx);
                mn = imin(p0, mn);
                mn = imin(p1, mn);
                const int d2 = (dir + 2) & 7, d6 = (dir + 6) & 7;
                const int o2 = dir_dy[d2][k] * istride + dir_dx[d2][k], o6 = dir_dy[d6][k] * istride + dir_dx[d6][k];
                const int16_t s[4] = {(int16_t)in[i * istride + j + o2], (int16_t)in[i * istride + j - o2],
                                      (int16_t)in[i * istride + j + o6], (int16_t)in[i * istride + j - o6]};
                for (int t = 0; t < 4; t++) {
                    if (s[t] != VERY_LARGE) mx = imax(s[t], mx);
                    mn = imin(s[t], mn);
                    sum = (int16_t)(sum + (int16_t)(sec_taps_t[k] * constrain(s[t] - x, sec_strength, sec_damping)));
                }
            }
            int y = x + ((8 + sum - (sum < 0)) >> 4);
            y = y < mn ? mn : (y > mx ? mx : y);
            if (dst8) dst8[i * dstride + j] = (uint8_t)y;
            else dst16[i * dstride + j] = (uint16_t)y;
        }
}

/* luma perceptual distortion of one 8x8 block: dist_8x8_{8bit,16bit}_c, Encoder/Codec/EbEncCdef.c:25-52,79-105.
 * `a` / `b` play the roles of the reference's (src = filtered, dst = source picture); the formula is symmetric. */
uint64_t orc_cdef_dist_8x8(const uint16_t *a, int astride, const uint16_t *b, int bstride, int coeff_shift) {
    uint64_t sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const uint64_t s = a[i * astride + j], d = b[i * bstride + j];
            sum_s += s; sum_d += d; sum_s2 += s * s; sum_d2 += d * d; sum_sd += s * d;
        }
    const uint64_t svar = sum_s2 - ((sum_s * sum_s + 32) >> 6), dvar = sum_d2 - ((sum_d * sum_d + 32) >> 6);
    return (uint64_t)floor(.5 + (sum_d2 + sum_s2 - 2 * sum_sd) * .5 * (svar + dvar + (400 << 2 * coeff_shift)) /
                                    (sqrt((20000 << 4 * coeff_shift) + svar * (double)dvar)));
}

typedef struct {
    const void *plane; int pix_bytes, stride, w, h; /* plane picture, visible size */
} PlaneView;
static inline int sample(const PlaneView *p, int x, int y) {
    if (x < 0 || y < 0 || x >= p->w || y >= p->h) return VERY_LARGE;
    return p->pix_bytes == 1 ? ((const uint8_t *)p->plane)[(size_t)y * p->stride + x] : ((const uint16_t *)p->plane)[(size_t)y * p->stride + x];
}
/* stage the (bw+16) x (bh+6) neighbourhood of a block like the reference's `in` buffer */
static void stage_block(const PlaneView *p, int x0, int y0, int bw, int bh, uint16_t *buf, int bstride) {
    for (int y = -3; y < bh + 3; y++)
        for (int x = -8; x < bw + 8; x++) buf[(y + 3) * bstride + x + 8] = (uint16_t)sample(p, x0 + x, y0 + y);
}

/* strength index -> (pri, sec) : get_cdef_filter_strengths, Common/Codec/EbDefinitions.h:1696-1718;
 * pick_method 0 = full (64), 1..3 = reduced sets (32 / 20 / 10) */
int orc_cdef_strength_count(int pick_method) { static const int n[4] = {64, 32, 20, 10}; return n[pick_method]; }
void orc_cdef_strength(int pick_method, int gi, int *pri, int *sec) {
    static const int priconv1[8] = {0, 1, 2, 3, 5, 7, 10, 13}, priconv2[5] = {0, 2, 4, 8, 14}, secconv3[2] = {0, 2};
    const int tot_sec = pick_method == 3 ? 2 : 4;
    const int pi = gi / tot_sec, si = gi % tot_sec;
    *pri = pi; *sec = si;
    if (pick_method == 1) *pri = priconv1[pi];
    else if (pick_method == 2) *pri = priconv2[pi];
    else if (pick_method == 3) { *pri = priconv2[pi]; *sec = secconv3[si]; }
}

/* Search: cdef_seg_search, Encoder/Codec/EbCdefProcess.c:80-280 (8-bit) / :281-475 (16-bit), for every fb.
 *   rec[3]/src[3]: deblocked reconstruction and source planes (4:2:0), luma size w x h (multiples of 8)
 *   skip8: [h/8][w/8] 1 = the 8x8 luma block is entirely "skip" (is_8x8_block_skip, EbEncCdef.c:239)
 *   mse: [2][nfb][64] (plane 0 = Y, 1 = U + V), untouched for fbs that are all-skip (svt_sb_all_skip). */
void orc_cdef_search_frame(const void *const rec[3], const int rec_stride[3], const void *const src[3], const int src_stride[3],
                           int pix_bytes, int w, int h, const uint8_t *skip8, int pri_damping, int bd, int pick_method,
                           uint64_t *mse, int fb_begin, int fb_end) {
    const int cs = bd - 8, nhfb = (w + 63) / 64, c8 = w / 8, ng = orc_cdef_strength_count(pick_method);
    uint16_t buf[(8 + 6) * 32], filt[64], srcb[64];
    for (int fb = fb_begin; fb < fb_end; fb++) {
        const int fbr = fb / nhfb, fbc = fb % nhfb;
        const int nb_x = imin(8, c8 - 8 * fbc), nb_y = imin(8, h / 8 - 8 * fbr);
        int dir[8][8], var[8][8], any = 0;
        for (int by = 0; by < nb_y; by++)
            for (int bx = 0; bx < nb_x; bx++) any |= !skip8[(8 * fbr + by) * c8 + 8 * fbc + bx];
        if (!any) continue;
        for (int pli = 0; pli < 3; pli++) {
            const int dec = pli ? 1 : 0, bs = 8 >> dec;
            const PlaneView rv = {rec[pli], pix_bytes, rec_stride[pli], w >> dec, h >> dec};
            const PlaneView sv = {src[pli], pix_bytes, src_stride[pli], w >> dec, h >> dec};
            const int damping = pri_damping + cs - (pli != 0); /* EbCdef.c:306-307 */
            for (int gi = 0; gi < ng; gi++) {
                int pri, sec;
                orc_cdef_strength(pick_method, gi, &pri, &sec);
                sec += sec == 3; /* EbCdefProcess.c:249 */
                uint64_t sum = 0;
                for (int by = 0; by < nb_y; by++)
                    for (int bx = 0; bx < nb_x; bx++) {
                        if (skip8[(8 * fbr + by) * c8 + 8 * fbc + bx]) continue;
                        const int x0 = (64 * fbc + 8 * bx) >> dec, y0 = (64 * fbr + 8 * by) >> dec;
                        stage_block(&rv, x0, y0, bs, bs, buf, 32);
                        const uint16_t *in = buf + 3 * 32 + 8;
                        if (pli == 0 && gi == 0) dir[by][bx] = orc_cdef_find_dir(in, 32, &var[by][bx], cs);
                        const int t = pri << cs, s = sec << cs;
                        if (t == 0 && s == 0) { /* EbCdef.c:311-333: plain copy */
                            for (int i = 0; i < bs; i++) for (int j = 0; j < bs; j++) filt[i * bs + j] = in[i * 32 + j];
                        } else
                            orc_cdef_filter_block(NULL, filt, bs, in, 32, pli ? t : orc_cdef_adjust_strength(t, var[by][bx]), s,
                                                  t ? dir[by][bx] : 0, damping, damping, bs, bs, cs);
                        for (int i = 0; i < bs; i++) for (int j = 0; j < bs; j++) srcb[i * bs + j] = (uint16_t)sample(&sv, x0 + j, y0 + i);
                        if (pli == 0) sum += orc_cdef_dist_8x8(filt, 8, srcb, 8, cs);
                        else
                            for (int i = 0; i < bs * bs; i++) { const int e = (int)srcb[i] - (int)filt[i]; sum += (uint64_t)(e * e); }
                    }
                /* compute_cdef_dist_*: ">> 2*coeff_shift" per plane call (EbEncCdef.c:175,219); U and V are
                 * separate calls that are then added (EbCdefProcess.c:268-271) */
                sum >>= 2 * cs;
                uint64_t *out = mse + ((size_t)(pli ? 1 : 0) * ((size_t)nhfb * ((h + 63) / 64)) + fb) * 64 + gi;
                if (pli == 2) *out += sum; else *out = sum;
            }
        }
    }
}

/* Apply: svt_av1_cdef_frame, Encoder/Codec/EbEncCdef.c:292-661 (8-bit) / :663-1031 (16-bit).
 *   y_strength / uv_strength: [nfb] the frame-header strength value (pri*4 + sec_idx, 0..63) selected for
 *   the fb; fbs with both 0, or all-skip, are left untouched (:434-441).  in[] = pre-CDEF planes, out[] =
 *   result planes (must start as a copy of in[]). */
void orc_cdef_apply_frame(const void *const in[3], void *const out[3], const int stride[3], int pix_bytes, int w, int h,
                          const uint8_t *skip8, const uint8_t *y_strength, const uint8_t *uv_strength, int damping_hdr, int bd) {
    const int cs = bd - 8, nhfb = (w + 63) / 64, nvfb = (h + 63) / 64, c8 = w / 8;
    uint16_t buf[(8 + 6) * 32];
    for (int fb = 0; fb < nhfb * nvfb; fb++) {
        const int fbr = fb / nhfb, fbc = fb % nhfb;
        const int nb_x = imin(8, c8 - 8 * fbc), nb_y = imin(8, h / 8 - 8 * fbr);
        int lv[2] = {y_strength[fb] / 4, uv_strength[fb] / 4}, sc[2] = {y_strength[fb] % 4, uv_strength[fb] % 4};
        sc[0] += sc[0] == 3; sc[1] += sc[1] == 3;
        if (lv[0] == 0 && sc[0] == 0 && lv[1] == 0 && sc[1] == 0) continue;
        int dir[8][8], var[8][8];
        for (int pli = 0; pli < 3; pli++) {
            const int dec = pli ? 1 : 0, bs = 8 >> dec;
            const PlaneView rv = {in[pli], pix_bytes, stride[pli], w >> dec, h >> dec};
            const int damping = damping_hdr + cs - (pli != 0);
            const int t = lv[pli ? 1 : 0] << cs, s = sc[pli ? 1 : 0] << cs;
            for (int by = 0; by < nb_y; by++)
                for (int bx = 0; bx < nb_x; bx++) {
                    if (skip8[(8 * fbr + by) * c8 + 8 * fbc + bx]) continue;
                    const int x0 = (64 * fbc + 8 * bx) >> dec, y0 = (64 * fbr + 8 * by) >> dec;
                    stage_block(&rv, x0, y0, bs, bs, buf, 32);
                    const uint16_t *src = buf + 3 * 32 + 8;
                    if (pli == 0) dir[by][bx] = orc_cdef_find_dir(src, 32, &var[by][bx], cs);
                    uint8_t *o8 = pix_bytes == 1 ? (uint8_t *)out[pli] + (size_t)y0 * stride[pli] + x0 : NULL;
                    uint16_t *o16 = pix_bytes == 2 ? (uint16_t *)out[pli] + (size_t)y0 * stride[pli] + x0 : NULL;
                    orc_cdef_filter_block(o8, o16, stride[pli], src, 32, pli ? t : orc_cdef_adjust_strength(t, var[by][bx]), s,
                                          t ? dir[by][bx] : 0, damping, damping, bs, bs, cs);
                }
        }
    }
}

/* finish_cdef_search after its four joint_strength_search_dual calls (Encoder/Codec/EbEncCdef.c:1258-1298): the count of signalled strength pairs by
 * RDCOST (EbRateDistortionCost.h:106, av1_cost_literal EbMdRateEstimation.h:35), then every filter block's pair index.  lev0 / lev1 / tot: [4][8] / [4]
 * = the searches' results for 1, 2, 4, 8 pairs.  Returns the chosen log2 count; y / uv [8] = its pairs; sel[sb_count] = the index per filter block. */
int orc_cdef_finish(const uint64_t *mse0, const uint64_t *mse1, int sb_count, const int32_t (*lev0)[8], const int32_t (*lev1)[8], const uint64_t *tot, uint64_t lambda,
                    int32_t *y, int32_t *uv, int32_t *sel, uint64_t *best_cost) {
    uint64_t best = (uint64_t)1 << 63;
    int bits = 0;
    for (int i = 0; i <= 3; i++) {
        const int nb = 1 << i;
        const int total_bits = sb_count * i + nb * 6 * 2;   /* CDEF_STRENGTH_BITS */
        const int rate_cost = total_bits * (1 << 9);
        const uint64_t dist = tot[i] * 16;
        const uint64_t cost = ((((uint64_t)rate_cost) * lambda + 256) >> 9) + dist * (1 << 7);
        if (cost < best) { best = cost; bits = i; }
    }
    const int nb = 1 << bits;
    for (int g = 0; g < 8; g++) { y[g] = g < nb ? lev0[bits][g] : 0; uv[g] = g < nb ? lev1[bits][g] : 0; }
    for (int i = 0; i < sb_count; i++) {
        uint64_t bm = (uint64_t)1 << 63;
        int bg = 0;
        for (int g = 0; g < nb; g++) {
            const uint64_t c = mse0[(size_t)i * 64 + y[g]] + mse1[(size_t)i * 64 + uv[g]];
            if (c < bm) { bm = c; bg = g; }
        }
        sel[i] = bg;
    }
    if (best_cost) *best_cost = best;
    return bits;
}
