/*
 * txfm_oracle.c — CPU restatement of SVT-AV1's forward / inverse 2-D transforms (4..64 points).
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).  Citations: file:line under /root/reference/Source/Lib.
 *
 * The reference spells every 1-D transform out as straight-line butterfly code, one function per
 * size (Encoder/Codec/EbTransforms.c:75-2278 forward, Common/Codec/EbInvTransforms.c:75-2358
 * inverse).  This restatement instead *constructs* the same flow graphs from their recursive
 * structure at run time (loops over stages/groups), which makes it an independent check of the
 * GPU templates.  It is pinned against the reference's own 1-D and 2-D functions for every size
 * and type by tests/test_oracle_vs_ref.py::test_txfm_*.
 */
#include "svt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------- tables ---- */
/* Common/Codec/EbInvTransforms.c:3325 "av1_cospi_arr[i][j] = round(cos(M_PI*j/128) * (1<<(10+i)))" */
static int32_t g_cospi[7][64];
static int     g_tables_ready = 0;
/* Common/Codec/EbInvTransforms.c:3360-3367 (values hand-adjusted there so that [1]+[2]==[4]) */
static const int32_t g_sinpi[7][5] = {{0, 330, 621, 836, 951},        {0, 660, 1241, 1672, 1901},
                                      {0, 1321, 2482, 3344, 3803},    {0, 2642, 4964, 6689, 7606},
                                      {0, 5283, 9929, 13377, 15212},  {0, 10566, 19858, 26755, 30424},
                                      {0, 21133, 39716, 53510, 60849}};
static void init_tables(void) {
    if (g_tables_ready) return;
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 64; j++) g_cospi[i][j] = (int32_t)llround(cos(M_PI * j / 128.0) * (double)(1 << (10 + i)));
    g_tables_ready = 1;
}
const int32_t *orc_cospi_arr(int bit) {
    init_tables();
    return g_cospi[bit - 10];
}

/* Common/Codec/EbInvTransforms.h:285-310: 32-bit wrapping products, 64-bit sum and rounding. */
static inline int32_t hb(int32_t w0, int32_t in0, int32_t w1, int32_t in1, int bit) {
    int32_t p0 = (int32_t)((uint32_t)w0 * (uint32_t)in0);
    int32_t p1 = (int32_t)((uint32_t)w1 * (uint32_t)in1);
    int64_t s  = (int64_t)p0 + (int64_t)p1 + (1LL << (bit - 1));
    return (int32_t)(s >> bit);
}
static inline int32_t rshift_round(int64_t v, int bit) { return (int32_t)((v + (1LL << (bit - 1))) >> bit); }
/* Common/Codec/EbInvTransforms.c:2418 svt_av1_round_shift_array_c */
static void round_shift_array(int32_t *a, int n, int bit) {
    if (bit == 0) return;
    if (bit > 0)
        for (int i = 0; i < n; i++) a[i] = rshift_round(a[i], bit);
    else
        for (int i = 0; i < n; i++) a[i] = (int32_t)((uint32_t)a[i] * (1u << (-bit)));
}
/* Common/Codec/EbInvTransforms.c:62-68 clamp_value */
static inline int32_t clampv(int32_t v, int bit) {
    if (bit <= 0) return v;
    const int64_t hi = (1LL << (bit - 1)) - 1, lo = -(1LL << (bit - 1));
    return (int32_t)(v < lo ? lo : (v > hi ? hi : v));
}
static inline int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }
static inline int brev(int x, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }
static inline int32_t addw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t subw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

/* ------------------------------------------------------------------------------------ DCT ----- */
/* Odd half of the forward DCT-N (elements N/2..N-1, passed as w[0..M-1], M = N/2):
 *   R0 (c32 on the middle half), then alternating butterfly stages B_k on groups of M>>k (every
 *   second group mirrored) and rotation stages R_k, then the final rotations.
 * e.g. fdct16 (EbTransforms.c:183-338): stage2 = R0, stage3 = B1, stage4 = R1, stage5 = B2, stage6 = final. */
static void fdct_odd(int32_t *w, int M, int N, const int32_t *c, int bit) {
    const int K = ilog2(M) - 1;
    if (M >= 4)
        for (int i = M / 4; i < M / 2; i++) {
            const int32_t lo = w[i], hi = w[M - 1 - i];
            w[i]         = hb(-c[32], lo, c[32], hi, bit);
            w[M - 1 - i] = hb(c[32], hi, c[32], lo, bit);
        }
    for (int k = 1; k <= K; k++) {
        const int S = M >> k;
        for (int g = 0; g < (1 << k); g++)
            for (int i = 0; i < S / 2; i++) {
                const int a = g * S + i, b = g * S + S - 1 - i;
                const int32_t lo = w[a], hi = w[b];
                if (g & 1) { w[a] = subw(hi, lo); w[b] = addw(hi, lo); }
                else       { w[a] = addw(lo, hi); w[b] = subw(lo, hi); }
            }
        if (k < K) {
            const int C = M >> k, q = C / 4;
            for (int j = 0; j < (1 << (k - 1)); j++) {
                const int a = (32 >> k) * (1 + 4 * brev(j, k - 1));
                for (int i = q; i < 3 * q; i++) {
                    const int idx = j * C + i, mir = M - 1 - idx;
                    const int32_t lo = w[idx], hi = w[mir];
                    if (i < 2 * q) { w[idx] = hb(-c[a], lo, c[64 - a], hi, bit);      w[mir] = hb(c[a], hi, c[64 - a], lo, bit); }
                    else           { w[idx] = hb(-c[64 - a], lo, -c[a], hi, bit);     w[mir] = hb(c[64 - a], hi, -c[a], lo, bit); }
                }
            }
        }
    }
    const int s = 64 / N, hb_bits = ilog2(M / 2 > 0 ? M / 2 : 1);
    for (int i = 0; i < M / 2; i++) {
        const int a = s * (1 + 4 * brev(i, hb_bits));
        const int32_t lo = w[i], hi = w[M - 1 - i];
        w[i]         = hb(c[64 - a], lo, c[a], hi, bit);
        w[M - 1 - i] = hb(c[64 - a], hi, -c[a], lo, bit);
    }
}
static void fdct_rec(int32_t *v, int n, const int32_t *c, int bit) {
    if (n == 2) {
        const int32_t a = v[0], b = v[1];
        v[0] = hb(c[32], a, c[32], b, bit);
        v[1] = hb(-c[32], b, c[32], a, bit);
        return;
    }
    for (int i = 0; i < n / 2; i++) {
        const int32_t lo = v[i], hi = v[n - 1 - i];
        v[i]         = addw(lo, hi);
        v[n - 1 - i] = subw(lo, hi);
    }
    fdct_rec(v, n / 2, c, bit);
    if (n == 4) { /* odd half of DCT-4 is one rotation (EbTransforms.c:98-99) */
        const int32_t lo = v[2], hi = v[3];
        v[2] = hb(c[48], lo, c[16], hi, bit);
        v[3] = hb(c[48], hi, -c[16], lo, bit);
    } else
        fdct_odd(v + n / 2, n / 2, n, c, bit);
}
/* svt_av1_fdct{4,8,16,32,64}_new (EbTransforms.c:75,110,183,338,678) */
void orc_fdct(const int32_t *in, int32_t *out, int n, int cos_bit) {
    int32_t v[64];
    const int32_t *c = orc_cospi_arr(cos_bit);
    memcpy(v, in, sizeof(int32_t) * n);
    fdct_rec(v, n, c, cos_bit);
    const int bits = ilog2(n);
    for (int i = 0; i < n; i++) out[i] = v[brev(i, bits)];
}

/* Inverse: the transposed graph, with clamp_value() on every add/sub stage
 * (svt_av1_idct{4..64}_new, EbInvTransforms.c:75,114,193,356,1542). */
static void idct_odd(int32_t *w, int M, int N, const int32_t *c, int bit, int clamp_bit) {
    const int K = ilog2(M) - 1;
    const int s = 64 / N, hb_bits = ilog2(M / 2 > 0 ? M / 2 : 1);
    for (int i = 0; i < M / 2; i++) {
        const int a = s * (1 + 4 * brev(i, hb_bits));
        const int32_t lo = w[i], hi = w[M - 1 - i];
        w[i]         = hb(c[64 - a], lo, -c[a], hi, bit);
        w[M - 1 - i] = hb(c[a], lo, c[64 - a], hi, bit);
    }
    for (int k = K; k >= 1; k--) {
        const int S = M >> k;
        for (int g = 0; g < (1 << k); g++)
            for (int i = 0; i < S / 2; i++) {
                const int a = g * S + i, b = g * S + S - 1 - i;
                const int32_t lo = w[a], hi = w[b];
                if (g & 1) { w[a] = clampv(subw(hi, lo), clamp_bit); w[b] = clampv(addw(lo, hi), clamp_bit); }
                else       { w[a] = clampv(addw(lo, hi), clamp_bit); w[b] = clampv(subw(lo, hi), clamp_bit); }
            }
        if (k > 1) {
            const int kk = k - 1, C = M >> kk, q = C / 4;
            for (int j = 0; j < (1 << (kk - 1)); j++) {
                const int a = (32 >> kk) * (1 + 4 * brev(j, kk - 1));
                for (int i = q; i < 3 * q; i++) {
                    const int idx = j * C + i, mir = M - 1 - idx;
                    const int32_t lo = w[idx], hi = w[mir];
                    if (i < 2 * q) { w[idx] = hb(-c[a], lo, c[64 - a], hi, bit);   w[mir] = hb(c[64 - a], lo, c[a], hi, bit); }
                    else           { w[idx] = hb(-c[64 - a], lo, -c[a], hi, bit);  w[mir] = hb(-c[a], lo, c[64 - a], hi, bit); }
                }
            }
        }
    }
    if (M >= 4)
        for (int i = M / 4; i < M / 2; i++) {
            const int32_t lo = w[i], hi = w[M - 1 - i];
            w[i]         = hb(-c[32], lo, c[32], hi, bit);
            w[M - 1 - i] = hb(c[32], lo, c[32], hi, bit);
        }
}
static void idct_rec(int32_t *v, int n, const int32_t *c, int bit, int clamp_bit) {
    if (n == 2) {
        const int32_t a = v[0], b = v[1];
        v[0] = hb(c[32], a, c[32], b, bit);
        v[1] = hb(c[32], a, -c[32], b, bit);
        return;
    }
    idct_rec(v, n / 2, c, bit, clamp_bit);
    if (n == 4) {
        const int32_t lo = v[2], hi = v[3];
        v[2] = hb(c[48], lo, -c[16], hi, bit);
        v[3] = hb(c[16], lo, c[48], hi, bit);
    } else
        idct_odd(v + n / 2, n / 2, n, c, bit, clamp_bit);
    for (int i = 0; i < n / 2; i++) {
        const int32_t lo = v[i], hi = v[n - 1 - i];
        v[i]         = clampv(addw(lo, hi), clamp_bit);
        v[n - 1 - i] = clampv(subw(lo, hi), clamp_bit);
    }
}
void orc_idct(const int32_t *in, int32_t *out, int n, int cos_bit, int clamp_bit) {
    int32_t v[64];
    const int32_t *c = orc_cospi_arr(cos_bit);
    const int bits = ilog2(n);
    for (int i = 0; i < n; i++) v[i] = in[brev(i, bits)];
    idct_rec(v, n, c, cos_bit, clamp_bit);
    memcpy(out, v, sizeof(int32_t) * n);
}

/* ------------------------------------------------------------------------------------ ADST ---- */
/* Output permutation of the inverse ADST (EbInvTransforms.c:879-900 for N=8, :1080-1105 for N=16):
 * out[j] = (-1)^j * v[p[j]]; the forward input stage (EbTransforms.c:1547-1556, :1645-1662) is the
 * inverse mapping v[p[j]] = (-1)^j * x[j]. */
static const uint8_t adst_p8[8]   = {0, 4, 6, 2, 3, 7, 5, 1};
static const uint8_t adst_p16[16] = {0, 8, 12, 4, 6, 14, 10, 2, 3, 11, 15, 7, 5, 13, 9, 1};

/* rotation stage on the upper half of every group of G elements (G = 4, 8, 16) */
static void adst_rot_groups(int32_t *v, int n, int G, const int32_t *c, int bit) {
    const int np = G / 4, half = np > 1 ? np / 2 : 1;
    for (int g = 0; g < n; g += G)
        for (int j = 0; j < np; j++) {
            const int a = (128 / G) * (1 + 4 * brev(j % half, ilog2(half)));
            const int l = g + G / 2 + 2 * j;
            const int32_t lo = v[l], hi = v[l + 1];
            if (np == 1 || j < half) { v[l] = hb(c[a], lo, c[64 - a], hi, bit);   v[l + 1] = hb(c[64 - a], lo, -c[a], hi, bit); }
            else                     { v[l] = hb(-c[64 - a], lo, c[a], hi, bit);  v[l + 1] = hb(c[a], lo, c[64 - a], hi, bit); }
        }
}
static void adst_final_rot(int32_t *v, int n, const int32_t *c, int bit) {
    for (int j = 0; j < n / 2; j++) {
        const int a = (32 / n) * (1 + 4 * j);
        const int32_t lo = v[2 * j], hi = v[2 * j + 1];
        v[2 * j]     = hb(c[a], lo, c[64 - a], hi, bit);
        v[2 * j + 1] = hb(c[64 - a], lo, -c[a], hi, bit);
    }
}
/* svt_av1_fadst4_new (EbTransforms.c:1445-1533) */
static void fadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *sp = g_sinpi[bit - 10];
    const int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    if (!(x0 | x1 | x2 | x3)) { out[0] = out[1] = out[2] = out[3] = 0; return; }
#define MULW(a, b) ((int32_t)((uint32_t)(a) * (uint32_t)(b)))
    int32_t s0 = MULW(sp[1], x0), s1 = MULW(sp[4], x0), s2 = MULW(sp[2], x1), s3 = MULW(sp[1], x1);
    int32_t s4 = MULW(sp[3], x2), s5 = MULW(sp[4], x3), s6 = MULW(sp[2], x3), s7 = subw(addw(x0, x1), x3);
    int32_t y0 = addw(addw(s0, s2), s5), y1 = MULW(sp[3], s7), y2 = addw(subw(s1, s3), s6), y3 = s4;
    out[0] = rshift_round(addw(y0, y3), bit);
    out[1] = rshift_round(y1, bit);
    out[2] = rshift_round(subw(y2, y3), bit);
    out[3] = rshift_round(addw(subw(y2, y0), y3), bit);
}
/* svt_av1_iadst4_new (EbInvTransforms.c:707-792) */
static void iadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *sp = g_sinpi[bit - 10];
    const int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    if (!(x0 | x1 | x2 | x3)) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    int32_t s0 = MULW(sp[1], x0), s1 = MULW(sp[2], x0), s2 = MULW(sp[3], x1), s3 = MULW(sp[4], x2);
    int32_t s4 = MULW(sp[1], x2), s5 = MULW(sp[2], x3), s6 = MULW(sp[4], x3), s7 = addw(subw(x0, x2), x3);
    s0 = addw(addw(s0, s3), s5);
    s1 = subw(subw(s1, s4), s6);
    s3 = s2;
    s2 = MULW(sp[3], s7);
    out[0] = rshift_round(addw(s0, s3), bit);
    out[1] = rshift_round(addw(s1, s3), bit);
    out[2] = rshift_round(s2, bit);
    out[3] = rshift_round(subw(addw(s0, s1), s3), bit);
#undef MULW
}
/* svt_av1_fadst{8,16}_new (EbTransforms.c:1535,1633) */
void orc_fadst(const int32_t *in, int32_t *out, int n, int cos_bit) {
    init_tables();
    if (n == 4) { fadst4(in, out, cos_bit); return; }
    const int32_t *c = orc_cospi_arr(cos_bit);
    const uint8_t *p = n == 8 ? adst_p8 : adst_p16;
    int32_t v[16];
    for (int j = 0; j < n; j++) v[p[j]] = (j & 1) ? subw(0, in[j]) : in[j];
    for (int G = 4; G <= n; G *= 2) {
        adst_rot_groups(v, n, G, c, cos_bit);
        for (int g = 0; g < n; g += G)
            for (int i = 0; i < G / 2; i++) {
                const int32_t lo = v[g + i], hi = v[g + i + G / 2];
                v[g + i] = addw(lo, hi);
                v[g + i + G / 2] = subw(lo, hi);
            }
    }
    adst_final_rot(v, n, c, cos_bit);
    for (int k = 0; k < n / 2; k++) { out[2 * k] = v[2 * k + 1]; out[2 * k + 1] = v[n - 2 - 2 * k]; }
}
/* svt_av1_iadst{8,16}_new (EbInvTransforms.c:797,902) */
void orc_iadst(const int32_t *in, int32_t *out, int n, int cos_bit, int clamp_bit) {
    init_tables();
    if (n == 4) { iadst4(in, out, cos_bit); return; }
    const int32_t *c = orc_cospi_arr(cos_bit);
    const uint8_t *p = n == 8 ? adst_p8 : adst_p16;
    int32_t v[16];
    for (int k = 0; k < n / 2; k++) { v[2 * k] = in[n - 1 - 2 * k]; v[2 * k + 1] = in[2 * k]; }
    adst_final_rot(v, n, c, cos_bit);
    for (int G = n; G >= 4; G /= 2) {
        for (int g = 0; g < n; g += G)
            for (int i = 0; i < G / 2; i++) {
                const int32_t lo = v[g + i], hi = v[g + i + G / 2];
                v[g + i] = clampv(addw(lo, hi), clamp_bit);
                v[g + i + G / 2] = clampv(subw(lo, hi), clamp_bit);
            }
        adst_rot_groups(v, n, G, c, cos_bit);
    }
    for (int j = 0; j < n; j++) out[j] = (j & 1) ? subw(0, v[p[j]]) : v[p[j]];
}

/* -------------------------------------------------------------------------------- identity ---- */
/* svt_av1_fidentity*_c (EbTransforms.c:2239-2278) == svt_av1_iidentity*_c (EbInvTransforms.c:2321-2358) */
void orc_identity(const int32_t *in, int32_t *out, int n) {
    for (int i = 0; i < n; i++) {
        switch (n) {
        case 4: out[i] = rshift_round((int64_t)in[i] * 5793, 12); break;
        case 8: out[i] = (int32_t)((int64_t)in[i] * 2); break;
        case 16: out[i] = rshift_round((int64_t)in[i] * 2 * 5793, 12); break;
        case 32: out[i] = (int32_t)((int64_t)in[i] * 4); break;
        default: out[i] = rshift_round((int64_t)in[i] * 4 * 5793, 12); break;
        }
    }
}

/* --------------------------------------------------------------------------------- 2-D core --- */
/* TxSize order: Common/Codec/EbDefinitions.h (TX_4X4 .. TX_64X16) */
static const uint8_t tx_w[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
static const uint8_t tx_h[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};
int orc_tx_width(int tx_size) { return tx_w[tx_size]; }
int orc_tx_height(int tx_size) { return tx_h[tx_size]; }
/* 1-D kind per TxType: 0 DCT, 1 ADST, 2 FLIPADST, 3 IDTX  (vtx_tab / htx_tab, EbInvTransforms.h:71-106) */
static const uint8_t vtx[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
static const uint8_t htx[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};
/* Encoder/Codec/EbTransforms.h:26-44 */
static const int8_t fwd_shift[19][3] = {{2, 0, 0},  {2, -1, 0}, {2, -2, 0}, {2, -4, 0},  {0, -2, -2}, {2, -1, 0}, {2, -1, 0},
                                        {2, -2, 0}, {2, -2, 0}, {2, -4, 0}, {2, -4, 0},  {0, -2, -2}, {2, -4, -2}, {2, -1, 0},
                                        {2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {0, -2, 0},  {2, -4, 0}};
/* Encoder/Codec/EbTransforms.h:46-57, indexed [log2(w)-2][log2(h)-2] */
static const int8_t fwd_cos_col[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
static const int8_t fwd_cos_row[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
/* Common/Codec/EbInvTransforms.h:51-69 */
static const int8_t inv_shift[19][2] = {{0, -4},  {-1, -4}, {-2, -4}, {-2, -4}, {-2, -4}, {0, -4},  {0, -4},  {-1, -4}, {-1, -4}, {-1, -4},
                                        {-1, -4}, {-1, -4}, {-1, -4}, {-1, -4}, {-1, -4}, {-2, -4}, {-2, -4}, {-2, -4}, {-2, -4}};

static void fwd_1d(int kind, const int32_t *in, int32_t *out, int n, int cos_bit) {
    if (kind == 0) orc_fdct(in, out, n, cos_bit);
    else if (kind == 3) orc_identity(in, out, n);
    else orc_fadst(in, out, n, cos_bit);
}
static void inv_1d(int kind, const int32_t *in, int32_t *out, int n, int cos_bit, int clamp_bit) {
    if (kind == 0) orc_idct(in, out, n, cos_bit, clamp_bit);
    else if (kind == 3) orc_identity(in, out, n);
    else orc_iadst(in, out, n, cos_bit, clamp_bit);
}

/* av1_tranform_two_d_core_c (Encoder/Codec/EbTransforms.c:2301-2370) with av1_transform_config (:2732).
 * Output: packed W x H int32, row-major (the full 64-wide output for 64-pt sizes; use
 * orc_handle_transform() for the reference's zero-out / re-pack step). */
void orc_fwd_txfm2d(const int16_t *input, int32_t *output, uint32_t stride, int tx_type, int tx_size, int bd) {
    (void)bd;
    const int W = tx_w[tx_size], H = tx_h[tx_size];
    const int8_t *sh = fwd_shift[tx_size];
    const int cbc = fwd_cos_col[ilog2(W) - 2][ilog2(H) - 2], cbr = fwd_cos_row[ilog2(W) - 2][ilog2(H) - 2];
    const int kc = vtx[tx_type], kr = htx[tx_type];
    const int ud = kc == 2, lr = kr == 2;
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * W * H);
    int32_t tin[64], tout[64];
    for (int c = 0; c < W; c++) {
        for (int r = 0; r < H; r++) tin[r] = input[(ud ? H - 1 - r : r) * stride + c];
        round_shift_array(tin, H, -sh[0]);
        fwd_1d(kc, tin, tout, H, cbc);
        round_shift_array(tout, H, -sh[1]);
        for (int r = 0; r < H; r++) buf[r * W + (lr ? W - 1 - c : c)] = tout[r];
    }
    const int rect2 = (W == 2 * H) || (H == 2 * W);
    for (int r = 0; r < H; r++) {
        fwd_1d(kr, buf + r * W, output + r * W, W, cbr);
        round_shift_array(output + r * W, W, -sh[2]);
        if (rect2)
            for (int c = 0; c < W; c++) output[r * W + c] = rshift_round((int64_t)output[r * W + c] * 5793, 12);
    }
    free(buf);
}

/* svt_handle_transform64x64_c / 64x32 / 32x64 / 64x16 / 16x64 (EbTransforms.c:2763-2931):
 * energy of the discarded region, zero it, re-pack the kept min(W,32) x min(H,32) block. */
uint64_t orc_handle_transform(int32_t *coeff, int tx_size) {
    const int W = tx_w[tx_size], H = tx_h[tx_size];
    const int kw = W > 32 ? 32 : W, kh = H > 32 ? 32 : H;
    if (kw == W && kh == H) return 0;
    uint64_t e = 0;
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++)
            if (r >= kh || c >= kw) {
                e += (uint64_t)((int64_t)coeff[r * W + c] * (int64_t)coeff[r * W + c]);
                coeff[r * W + c] = 0;
            }
    if (kw != W)
        for (int r = 1; r < kh; r++) memmove(coeff + r * kw, coeff + r * W, sizeof(int32_t) * kw);
    return e;
}

/* av1_estimate_transform (Encoder/Codec/EbTransforms.c:3613-3670) for a coefficient shape (EB_TRANS_COEFF_SHAPE, EbDefinitions.h:2610-2614:
 * 0 DEFAULT_SHAPE, 1 N2_SHAPE, 2 N4_SHAPE, 3 ONLY_DC_SHAPE): forward transform + the 64-point zero-out / re-pack, output = the packed
 * min(W,32) x min(H,32) block, return value = *three_quad_energy.
 * The N2 / N4 transform families (:3840-7250) are the default butterfly graphs pruned to the outputs of the top-left W/2 x H/2 (W/4 x H/4)
 * corner: those coefficients equal the default transform's, every other one is written as zero, and handle_transform*_N2_N4 (:2933-2964)
 * only re-packs (energy 0).  ONLY_DC (:3407-3431) runs N4 and then clears everything except coefficient 0. */
uint64_t orc_estimate_transform(const int16_t *residual, uint32_t stride, int32_t *coeff, int tx_type, int tx_size, int bd, int shape) {
    const int W = tx_w[tx_size], H = tx_h[tx_size];
    const int kw = W > 32 ? 32 : W, kh = H > 32 ? 32 : H;
    int32_t *full = (int32_t *)malloc(sizeof(int32_t) * W * H);
    orc_fwd_txfm2d(residual, full, stride, tx_type, tx_size, bd);
    uint64_t energy = 0;
    if (shape == 0) energy = orc_handle_transform(full, tx_size);
    else {
        const int cw = shape == 3 ? 1 : W >> shape, ch = shape == 3 ? 1 : H >> shape;
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++)
                if (r >= ch || c >= cw) full[r * W + c] = 0;
        if (kw != W)
            for (int r = 1; r < kh; r++) memmove(full + r * kw, full + r * W, sizeof(int32_t) * kw);
    }
    memcpy(coeff, full, sizeof(int32_t) * kw * kh);
    free(full);
    return energy;
}

/* inv_txfm2d_add_c (Common/Codec/EbInvTransforms.c:2455-2533), svt_av1_get_inv_txfm_cfg (:2432),
 * svt_av1_gen_inv_stage_range (:23-60), the 64-pt input re-mapping (:2648-2714), pixel add with
 * highbd_clip_pixel_add/check_range (:2398-2416).  `input` is the packed min(W,32) x min(H,32) block. */
void orc_inv_txfm2d_add(const int32_t *input, const uint16_t *pred, int32_t stride_r, uint16_t *recon,
                        int32_t stride_w, int tx_type, int tx_size, int bd) {
    const int W = tx_w[tx_size], H = tx_h[tx_size];
    const int kw = W > 32 ? 32 : W, kh = H > 32 ? 32 : H;
    const int8_t *sh = inv_shift[tx_size];
    const int cos_bit = 12; /* INV_COS_BIT, EbInvTransforms.h:39-50 */
    const int kc = vtx[tx_type], kr = htx[tx_type];
    const int ud = kc == 2, lr = kr == 2;
    const int rng_row = bd == 8 ? 16 : (bd == 10 ? 18 : 20), rng_col = bd == 12 ? 18 : 16;
    const int rect2 = (W == 2 * H) || (H == 2 * W);
    int32_t *buf = (int32_t *)calloc((size_t)W * H, sizeof(int32_t));
    int32_t tin[64], tout[64];
    for (int r = 0; r < H; r++) {
        for (int c = 0; c < W; c++) {
            int32_t v = (r < kh && c < kw) ? input[r * kw + c] : 0;
            if (rect2) v = rshift_round((int64_t)v * 2896, 12);
            tin[c] = clampv(v, bd + 8);
        }
        inv_1d(kr, tin, buf + r * W, W, cos_bit, rng_row);
        round_shift_array(buf + r * W, W, -sh[0]);
    }
    const int col_clamp = bd + 6 > 16 ? bd + 6 : 16;
    const int32_t pix_max = (1 << bd) - 1;
    const int64_t res_max = (1 << (7 + bd)) - 1 + (914 << (bd - 7)), res_min = -res_max - 1;
    for (int c = 0; c < W; c++) {
        for (int r = 0; r < H; r++) tin[r] = clampv(buf[r * W + (lr ? W - 1 - c : c)], col_clamp);
        inv_1d(kc, tin, tout, H, cos_bit, rng_col);
        round_shift_array(tout, H, -sh[1]);
        for (int r = 0; r < H; r++) {
            int64_t t = tout[ud ? H - 1 - r : r];
            t = t < res_min ? res_min : (t > res_max ? res_max : t);
            int32_t p = (int32_t)pred[r * stride_r + c] + (int32_t)t;
            recon[r * stride_w + c] = (uint16_t)(p < 0 ? 0 : (p > pix_max ? pix_max : p));
        }
    }
    free(buf);
}

/* svt_av1_inv_txfm_add_c (EbInvTransforms.c:3302-3323): 8-bit pixels through the 16-bit core. */
void orc_inv_txfm_add_8bit(const int32_t *input, const uint8_t *pred, int32_t stride_r, uint8_t *recon,
                           int32_t stride_w, int tx_type, int tx_size) {
    const int W = tx_w[tx_size], H = tx_h[tx_size];
    uint16_t tmp[64 * 64];
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) tmp[r * 64 + c] = pred[r * stride_r + c];
    orc_inv_txfm2d_add(input, tmp, 64, tmp, 64, tx_type, tx_size, 8);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) recon[r * stride_w + c] = (uint8_t)tmp[r * 64 + c];
}

/* svt_residual_kernel8bit_c (Common/C_DEFAULT/EbPictureOperators_C.c) */
void orc_residual_8bit(const uint8_t *src, uint32_t src_stride, const uint8_t *pred, uint32_t pred_stride,
                       int16_t *res, uint32_t res_stride, uint32_t w, uint32_t h) {
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++) res[r * res_stride + c] = (int16_t)((int)src[r * src_stride + c] - (int)pred[r * pred_stride + c]);
}

/* Frame-level driver for tests / bench cpu_baseline: residual -> fwd txfm -> (64-pt re-pack) -> quantize
 * -> inverse txfm -> recon for a list of blocks of one transform size (descriptor = x | y << 14 |
 * tx_type << 28, the layout of SVT_HIP_TX_DESC). 8-bit planes. */
void orc_quantize(int variant, const int32_t *coeff, int n, const int16_t *zbin, const int16_t *round,
                  const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                  const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, int log_scale);
void orc_txfm_chain_8bit(const uint8_t *src, int src_stride, const uint8_t *pred, int pred_stride, uint8_t *recon,
                         int recon_stride, const uint32_t *descs, int begin, int end, int tx_size, int variant,
                         const int16_t qp[7][2], const int16_t *const scans[3], int log_scale, int32_t *qcoeff_out,
                         uint16_t *eob_out) {
    const int W = tx_w[tx_size], H = tx_h[tx_size], kw = W > 32 ? 32 : W, kh = H > 32 ? 32 : H, nk = kw * kh;
    int16_t res[64 * 64];
    int32_t *co = (int32_t *)malloc(sizeof(int32_t) * 64 * 64 * 3), *q = co + 4096, *dq = q + 4096;
    for (int i = begin; i < end; i++) {
        const int x = descs[i] & 0x3FFF, y = (descs[i] >> 14) & 0x3FFF, tt = descs[i] >> 28;
        orc_residual_8bit(src + (size_t)y * src_stride + x, src_stride, pred + (size_t)y * pred_stride + x, pred_stride, res, W, W, H);
        orc_fwd_txfm2d(res, co, W, tt, tx_size, 8);
        orc_handle_transform(co, tx_size);
        const int cls = (W <= 16 && H <= 16) ? (tt < 10 ? 0 : ((tt & 1) ? 2 : 1)) : 0;
        uint16_t eob;
        const int16_t *rnd = variant >= 2 ? qp[5] : qp[1], *qnt = variant >= 2 ? qp[6] : qp[2];
        orc_quantize(variant, co, nk, qp[0], rnd, qnt, qp[3], q, dq, qp[4], &eob, scans[cls], log_scale);
        if (qcoeff_out) memcpy(qcoeff_out + (size_t)i * nk, q, sizeof(int32_t) * nk);
        if (eob_out) eob_out[i] = eob;
        orc_inv_txfm_add_8bit(dq, pred + (size_t)y * pred_stride + x, pred_stride, recon + (size_t)y * recon_stride + x, recon_stride, tt, tx_size);
    }
    free(co);
}
