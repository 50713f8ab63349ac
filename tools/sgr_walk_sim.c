/* tools/sgr_walk_sim.c — CPU study of the restoration walk's speculation policy (development tool, not product code).
 *
 * The device walk (svt-av1_amd/csrc/sgr_walk.hip) replays the reference's coordinate descent (finer_search_pixel_proj_error,
 * Encoder/Codec/EbRestorationPick.c:353-446) on a cache of exactly evaluated points; at the first unknown point it walks on the quadratic model the
 * five projection sums give and collects the points it visits; a pass over the unit's samples evaluates them exactly.  WHICH points a replay asks for
 * decides how many passes (fixed cost each) and how many evaluated points (cost each) a walk takes — and can never change the result.  This file
 * restates the replay (state machine and all) on the oracle's filters and lets several request policies run on the same units, counting passes and
 * points, so that a policy can be chosen without a GPU.  Build / run: tools/sgr_walk_sim.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* oracle (liboracle.so) */
void orc_sgr_filter(const void *dgd, int pix_bytes, int w, int h, int stride, int32_t *flt0, int32_t *flt1, int flt_stride, int ep, int bd);
int  orc_rest_units(int size, int unit_size);
int  orc_rest_unit_limits(int pw, int ph, int ss_y, int unit_size, int32_t *limits);
extern const int32_t orc_sgr_params[16][4];

#define MAXC 64
long g_cls_pass[32][3], g_cls_pts[32][3], g_cls_ref[3], g_cls_minpass[3];   /* per class (0: both filters, 1: r1 only, 2: r0 only) and policy index */
typedef struct {
    int    mode;       /* 0 = the device's policy (model path); 1 = best-first over the decision tree; 2 = model path + its one-step alternatives */
    int    cap;        /* points per pass */
    double sigma_k;    /* mode 1: std of (exact - model) of a probe relative to its current point = sigma_k * sqrt(err / n_samples) * sqrt(n_samples) ... see noise_sigma */
    double min_p;      /* mode 1: branches below this probability are not expanded */
    int    stop_known; /* mode 1: 1 = stop a branch at its end (always) */
} Policy;
typedef struct { long walks, passes, points, replays; long hist[16]; double cost; long ref_points; } Stats;

typedef struct { int s, p, up, q0, q1, moved, have_err; double err; int run; } WS;   /* run: the last decision accepted a step-2 probe (the walk repeats the probe one step further) */

typedef struct {
    int n, w, h;
    int32_t *d0, *d1, *e0;     /* flt0 - u, flt1 - u, dat - src of the unit */
    int has0, has1, ep;
    double H00, H01, H11, C0, C1;
    int64_t memo[256 * 256];   /* exact error per (x + 128, y + 128), -1 = not computed yet */
    long n_exact;
} Unit;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int64_t exact_err(Unit *U, int x, int y) {
    int64_t *m = &U->memo[(y + 128) * 256 + x + 128];
    if (*m >= 0) return *m;
    const int xq0 = U->has0 ? x : 0, xq1 = !U->has1 ? 0 : (U->has0 ? 128 - x - y : 128 - y);   /* svt_decode_xq */
    int64_t err = 0;
    const int32_t *d0 = U->d0, *d1 = U->d1, *e0 = U->e0;
    for (int i = 0; i < U->n; i++) {
        const int32_t e = e0[i] + ((xq0 * d0[i] + xq1 * d1[i] + 1024) >> 11);
        err += (int64_t)e * e;
    }
    U->n_exact++;
    return *m = err;
}
static double model(const Unit *U, int x, int y) {
    const double a = U->has0 ? x : 0, b = !U->has1 ? 0 : (U->has0 ? 128 - x - y : 128 - y);
    return a * a * U->H00 + 2 * a * b * U->H01 + b * b * U->H11 - 256.0 * (a * U->C0 + b * U->C1);
}

/* the cache of one simulated walk */
typedef struct { int x[512], y[512]; int n; int wx[MAXC], wy[MAXC]; int nw, cap; } Cache;
static int cached(const Cache *K, int x, int y) { for (int i = 0; i < K->n; i++) if (K->x[i] == x && K->y[i] == y) return 1; return 0; }
static void want(Cache *K, int x, int y) {
    for (int i = 0; i < K->nw; i++) if (K->wx[i] == x && K->wy[i] == y) return;
    if (K->nw < K->cap) { K->wx[K->nw] = x; K->wy[K->nw] = y; K->nw++; }
}

/* one decision step of the walk on `value`s; returns 0 when the walk is over.  probe(): fills c0, c1 with the next probe or returns 0 when the state
 * advances without a probe. */
static int next_probe(WS *T, const Unit *U, int *c0, int *c1) {
    for (;;) {
        if (T->s < 1) return 0;
        if (T->p >= 2) { T->s >>= 1; T->p = 0; T->up = 0; T->moved = 0; continue; }
        if (T->p == 0 ? !U->has0 : !U->has1) { T->p++; T->up = 0; T->moved = 0; continue; }
        const int tmin = T->p == 0 ? -96 : -32, tmax = T->p == 0 ? 31 : 95;
        const int qp = T->p == 0 ? T->q0 : T->q1, d = T->up ? T->s : -T->s;
        if (T->up ? qp + T->s <= tmax : qp - T->s >= tmin) {
            *c0 = T->p == 0 ? T->q0 + d : T->q0; *c1 = T->p == 0 ? T->q1 : T->q1 + d;
            return 1;
        }
        /* no probe possible: as a rejected probe */
        if (!T->up) { if (T->moved) T->p = 2; else T->up = 1; }
        else { T->p++; T->up = 0; T->moved = 0; }
    }
}
static void decide(WS *T, int c0, int c1, double err2, int accept) {
    int again = 0;
    if (accept) { T->q0 = c0; T->q1 = c1; T->err = err2; if (!T->up) T->moved = 1; again = T->s == 2; }
    T->run = again ? T->run + 1 : 0;
    if (!again) {
        if (!T->up) { if (T->moved) T->p = 2; else T->up = 1; }
        else { T->p++; T->up = 0; T->moved = 0; }
    }
}

/* advance W on exact (cached) values as far as possible; returns 1 when the walk finished */
static int advance_exact(Unit *U, Cache *K, WS *W) {
    if (!W->have_err) {
        if (!cached(K, W->q0, W->q1)) return 0;
        W->err = (double)exact_err(U, W->q0, W->q1); W->have_err = 1;
    }
    for (;;) {
        WS T = *W;
        int c0, c1;
        if (!next_probe(&T, U, &c0, &c1)) { *W = T; return 1; }
        if (!cached(K, c0, c1)) { *W = T; return 0; }   /* T = W advanced to right before the probe */
        const double e2 = (double)exact_err(U, c0, c1);
        decide(&T, c0, c1, e2, !(e2 > T.err));
        *W = T;
    }
}

static double sigma_unit(const Unit *U) { return sqrt((double)exact_err((Unit *)U, 0, 0) / 12.0 + 1.0) * 4194304.0; }
static double phi(double z) { return 0.5 * erfc(-z / sqrt(2.0)); }

/* the speculative part of a replay: fills K->wx / wy */
typedef struct { WS T; double lp; } Node;
static void speculate(Unit *U, Cache *K, const WS *W, const Policy *P, double sigma) {
    K->nw = 0; K->cap = P->cap;
    if (!W->have_err) { want(K, W->q0, W->q1); }
    if (P->mode == 4) {   /* model path, but a run of accepted step-2 probes is assumed to go on while the model does not clearly object */
        WS T = *W;
        T.err = model(U, T.q0, T.q1); T.have_err = 1;
        while (K->nw < K->cap) {
            int c0, c1;
            if (!next_probe(&T, U, &c0, &c1)) break;
            if (!cached(K, c0, c1)) want(K, c0, c1);
            const double e2 = model(U, c0, c1);
            int acc = !(e2 > T.err);
            if (!acc && T.s == 2 && T.run >= (int)P->min_p && e2 - T.err < P->sigma_k * sigma_unit(U)) acc = 1;
            decide(&T, c0, c1, acc ? (e2 < T.err ? e2 : T.err) : e2, acc);
        }
        return;
    }
    if (P->mode == 0 || P->mode == 2) {
        WS T = *W;
        T.err = model(U, T.q0, T.q1); T.have_err = 1;
        while (K->nw < K->cap) {
            int c0, c1;
            if (!next_probe(&T, U, &c0, &c1)) break;
            if (!cached(K, c0, c1)) want(K, c0, c1);
            const double e2 = model(U, c0, c1);
            const int acc = !(e2 > T.err);
            if (P->mode == 2 && K->nw < K->cap) {   /* the other branch's next probe */
                WS A = T; int a0, a1;
                decide(&A, c0, c1, e2, !acc);
                if (next_probe(&A, U, &a0, &a1) && !cached(K, a0, a1)) want(K, a0, a1);
            }
            decide(&T, c0, c1, e2, acc);
        }
        return;
    }
    /* mode 1: best-first expansion of the decision tree; a decision goes the model's way with probability Phi(|d| / sigma) */
    static Node q[4096];
    int nq = 0;
    q[nq].T = *W; q[nq].T.err = model(U, W->q0, W->q1); q[nq].T.have_err = 1; q[nq].lp = 0; nq++;
    int guard = 0;
    while (nq && K->nw < K->cap && guard++ < 2000) {
        int bi = 0;
        for (int i = 1; i < nq; i++) if (q[i].lp > q[bi].lp) bi = i;
        Node N = q[bi]; q[bi] = q[--nq];
        int c0, c1;
        if (!next_probe(&N.T, U, &c0, &c1)) continue;   /* this branch's walk is over */
        const int known = cached(K, c0, c1);
        if (!known) want(K, c0, c1);
        const double e2 = model(U, c0, c1), d = e2 - N.T.err;
        double pa = phi(-d / sigma);   /* P(exact probe <= exact current) */
        if (pa < 1e-9) pa = 1e-9; if (pa > 1 - 1e-9) pa = 1 - 1e-9;
        for (int br = 0; br < 2; br++) {
            const double pb = br ? 1 - pa : pa;
            if (N.lp + log(pb) < log(P->min_p) || nq >= 4090) continue;
            Node Cn = N;
            decide(&Cn.T, c0, c1, e2, br == 0);
            Cn.lp = N.lp + log(pb);
            q[nq++] = Cn;
        }
    }
}

long g_policy_mismatch;   /* walks whose simulated result (xqd pair, error) differs from the reference walk's: must stay 0 for every policy */
static void run_policy(Unit *U, const int start[2], const Policy *P, Stats *S, double A, double B, double sigma, const int ref_xy[2], int64_t ref_err) {
    Cache K; K.n = 0;
    WS W = {2, 0, 0, start[0], start[1], 0, 0, 0.0, 0};
    int passes = 0, points = 0;
    for (int it = 0; it < 64; it++) {
        S->replays++;
        if (advance_exact(U, &K, &W)) break;
        speculate(U, &K, &W, P, sigma);
        if (!K.nw) { fprintf(stderr, "policy asked for nothing\n"); exit(1); }
        for (int i = 0; i < K.nw; i++) { (void)exact_err(U, K.wx[i], K.wy[i]); K.x[K.n] = K.wx[i]; K.y[K.n] = K.wy[i]; K.n++; }
        passes++; points += K.nw;
    }
    if (W.q0 != ref_xy[0] || W.q1 != ref_xy[1] || (int64_t)W.err != ref_err || W.s >= 1) g_policy_mismatch++;
    S->walks++; S->passes += passes; S->points += points; S->hist[passes < 15 ? passes : 15]++;
    { const int cls = U->has0 && U->has1 ? 0 : (U->has1 ? 1 : 2); g_cls_pass[P->stop_known][cls] += passes; g_cls_pts[P->stop_known][cls] += points; }
    S->cost += A * passes + B * points;
}

/* reference walk: number of distinct points it evaluates */
long g_off[3][33][33];   /* [class][dy + 16][dx + 16]: walks whose reference walk evaluates start + (dx, dy) */
long g_cls_walks[3];
static int ref_points(Unit *U, const int start[2], int out_xy[2], int64_t *out_err) {
    Cache K; K.n = 0;
    WS W = {2, 0, 0, start[0], start[1], 0, 1, 0.0, 0};
    W.err = (double)exact_err(U, start[0], start[1]); K.x[0] = start[0]; K.y[0] = start[1]; K.n = 1;
    for (;;) {
        int c0, c1;
        if (!next_probe(&W, U, &c0, &c1)) break;
        if (!cached(&K, c0, c1)) { K.x[K.n] = c0; K.y[K.n] = c1; K.n++; }
        const double e2 = (double)exact_err(U, c0, c1);
        if (getenv("SIM_TRACE") && g_cls_walks[0] + g_cls_walks[1] + g_cls_walks[2] < atoi(getenv("SIM_TRACE")))
            fprintf(stderr, "  ep %2d s %d p %d up %d cur (%d,%d) probe (%d,%d): exact diff %.0f model diff %.0f%s\n", U->ep, W.s, W.p, W.up, W.q0, W.q1, c0, c1, e2 - W.err,
                    (model(U, c0, c1) - model(U, W.q0, W.q1)) / 4194304.0, !(e2 > W.err) ? "  ACCEPT" : "");
        decide(&W, c0, c1, e2, !(e2 > W.err));
    }
    out_xy[0] = W.q0; out_xy[1] = W.q1; *out_err = (int64_t)W.err;
    {
        const int cls = U->has0 && U->has1 ? 0 : (U->has1 ? 1 : 2);
        g_cls_walks[cls]++; g_cls_ref[cls] += K.n; g_cls_minpass[cls] += (K.n + 7) / 8;
        for (int i = 0; i < K.n; i++) {
            const int dx = K.x[i] - start[0], dy = K.y[i] - start[1];
            if (dx >= -16 && dx <= 16 && dy >= -16 && dy <= 16) g_off[cls][dy + 16][dx + 16]++;
        }
    }
    return K.n;
}

static void solve_and_encode(const int64_t *sums, int size, int ep, int xqd[2]) {
    double H00 = (double)sums[0], H01 = (double)sums[1], H11 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    const double dsize = (double)size;
    H00 /= dsize; H01 /= dsize; H11 /= dsize; C0 /= dsize; C1 /= dsize;
    const double H10 = H01;
    int xq[2] = {0, 0};
    const int has0 = ep < 10 || ep >= 14, has1 = ep < 14;
    if (!has0) { if (!(H11 < 1e-8)) xq[1] = (int)rint((C1 / H11) * 128.0); }
    else if (!has1) { if (!(H00 < 1e-8)) xq[0] = (int)rint((C0 / H00) * 128.0); }
    else {
        const double det = H00 * H11 - H01 * H10;
        if (!(det < 1e-8)) {
            const double x0 = (H11 * C0 - H01 * C1) / det, x1 = (H00 * C1 - H10 * C0) / det;
            xq[0] = (int)rint(x0 * 128.0); xq[1] = (int)rint(x1 * 128.0);
        }
    }
    if (!has0) { xqd[0] = 0; xqd[1] = clampi(128 - xq[1], -32, 95); }
    else if (!has1) { xqd[0] = clampi(xq[0], -96, 31); xqd[1] = clampi(128 - xqd[0], -32, 95); }
    else { xqd[0] = clampi(xq[0], -96, 31); xqd[1] = clampi(128 - xqd[0] - xq[1], -32, 95); }
}

/* dgd: pixel (0, 0) of the 3-sample extended CDEF output (8-bit).  stats[n_pol].  noise[4]: sum of squared (exact - model) probe-vs-current
 * differences normalised by err_model / 12 ... (diagnostics: [0] count, [1] sum of z^2 with z = diff / sqrt(n / 12 ... )) */
int32_t *g_ref_xqd; int64_t *g_ref_err;   /* optional: [unit][16][2] / [unit][16] results of the reference walk (the caller compares them with the oracle's search) */
int sim_plane(const uint8_t *dgd, int stride, const uint8_t *src, int src_stride, int pw, int ph, int ss, int unit_size, uint32_t ep_mask,
              const Policy *pol, int n_pol, Stats *stats, double A, double B, double *noise, int max_units) {
    const int nu = orc_rest_units(pw, unit_size) * orc_rest_units(ph, unit_size);
    int32_t *lim = (int32_t *)malloc(sizeof(int32_t) * 4 * nu);
    orc_rest_unit_limits(pw, ph, ss, unit_size, lim);
    const int puw = 64 >> ss, puh = 64 >> ss;
    Unit *U = (Unit *)malloc(sizeof(Unit));
    for (int u = 0; u < nu && u < max_units; u++) {
        const int x0 = lim[4 * u], x1 = lim[4 * u + 1], y0 = lim[4 * u + 2], y1 = lim[4 * u + 3], w = x1 - x0, h = y1 - y0;
        const int fs = ((w + 7) & ~7) + 8;
        int32_t *f0 = (int32_t *)calloc((size_t)2 * fs * h, sizeof(int32_t)), *f1 = f0 + (size_t)fs * h;
        const uint8_t *d = dgd + (size_t)y0 * stride + x0, *s = src + (size_t)y0 * src_stride + x0;
        U->n = w * h; U->w = w; U->h = h;
        U->d0 = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)U->n); U->d1 = U->d0 + U->n; U->e0 = U->d1 + U->n;
        for (int ep = 0; ep < 16; ep++) {
            if (!((ep_mask >> ep) & 1)) continue;
            for (int i = 0; i < h; i += puh)
                for (int j = 0; j < w; j += puw)
                    orc_sgr_filter(d + (size_t)i * stride + j, 1, w - j < puw ? w - j : puw, h - i < puh ? h - i : puh, stride, f0 + (size_t)i * fs + j, f1 + (size_t)i * fs + j, fs, ep, 8);
            U->has0 = orc_sgr_params[ep][0] > 0; U->has1 = orc_sgr_params[ep][1] > 0; U->ep = ep;
            int64_t sums[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < h; i++)
                for (int j = 0; j < w; j++) {
                    const int32_t uu = d[(size_t)i * stride + j] << 4, sv = (s[(size_t)i * src_stride + j] << 4) - uu;
                    const int32_t a = U->has0 ? f0[(size_t)i * fs + j] - uu : 0, b = U->has1 ? f1[(size_t)i * fs + j] - uu : 0;
                    U->d0[i * w + j] = a; U->d1[i * w + j] = b; U->e0[i * w + j] = (int32_t)d[(size_t)i * stride + j] - (int32_t)s[(size_t)i * src_stride + j];
                    sums[0] += (int64_t)a * a; sums[1] += (int64_t)a * b; sums[2] += (int64_t)b * b; sums[3] += (int64_t)a * sv; sums[4] += (int64_t)b * sv;
                }
            U->H00 = (double)sums[0]; U->H01 = (double)sums[1]; U->H11 = (double)sums[2]; U->C0 = (double)sums[3]; U->C1 = (double)sums[4];
            for (int i = 0; i < 256 * 256; i++) U->memo[i] = -1;
            U->n_exact = 0;
            int start[2];
            solve_and_encode(sums, w * h, ep, start);
            int rxy[2]; int64_t rerr;
            const int nref = ref_points(U, start, rxy, &rerr);
            if (g_ref_xqd) { g_ref_xqd[((size_t)u * 16 + ep) * 2] = rxy[0]; g_ref_xqd[((size_t)u * 16 + ep) * 2 + 1] = rxy[1]; g_ref_err[(size_t)u * 16 + ep] = rerr; }
            /* noise of the model: for the probes of the reference walk, (exact diff - model diff / 2^22) against sqrt(err) */
            {
                /* model is in units of 2^22 x error (xq scaled by 2^-11, squared) up to a constant: exact ~ model / 2^22 + const */
                const double e_s = (double)exact_err(U, start[0], start[1]);
                const int probes[4][2] = {{-2, 0}, {2, 0}, {0, -2}, {0, 2}};
                for (int k = 0; k < 4; k++) {
                    if ((k < 2 && !U->has0) || (k >= 2 && !U->has1)) continue;
                    const int cx = clampi(start[0] + probes[k][0], -96, 31), cy = clampi(start[1] + probes[k][1], -32, 95);
                    const double ex = (double)exact_err(U, cx, cy) - e_s, mo = (model(U, cx, cy) - model(U, start[0], start[1])) / 4194304.0;
                    const double z = (ex - mo) / sqrt(e_s / 12.0 + 1.0);
                    noise[0] += 1; noise[1] += z * z; noise[2] += fabs(ex); noise[3] += fabs(mo);
                }
            }
            const double e_start = (double)exact_err(U, start[0], start[1]);
            for (int k = 0; k < n_pol; k++) {
                stats[k].ref_points += nref;
                /* sigma in model units: sigma_k x sqrt(err / 12) x 2^22  (2 sum(m delta) has variance 4 sum(m^2) / 12; differences of two nearby points correlate) */
                const double sigma = pol[k].sigma_k * sqrt(e_start / 12.0 + 1.0) * 4194304.0;
                run_policy(U, start, &pol[k], &stats[k], A, B, sigma, rxy, rerr);
            }
        }
        free(U->d0);
        free(f0);
    }
    free(U); free(lim);
    return nu;
}
