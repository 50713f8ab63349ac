/* tests/resident_table_test.c — drives integration/svt_hip_resident.c (the table behind SVT_HIP_RESIDENT) without the encoder: the three library calls it makes
 * are stubbed here on host memory, with failure injection.  Built and run by tests/test_resident_table.py; prints "ok" and exits 0 when every check holds. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svt_hip_resident.h"

static int  g_fail_alloc, g_fail_copy;
static long g_live_blocks, g_copies;
static pthread_mutex_t g_stub_mu = PTHREAD_MUTEX_INITIALIZER;
int svt_hip_malloc(SvtHipCtx *c, void **p, size_t bytes) {
    (void)c;
    if (g_fail_alloc) { *p = NULL; return SVT_HIP_ERR_RUNTIME; }
    *p = malloc(bytes);
    pthread_mutex_lock(&g_stub_mu); g_live_blocks++; pthread_mutex_unlock(&g_stub_mu);
    return *p ? SVT_HIP_OK : SVT_HIP_ERR_RUNTIME;
}
int svt_hip_free(SvtHipCtx *c, void *p) {
    (void)c;
    if (p) { pthread_mutex_lock(&g_stub_mu); g_live_blocks--; pthread_mutex_unlock(&g_stub_mu); }
    free(p);
    return SVT_HIP_OK;
}
int svt_hip_memcpy_h2d(SvtHipCtx *c, void *d, const void *h, size_t n) {
    (void)c;
    if (g_fail_copy) return SVT_HIP_ERR_RUNTIME;
    memcpy(d, h, n);
    pthread_mutex_lock(&g_stub_mu); g_copies++; pthread_mutex_unlock(&g_stub_mu);
    return SVT_HIP_OK;
}

static long g_pins;
int svt_hip_host_register(SvtHipCtx *c, void *h, size_t n) { (void)c; (void)h; (void)n; pthread_mutex_lock(&g_stub_mu); g_pins++; pthread_mutex_unlock(&g_stub_mu); return SVT_HIP_OK; }
int svt_hip_host_unregister(SvtHipCtx *c, void *h) { (void)c; (void)h; pthread_mutex_lock(&g_stub_mu); g_pins--; pthread_mutex_unlock(&g_stub_mu); return SVT_HIP_OK; }
static size_t pow2_block(size_t n) { size_t b = 256; while (b < n) b <<= 1; return b; }

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #x); exit(1); } } while (0)
static SvtHipCtx *const HIP = (SvtHipCtx *)(uintptr_t)0x1000;   /* opaque to the table */

static SvtHipResidentStats stats(void) { SvtHipResidentStats s; svt_hip_resident_stats(&s); return s; }

typedef struct { uint8_t *plane[4]; size_t bytes; int iters; } Hammer;
static void *hammer(void *arg) {
    Hammer *h = (Hammer *)arg;
    unsigned seed = (unsigned)(uintptr_t)&seed;
    for (int i = 0; i < h->iters; i++) {
        const int      k = (int)(rand_r(&seed) % 4);
        const uint8_t *d = (const uint8_t *)svt_hip_resident_acquire(HIP, h->plane[k], h->bytes);
        if (d) {
            /* an acquired copy is complete and belongs to the plane asked for (each plane is filled with its own byte) */
            CHECK(d[0] == (uint8_t)(k + 1) && d[h->bytes - 1] == (uint8_t)(k + 1) && d[h->bytes / 2] == (uint8_t)(k + 1));
            svt_hip_resident_release(h->plane[k]);
        }
    }
    return NULL;
}

int main(void) {
    enum { N = 1 << 16 };
    uint8_t *a = (uint8_t *)malloc(N), *b = (uint8_t *)malloc(N), *c = (uint8_t *)malloc(2 * N);
    memset(a, 1, N); memset(b, 2, N); memset(c, 3, 2 * N);

    /* off: nothing is ever resident */
    svt_hip_resident_configure(0, 0, 0, NULL, NULL);
    svt_hip_resident_note(a, N);
    CHECK(!svt_hip_resident_acquire(HIP, a, N));

    svt_hip_resident_configure(1, (size_t)3 * N, 0, NULL, NULL);
    /* never announced -> NULL; announced -> one upload, then hits on the same block */
    CHECK(!svt_hip_resident_acquire(HIP, a, N));
    svt_hip_resident_note(a, N);
    const uint8_t *da = (const uint8_t *)svt_hip_resident_acquire(HIP, a, N);
    CHECK(da && da != a && !memcmp(da, a, N) && stats().uploads == 1 && stats().hits == 0);
    CHECK(svt_hip_resident_acquire(HIP, a, N) == da && stats().uploads == 1 && stats().hits == 1);
    CHECK(!svt_hip_resident_acquire(HIP, a, N + 1));   /* further than announced */
    svt_hip_resident_release(a); svt_hip_resident_release(a);

    /* a host write without an announcement is not seen (the writers' contract); the announcement makes the next acquire upload again, into the same block */
    a[5] = 77;
    CHECK(svt_hip_resident_acquire(HIP, a, N) == da && da[5] == 1);
    svt_hip_resident_release(a);
    svt_hip_resident_note(a, N);
    CHECK(svt_hip_resident_acquire(HIP, a, N) == da && da[5] == 77 && stats().uploads == 2);

    /* announced again while the copy is being read: not resident for new callers until the reader has released; the reader's copy is untouched */
    a[6] = 88;
    svt_hip_resident_note(a, N);
    CHECK(!svt_hip_resident_acquire(HIP, a, N) && da[6] == 1);
    svt_hip_resident_release(a);
    CHECK(svt_hip_resident_acquire(HIP, a, N) == da && da[6] == 88 && stats().uploads == 3);
    svt_hip_resident_release(a);

    /* budget 3 N: a (N) + c (2 N) fit; b then evicts the least recently acquired unused copy (a), never one in use */
    svt_hip_resident_note(c, 2 * N); svt_hip_resident_note(b, N);
    const uint8_t *dc = (const uint8_t *)svt_hip_resident_acquire(HIP, c, 2 * N);
    CHECK(dc && !memcmp(dc, c, 2 * N) && stats().evictions == 0 && g_live_blocks == 2);
    const uint8_t *db = (const uint8_t *)svt_hip_resident_acquire(HIP, b, N);   /* c is in use: a goes */
    CHECK(db && !memcmp(db, b, N) && stats().evictions == 1 && g_live_blocks == 2);
    CHECK(!svt_hip_resident_acquire(HIP, a, N));                                /* b and c in use: no room, nothing to evict */
    svt_hip_resident_release(c);
    const uint8_t *da2 = (const uint8_t *)svt_hip_resident_acquire(HIP, a, N);  /* c is free now and goes; a is uploaded afresh (still announced) */
    CHECK(da2 && !memcmp(da2, a, N) && stats().evictions == 2);
    svt_hip_resident_release(a); svt_hip_resident_release(b);

    /* a larger extent announced for a known plane: the block is replaced */
    uint8_t *big = (uint8_t *)malloc(2 * N);
    memset(big, 9, 2 * N);
    svt_hip_resident_release_all(HIP);
    CHECK(g_live_blocks == 0 && stats().resident_mb == 0.0);
    svt_hip_resident_note(big, N);
    CHECK(svt_hip_resident_acquire(HIP, big, N));
    svt_hip_resident_release(big);
    svt_hip_resident_note(big, 2 * N);
    const uint8_t *dbig = (const uint8_t *)svt_hip_resident_acquire(HIP, big, 2 * N);
    CHECK(dbig && dbig[2 * N - 1] == 9 && g_live_blocks == 1);
    svt_hip_resident_release(big);

    /* failures leave the table consistent: a failed allocation or copy -> NULL, the plane stays announced, a later acquire succeeds */
    svt_hip_resident_release_all(HIP);
    svt_hip_resident_note(a, N);
    g_fail_alloc = 1; CHECK(!svt_hip_resident_acquire(HIP, a, N)); g_fail_alloc = 0;
    g_fail_copy = 1;  CHECK(!svt_hip_resident_acquire(HIP, a, N)); g_fail_copy = 0;
    CHECK(svt_hip_resident_acquire(HIP, a, N) && g_live_blocks == 1);
    svt_hip_resident_release(a);

    /* the test knob: re-announcements ignored -> the copy goes stale (what tests/test_encode_e2e.py::test_resident_planes_announcements_matter relies on) */
    svt_hip_resident_release_all(HIP);
    svt_hip_resident_configure(1, (size_t)64 * N, 1, NULL, NULL);
    memset(a, 1, N);
    svt_hip_resident_note(a, N);
    const uint8_t *ds = (const uint8_t *)svt_hip_resident_acquire(HIP, a, N);
    svt_hip_resident_release(a);
    a[0] = 200; svt_hip_resident_note(a, N);
    CHECK(svt_hip_resident_acquire(HIP, a, N) == ds && ds[0] == 1);
    svt_hip_resident_release(a);

    /* the budget counts what the allocator hands out (power-of-two blocks here: a plane of N + 256 bytes occupies 2 N), and host ranges are page-locked once */
    svt_hip_resident_release_all(HIP);
    svt_hip_resident_configure(1, (size_t)4 * N, 0, NULL, NULL);
    svt_hip_resident_configure_blocks(pow2_block, 1);
    memset(a, 1, N); memset(b, 2, N); memset(c, 3, 2 * N);
    svt_hip_resident_note(a, N); svt_hip_resident_note(b, N); svt_hip_resident_note(c, N);
    CHECK(svt_hip_resident_acquire(HIP, a, N) && svt_hip_resident_acquire(HIP, b, N) && stats().resident_mb == 4.0 * N / 1048576.0 && g_pins == 2);
    CHECK(!svt_hip_resident_acquire(HIP, c, N));   /* a third 2 N block does not fit 4 N */
    svt_hip_resident_release(a);
    svt_hip_resident_note(a, N);
    CHECK(svt_hip_resident_acquire(HIP, a, N) && g_pins == 2);   /* uploaded again, page-locked once */
    svt_hip_resident_release(a); svt_hip_resident_release(b);
    svt_hip_resident_unpin_all(HIP, 0);
    CHECK(g_pins == 0);
    svt_hip_resident_release_all(HIP);
    svt_hip_resident_configure_blocks(NULL, 0);

    /* eight threads over four planes with a budget of two: every acquired copy is complete and the right one; nothing is left in use */
    svt_hip_resident_release_all(HIP);
    svt_hip_resident_configure(1, (size_t)2 * N, 0, NULL, NULL);
    Hammer h; h.bytes = N; h.iters = 20000;
    for (int k = 0; k < 4; k++) { h.plane[k] = (uint8_t *)malloc(N); memset(h.plane[k], k + 1, N); svt_hip_resident_note(h.plane[k], N); }
    pthread_t th[8];
    for (int i = 0; i < 8; i++) CHECK(!pthread_create(&th[i], NULL, hammer, &h));
    for (int i = 0; i < 8; i++) pthread_join(th[i], NULL);
    const SvtHipResidentStats s = stats();
    CHECK(g_live_blocks <= 2 && s.resident_mb <= 2.0 * N / 1048576.0 && s.hits > 0 && s.evictions > 0);
    svt_hip_resident_release_all(HIP);   /* would free blocks in use twice if a user count were off: valgrind / ASan runs of this file see it */
    CHECK(g_live_blocks == 0);
    for (int k = 0; k < 4; k++) free(h.plane[k]);
    free(a); free(b); free(c); free(big);
    printf("ok notes=%ld uploads=%ld hits=%ld evictions=%ld refused=%ld\n", s.notes, s.uploads, s.hits, s.evictions, s.refused);
    return 0;
}
