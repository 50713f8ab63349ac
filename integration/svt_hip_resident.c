/* svt_hip_resident.c — see svt_hip_resident.h.  Host bookkeeping only; the copies are made by the library's svt_hip_memcpy_h2d. */
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include "svt_hip_resident.h"

/* an encoder instance announces three planes per analysis reference object and three per input picture: a few hundred at the deepest look-ahead */
#define RES_MAX 2048
typedef struct {
    const uint8_t     *host;
    size_t             bytes, dev_bytes;   /* announced extent; extent the block was allocated for */
    void              *dev;
    int                stale, users;
    unsigned long long last;
} ResEntry;
static ResEntry             g_res[RES_MAX];
static pthread_mutex_t      g_mu = PTHREAD_MUTEX_INITIALIZER;
static int                  g_on, g_ignore_renotes;
static size_t               g_limit = (size_t)16384 << 20, g_dev_bytes;
static unsigned long long   g_clock;
static SvtHipResidentStats  g_st;
static SvtHipResidentMalloc g_alloc;
static SvtHipResidentFree   g_free;

static int  block_alloc(SvtHipCtx *hip, void **p, size_t bytes) { return g_alloc ? g_alloc(hip, p, bytes) : svt_hip_malloc(hip, p, bytes); }
static void block_free(SvtHipCtx *hip, void *p) { if (g_free) g_free(hip, p); else (void)svt_hip_free(hip, p); }

void svt_hip_resident_configure(int on, size_t limit_bytes, int ignore_renotes, SvtHipResidentMalloc alloc, SvtHipResidentFree release) {
    pthread_mutex_lock(&g_mu);
    g_on = on; g_ignore_renotes = ignore_renotes;
    if (limit_bytes) g_limit = limit_bytes;
    g_alloc = alloc; g_free = release;
    pthread_mutex_unlock(&g_mu);
}
int svt_hip_resident_enabled(void) { return g_on; }

static ResEntry *find(const void *host) {   /* g_mu held */
    for (int i = 0; i < RES_MAX; i++)
        if (g_res[i].host == (const uint8_t *)host) return &g_res[i];
    return NULL;
}

void svt_hip_resident_note(const void *host, size_t bytes) {
    if (!g_on || !host || !bytes) return;
    pthread_mutex_lock(&g_mu);
    g_st.notes++;
    ResEntry *e = find(host);
    if (e) {
        if (!g_ignore_renotes) { e->stale = 1; e->bytes = bytes; }   /* the block (if any) is kept: the next acquire overwrites it */
    } else {
        for (int i = 0; i < RES_MAX && !e; i++)
            if (!g_res[i].host) e = &g_res[i];
        if (e) { memset(e, 0, sizeof(*e)); e->host = (const uint8_t *)host; e->bytes = bytes; e->stale = 1; }
        /* table full: the plane is simply never resident */
    }
    pthread_mutex_unlock(&g_mu);
}

static void drop_block(SvtHipCtx *hip, ResEntry *e) {   /* g_mu held, e->users == 0 */
    block_free(hip, e->dev);
    e->dev = NULL; e->stale = 1;
    g_dev_bytes -= e->dev_bytes;
    e->dev_bytes = 0;
}
/* copies nobody uses go back, least recently acquired first, until `need` more bytes fit the budget (g_mu held) */
static void make_room(SvtHipCtx *hip, size_t need, const ResEntry *keep) {
    while (g_dev_bytes + need > g_limit) {
        ResEntry *v = NULL;
        for (int i = 0; i < RES_MAX; i++)
            if (g_res[i].dev && !g_res[i].users && &g_res[i] != keep && (!v || g_res[i].last < v->last)) v = &g_res[i];
        if (!v) return;
        drop_block(hip, v);
        g_st.evictions++;
    }
}

const void *svt_hip_resident_acquire(SvtHipCtx *hip, const void *host, size_t bytes) {
    if (!g_on || !hip || !host) return NULL;
    const void *ret = NULL;
    pthread_mutex_lock(&g_mu);
    ResEntry *e = find(host);
    do {
        if (!e) break;                         /* never announced */
        if (e->bytes < bytes) break;           /* the caller reads further than what was announced */
        if (e->stale && e->users) break;       /* announced again while the old copy is being read */
        if (e->stale || !e->dev) {
            if (e->dev && e->dev_bytes < e->bytes) drop_block(hip, e);   /* announced again with a larger extent */
            if (!e->dev) {
                make_room(hip, e->bytes, e);
                if (g_dev_bytes + e->bytes > g_limit) break;
                /* + 256: dword-aligned window loads of the search kernels may run a few bytes past the last row */
                if (block_alloc(hip, &e->dev, e->bytes + 256) != SVT_HIP_OK) { e->dev = NULL; break; }
                e->dev_bytes = e->bytes;
                g_dev_bytes += e->bytes;
            }
            if (svt_hip_memcpy_h2d(hip, e->dev, e->host, e->bytes) != SVT_HIP_OK) break;   /* stays out of date */
            e->stale = 0;
            g_st.uploads++;
            g_st.uploaded_mb += e->bytes / 1048576.0;
        } else
            g_st.hits++;
        e->users++;
        e->last = ++g_clock;
        ret = e->dev;
    } while (0);
    if (!ret && e) g_st.refused++;
    pthread_mutex_unlock(&g_mu);
    return ret;
}

void svt_hip_resident_release(const void *host) {
    if (!g_on || !host) return;
    pthread_mutex_lock(&g_mu);
    ResEntry *e = find(host);
    if (e && e->users > 0) e->users--;
    pthread_mutex_unlock(&g_mu);
}

void svt_hip_resident_release_all(SvtHipCtx *hip) {
    pthread_mutex_lock(&g_mu);
    for (int i = 0; i < RES_MAX; i++) {
        if (g_res[i].dev) block_free(hip, g_res[i].dev);
        memset(&g_res[i], 0, sizeof(g_res[i]));
    }
    g_dev_bytes = 0;
    pthread_mutex_unlock(&g_mu);
}

void svt_hip_resident_stats(SvtHipResidentStats *out) {
    pthread_mutex_lock(&g_mu);
    *out = g_st;
    out->resident_mb = g_dev_bytes / 1048576.0;
    pthread_mutex_unlock(&g_mu);
}
