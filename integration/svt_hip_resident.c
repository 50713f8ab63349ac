/* svt_hip_resident.c — see svt_hip_resident.h.  Host bookkeeping only; the copies are made by the library's svt_hip_memcpy_h2d. */
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include "svt_hip_resident.h"

/* an encoder instance announces three planes per analysis reference object and three per input picture: a few hundred at the deepest look-ahead.
 * Open addressing on the host pointer; entries are only ever removed all at once (svt_hip_resident_release_all), so there are no tombstones. */
#define RES_SLOTS 4096   /* power of two */
#define RES_MAX   2048   /* at most half full */
typedef struct {
    const uint8_t     *host;
    size_t             bytes, dev_bytes;   /* announced extent; what the block counts against the budget (the allocator's block size) */
    void              *dev;
    int                stale, users, uploading, pinned;
    unsigned           gen;                /* announcements so far: an upload that was overtaken by one stays out of date */
    unsigned long long last;
} ResEntry;
static ResEntry             g_res[RES_SLOTS];
static int                  g_res_n;
static pthread_mutex_t      g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t       g_cv = PTHREAD_COND_INITIALIZER;   /* an upload has ended */
static int                  g_on, g_ignore_renotes, g_pin;
static size_t               g_limit = (size_t)16384 << 20, g_dev_bytes;
static unsigned long long   g_clock;
static SvtHipResidentStats  g_st;
static SvtHipResidentMalloc g_alloc;
static SvtHipResidentFree   g_free;
static SvtHipResidentBlockSize g_block_size;

static int    block_alloc(SvtHipCtx *hip, void **p, size_t bytes) { return g_alloc ? g_alloc(hip, p, bytes) : svt_hip_malloc(hip, p, bytes); }
static void   block_free(SvtHipCtx *hip, void *p) { if (g_free) g_free(hip, p); else (void)svt_hip_free(hip, p); }
/* + 256: dword-aligned window loads of the search kernels may run a few bytes past the last row */
static size_t block_bytes(size_t plane_bytes) { return g_block_size ? g_block_size(plane_bytes + 256) : plane_bytes; }

void svt_hip_resident_configure(int on, size_t limit_bytes, int ignore_renotes, SvtHipResidentMalloc alloc, SvtHipResidentFree release) {
    pthread_mutex_lock(&g_mu);
    g_on = on; g_ignore_renotes = ignore_renotes;
    if (limit_bytes) g_limit = limit_bytes;
    g_alloc = alloc; g_free = release;
    pthread_mutex_unlock(&g_mu);
}
void svt_hip_resident_configure_blocks(SvtHipResidentBlockSize block_size, int pin_host) {
    pthread_mutex_lock(&g_mu);
    g_block_size = block_size; g_pin = pin_host;
    pthread_mutex_unlock(&g_mu);
}
int svt_hip_resident_enabled(void) { return g_on; }

static unsigned slot_of(const void *host) { return (unsigned)(((uintptr_t)host >> 6) * 2654435761u) & (RES_SLOTS - 1); }
static ResEntry *find(const void *host) {   /* g_mu held */
    for (unsigned i = slot_of(host), n = 0; n < RES_SLOTS; i = (i + 1) & (RES_SLOTS - 1), n++) {
        if (g_res[i].host == (const uint8_t *)host) return &g_res[i];
        if (!g_res[i].host) return NULL;
    }
    return NULL;
}
static ResEntry *insert(const void *host) {   /* g_mu held, host not in the table */
    if (g_res_n >= RES_MAX) return NULL;      /* table full: the plane is simply never resident */
    unsigned i = slot_of(host);
    while (g_res[i].host) i = (i + 1) & (RES_SLOTS - 1);
    memset(&g_res[i], 0, sizeof(g_res[i]));
    g_res[i].host = (const uint8_t *)host;
    g_res_n++;
    return &g_res[i];
}

void svt_hip_resident_note(const void *host, size_t bytes) {
    if (!g_on || !host || !bytes) return;
    pthread_mutex_lock(&g_mu);
    g_st.notes++;
    ResEntry *e = find(host);
    if (e) {
        if (!g_ignore_renotes) { e->stale = 1; e->bytes = bytes; e->gen++; }   /* the block (if any) is kept: the next acquire overwrites it */
    } else if ((e = insert(host))) {
        e->bytes = bytes; e->stale = 1;
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
        for (int i = 0; i < RES_SLOTS; i++)
            if (g_res[i].dev && !g_res[i].users && !g_res[i].uploading && &g_res[i] != keep && (!v || g_res[i].last < v->last)) v = &g_res[i];
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
        while (e->uploading) pthread_cond_wait(&g_cv, &g_mu);   /* another context is bringing this plane up: its copy serves this caller too */
        if (e->bytes < bytes) break;           /* the caller reads further than what was announced */
        if (e->stale && e->users) break;       /* announced again while the old copy is being read */
        if (e->stale || !e->dev) {
            const size_t need = block_bytes(e->bytes);
            if (e->dev && e->dev_bytes < need) drop_block(hip, e);   /* announced again with a larger extent */
            int fresh = 0;
            if (!e->dev) {
                make_room(hip, need, e);
                if (g_dev_bytes + need > g_limit) break;
                g_dev_bytes += need;           /* reserved: the block is allocated outside the lock */
                e->dev_bytes = need;
                fresh = 1;
            }
            /* The allocation, the page-locking of the host range (once per range) and the copy run on the caller's context WITHOUT the table's lock: other planes'
             * readers are not held up; readers of this plane wait for `uploading` to clear.  svt_hip_memcpy_h2d drains the caller's stream, so the copy is
             * complete, and visible to every context of the device, before the entry is published. */
            e->uploading = 1;
            const unsigned gen = e->gen;
            const size_t   n = e->bytes;
            const int      pin = g_pin && !e->pinned;
            void          *dev = e->dev;
            pthread_mutex_unlock(&g_mu);
            int ok = 1, pinned = 0;
            if (fresh && block_alloc(hip, &dev, n + 256) != SVT_HIP_OK) { dev = NULL; ok = 0; }
            if (ok && pin) pinned = svt_hip_host_register(hip, (void *)(uintptr_t)host, n) == SVT_HIP_OK;   /* a refusal only costs speed */
            if (ok && svt_hip_memcpy_h2d(hip, dev, host, n) != SVT_HIP_OK) ok = 0;
            pthread_mutex_lock(&g_mu);
            e->uploading = 0;
            if (pinned) e->pinned = 1;
            if (fresh) {
                if (dev) e->dev = dev;
                else { g_dev_bytes -= e->dev_bytes; e->dev_bytes = 0; }
            }
            pthread_cond_broadcast(&g_cv);
            if (!ok) break;                    /* stays out of date */
            g_st.uploads++;
            g_st.uploaded_mb += n / 1048576.0;
            if (e->gen != gen) break;          /* announced again during the copy: what was copied may be torn */
            e->stale = 0;
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

/* the host ranges stop being page-locked (before the encoder frees them); the device copies stay */
void svt_hip_resident_unpin_all(SvtHipCtx *hip, int keep_pinning) {
    pthread_mutex_lock(&g_mu);
    for (int i = 0; i < RES_SLOTS; i++)
        if (g_res[i].host && g_res[i].pinned) { (void)svt_hip_host_unregister(hip, (void *)(uintptr_t)g_res[i].host); g_res[i].pinned = 0; }
    if (!keep_pinning) g_pin = 0;   /* keep_pinning: other encoder instances go on; their planes are page-locked again by their next upload */
    pthread_mutex_unlock(&g_mu);
}

void svt_hip_resident_release_all(SvtHipCtx *hip) {
    pthread_mutex_lock(&g_mu);
    for (int i = 0; i < RES_SLOTS; i++) {
        if (g_res[i].host && g_res[i].pinned) (void)svt_hip_host_unregister(hip, (void *)(uintptr_t)g_res[i].host);
        if (g_res[i].dev) block_free(hip, g_res[i].dev);
        memset(&g_res[i], 0, sizeof(g_res[i]));
    }
    g_res_n = 0;
    g_dev_bytes = 0;
    pthread_mutex_unlock(&g_mu);
}

void svt_hip_resident_stats(SvtHipResidentStats *out) {
    pthread_mutex_lock(&g_mu);
    *out = g_st;
    out->resident_mb = g_dev_bytes / 1048576.0;
    pthread_mutex_unlock(&g_mu);
}
