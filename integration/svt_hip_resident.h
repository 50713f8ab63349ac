/*
 * svt_hip_resident.h — the table of resident planes behind SVT_HIP_RESIDENT (integration/svt_hip_hooks.h says who announces and who asks).
 *
 * Plain C on the library's C ABI (svt_hip.h) and pthreads only — no reference types — so that tests/test_resident_table.py can drive it against the CPU test
 * double without the encoder.  A plane is a host range [host, host + bytes) whose writer announces every (re)write AFTER it is complete:
 *   note    : the range was written.  Its device copy (if any) is out of date; the block is kept and overwritten by the next acquire.
 *   acquire : the device copy — uploaded now, on the caller's context, when the last announcement is newer than the copy — or NULL: never announced, smaller than
 *             `bytes`, announced again while other callers still read the old copy (cannot happen by the reference's life-time rules; the caller uploads what it
 *             needs itself), over the budget with nothing to evict, allocation or copy failed.  The upload is synchronous (svt_hip_memcpy_h2d drains the context),
 *             so every context of the device may read the copy when acquire returns; it runs on the caller's context OUTSIDE the table's lock (the entry is marked
 *             "uploading": callers that want the same plane wait for it, everybody else passes), from the host range page-locked in place on first use.
 *   release : pairs with a successful acquire, after the launches that read the copy have completed.  Copies in use are never evicted.
 * The budget evicts the least recently acquired copies nobody uses and counts what a block really occupies (the allocator's block size).  All functions are thread-safe.
 */
#ifndef SVT_HIP_RESIDENT_H
#define SVT_HIP_RESIDENT_H
#include <stddef.h>
#include "svt_hip.h"

typedef int  (*SvtHipResidentMalloc)(SvtHipCtx *hip, void **p, size_t bytes);
typedef void (*SvtHipResidentFree)(SvtHipCtx *hip, void *p);
typedef struct { long notes, uploads, hits, evictions, refused; double uploaded_mb, resident_mb; } SvtHipResidentStats;

/* on = 0: every acquire returns NULL.  limit_bytes: budget of device copies.  ignore_renotes: test knob (SVT_HIP_RESIDENT_FAULT) — only a plane's first
 * announcement counts, its copy goes stale.  alloc / release: device blocks (NULL = svt_hip_malloc / svt_hip_free). */
void        svt_hip_resident_configure(int on, size_t limit_bytes, int ignore_renotes, SvtHipResidentMalloc alloc, SvtHipResidentFree release);
/* block_size: what the allocator really hands out for a request (NULL = the request itself) — the budget counts that.  pin_host: page-lock a plane's host range
 * in place at its first upload (svt_hip_host_register; released by svt_hip_resident_unpin_all / _release_all). */
typedef size_t (*SvtHipResidentBlockSize)(size_t bytes);
void        svt_hip_resident_configure_blocks(SvtHipResidentBlockSize block_size, int pin_host);
void        svt_hip_resident_unpin_all(SvtHipCtx *hip, int keep_pinning);   /* before an encoder instance frees its pictures: no host range stays page-locked; keep_pinning = 0: and no new one is */
int         svt_hip_resident_enabled(void);
void        svt_hip_resident_note(const void *host, size_t bytes);
const void *svt_hip_resident_acquire(SvtHipCtx *hip, const void *host, size_t bytes);
void        svt_hip_resident_release(const void *host);
void        svt_hip_resident_release_all(SvtHipCtx *hip);   /* no plane is in use any more: every block goes back, every announcement is forgotten */
void        svt_hip_resident_stats(SvtHipResidentStats *out);
#endif
