// svt_hip_host.h — internal: host-only pieces shared by svt_hip_api.cpp and svt_hip_host.cpp (not part of the public ABI).
#pragma once
#include <stdint.h>
#include "../../include/svt_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* one probe of the filter-level search: the SSE of the plane deblocked at (level_v, level_h) against the source; < 0 = failure */
typedef int64_t (*SvtHipTryLevelFn)(void* user, int level_v, int level_h);
int svt_hip_dlf_search_levels_host(const SvtHipDlfSearch* p, SvtHipTryLevelFn try_fn, void* user, int* best_level, int64_t* best_err);
int svt_hip_dlf_search_plan(const SvtHipDlfSearch* p, const int64_t* ss_err, int need[2], int* best_level, int64_t* best_err);   /* the walk replayed on known errors: the levels it needs next (0 = done) */
void svt_hip_dlf_search_probe_levels(const SvtHipDlfSearch* p, int lvl, int* lv_v, int* lv_h);
#ifdef __cplusplus
}
#endif
