// lds_stage.h — byte-misalignment-removing global -> LDS staging shared by the SAD search kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// total elements, element i -> value load(i), then store(i, value): U independent global loads are issued before the first store, so a
// thread pays one memory latency per U elements instead of one per element (the compiler does not pipeline the plain loop).
template <int U, typename T, typename LoadF, typename StoreF>
__device__ __forceinline__ void batched_stage(int total, int tid, int nthreads, LoadF load, StoreF store) {
    for (int i0 = tid; i0 < total; i0 += nthreads * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = load(min(i0 + u * nthreads, total - 1));   // unconditional (a clamped index): a load under `if (i < total)` becomes a branch with its own s_waitcnt vmcnt(0) and the U loads run one after the other
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * nthreads; if (i < total) store(i, v[u]); }
    }
}

struct __attribute__((aligned(4))) Dw2 { uint32_t x, y; };  // 8-byte value that is only 4-byte aligned in LDS
__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// Stage `rows` x `row_dw` dwords starting at byte address `base` (row pitch `pitch` bytes, pitch % 4
// == 0) into LDS with the (base & 3) misalignment removed.  Loads are aligned dwords; a dword is
// fetched only if it overlaps [0, need_bytes) of its row so nothing outside the window's aligned
// footprint is touched.
template <int U = 8>
__device__ __forceinline__ void stage_rows(uint32_t* lds, int lds_stride_dw, const uint8_t* base, int pitch,
                                           int rows, int row_dw, int need_bytes, int tid, int nthreads) {
    const uint32_t shift = (uint32_t)((uintptr_t)base & 3);
    const uint32_t* g0   = (const uint32_t*)(base - shift);
    const int last_dw    = (need_bytes + (int)shift + 3) / 4;  // dwords [0,last_dw) overlap the needed bytes
    const int total      = rows * row_dw;
    // U (template parameter; registers: 3 per unit) independent loads are issued before the first one is consumed: a plain one-element-per-iteration loop exposes one full
    // global-memory latency per element (the compiler does not software-pipeline it), which made staging ~1/4 of a search kernel
    for (int i0 = tid; i0 < total; i0 += nthreads * U) {
        uint32_t lo[U], hi[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * nthreads;
            dst[u] = -1; lo[u] = 0; hi[u] = 0;
            if (i < total) {
                const int r = i / row_dw, j = i - r * row_dw;
                const uint32_t* g = g0 + (size_t)r * (pitch >> 2) + j;
                dst[u] = r * lds_stride_dw + j;
                if (j < last_dw) lo[u] = g[0];
                if (shift && j + 1 < last_dw) hi[u] = g[1];   // aligned windows need no second dword
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (dst[u] >= 0) lds[dst[u]] = __builtin_amdgcn_alignbyte(hi[u], lo[u], shift);
    }
}
