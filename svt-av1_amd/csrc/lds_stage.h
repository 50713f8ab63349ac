// lds_stage.h — byte-misalignment-removing global -> LDS staging shared by the SAD search kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct __attribute__((aligned(4))) Dw2 { uint32_t x, y; };  // 8-byte value that is only 4-byte aligned in LDS
__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// Stage `rows` x `row_dw` dwords starting at byte address `base` (row pitch `pitch` bytes, pitch % 4
// == 0) into LDS with the (base & 3) misalignment removed.  Loads are aligned dwords; a dword is
// fetched only if it overlaps [0, need_bytes) of its row so nothing outside the window's aligned
// footprint is touched.
__device__ __forceinline__ void stage_rows(uint32_t* lds, int lds_stride_dw, const uint8_t* base, int pitch,
                                           int rows, int row_dw, int need_bytes, int tid, int nthreads) {
    const uint32_t shift = (uint32_t)((uintptr_t)base & 3);
    const uint32_t* g0   = (const uint32_t*)(base - shift);
    const int last_dw    = (need_bytes + (int)shift + 3) / 4;  // dwords [0,last_dw) overlap the needed bytes
    const int total      = rows * row_dw;
    for (int i = tid; i < total; i += nthreads) {
        const int r = i / row_dw, j = i - r * row_dw;
        const uint32_t* g = g0 + (size_t)r * (pitch >> 2) + j;
        uint32_t lo = (j < last_dw) ? g[0] : 0u;
        uint32_t hi = (j + 1 < last_dw) ? g[1] : 0u;
        lds[r * lds_stride_dw + j] = __builtin_amdgcn_alignbyte(hi, lo, shift);
    }
}

