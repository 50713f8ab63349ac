#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t bytes4(uint32_t d0, uint32_t d1, uint32_t d2, int sh) {
    return sh < 4 ? __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)sh) : (sh < 8 ? __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)(sh - 4)) : d2);
}
__global__ void k(const uint32_t* in, int o, int TA, int TB, uint32_t* out) {
    uint32_t d0 = in[0] ^ 0x80808080u, d1 = in[1] ^ 0x80808080u, d2 = in[2] ^ 0x80808080u;
    for (int q = 0; q < 4; q++) {
        int sum = 128 * 128 + 64;
        uint32_t A = bytes4(d0, d1, d2, o + q), B = bytes4(d0, d1, d2, o + q + 4);
        out[8 * q] = A; out[8 * q + 1] = B;
        int s1 = __builtin_amdgcn_sdot4((int)A, TA, sum, false);
        int s2 = __builtin_amdgcn_sdot4((int)B, TB, s1, false);
        out[8 * q + 2] = s1; out[8 * q + 3] = s2; out[8 * q + 4] = min(max(s2 >> 7, 0), 255);
    }
}
int main() {
    uint32_t h[3] = {0x64646464, 0x64646464, 0x64646464}, *d, *o, r[32];
    hipMalloc(&d, 12); hipMalloc(&o, 128); hipMemcpy(d, h, 12, hipMemcpyHostToDevice);
    int f[8] = {0, 0, -12, 110, 38, -8, 0, 0};
    int TA = (f[0] & 0xff) | (f[1] & 0xff) << 8 | (f[2] & 0xff) << 16 | (f[3] & 0xff) << 24, TB = (f[4] & 0xff) | (f[5] & 0xff) << 8 | (f[6] & 0xff) << 16 | (f[7] & 0xff) << 24;
    for (int oo = 0; oo < 2; oo++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, oo, TA, TB, o); hipMemcpy(r, o, 128, hipMemcpyDeviceToHost);
        for (int q = 0; q < 4; q++) printf("o=%d q=%d A=%08x B=%08x s1=%d s2=%d out=%u\n", oo, q, r[8*q], r[8*q+1], (int)r[8*q+2], (int)r[8*q+3], r[8*q+4]);
    }
    printf("TA=%08x TB=%08x\n", TA, TB);
}
