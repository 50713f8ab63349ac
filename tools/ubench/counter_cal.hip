// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per access width (/opt/skills/guides/MI355X_MICROARCH.md "HBM": the counters are derived from the
// L2's fabric-side request counters; a 16-byte-per-lane coalesced read is reported at exactly half its bytes, other widths are uncalibrated).  Every kernel below
// moves a KNOWN number of bytes, coalesced, with one access width per lane: tools/calibrate_counters.sh runs this under the two PMC passes and writes
// profiles/<round>/counter_calibration.json = true bytes / counter bytes per (direction, width), which tools/roofline_defs.py applies to each kernel's traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/counter_cal.hip -o tools/ubench/counter_cal
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T> struct Acc { static __device__ unsigned fold(T v); };
template <> __device__ unsigned Acc<uint8_t>::fold(uint8_t v) { return v; }
template <> __device__ unsigned Acc<uint16_t>::fold(uint16_t v) { return v; }
template <> __device__ unsigned Acc<uint32_t>::fold(uint32_t v) { return v; }
template <> __device__ unsigned Acc<uint2>::fold(uint2 v) { return v.x ^ v.y; }
template <> __device__ unsigned Acc<uint4>::fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// every lane reads n / (threads) elements of type T, consecutive lanes consecutive elements; the xor goes out so that nothing is optimised away
template <typename T> __global__ void __launch_bounds__(256) cal_read(const T* __restrict__ p, size_t n, unsigned* out, unsigned magic) {
    unsigned a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a ^= Acc<T>::fold(p[i]);
    if (a == magic) out[0] = a;   // magic is a run-time value: the loads stay
}
template <typename T> __device__ T make(unsigned v);
template <> __device__ uint16_t make<uint16_t>(unsigned v) { return (uint16_t)v; }
template <> __device__ uint32_t make<uint32_t>(unsigned v) { return v; }
template <> __device__ uint2 make<uint2>(unsigned v) { return make_uint2(v, v + 1); }
template <> __device__ uint4 make<uint4>(unsigned v) { return make_uint4(v, v + 1, v + 2, v + 3); }
template <> __device__ uint8_t make<uint8_t>(unsigned v) { return (uint8_t)v; }
template <typename T> __global__ void __launch_bounds__(256) cal_write(T* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make<T>((unsigned)i);
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB: four times the Infinity Cache
    void* buf; unsigned* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((cal_read<uint8_t>), dim3(grid), dim3(256), 0, 0, (const uint8_t*)buf, bytes / 8, out, 0x00ABu + (unsigned)rep);          // the narrow kernels move 1/8 GiB .. 1 GiB
        hipLaunchKernelGGL((cal_read<uint16_t>), dim3(grid), dim3(256), 0, 0, (const uint16_t*)buf, bytes / 8, out, 0x00ABu + (unsigned)rep);
        hipLaunchKernelGGL((cal_read<uint32_t>), dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 8, out, 0x00ABu + (unsigned)rep);
        hipLaunchKernelGGL((cal_read<uint2>), dim3(grid), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out, 0x00ABu + (unsigned)rep);
        hipLaunchKernelGGL((cal_read<uint4>), dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out, 0x00ABu + (unsigned)rep);
        hipLaunchKernelGGL((cal_write<uint8_t>), dim3(grid), dim3(256), 0, 0, (uint8_t*)buf, bytes / 8);
        hipLaunchKernelGGL((cal_write<uint16_t>), dim3(grid), dim3(256), 0, 0, (uint16_t*)buf, bytes / 8);
        hipLaunchKernelGGL((cal_write<uint32_t>), dim3(grid), dim3(256), 0, 0, (uint32_t*)buf, bytes / 8);
        hipLaunchKernelGGL((cal_write<uint2>), dim3(grid), dim3(256), 0, 0, (uint2*)buf, bytes / 8);
        hipLaunchKernelGGL((cal_write<uint4>), dim3(grid), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    }
    CK(hipDeviceSynchronize());
    // true bytes per launch: elements x sizeof(T)
    printf("TRUE cal_read<unsigned char> %zu\nTRUE cal_read<unsigned short> %zu\nTRUE cal_read<unsigned int> %zu\nTRUE cal_read<uint2> %zu\nTRUE cal_read<uint4> %zu\n",
           bytes / 8, bytes / 4, bytes / 2, bytes, bytes);
    printf("TRUE cal_write<unsigned char> %zu\nTRUE cal_write<unsigned short> %zu\nTRUE cal_write<unsigned int> %zu\nTRUE cal_write<uint2> %zu\nTRUE cal_write<uint4> %zu\n",
           bytes / 8, bytes / 4, bytes / 2, bytes, bytes);
    return 0;
}
