// Sustained HBM WRITE bandwidth of plain / nontemporal 16-byte stores (grid-stride fill of a buffer far larger than L2 + MALL),
// next to a copy (read + write) and a pure read of the same size.
//   hipcc -O3 --offload-arch=gfx950 write_bw_probe.hip -o write_bw_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 plain store, 1 nontemporal store, 2 copy, 3 read
__global__ __launch_bounds__(256) void k(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16, u32x4* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        if (MODE == 0) dst[i] = u32x4{(unsigned)i, 1, 2, 3};
        if (MODE == 1) __builtin_nontemporal_store(u32x4{(unsigned)i, 1, 2, 3}, dst + i);
        if (MODE == 2) dst[i] = src[i];
        if (MODE == 3) { const u32x4 v = src[i]; acc ^= v; }
    }
    if (MODE == 3 && acc.x == 0x12345u) sink[0] = acc;
}

template <typename F>
double time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    const size_t bytes = 4ull << 30, n16 = bytes / 16;
    u32x4 *a, *b;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(b, 1, bytes));
    for (int wgs : {1024, 4096, 16384, 65536}) {
        double us;
        us = time_us([&] { hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, a, b, n16, a); }, 3);
        printf("%6d WGs  plain store   %8.1f us  write %5.2f TB/s\n", wgs, us, bytes / us / 1e6);
        us = time_us([&] { hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, a, b, n16, a); }, 3);
        printf("%6d WGs  nt store      %8.1f us  write %5.2f TB/s\n", wgs, us, bytes / us / 1e6);
        us = time_us([&] { hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, a, b, n16, a); }, 3);
        printf("%6d WGs  copy          %8.1f us  read + write %5.2f TB/s\n", wgs, us, 2.0 * bytes / us / 1e6);
        us = time_us([&] { hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, a, b, n16, a); }, 3);
        printf("%6d WGs  read          %8.1f us  read %5.2f TB/s\n", wgs, us, bytes / us / 1e6);
    }
    CHECK(hipMemsetAsync(a, 0, bytes, 0));
    double us = time_us([&] { CHECK(hipMemsetAsync(a, 0, bytes, 0)); }, 3);
    printf("hipMemsetAsync          %8.1f us  write %5.2f TB/s\n", us, bytes / us / 1e6);
    return 0;
}
