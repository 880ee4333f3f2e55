// What do 16-byte global stores at 2-byte alignment cost?  The pattern tap of the T = 577 attention kernel leaves as 4 rows x 256
// contiguous bytes per wave-instruction, rows 1154 B apart (only 2-byte aligned).  Same traffic, three forms:
//   aligned   rows padded to 1168 B (16-byte aligned pieces)
//   odd       rows 1154 B apart, pieces at the rows' own alignment (what the kernel does)
//   flat      the same bytes as one flat aligned stream (what a funnel-shifted writer would emit)
//   hipcc -O3 --offload-arch=gfx950 unaligned_store_probe.hip -o unaligned_store_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct __attribute__((packed, aligned(2))) U4a2 { unsigned x, y, z, w; };

// one wave per 32-row strip of a [rows][T] bf16 matrix; per window of 128 keys: 8 instructions of 4 rows x 256 B
template <int MODE>
__global__ __launch_bounds__(256) void k_store(unsigned char* __restrict__ dst, int T, int row_bytes, int n_strips, int repeat) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wave;
    if (strip >= n_strips) return;
    const int st_row = lane >> 4, st_ch = lane & 15;
    const uint4 v = make_uint4(lane, strip, 0x3c003c00u, 0x3c003c00u);
    for (int rep = 0; rep < repeat; ++rep) {
        if (MODE == 2) {
            unsigned char* base = dst + (size_t)strip * 32 * row_bytes;
            for (int off = lane * 16; off + 16 <= 32 * row_bytes; off += 1024) *reinterpret_cast<uint4*>(base + off) = v;
            continue;
        }
        for (int k0 = 0; k0 < T; k0 += 128) {
            const int nb = min(256, (T - k0) * 2);
#pragma unroll 2
            for (int it = 0; it < 8; ++it) {
                const int row = st_row + 4 * it;
                if (st_ch * 16 + 16 <= nb) {
                    unsigned char* d = dst + (size_t)(strip * 32 + row) * row_bytes + k0 * 2 + st_ch * 16;
                    if (MODE == 0) *reinterpret_cast<uint4*>(d) = v;
                    else *reinterpret_cast<U4a2*>(d) = U4a2{v.x, v.y, v.z, v.w};
                }
            }
        }
    }
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
    const int T = 577, heads = 2048;
    const int n_rows = heads * T, n_strips = (n_rows + 31) / 32;
    unsigned char* dst;
    CHECK(hipMalloc(&dst, (size_t)(n_rows + 64) * 1168 + 4096));
    const double bytes = (double)n_rows * T * 2;
    const dim3 grid((n_strips + 3) / 4), block(256);
    double us;
    us = time_us([&] { hipLaunchKernelGGL(k_store<0>, grid, block, 0, 0, dst, T, 1168, n_strips, 1); }, 5);
    printf("aligned rows (1168 B pitch):   %8.1f us  %6.2f TB/s\n", us, bytes / us / 1e6);
    us = time_us([&] { hipLaunchKernelGGL(k_store<1>, grid, block, 0, 0, dst, T, 1154, n_strips, 1); }, 5);
    printf("2-byte aligned rows (1154 B):  %8.1f us  %6.2f TB/s\n", us, bytes / us / 1e6);
    us = time_us([&] { hipLaunchKernelGGL(k_store<1>, grid, block, 0, 0, dst, T, 1168, n_strips, 1); }, 5);
    printf("unaligned TYPE, aligned pitch: %8.1f us  %6.2f TB/s\n", us, bytes / us / 1e6);
    us = time_us([&] { hipLaunchKernelGGL(k_store<2>, grid, block, 0, 0, dst, T, 1154, n_strips, 1); }, 5);
    printf("flat aligned stream:           %8.1f us  %6.2f TB/s\n", us, bytes / us / 1e6);
    return 0;
}
