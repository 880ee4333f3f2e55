// How fast can one CU pull L2-resident bytes, (a) into registers with global_load_dwordx4, (b) into LDS with
// buffer_load_dwordx4 ... lds (the LDS-DMA path the GEMM K loops use)?  One 512-thread workgroup per CU re-reads its own
// slice (larger than the 32 KB L1, so every pass is L2-served); `slice` and the number of workgroups are arguments.
//   hipcc -O3 --offload-arch=gfx950 l2_to_cu_probe.hip -o l2_to_cu_probe && ./l2_to_cu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int UNROLL>
__global__ __launch_bounds__(512) void k_vgpr(const uint4* __restrict__ src, uint4* __restrict__ out, int passes, int slice16) {
    const uint4* p = src + (size_t)blockIdx.x * slice16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int pass = 0; pass < passes; ++pass) {
        for (int i = threadIdx.x; i < slice16; i += 512 * UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + i + u * 512));
                v[u] = make_uint4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if (acc.x == 0x12345678u) out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(512) void k_vgpr_plain(const uint4* __restrict__ src, uint4* __restrict__ out, int passes, int slice16) {
    const uint4* p = src + (size_t)blockIdx.x * slice16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int pass = 0; pass < passes; ++pass) {
        for (int i = threadIdx.x; i < slice16; i += 512 * UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 512];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
        asm volatile("" ::: "memory");
    }
    if (acc.x == 0x12345678u) out[blockIdx.x * 512 + threadIdx.x] = acc;
}

// LDS-DMA: every wave-instruction moves 1 KiB (64 lanes x 16 B) into a 64 KB LDS ring; DEPTH instructions per wave in flight
template <int DEPTH>
__global__ __launch_bounds__(512) void k_dma(const uint4* __restrict__ src, uint4* __restrict__ out, int passes, int slice16) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[8 * DEPTH * 1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(src + (size_t)blockIdx.x * slice16), 0, slice16 * 16, 0x00020000);
    const int n_inst = slice16 / 64;                        // wave-instructions per pass over the slice
    for (int pass = 0; pass < passes; ++pass) {
        for (int i = wave; i < n_inst; i += 8 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(ring + (d * 8 + wave) * 1024), 16,
                                                         (unsigned)((i + d * 8) * 1024 + lane * 16), 0, 0, 0);
            if (DEPTH >= 4) __builtin_amdgcn_s_waitcnt(0x0F70 | (DEPTH / 2)); else __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (ring[threadIdx.x] == 0x5a && ring[threadIdx.x + 1] == 0x77) out[blockIdx.x * 512 + threadIdx.x] = make_uint4(1, 2, 3, 4);
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

int main(int argc, char** argv) {
    const int passes = 64;
    uint4 *src, *out;
    const size_t total = 256ull << 20;
    CHECK(hipMalloc(&src, total)); CHECK(hipMalloc(&out, 1024 * 512 * 16));
    CHECK(hipMemset(src, 1, total));
    printf("workgroups, slice KB, kernel: us per launch, GB/s per CU, TB/s chip\n");
    for (int wgs : {64, 128, 256}) {
        for (int slice_kb : {64, 128, 512}) {
            const int slice16 = slice_kb * 1024 / 16;
            const double bytes = (double)wgs * slice_kb * 1024.0 * passes;
            auto report = [&](const char* name, double us) {
                printf("%4d WGs %4d KB %-22s %9.1f us %7.1f GB/s/CU %6.2f TB/s\n", wgs, slice_kb, name, us, bytes / wgs / us / 1e3, bytes / us / 1e6);
            };
            report("global_load nt x4", time_us([&] { hipLaunchKernelGGL(k_vgpr<4>, dim3(wgs), dim3(512), 0, 0, src, out, passes, slice16); }, 5));
            report("global_load plain x4", time_us([&] { hipLaunchKernelGGL(k_vgpr_plain<4>, dim3(wgs), dim3(512), 0, 0, src, out, passes, slice16); }, 5));
            report("global_load plain x8", time_us([&] { hipLaunchKernelGGL(k_vgpr_plain<8>, dim3(wgs), dim3(512), 0, 0, src, out, passes, slice16); }, 5));
            report("lds-dma depth 2", time_us([&] { hipLaunchKernelGGL(k_dma<2>, dim3(wgs), dim3(512), 0, 0, src, out, passes, slice16); }, 5));
            report("lds-dma depth 4", time_us([&] { hipLaunchKernelGGL(k_dma<4>, dim3(wgs), dim3(512), 0, 0, src, out, passes, slice16); }, 5));
            report("lds-dma depth 8", time_us([&] { hipLaunchKernelGGL(k_dma<8>, dim3(wgs), dim3(512), 0, 0, src, out, passes, slice16); }, 5));
        }
    }
    return 0;
}
