// What does a CU-masked stream (hipExtStreamCreateWithCUMask) give on MI355X?
// (1) census: which (XCC, SE, CU) do the workgroups of a launch land on, for masks of the first n bits and of bits b with (b % 8) < x
//     (the KFD deals mask bits round-robin over the XCCs: bit b -> XCC b % 8 -- checked here, not assumed);
// (2) what a streaming copy (16 B per lane, 1 GB) sustains on n CUs alone, and two such copies side by side on disjoint masks;
// (3) the same copy on n CUs beside an MFMA-only kernel on the other 256 - n CUs (does the matrix side slow the memory side?).
//   hipcc -O3 --offload-arch=gfx950 cu_mask_probe.hip -o cu_mask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_census(uint32_t* out, int spin) {
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID[3:0]
    // stay resident a while so that the launch spreads over every CU the mask allows
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 100000 && (int64_t)(__builtin_readcyclecounter() - t0) < spin; ++it) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}

__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(src[i], dst + i);
}

// MFMA-only load: each wave runs `iters` x 16 independent-accumulator 32x32x16 bf16 MFMAs on random-ish operands
__global__ __launch_bounds__(512) void k_mfma(float* sink, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.37f + 0.01f * ((threadIdx.x * 7 + e * 3) % 61)); b[e] = (__bf16)(-0.41f + 0.013f * ((threadIdx.x * 5 + e) % 53)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) sink[threadIdx.x] = s;
}

static hipStream_t masked(const std::vector<int>& bits) {
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b : bits) m[b >> 5] |= 1u << (b & 31);
    hipStream_t s;
    CHECK(hipExtStreamCreateWithCUMask(&s, 8, m));
    return s;
}
static std::vector<int> first_n(int n) { std::vector<int> v; for (int b = 0; b < n; ++b) v.push_back(b); return v; }
static std::vector<int> range(int lo, int hi) { std::vector<int> v; for (int b = lo; b < hi; ++b) v.push_back(b); return v; }
static std::vector<int> xcc_lt(int x0, int x1) { std::vector<int> v; for (int b = 0; b < 256; ++b) if ((b % 8) >= x0 && (b % 8) < x1) v.push_back(b); return v; }

static void census(const char* label, hipStream_t s, uint32_t* dev) {
    const int nwg = 2048;
    CHECK(hipMemsetAsync(dev, 0xff, nwg * 8, s));
    hipLaunchKernelGGL(k_census, dim3(nwg), dim3(256), 0, s, dev, 200000);
    CHECK(hipStreamSynchronize(s));
    std::vector<uint32_t> h(nwg * 2);
    CHECK(hipMemcpy(h.data(), dev, nwg * 8, hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    std::set<uint32_t> cus;
    std::set<uint32_t> per_xcc_cus[16];
    for (int i = 0; i < nwg; ++i) {
        const uint32_t hw = h[i * 2], xcc = h[i * 2 + 1] & 15;
        const uint32_t cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        cus.insert(key); per_xcc_cus[xcc].insert(key); per_xcc[xcc]++;
    }
    printf("%-34s distinct CUs %3zu | CUs per XCC:", label, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %2zu", per_xcc_cus[x].size());
    printf(" | workgroups per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("\n");
}

static double time_copy(hipStream_t s, int grid, const u32x4* src, u32x4* dst, size_t n16, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s, src, dst, n16);
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s, src, dst, n16);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char* what = argc > 1 ? argv[1] : "all";
    auto want = [&](const char* sec) { return !strcmp(what, "all") || !strcmp(what, sec); };
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    uint32_t* dev;
    CHECK(hipMalloc(&dev, 2048 * 8));
    hipStream_t plain;
    CHECK(hipStreamCreate(&plain));
    if (want("census")) {
    census("no mask", plain, dev);
    struct { const char* label; std::vector<int> bits; } cases[] = {
        {"bits 0..127", first_n(128)}, {"bits 128..255", range(128, 256)}, {"bits 0..63", first_n(64)}, {"bits 0..191", first_n(192)},
        {"bits b%8 < 4", xcc_lt(0, 4)}, {"bits b%8 >= 4", xcc_lt(4, 8)}, {"bits b%8 < 6", xcc_lt(0, 6)}, {"bits b%8 < 2", xcc_lt(0, 2)},
        {"bits 0..31", first_n(32)}, {"bits 0..7", first_n(8)},
    };
    for (auto& c : cases) { hipStream_t s = masked(c.bits); census(c.label, s, dev); CHECK(hipStreamDestroy(s)); }
    }

    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    u32x4 *srcA, *dstA, *srcB, *dstB;
    CHECK(hipMalloc(&srcA, bytes)); CHECK(hipMalloc(&dstA, bytes)); CHECK(hipMalloc(&srcB, bytes)); CHECK(hipMalloc(&dstB, bytes));
    CHECK(hipMemset(srcA, 1, bytes)); CHECK(hipMemset(srcB, 2, bytes));
    float* sink; CHECK(hipMalloc(&sink, 4096));
    if (want("copy")) {
    printf("\nstreaming copy, 1 GiB read + 1 GiB written, grid = 8 workgroups per CU of the mask\n");
    for (int n : {32, 64, 96, 128, 192, 256}) {
        for (int layout = 0; layout < 2; ++layout) {
            if (layout == 1 && n % 32) continue;
            hipStream_t s = masked(layout == 0 ? first_n(n) : xcc_lt(0, n / 32));
            const double ms = time_copy(s, n * 8, srcA, dstA, n16, 10);
            printf("  %3d CUs (%s): %7.3f ms  %7.1f GB/s (read + write)\n", n, layout == 0 ? "first n bits = a slice of every XCC" : "whole XCCs               ", ms, 2.0 * bytes / ms / 1e6);
            CHECK(hipStreamDestroy(s));
        }
    }
    }
    if (want("pair")) {
    printf("\ntwo copies side by side on disjoint masks (each 1 GiB + 1 GiB)\n");
    for (int layout = 0; layout < 2; ++layout) {
        for (int n : {64, 128, 192}) {
            if (layout == 1 && n % 32) continue;
            hipStream_t a = masked(layout == 0 ? first_n(n) : xcc_lt(0, n / 32));
            hipStream_t b = masked(layout == 0 ? range(n, 256) : xcc_lt(n / 32, 8));
            hipEvent_t e0, e1, eb;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&eb));
            hipLaunchKernelGGL(k_copy, dim3(n * 8), dim3(256), 0, a, srcA, dstA, n16);
            hipLaunchKernelGGL(k_copy, dim3((256 - n) * 8), dim3(256), 0, b, srcB, dstB, n16);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, a));
            CHECK(hipStreamWaitEvent(b, e0, 0));
            for (int r = 0; r < 10; ++r) {
                hipLaunchKernelGGL(k_copy, dim3(n * 8), dim3(256), 0, a, srcA, dstA, n16);
                hipLaunchKernelGGL(k_copy, dim3((256 - n) * 8), dim3(256), 0, b, srcB, dstB, n16);
            }
            CHECK(hipEventRecord(eb, b));
            CHECK(hipStreamWaitEvent(a, eb, 0));
            CHECK(hipEventRecord(e1, a));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %3d | %3d CUs (%s): %7.3f ms per pair  %7.1f GB/s total\n", n, 256 - n, layout == 0 ? "slices" : "whole XCCs", ms / 10, 4.0 * bytes / (ms / 10) / 1e6);
            CHECK(hipStreamDestroy(a)); CHECK(hipStreamDestroy(b));
        }
    }
    }
    if (want("mfma")) {
    printf("\ncopy on n CUs beside an MFMA-only kernel on the other 256 - n (one 512-thread workgroup per CU)\n");
    for (int n : {64, 128}) {
        hipStream_t a = masked(first_n(n)), b = masked(range(n, 256));
        const double alone = time_copy(a, n * 8, srcA, dstA, n16, 5);
        // MFMA alone
        hipEvent_t m0, m1; CHECK(hipEventCreate(&m0)); CHECK(hipEventCreate(&m1));
        const int iters = 40000;
        hipLaunchKernelGGL(k_mfma, dim3(256 - n), dim3(512), 0, b, sink, 1000);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(m0, b));
        hipLaunchKernelGGL(k_mfma, dim3(256 - n), dim3(512), 0, b, sink, iters);
        CHECK(hipEventRecord(m1, b));
        CHECK(hipDeviceSynchronize());
        float mfma_alone; CHECK(hipEventElapsedTime(&mfma_alone, m0, m1));
        // both
        CHECK(hipEventRecord(m0, b));
        hipLaunchKernelGGL(k_mfma, dim3(256 - n), dim3(512), 0, b, sink, iters);
        CHECK(hipEventRecord(m1, b));
        const double beside = time_copy(a, n * 8, srcA, dstA, n16, 5);
        CHECK(hipDeviceSynchronize());
        float mfma_beside; CHECK(hipEventElapsedTime(&mfma_beside, m0, m1));
        const double fl = 2.0 * 32 * 32 * 16 * 16.0 * iters * 8 * (256 - n);
        printf("  copy on %3d CUs: alone %7.3f ms, beside MFMA %7.3f ms | MFMA on %3d CUs: alone %7.2f ms (%6.0f TF), beside the copy %7.2f ms (%6.0f TF)\n",
               n, alone, beside, 256 - n, mfma_alone, fl / mfma_alone / 1e9, mfma_beside, fl / mfma_beside / 1e9);
        CHECK(hipStreamDestroy(a)); CHECK(hipStreamDestroy(b));
    }
    }
    return 0;
}
