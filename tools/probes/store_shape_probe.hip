// (1) What does the SHAPE of a wave's 16-byte stores cost?  A [M][N] bf16 matrix (N = 3072: the MLP-1 outputs) written by one wave per
//     32 x 64 block, as
//       lines   8 rows x 128 contiguous bytes per instruction (what the LDS-staged epilogue emits)
//       pieces  32 rows x 32 bytes per instruction: lanes l and l + 32 write adjacent 16-byte chunks of row l (what a
//               row-per-lane epilogue behind v_permlane32_swap emits; four instructions fill a row's 128-byte line)
//     each with plain and nontemporal stores.
// (2) Is "acc + bias" by ONE extra MFMA (bias as the k = 0 element of one operand, 1.0 as the k = 0 element of the other, zeros elsewhere)
//     bit-identical to v_add_f32(acc, bias)?  Checked over random accumulators / bf16 biases for both operand orders.
//   hipcc -O3 --offload-arch=gfx950 store_shape_probe.hip -o store_shape_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k_store(unsigned char* __restrict__ dst, int M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nbn = N / 64;
    const int blk = blockIdx.x * 4 + wave;
    const int bm = blk / nbn, bn = blk - bm * nbn;
    if (bm * 32 >= M) return;
    unsigned char* base = dst + ((size_t)bm * 32 * N + (size_t)bn * 64) * 2;
    const u32x4 v = {(unsigned)lane, (unsigned)blk, 0x3c003c00u, 0x3c003c00u};
    if (MODE == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            u32x4* p = reinterpret_cast<u32x4*>(base + (size_t)(it * 8 + (lane >> 3)) * N * 2 + (lane & 7) * 16);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4* p = reinterpret_cast<u32x4*>(base + (size_t)(lane & 31) * N * 2 + (c * 2 + (lane >> 5)) * 16);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

__global__ void k_bias(const float* accs, const uint16_t* bias, float* out_add, float* out_mfma_a, float* out_mfma_b) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = accs[(blockIdx.x * 16 + e) * 64 + lane];
    // plain layout: lane = column (l31), registers = rows: bias per lane = bias[l31]
    const float b = __uint_as_float(((uint32_t)bias[blockIdx.x * 32 + l31]) << 16);
    for (int e = 0; e < 16; ++e) out_add[(blockIdx.x * 16 + e) * 64 + lane] = acc[e] + b;
    // (a) bias as the SECOND operand's k = 0 element (column operand), ones as the first's
    u32x4 ones = {half == 0 ? 0x3F80u : 0u, 0, 0, 0}, bf = {half == 0 ? (uint32_t)bias[blockIdx.x * 32 + l31] : 0u, 0, 0, 0};
    f32x16 ra = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) out_mfma_a[(blockIdx.x * 16 + e) * 64 + lane] = ra[e];
    // (b) swapped roles: bias on the FIRST operand -> it is added along the register (row) index: element e of lane gets bias[row(e, half)]
    f32x16 rb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, ones), acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) out_mfma_b[(blockIdx.x * 16 + e) * 64 + lane] = rb[e];
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
    const int M = 25600, N = 3072;
    unsigned char* dst;
    CHECK(hipMalloc(&dst, (size_t)M * N * 2 * 2));
    const double bytes = (double)M * N * 2;
    const dim3 grid((M / 32) * (N / 64) / 4), block(256);
    for (int rep = 0; rep < 2; ++rep) {
        printf("lines  plain: %7.1f us %5.2f TB/s\n", time_us([&] { hipLaunchKernelGGL((k_store<0, false>), grid, block, 0, 0, dst, M, N); }, 20), bytes / time_us([&] { hipLaunchKernelGGL((k_store<0, false>), grid, block, 0, 0, dst, M, N); }, 20) / 1e6);
        printf("lines  nt   : %7.1f us\n", time_us([&] { hipLaunchKernelGGL((k_store<0, true>), grid, block, 0, 0, dst, M, N); }, 20));
        printf("pieces plain: %7.1f us\n", time_us([&] { hipLaunchKernelGGL((k_store<1, false>), grid, block, 0, 0, dst, M, N); }, 20));
        printf("pieces nt   : %7.1f us\n", time_us([&] { hipLaunchKernelGGL((k_store<1, true>), grid, block, 0, 0, dst, M, N); }, 20));
    }
    // ---- bias by MFMA
    const int NB = 4096;
    std::vector<float> h_acc((size_t)NB * 16 * 64);
    std::vector<uint16_t> h_bias((size_t)NB * 32);
    srand(1);
    for (auto& v : h_acc) { v = ((rand() % 20001) - 10000) * 1e-3f * ((rand() & 7) == 0 ? 1e-3f : 1.0f); }
    for (auto& v : h_bias) { float f = ((rand() % 4001) - 2000) * 1e-3f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    float *d_acc, *d_add, *d_a, *d_b; uint16_t* d_bias;
    const size_t nb = h_acc.size() * 4;
    CHECK(hipMalloc(&d_acc, nb)); CHECK(hipMalloc(&d_add, nb)); CHECK(hipMalloc(&d_a, nb)); CHECK(hipMalloc(&d_b, nb)); CHECK(hipMalloc(&d_bias, h_bias.size() * 2));
    CHECK(hipMemcpy(d_acc, h_acc.data(), nb, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_bias, h_bias.data(), h_bias.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bias, dim3(NB), dim3(64), 0, 0, d_acc, d_bias, d_add, d_a, d_b);
    CHECK(hipDeviceSynchronize());
    std::vector<float> r_add(h_acc.size()), r_a(h_acc.size()), r_b(h_acc.size());
    CHECK(hipMemcpy(r_add.data(), d_add, nb, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r_a.data(), d_a, nb, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r_b.data(), d_b, nb, hipMemcpyDeviceToHost));
    size_t bad_a = 0, bad_b = 0;
    for (size_t i = 0; i < r_add.size(); ++i) if (memcmp(&r_add[i], &r_a[i], 4)) ++bad_a;
    // (b): element (blk, e, lane) must equal acc + bias[row], row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    for (int blk = 0; blk < NB; ++blk)
        for (int e = 0; e < 16; ++e)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                uint32_t u = ((uint32_t)h_bias[blk * 32 + row]) << 16; float b; memcpy(&b, &u, 4);
                const float want = h_acc[((size_t)blk * 16 + e) * 64 + lane] + b;
                if (memcmp(&want, &r_b[((size_t)blk * 16 + e) * 64 + lane], 4)) ++bad_b;
            }
    printf("bias by MFMA: column-operand form %zu / %zu differ from v_add_f32; row-operand form %zu differ\n", bad_a, r_add.size(), bad_b);
    return 0;
}
