// Row-wise HBM-bound kernels: LayerNorm (+ embed assembly) with fused tap stores, L2 normalise,
// batched transpose (weight shadow packing).  One 64-lane wave owns one row; 16-byte vector
// loads/stores; every tensor is read once and every requested tap written once.
#include "rowops.hpp"
#include "prof.hpp"

namespace {

constexpr int LN_MAX_CHUNKS = 4;   // 8-element chunks per lane -> d <= 64 * 8 * 4 = 2048

// LayerNorm exactly as models/layers/layer_norm.py:75-93:
//   x <- x - mean(x);  scale = sqrt(mean(x^2) + eps);  y = x / scale * w + b
// In bf16 mode the input is up-cast to fp32 first (:84-85), hook_scale / hook_normalized fire on
// fp32 values (:88-93) and only the returned tensor is cast back to bf16.
// EMBED mode builds the row on the fly (models/base_vit.py:171-181):
//   row(b, t) = (t == 0 ? cls_token : patch_embed[b, t-1]) + W_pos[t]   -> hook_full_embed
template <typename T, bool EMBED>
__global__ __launch_bounds__(256) void ln_kernel(const LnParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int d = p.d;
    const int nchunks = d >> 3;

    float x[LN_MAX_CHUNKS][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunks) {
            if constexpr (EMBED) {
                const int b = row / p.T, t = row - b * p.T;
                float e[8], pos[8];
                if (p.use_cls && t == 0) {
                    load8(reinterpret_cast<const T*>(p.cls) + ch * 8, e);
                } else {
                    const int64_t prow = (int64_t)b * (p.T - p.use_cls) + (t - p.use_cls);
                    load8(reinterpret_cast<const T*>(p.x) + prow * d + ch * 8, e);
                }
                load8(reinterpret_cast<const T*>(p.pos) + (int64_t)t * d + ch * 8, pos);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[c][i] = DT<T>::round(e[i] + pos[i]);
                if (p.full_out) store8(reinterpret_cast<T*>(p.full_out) + (int64_t)row * d + ch * 8, x[c]);
            } else {
                load8(reinterpret_cast<const T*>(p.x) + (int64_t)row * p.ldx + ch * 8, x[c]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += x[c][i];
        }
    }
    if (!p.do_ln) return;
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        if (lane + 64 * c < nchunks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[c][i] -= mean;
                sq += x[c][i] * x[c][i];
            }
        }
    }
    const float scale = sqrtf(wave_sum(sq) / (float)d + p.eps);
    if (p.scale_out && lane == 0) p.scale_out[row] = scale;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunks) {
            float w[8], b[8], y[8];
            load8(reinterpret_cast<const T*>(p.w) + ch * 8, w);
            load8(reinterpret_cast<const T*>(p.b) + ch * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = (x[c][i] / scale) * w[i] + b[i];
            if (p.norm_f32_out) store8_stream(p.norm_f32_out + (int64_t)row * d + ch * 8, y);   // (the fp32 tap of bf16 mode)
            if (p.out) store8(reinterpret_cast<T*>(p.out) + (int64_t)row * d + ch * 8, y);
        }
    }
}

// F.normalize(x, dim=-1) (models/base_vit.py:214-215): x / max(||x||_2, 1e-12); one wave per row
template <typename T>
__global__ __launch_bounds__(256) void l2norm_kernel(const T* __restrict__ x, T* __restrict__ out, int rows, int n) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float sq = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float v = DT<T>::load(x + (int64_t)row * n + i);
        sq += v * v;
    }
    const float nrm = fmaxf(sqrtf(wave_sum(sq)), 1e-12f);
    for (int i = lane; i < n; i += 64) {
        const float v = DT<T>::load(x + (int64_t)row * n + i);
        DT<T>::store(out + (int64_t)row * n + i, v / nrm);
    }
}

// out[b][c][r] = in[b][r][c]; 32 x 32 tiles through LDS (+1 pad), coalesced on both sides
template <typename E>
__global__ __launch_bounds__(256) void transpose_kernel(const E* __restrict__ in, E* __restrict__ out, int R, int C) {
    __shared__ E tile[32][33];
    const int64_t boff = (int64_t)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < R && c < C) tile[ty + 8 * i][tx] = in[boff + (int64_t)r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < C) out[boff + (int64_t)c * R + r] = tile[tx][ty + 8 * i];
    }
}

}  // namespace

int pv_launch_ln(int dtype, const LnParams& p, hipStream_t stream) {
    PV_REQUIRE(p.d % 8 == 0 && p.d <= 64 * 8 * LN_MAX_CHUNKS, "layernorm width must be a multiple of 8 and <= 2048");
    PV_REQUIRE(p.rows > 0, "layernorm rows");
    if (!p.embed) PV_REQUIRE(p.ldx % 8 == 0, "layernorm row stride must be a multiple of 8");
    const dim3 grid((p.rows + 3) / 4), block(256);
    const double eb = dtype == PV_DTYPE_BF16 ? 2.0 : 4.0;
    const double rd = (double)p.rows * p.d;
    ProfScope prof(PV_PROF_LN, stream, 8.0 * rd,
                   rd * eb * (1.0 + (p.out ? 1.0 : 0.0) + (p.full_out ? 1.0 : 0.0)) + (p.norm_f32_out ? 4.0 * rd : 0.0) +
                       (p.scale_out ? 4.0 * p.rows : 0.0));
    if (dtype == PV_DTYPE_BF16) {
        if (p.embed) hipLaunchKernelGGL((ln_kernel<bf16_t, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((ln_kernel<bf16_t, false>), grid, block, 0, stream, p);
    } else {
        if (p.embed) hipLaunchKernelGGL((ln_kernel<float, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((ln_kernel<float, false>), grid, block, 0, stream, p);
    }
    PV_LAUNCH_CHECK("ln_kernel");
    return PV_OK;
}

int pv_launch_l2norm(int dtype, const void* x, void* out, int rows, int n, hipStream_t stream) {
    const dim3 grid((rows + 3) / 4), block(256);
    if (dtype == PV_DTYPE_BF16)
        hipLaunchKernelGGL((l2norm_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)x, (bf16_t*)out, rows, n);
    else
        hipLaunchKernelGGL((l2norm_kernel<float>), grid, block, 0, stream, (const float*)x, (float*)out, rows, n);
    PV_LAUNCH_CHECK("l2norm_kernel");
    return PV_OK;
}

namespace {
// bf16 patches as GEMM rows (patch size other than 32, even): image row (b, c, y) is S contiguous pixels = G runs of p; run
// px goes to row b*P + (y/p)*G + px of the [B*P][Kp] matrix at column c*p*p + (y%p)*p.  One thread per pixel PAIR (a dword:
// p and every column offset are even), so the reads of a wave are one contiguous stretch of the image row.
// Columns [K, Kp) (Kp = K rounded up to 8: whole 16-byte chunks for the DMA of the tiled GEMM) are zeroed.
__global__ __launch_bounds__(256) void patch_pack_kernel(const uint32_t* __restrict__ img, uint32_t* __restrict__ out, int B, int C,
                                                         int S, int p, int G, int Kp) {
    const int half_row = S / 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n_rows = (int64_t)B * C * S;
    if (t >= n_rows * half_row) return;
    const int w = (int)(t % half_row);
    const int64_t r = t / half_row;                    // (b, c, y)
    const int y = (int)(r % S);
    const int c = (int)((r / S) % C);
    const int b = (int)(r / ((int64_t)S * C));
    const int px = 2 * w / p, j = 2 * w - px * p, py = y / p, i = y - py * p;
    if (px >= G || py >= G) return;                    // pixels beyond the last whole patch
    const int64_t m = ((int64_t)b * G + py) * G + px;
    out[(m * Kp + (c * p + i) * p + j) >> 1] = img[t];
}
__global__ __launch_bounds__(256) void patch_pad_kernel(uint32_t* __restrict__ out, int64_t rows, int K, int Kp) {
    const int padw = (Kp - K) / 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * padw) return;
    out[((t / padw) * Kp + K) / 2 + t % padw] = 0u;
}
// rows of [R][K] -> [R][Kp], zero-padded (the patch-embedding weights beside patch_pack_kernel's rows)
__global__ __launch_bounds__(256) void pad_rows_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int K, int Kp) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)R * Kp) return;
    const int col = (int)(t % Kp);
    out[t] = col < K ? in[(t / Kp) * K + col] : (uint16_t)0;
}
}  // namespace

int pv_launch_patch_pack_bf16(const void* images, void* out, int B, int C, int S, int p, int G, int Kp, hipStream_t stream) {
    PV_REQUIRE(p % 2 == 0 && S % 2 == 0 && Kp % 2 == 0, "patch_pack needs an even patch and image size");
    PV_REQUIRE(((uintptr_t)images & 3) == 0 && ((uintptr_t)out & 3) == 0, "patch_pack alignment");
    const int K = C * p * p;
    const int64_t n = (int64_t)B * C * S * (S / 2);
    hipLaunchKernelGGL(patch_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)images,
                       (uint32_t*)out, B, C, S, p, G, Kp);
    if (Kp > K) {
        const int64_t rows = (int64_t)B * G * G, m = rows * ((Kp - K) / 2);
        hipLaunchKernelGGL(patch_pad_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, (uint32_t*)out, rows, K, Kp);
    }
    PV_LAUNCH_CHECK("patch_pack_kernel");
    return PV_OK;
}

int pv_launch_pad_rows_bf16(const void* in, void* out, int R, int K, int Kp, hipStream_t stream) {
    const int64_t n = (int64_t)R * Kp;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint16_t*)in, (uint16_t*)out, R, K, Kp);
    PV_LAUNCH_CHECK("pad_rows_kernel");
    return PV_OK;
}

int pv_launch_transpose(int elem_bytes, const void* in, void* out, int batch, int R, int C, hipStream_t stream) {
    PV_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "transpose element size must be 2 or 4");
    PV_REQUIRE(batch > 0 && R > 0 && C > 0 && batch < 65536, "transpose dims");
    const dim3 grid((C + 31) / 32, (R + 31) / 32, batch), block(256);
    if (elem_bytes == 2)
        hipLaunchKernelGGL((transpose_kernel<uint16_t>), grid, block, 0, stream, (const uint16_t*)in, (uint16_t*)out, R, C);
    else
        hipLaunchKernelGGL((transpose_kernel<uint32_t>), grid, block, 0, stream, (const uint32_t*)in, (uint32_t*)out, R, C);
    PV_LAUNCH_CHECK("transpose_kernel");
    return PV_OK;
}

namespace {
template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ in, T* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = *reinterpret_cast<const float4*>(in + i * 4);
    DT<T>::store(out + i * 4, v.x); DT<T>::store(out + i * 4 + 1, v.y);
    DT<T>::store(out + i * 4 + 2, v.z); DT<T>::store(out + i * 4 + 3, v.w);
}
}  // namespace

int pv_launch_cast_from_f32(int dtype, const float* in, void* out, int64_t n, hipStream_t stream) {
    PV_REQUIRE(in && out && n > 0 && n % 4 == 0 && pv_aligned16(in), "cast launcher arguments");
    const int64_t n4 = n / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
    if (dtype == PV_DTYPE_BF16) hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, grid, block, 0, stream, in, reinterpret_cast<bf16_t*>(out), n4);
    else hipLaunchKernelGGL(cast_from_f32_kernel<float>, grid, block, 0, stream, in, reinterpret_cast<float*>(out), n4);
    PV_LAUNCH_CHECK("cast_from_f32_kernel");
    return PV_OK;
}
