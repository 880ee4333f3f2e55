// CLIP validation transform on the GPU (/root/reference/src/vit_prisma/transforms/model_transforms.py:9-20):
//     Resize(S, BICUBIC, antialias) -> CenterCrop(S) -> ToTensor -> Normalize(mean, std)
// for batches of decoded uint8 RGB images [B][H][W][3], written straight in the model's layout and dtype [B][3][S][S].
// The reference resizes PIL images, i.e. with Pillow's two-pass 8-bit resampler (Resample.c): fixed-point taps
// (PRECISION_BITS = 22), horizontal pass first, EACH pass rounded and clipped to uint8.  This kernel does exactly that
// arithmetic (the tap tables come from the host: vit_prisma_amd/transforms.py builds them the way precompute_coeffs /
// normalize_coeffs_8bpc do, in double precision), so the result is bit-identical to the reference's CPU pipeline -- only the
// pixels of the centre crop are computed.
//   workgroup = 16 x 16 output pixels: the horizontally resampled rows the tile needs go through LDS once (each is shared by
//   up to ksize_y output rows), then every thread runs its vertical taps out of LDS and normalises.
#include <hip/hip_runtime.h>

#include "pv_common.hpp"

namespace {

constexpr int PP_BITS = 22;
constexpr int PP_TILE = 16;
constexpr int PP_MAXR = 192;            // intermediate rows a tile may hold in LDS (more: taps straight from global)

struct PreParams {
    const uint8_t* img;                 // [B][H][W][3]
    const int32_t *xb, *xk, *yb, *yk;   // bounds [n][2] = (first input index, taps), taps [n][ksize]
    int B, H, W, kx, ky, left, top, S;
    float mean[3], std[3];
    void* out;                          // [B][3][S][S] f32 / bf16
    int out_bf16;
};

__device__ __forceinline__ int clip8(int v) { return min(max(v >> PP_BITS, 0), 255); }

// one horizontally resampled pixel (3 channels) of input row `row`, output column x
__device__ __forceinline__ void hpix(const PreParams& p, const uint8_t* img_b, int row, int x, int (&t)[3]) {
    const int x0 = p.xb[2 * x], nx = p.xb[2 * x + 1];
    const int32_t* k = p.xk + (int64_t)x * p.kx;
    const uint8_t* src = img_b + ((int64_t)row * p.W + x0) * 3;
    int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
    for (int i = 0; i < nx; ++i) {
        const int kk = k[i];
        a0 += (int)src[3 * i] * kk;
        a1 += (int)src[3 * i + 1] * kk;
        a2 += (int)src[3 * i + 2] * kk;
    }
    t[0] = clip8(a0); t[1] = clip8(a1); t[2] = clip8(a2);
}

__global__ __launch_bounds__(256) void clip_preprocess_kernel(const PreParams p) {
    __shared__ uint8_t mid[PP_MAXR][PP_TILE][4];
    const int b = blockIdx.z;
    const int ox0 = blockIdx.x * PP_TILE, oy0 = blockIdx.y * PP_TILE;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const uint8_t* img_b = p.img + (int64_t)b * p.H * p.W * 3;
    // rows of the horizontally resampled image this tile's vertical taps touch (bounds are monotone in y)
    const int y_first = p.top + oy0, y_last = p.top + min(oy0 + PP_TILE, p.S) - 1;
    const int r_lo = p.yb[2 * y_first], r_hi = p.yb[2 * y_last] + p.yb[2 * y_last + 1];
    const int nrows = r_hi - r_lo;
    const bool tiled = nrows <= PP_MAXR;
    if (tiled) {
        for (int it = threadIdx.x; it < nrows * PP_TILE; it += 256) {
            const int r = it >> 4, cx = it & 15;
            const int ox = ox0 + cx;
            int t[3] = {0, 0, 0};
            if (ox < p.S) hpix(p, img_b, r_lo + r, p.left + ox, t);
            mid[r][cx][0] = (uint8_t)t[0]; mid[r][cx][1] = (uint8_t)t[1]; mid[r][cx][2] = (uint8_t)t[2];
        }
        __syncthreads();
    }
    const int ox = ox0 + tx, oy = oy0 + ty;
    if (ox >= p.S || oy >= p.S) return;
    const int y = p.top + oy;
    const int y0 = p.yb[2 * y], ny = p.yb[2 * y + 1];
    const int32_t* k = p.yk + (int64_t)y * p.ky;
    int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
    for (int j = 0; j < ny; ++j) {
        int t[3];
        if (tiled) {
            const uint8_t* m = mid[y0 - r_lo + j][tx];
            t[0] = m[0]; t[1] = m[1]; t[2] = m[2];
        } else {
            hpix(p, img_b, y0 + j, p.left + ox, t);
        }
        const int kk = k[j];
        a0 += t[0] * kk; a1 += t[1] * kk; a2 += t[2] * kk;
    }
    const int v[3] = {clip8(a0), clip8(a1), clip8(a2)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // ToTensor: uint8 / 255 (a true fp32 division, like torch's div); Normalize: (x - mean) / std
        const float f = ((float)v[c] / 255.0f - p.mean[c]) / p.std[c];
        const int64_t o = (((int64_t)b * 3 + c) * p.S + oy) * p.S + ox;
        if (p.out_bf16) reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(f);
        else reinterpret_cast<float*>(p.out)[o] = f;
    }
}

}  // namespace

extern "C" int pv_clip_preprocess(const uint8_t* images, int32_t B, int32_t H, int32_t W, const int32_t* xbounds, const int32_t* xtaps,
                                  int32_t ksize_x, const int32_t* ybounds, const int32_t* ytaps, int32_t ksize_y, int32_t new_w,
                                  int32_t new_h, int32_t left, int32_t top, int32_t S, const float* mean3, const float* std3,
                                  int32_t out_dtype, void* out, void* stream) {
    PV_REQUIRE(images && xbounds && xtaps && ybounds && ytaps && mean3 && std3 && out, "null argument");
    PV_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && S > 0 && ksize_x > 0 && ksize_y > 0, "dims");
    PV_REQUIRE(left >= 0 && top >= 0 && left + S <= new_w && top + S <= new_h, "the crop must lie inside the resized image");
    PV_REQUIRE(out_dtype == PV_DTYPE_F32 || out_dtype == PV_DTYPE_BF16, "output dtype must be fp32 or bf16");
    PV_REQUIRE((int64_t)H * W * 3 < (1ll << 31), "image too large");
    PreParams p;
    p.img = images; p.xb = xbounds; p.xk = xtaps; p.yb = ybounds; p.yk = ytaps;
    p.B = B; p.H = H; p.W = W; p.kx = ksize_x; p.ky = ksize_y; p.left = left; p.top = top; p.S = S;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.std[c] = std3[c]; }
    p.out = out; p.out_bf16 = out_dtype == PV_DTYPE_BF16;
    const dim3 grid((S + PP_TILE - 1) / PP_TILE, (S + PP_TILE - 1) / PP_TILE, B);
    hipLaunchKernelGGL(clip_preprocess_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    PV_LAUNCH_CHECK("clip_preprocess_kernel");
    return PV_OK;
}
