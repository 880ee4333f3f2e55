// Dense SAE train step for the activations that are NOT k-sparse: ReLU + L1 (gfx950; C ABI: pv_sae_dense_step).
//
// Reference semantics: StandardSparseAutoencoder.forward with activation_fn_str = "relu" (/root/reference/src/vit_prisma/
// sae/sae.py:557-645: encode :557-581, decode :583-595, mse :144-149, L1 sparsity :617-626 with lp_norm = 1) and the body of
// VisionSAETrainer.train_step between zero_grad and clip (sae/train_sae.py:328-392).  Every published CLIP SAE of the
// reference is of this kind (docs/sae_table.md:9-36).  A token keeps hundreds to thousands of its d_sae features, so the
// k-sparse machinery of sae.hip does not apply: the step is five dense GEMMs of 2 N d_in d_sae FLOP each.  They run on the
// EXACT fp32 matrix instruction (v_mfma_f32_32x32x2_f32 == an fmaf chain per output: no reduced-precision operand, results
// equal to the reference's fp32 matmuls up to summation order, measured <= 2e-6 relative), and everything elementwise rides
// in their epilogues -- no [N, d_sae] tensor is read or written except f itself (once) and dH over it (in place):
//
//   prep      (sae.hip)  LN-in, sae_in = x_hat - b_dec, loss normaliser
//   G1  f  = relu(sae_in W_enc + b_enc)  (W_enc^T read)  epilogue: bias, ReLU, store f; per-64-row column counts of f > 0
//                                                        (firing statistics, l0) and per-wave sums of f (the L1 term)
//   G2  pre = f W_dec  (split-K: [tokens x d_in] is only 192 tiles)   -> partial sums
//   finish    sae_out = (pre + b_dec) std + mu, err, mse partials, dY                  (one wave per token)
//   G4  gW_dec   = f^T dY                               straight into the flat gradient buffer
//   G3  dH = (dY W_dec^T + l1 / N) [f > 0]              epilogue: gate by the stored f, written OVER f; per-64-row column sums
//                                                        (gb_enc)
//   G5  gW_enc^T = dH^T sae_in                          straight into the flat gradient buffer (transposed encoder layout)
//   gb_dec = colsum(dY) - W_enc gb_enc   (sae.hip)
// then pv_sae_grad_sqnorm / pv_sae_apply (clip -> project -> Adam) as for the top-k step.
//
// Operand layouts: the fp32 matrix instruction takes ONE float per lane and operand, so a matrix can be fed from either of
// its two row-major layouts without a transposed copy -- K-contiguous rows ([rows][K]: one 16-byte LDS read = four k-steps) or
// row-contiguous k-slices ([K][rows]: four 4-byte LDS reads, lanes on consecutive rows).  f^T, dH^T, dY and sae_in of G4 / G5
// are therefore the buffers as they lie, read "the other way".
#include <algorithm>
#include "sae.hpp"
#include <cstring>

namespace {

constexpr int DG_BM = 128, DG_BN = 128;
constexpr int DG_KSLAB = 32;                  // floats of K per stage
constexpr int DG_ROWB = 144;                  // K-contiguous tile: 128 rows x (128 + 16 pad) bytes
constexpr int DG_KROW = 528;                  // row-contiguous tile: 32 k-rows x (512 + 16 pad) bytes
constexpr int DG_TILE = DG_BM * DG_ROWB;      // 18432 >= 32 * 528
constexpr int DG_CS_LD = 68;                  // fp32 staging row of the epilogue (floats)
constexpr int DG_LDS = 4 * DG_TILE;           // As[2] + Bs[2] = 73728 >= 4 waves x 64 x 68 x 4

enum { DG_EPI_STORE = 0, DG_EPI_ENC = 1, DG_EPI_DH = 2, DG_EPI_MUL = 3, DG_EPI_GENC = 4, DG_EPI_EXPB = 5,      // EXPB: out = exp(acc + bias)
       DG_EPI_ENC_T = 6, DG_EPI_DH_T = 7 };   // ENC / DH with the tail of SURVEY.md 8(f) row 3 compiled in: tanh-relu, lp_norm > 1 (DenseGemm.act / lp)

struct DenseGemm {
    const float* A; int64_t lda;              // A_KM ? [K][lda] (M contiguous) : [M][lda] (K contiguous)
    const float* B; int64_t ldb;              // B_KN ? [K][ldb] (N contiguous) : [N][ldb] (K contiguous)
    int M, N, K;
    int k_chunk;                              // K range of one blockIdx.y (multiple of 32; K when not split)
    int group;                                // set by the launcher: row tiles per group of the tile order (see the kernel), 0 = N fastest
    float* out; int64_t ldo;                  // STORE: out + blockIdx.y * out_zstride
    int64_t out_zstride;
    const float* bias;                        // ENC: b_enc [N]
    float* colpart;                           // ENC / DH: [ceil(M / 64)][N] column partials of the wave's 64-row block
    float* rowpart;                           // ENC: [ceil(M / 64)][ceil(N / 64)] sums of f over the wave's 64 x 64 block
    float add;                                // DH: l1_coefficient / N_global
    // ghost gradients (sae.py:151-179): the dead features' columns, compacted
    const int32_t* dead_slot;                 // ENC / DH: [N] column -> compact slot of a dead feature, -1 otherwise; or NULL
    float* dead_act; int64_t ldd;             // ENC: exp(hidden_pre) of the dead columns -> dead_act[row][slot];  DH: the ghost term of
                                              //      d loss / d hidden_pre, added to dH there
    const float* mul;                         // MUL: out = acc * mul (same shape and leading dimension as out)
    // gated SAE (sae.py:699-716): GENC takes acc = sae_in @ W_enc and writes feature_acts = [acc + bias > 0] relu(acc e^cscale + bias2)
    // to out, relu(acc + bias) to out2 (same leading dimension); colpart = firing counts, colpart2 / rowpart = sums of out2.
    // DH with colpart2: additionally the column sums of dH * (the stored activation) (the gradient of r_mag)
    const float* bias2;
    const float* cscale;
    float* out2;
    float* colpart2;
    // pv_sae_relu_step: the step's mode word (device).  Non-NULL: the kernel runs only when *gate == 1 (the sparse attempt could not
    // hold the batch); NULL: always.
    const uint32_t* gate;
    // split-fp16 form (SPLIT kernels, round 6): DG_AMAX_SLOTS words each -- the bit patterns of partial maxima of |A| and |B| (any entry
    // may be 0); the largest gives the operand's power-of-two scale.  amax_out (ENC / GENC / DH epilogues, or NULL): the same for the
    // tensor this launch writes to `out` (atomicMax on bit patterns of non-negative floats: order-independent, so run-to-run identical)
    const uint32_t* a_max;
    const uint32_t* b_max;
    uint32_t* amax_out;
    // round 6: the tail of SURVEY.md 8(f) row 3 on the ENC / DH epilogues
    int act;                                  // PV_SAE_ACT_TANH_RELU: f = tanh(relu(.)) (sae.py:823-830); DH: x (1 - f^2)
    float lp;                                 // p of the sparsity term ||f_n||_p (sae.py:617); 0 / 1: the 1-norm
    float* lp_part;                           // ENC, lp > 1: [M][ceil(N / 64)] sums of f^p over the row's 64 columns of a wave
    const float* lp_tok;                      // DH, lp > 1: [M] l1 / N_global * ||f_n||_p^(1 - p), the per-token factor of d ||f_n||_p / d f
};

constexpr int DG_AMAX_SLOTS = 256;

// 16 bytes from global memory, or zeros (a plain branch: `ok ? *p : zero` makes hipcc select between two ADDRESSES and park the
// staging registers in scratch)
__device__ __forceinline__ uint4 ld16_or_zero(const float* ptr, bool ok) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ok) v = *reinterpret_cast<const uint4*>(ptr);
    return v;
}

// ---- the split-fp16 K loop (round 6) ------------------------------------------------------------------------------------------
// The fp32 matrix instruction runs at 1/16 of the fp16 one.  An fp32 value x scaled by a power of two s splits EXACTLY into
// s x = hi + lo + r with hi = fp16(s x), lo = fp16(s x - hi) and |r| <= 2^-20 |s x| (both conversions round toward zero:
// v_cvt_pkrtz_f16_f32, one instruction per pair), so a . b = (a_hi b_hi + a_hi b_lo + a_lo b_hi) / (s_a s_b) up to 2^-19 relative per
// product: three fp16 products with fp32 accumulation instead of one fp32 product -- the same tiles and epilogues at 16 / 3 of the
// matrix rate.  s = 2^(13 - floor(log2 max|.|)) per TENSOR (DenseGemm.a_max / b_max: the producers track the maxima), so the largest
// entry lands in [2^13, 2^14) and every entry within 2^-14 of it keeps 21+ bits (smaller ones lose relative, not absolute,
// accuracy: their lo part goes denormal at 2^-24 of the scaled range).  Conversion happens in the staging pass the fp32 kernel has
// anyway (global -> registers -> LDS); the LDS image of an operand is two planes (hi, lo) of [128 rows][32 k] halves -- 64-byte rows,
// 16-byte chunk c of row r at c ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragments, the filter GEMM's image -- whichever of
// its two row-major layouts the fp32 matrix lies in: K-contiguous rows are converted four k at a time, row-contiguous k-slices are
// transposed on the way (a lane loads the same four rows at four consecutive k and writes four 8-byte row pieces).
typedef _Float16 dg_f16x8 __attribute__((ext_vector_type(8)));
constexpr int DG_SPL_PLANE = 128 * 64;          // one operand plane of a stage
constexpr int DG_SPL_STAGE = 4 * DG_SPL_PLANE;  // A hi, A lo, B hi, B lo

// power-of-two scale of an operand from its tracked maximum (bit pattern of max|.|): the largest entry lands in [2^13, 2^14)
__device__ __forceinline__ void dg_scale_of(uint32_t max_bits, float& s, float& inv) {
    int e = (int)((max_bits >> 23) & 0xffu);
    e = e < 24 ? 24 : (e > 250 ? 250 : e);      // (zeros / denormal maxima: any scale is exact on them)
    s = __uint_as_float((uint32_t)(267 - e) << 23);
    inv = __uint_as_float((uint32_t)(e - 13) << 23);
}

// hi / lo halves of four scaled floats, packed as (k0 k1), (k2 k3)
// (both conversions round toward zero, v_cvt_pkrtz_f16_f32: one full-rate instruction per pair.  The round-to-nearest pair
// conversion of gfx950, v_cvt_pk_f16_f32 (-DPV_SPLIT_RNE), is 8 x more accurate per element and was measured 2.1 ms per step slower
// -- 6.02 against 3.90 ms for the ReLU + L1 step, profiles/r06_dense_split_fp16_ab.txt -- while the product's error is already below
// that of an fp32 accumulation either way: a numpy emulation over K = 768 gives 3.4e-7 of the largest entry for this form, 6e-8 for
// round-to-nearest, 6.2e-7 for numpy's own fp32 matmul.)
__device__ __forceinline__ void dg_split4(float x0, float x1, float x2, float x3, float s, uint2& hi, uint2& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 a = {x0 * s, x1 * s}, b = {x2 * s, x3 * s};
#ifndef PV_SPLIT_RNE
    typedef __fp16 hz2 __attribute__((ext_vector_type(2)));
    const hz2 z01 = __builtin_amdgcn_cvt_pkrtz(a[0], a[1]), z23 = __builtin_amdgcn_cvt_pkrtz(b[0], b[1]);
    const hz2 y01 = __builtin_amdgcn_cvt_pkrtz(a[0] - (float)z01[0], a[1] - (float)z01[1]);
    const hz2 y23 = __builtin_amdgcn_cvt_pkrtz(b[0] - (float)z23[0], b[1] - (float)z23[1]);
    const h2 h01 = __builtin_bit_cast(h2, z01), h23 = __builtin_bit_cast(h2, z23), l01 = __builtin_bit_cast(h2, y01), l23 = __builtin_bit_cast(h2, y23);
#else
    const h2 h01 = __builtin_convertvector(a, h2), h23 = __builtin_convertvector(b, h2);
    const h2 l01 = __builtin_convertvector(a - __builtin_convertvector(h01, f2), h2);
    const h2 l23 = __builtin_convertvector(b - __builtin_convertvector(h23, f2), h2);
#endif
    hi = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
    lo = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
}

template <bool A_KM, bool B_KN, int EPI_, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void dense_gemm_kernel(const DenseGemm p) {
    // (the tail variants are their own instantiations: tanhf / powf inlined into the plain ENC / DH epilogues cost the split-fp16 step
    // 2.2 ms of 3.8 although never executed there)
    constexpr bool TAIL = EPI_ == DG_EPI_ENC_T || EPI_ == DG_EPI_DH_T;
    constexpr int EPI = EPI_ == DG_EPI_ENC_T ? DG_EPI_ENC : (EPI_ == DG_EPI_DH_T ? DG_EPI_DH : EPI_);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (p.gate && *p.gate != 1u) return;                   // (uniform over the grid)
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * DG_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware bijective remap (block b runs on XCD b % 8): each XCD gets a contiguous run of tiles, N fastest
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int ntn = (p.N + DG_BN - 1) / DG_BN;
    // Tile order inside an XCD's run.  N fastest (group 0) lets the 64 tiles an XCD runs at a time share ONE row slab and stream 64
    // different column slabs: at a wide output (the encoder and dH GEMMs: 192 column tiles of 393 KB operand rows each) every XCD
    // drags the whole B matrix through its 4 MB L2 once per row tile -- 2.8 GB of HBM fetches for a 75 MB matrix, PMC-measured, and
    // the kernel was bound by them.  Grouped (group G > 0): G row tiles x all column tiles per group, the row tile fastest inside --
    // the G tiles that share a column slab run side by side, the slab is fetched once per group.
    int tile_m, tile_n;
    if (p.group > 0) {
        const int ntm = (p.M + DG_BM - 1) / DG_BM;
        const int per = p.group * ntn, g = swz / per, idx = swz - g * per;
        const int gc = min(p.group, ntm - g * p.group);
        tile_n = idx / gc;
        tile_m = g * p.group + (idx - tile_n * gc);
    } else {
        tile_m = swz / ntn;
        tile_n = swz - tile_m * ntn;
    }
    const int m0 = tile_m * DG_BM, n0 = tile_n * DG_BN;
    const int k_begin = blockIdx.y * p.k_chunk, k_end = min(p.K, k_begin + p.k_chunk);
    const int nk = (k_end - k_begin + DG_KSLAB - 1) / DG_KSLAB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int l31 = lane & 31, half = lane >> 5;
    float inv_a = 1.0f, inv_b = 1.0f;                      // SPLIT: the two inverse operand scales (applied in the epilogue)
    if constexpr (SPLIT) {
        // ---- operand scales: the largest of the tracked partial maxima (one slot per thread)
        float sa, sb;
        {
            uint32_t ma = p.a_max[tid], mb = p.b_max[tid];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ma = max(ma, (uint32_t)__shfl_xor((int)ma, o, 64));
                mb = max(mb, (uint32_t)__shfl_xor((int)mb, o, 64));
            }
            uint32_t* red = reinterpret_cast<uint32_t*>(smem);
            if (lane == 0) { red[wave] = ma; red[4 + wave] = mb; }
            __syncthreads();
            ma = max(max(red[0], red[1]), max(red[2], red[3]));
            mb = max(max(red[4], red[5]), max(red[6], red[7]));
            __syncthreads();
            dg_scale_of(ma, sa, inv_a);
            dg_scale_of(mb, sb, inv_b);
        }
        uint4 ra[4], rb[4];
        const int rgl = lane & 7, kql = lane >> 3;             // row-contiguous sources: rows wave * 32 + 4 rgl .. + 3 at k = 4 kql + i
        auto load_slab = [&](int kt) {
            const int kb = k_begin + kt * DG_KSLAB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i;
                if constexpr (A_KM) {
                    const int k = kb + 4 * kql + i, mm = m0 + wave * 32 + 4 * rgl;
                    ra[i] = ld16_or_zero(p.A + (int64_t)k * p.lda + mm, k < k_end && mm < p.M);
                } else {
                    const int mm = m0 + (c >> 3), k = kb + (c & 7) * 4;
                    ra[i] = ld16_or_zero(p.A + (int64_t)mm * p.lda + k, mm < p.M && k < k_end);
                }
                if constexpr (B_KN) {
                    const int k = kb + 4 * kql + i, nn = n0 + wave * 32 + 4 * rgl;
                    rb[i] = ld16_or_zero(p.B + (int64_t)k * p.ldb + nn, k < k_end && nn < p.N);
                } else {
                    const int nn = n0 + (c >> 3), k = kb + (c & 7) * 4;
                    rb[i] = ld16_or_zero(p.B + (int64_t)nn * p.ldb + k, nn < p.N && k < k_end);
                }
            }
        };
        auto put = [&](unsigned char* plane_hi, int row, int kgrp, float x0, float x1, float x2, float x3, float sc) {
            uint2 hi, lo;
#if defined(PV_DG_NOCVT)
            hi = make_uint2(__float_as_uint(x0), __float_as_uint(x1)); lo = make_uint2(__float_as_uint(x2), __float_as_uint(x3));
#elif defined(PV_DG_HALFCVT)                           // (timing ablation: the hi halves only -- 2^-11 products, the step stays dense)
            {
                typedef __fp16 hz2 __attribute__((ext_vector_type(2)));
                const hz2 z01 = __builtin_amdgcn_cvt_pkrtz(x0 * sc, x1 * sc), z23 = __builtin_amdgcn_cvt_pkrtz(x2 * sc, x3 * sc);
                hi = make_uint2(__builtin_bit_cast(uint32_t, z01), __builtin_bit_cast(uint32_t, z23));
                lo = make_uint2(0u, 0u);
            }
#else
            dg_split4(x0, x1, x2, x3, sc, hi, lo);
#endif
            const int off = row * 64 + ((((kgrp >> 1) ^ (row >> 2)) & 3) << 4) + (kgrp & 1) * 8;
            *reinterpret_cast<uint2*>(plane_hi + off) = hi;
            *reinterpret_cast<uint2*>(plane_hi + DG_SPL_PLANE + off) = lo;
        };
        auto store_operand = [&](unsigned char* plane_hi, const uint4 (&r)[4], bool km, float sc) {
            if (km) {
                const int row = wave * 32 + 4 * rgl;
                put(plane_hi, row + 0, kql, __uint_as_float(r[0].x), __uint_as_float(r[1].x), __uint_as_float(r[2].x), __uint_as_float(r[3].x), sc);
                put(plane_hi, row + 1, kql, __uint_as_float(r[0].y), __uint_as_float(r[1].y), __uint_as_float(r[2].y), __uint_as_float(r[3].y), sc);
                put(plane_hi, row + 2, kql, __uint_as_float(r[0].z), __uint_as_float(r[1].z), __uint_as_float(r[2].z), __uint_as_float(r[3].z), sc);
                put(plane_hi, row + 3, kql, __uint_as_float(r[0].w), __uint_as_float(r[1].w), __uint_as_float(r[2].w), __uint_as_float(r[3].w), sc);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = tid + 256 * i;
                    put(plane_hi, c >> 3, c & 7, __uint_as_float(r[i].x), __uint_as_float(r[i].y), __uint_as_float(r[i].z), __uint_as_float(r[i].w), sc);
                }
            }
        };
        auto store_slab = [&](int buf) {
            store_operand(smem + buf * DG_SPL_STAGE, ra, A_KM, sa);
            store_operand(smem + buf * DG_SPL_STAGE + 2 * DG_SPL_PLANE, rb, B_KN, sb);
        };
        if (nk > 0) {
            load_slab(0);
            store_slab(0);
        }
        __syncthreads();
        const int sw = (l31 >> 2) & 3;
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
#ifndef PV_DG_NOLOAD                                   // (timing ablations of tools/build_variant.sh builds: results are wrong with any of them)
            if (kt + 1 < nk) load_slab(kt + 1);
#endif
            const unsigned char* Ab = smem + buf * DG_SPL_STAGE;
            const unsigned char* Bb = Ab + 2 * DG_SPL_PLANE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int co = ((ks * 2 + half) ^ sw) << 4;
                dg_f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const unsigned char* a = Ab + (wm * 64 + mi * 32 + l31) * 64 + co;
                    ah[mi] = __builtin_bit_cast(dg_f16x8, *reinterpret_cast<const uint4*>(a));
                    al[mi] = __builtin_bit_cast(dg_f16x8, *reinterpret_cast<const uint4*>(a + DG_SPL_PLANE));
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const unsigned char* b = Bb + (wn * 64 + ni * 32 + l31) * 64 + co;
                    bh[ni] = __builtin_bit_cast(dg_f16x8, *reinterpret_cast<const uint4*>(b));
                    bl[ni] = __builtin_bit_cast(dg_f16x8, *reinterpret_cast<const uint4*>(b + DG_SPL_PLANE));
                }
#ifndef PV_DG_NOMFMA
                // (the two cross terms first: they are 2^-11 of the main term)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
#else
                acc[0][0][0] += (float)ah[0][0] + (float)al[1][1] + (float)bh[0][2] + (float)bl[1][3] + (float)ah[1][4] + (float)al[0][5] + (float)bh[1][6] + (float)bl[0][7];
#endif
            }
            if (kt + 1 < nk) store_slab(buf ^ 1);
            __syncthreads();
        }
    } else {
        uint4 ra[4], rb[4];
        auto load_slab = [&](int kt) {
            const int kb = k_begin + kt * DG_KSLAB;
    #pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i;
                if constexpr (A_KM) {                      // chunk c -> k row c >> 5, 4-float group c & 31 of the 128 rows
                    const int k = kb + (c >> 5), mm = m0 + (c & 31) * 4;
                    ra[i] = ld16_or_zero(p.A + (int64_t)k * p.lda + mm, k < k_end && mm < p.M);
                } else {                                   // chunk c -> row c >> 3, 4-float k group c & 7
                    const int mm = m0 + (c >> 3), k = kb + (c & 7) * 4;
                    ra[i] = ld16_or_zero(p.A + (int64_t)mm * p.lda + k, mm < p.M && k < k_end);
                }
                if constexpr (B_KN) {
                    const int k = kb + (c >> 5), nn = n0 + (c & 31) * 4;
                    rb[i] = ld16_or_zero(p.B + (int64_t)k * p.ldb + nn, k < k_end && nn < p.N);
                } else {
                    const int nn = n0 + (c >> 3), k = kb + (c & 7) * 4;
                    rb[i] = ld16_or_zero(p.B + (int64_t)nn * p.ldb + k, nn < p.N && k < k_end);
                }
            }
        };
        auto store_slab = [&](int buf) {
    #pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i;
                if constexpr (A_KM) *reinterpret_cast<uint4*>(As + buf * DG_TILE + (c >> 5) * DG_KROW + (c & 31) * 16) = ra[i];
                else *reinterpret_cast<uint4*>(As + buf * DG_TILE + (c >> 3) * DG_ROWB + (c & 7) * 16) = ra[i];
                if constexpr (B_KN) *reinterpret_cast<uint4*>(Bs + buf * DG_TILE + (c >> 5) * DG_KROW + (c & 31) * 16) = rb[i];
                else *reinterpret_cast<uint4*>(Bs + buf * DG_TILE + (c >> 3) * DG_ROWB + (c & 7) * 16) = rb[i];
            }
        };

        if (nk > 0) {
            load_slab(0);
            store_slab(0);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_slab(kt + 1);
            const unsigned char* Ab = As + buf * DG_TILE;
            const unsigned char* Bb = Bs + buf * DG_TILE;
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                // MFMA step (j, e) consumes k = 8 j + 4 half + e of the slab on BOTH operands
                float af[2][4], bf[2][4];
    #pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    if constexpr (A_KM) {
                        const float* ak = reinterpret_cast<const float*>(Ab + (8 * j + 4 * half) * DG_KROW) + wm * 64 + mi * 32 + l31;
    #pragma unroll
                        for (int e = 0; e < 4; ++e) af[mi][e] = ak[e * (DG_KROW / 4)];
                    } else {
                        const float4 a = *reinterpret_cast<const float4*>(Ab + (wm * 64 + mi * 32 + l31) * DG_ROWB + half * 16 + j * 32);
                        af[mi][0] = a.x; af[mi][1] = a.y; af[mi][2] = a.z; af[mi][3] = a.w;
                    }
                }
    #pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if constexpr (B_KN) {
                        const float* bk = reinterpret_cast<const float*>(Bb + (8 * j + 4 * half) * DG_KROW) + wn * 64 + ni * 32 + l31;
    #pragma unroll
                        for (int e = 0; e < 4; ++e) bf[ni][e] = bk[e * (DG_KROW / 4)];
                    } else {
                        const float4 b = *reinterpret_cast<const float4*>(Bb + (wn * 64 + ni * 32 + l31) * DG_ROWB + half * 16 + j * 32);
                        bf[ni][0] = b.x; bf[ni][1] = b.y; bf[ni][2] = b.z; bf[ni][3] = b.w;
                    }
                }
    #pragma unroll
                for (int e = 0; e < 4; ++e)
    #pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
    #pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][e], bf[ni][e], acc[mi][ni], 0, 0, 0);
            }
            if (kt + 1 < nk) store_slab(buf ^ 1);
            __syncthreads();
        }

    }

    // ---- epilogue: accumulators -> per-wave LDS block (the operand buffers are free: barrier above) -> a lane owns 8
    // consecutive columns of a row, 8 rows of the wave's 64 x 64 block
    float* Cs = reinterpret_cast<float*>(smem) + wave * (64 * DG_CS_LD);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                Cs[row * DG_CS_LD + ni * 32 + l31] = SPLIT ? acc[mi][ni][e] * inv_a * inv_b : acc[mi][ni][e];
            }
    __builtin_amdgcn_wave_barrier();                      // (wave-private block; a wave's LDS operations execute in order)
    const int cc = (lane & 7) * 8;
    const int gn = n0 + wn * 64 + cc;
    const bool col_ok = gn < p.N;                          // N % 8 == 0 is required by the launcher: a chunk is all or nothing
    float csum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) csum[i] = 0.f;
    float rsum = 0.f;
    float b8[8];
    if constexpr (EPI == DG_EPI_ENC || EPI == DG_EPI_GENC || EPI == DG_EPI_EXPB) {
#pragma unroll
        for (int i = 0; i < 8; ++i) b8[i] = 0.f;
        if (col_ok) load8(p.bias + gn, b8);
    }
    float csum2[8], bm8[8], er8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { csum2[i] = 0.f; bm8[i] = 0.f; er8[i] = 1.f; }
    if constexpr (EPI == DG_EPI_GENC) {
        if (col_ok) {
            load8(p.bias2 + gn, bm8);
            load8(p.cscale + gn, er8);
#pragma unroll
            for (int i = 0; i < 8; ++i) er8[i] = expf(er8[i]);
        }
    }
    int dslot[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dslot[i] = -1;
    if constexpr (EPI == DG_EPI_ENC || EPI == DG_EPI_DH) {
        if (p.dead_slot && col_ok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dslot[i] = p.dead_slot[gn + i];
        }
    }
    float* outz = p.out + (int64_t)blockIdx.y * p.out_zstride;
    float vmax = 0.f;
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int gm = m0 + wm * 64 + row;
        if (gm < p.M && col_ok) {
            float v[8];
            const float4 x0 = *reinterpret_cast<const float4*>(Cs + row * DG_CS_LD + cc);
            const float4 x1 = *reinterpret_cast<const float4*>(Cs + row * DG_CS_LD + cc + 4);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            float* o = outz + (int64_t)gm * p.ldo + gn;
            if constexpr (EPI == DG_EPI_ENC) {
                if (p.dead_slot) {                                     // ghost gradients: exp(hidden_pre) of the dead features (sae.py:163)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int sl = dslot[i];
                        if (sl >= 0) p.dead_act[(int64_t)gm * p.ldd + sl] = expf(v[i] + b8[i]);       // (the accurate exp: the ghost gradients cancel down to ~1e-9, where the fast exp's 1e-6 is visible)
                    }
                }
                float psum = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = fmaxf(v[i] + b8[i], 0.f);                   // hidden_pre + b_enc -> ReLU (sae.py:567-577)
                    if constexpr (TAIL) {
                        if (p.act == PV_SAE_ACT_TANH_RELU) v[i] = tanhf(v[i]);      // "tanh-relu" (sae.py:823-830)
                    }
                    csum[i] += v[i] > 0.f ? 1.f : 0.f;                // firing counts (train_sae.py:356-364)
                    rsum += v[i];                                     // ||f||_1 (sae.py:617)
                    if constexpr (TAIL) {
                        if (p.lp_part) psum += v[i] > 0.f ? powf(v[i], p.lp) : 0.f;
                    }
                }
                if constexpr (TAIL) {
                    if (p.lp_part) {                                   // sum of f^p over this row's 64 columns (the 8 lanes that share the row)
                        psum += __shfl_xor(psum, 1, 64);
                        psum += __shfl_xor(psum, 2, 64);
                        psum += __shfl_xor(psum, 4, 64);
                        if ((lane & 7) == 0) p.lp_part[(int64_t)gm * ((p.N + 63) / 64) + tile_n * 2 + wn] = psum;
                    }
                }
            } else if constexpr (EPI == DG_EPI_GENC) {
                float g8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float gate = v[i] + b8[i];                   // gating pre-activation (sae.py:703-706)
                    const float mag = v[i] * er8[i] + bm8[i];          // sae_in @ (W_enc * exp(r_mag)) + b_mag (:708-712)
                    g8[i] = fmaxf(gate, 0.f);                          // relu(gate): L1 + auxiliary path (:773-778)
                    v[i] = gate > 0.f ? fmaxf(mag, 0.f) : 0.f;         // feature_acts (:714-716)
                    csum[i] += v[i] > 0.f ? 1.f : 0.f;
                    csum2[i] += g8[i];
                    rsum += g8[i];
                }
                if (p.amax_out) {                                      // (out2 shares out's maximum: the two are ONE operand [2N][F] of the GEMMs behind)
#pragma unroll
                    for (int i = 0; i < 8; ++i) vmax = fmaxf(vmax, g8[i]);
                }
                store8(p.out2 + (int64_t)gm * p.ldo + gn, g8);
            } else if constexpr (EPI == DG_EPI_DH) {
                float f8[8];
                load8(o, f8);                                          // the stored activation: the ReLU gate of the backward
                float lpt = 0.f;
                if constexpr (TAIL) lpt = p.lp_tok ? p.lp_tok[gm] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    // d loss / d hidden_pre = (dF + l1 / N) [f > 0]; TAIL, lp > 1: the sparsity term's l1 / N ||f_n||_p^(1 - p) f^(p - 1)
                    // instead of the constant; TAIL, tanh-relu: x d tanh = 1 - f^2
                    float g = v[i] + p.add;
                    if constexpr (TAIL) {
                        if (p.lp_tok) g = v[i] + lpt * powf(f8[i], p.lp - 1.f);
                        if (p.act == PV_SAE_ACT_TANH_RELU) g *= 1.f - f8[i] * f8[i];
                    }
                    v[i] = f8[i] > 0.f ? g : 0.f;
                    if (p.dead_slot && dslot[i] >= 0) v[i] += p.dead_act[(int64_t)gm * p.ldd + dslot[i]];    // + the ghost term (not gated)
                    csum[i] += v[i];                                  // gb_enc
                    csum2[i] += v[i] * f8[i];
                }
            } else if constexpr (EPI == DG_EPI_EXPB) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = expf(v[i] + b8[i]);         // exp(hidden_pre) of the dead features (sae.py:163), as DG_EPI_ENC writes it
            } else if constexpr (EPI == DG_EPI_MUL) {
                float m8[8];
                load8(p.mul + (int64_t)gm * p.ldo + gn, m8);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= m8[i];
            }
            if (p.amax_out) {
#pragma unroll
                for (int i = 0; i < 8; ++i) vmax = fmaxf(vmax, fabsf(v[i]));
            }
            store8(o, v);
        }
    }
    if (p.amax_out) {                                      // max |out| of this wave's block -> one of the DG_AMAX_SLOTS partial maxima
        vmax = wave_max(vmax);
        if (lane == 0) atomicMax(p.amax_out + ((blockIdx.x * 4 + wave) & (DG_AMAX_SLOTS - 1)), __float_as_uint(vmax));
    }
    if constexpr (EPI == DG_EPI_ENC || EPI == DG_EPI_DH || EPI == DG_EPI_GENC) {
        // the wave's 64 rows of each column: lanes with equal (lane & 7) hold the same 8 columns
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            csum[i] += __shfl_xor(csum[i], 8, 64);
            csum[i] += __shfl_xor(csum[i], 16, 64);
            csum[i] += __shfl_xor(csum[i], 32, 64);
        }
        const int rblk = tile_m * 2 + wm;
        if (lane < 8 && col_ok && m0 + wm * 64 < p.M) store8(p.colpart + (int64_t)rblk * p.N + gn, csum);
        if (EPI != DG_EPI_ENC && p.colpart2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                csum2[i] += __shfl_xor(csum2[i], 8, 64);
                csum2[i] += __shfl_xor(csum2[i], 16, 64);
                csum2[i] += __shfl_xor(csum2[i], 32, 64);
            }
            if (lane < 8 && col_ok && m0 + wm * 64 < p.M) store8(p.colpart2 + (int64_t)rblk * p.N + gn, csum2);
        }
        if constexpr (EPI == DG_EPI_ENC || EPI == DG_EPI_GENC) {
            rsum = wave_sum(rsum);
            const int ncb = (p.N + 63) / 64;
            if (lane == 0 && m0 + wm * 64 < p.M && n0 + wn * 64 < p.N) p.rowpart[(int64_t)rblk * ncb + tile_n * 2 + wn] = rsum;
        }
    }
}

template <bool A_KM, bool B_KN, int EPI, bool SPLIT>
int launch_dense_gemm_form(const DenseGemm& p, int splits, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_gemm_kernel<A_KM, B_KN, EPI, SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DG_LDS);
        if (e != hipSuccess) {
            pv_set_error(std::string("hipFuncSetAttribute(dense_gemm_kernel): ") + hipGetErrorString(e));
            return PV_ERR_HIP;
        }
        attr_done = true;
    }
    const int ntm = (p.M + DG_BM - 1) / DG_BM, ntn = (p.N + DG_BN - 1) / DG_BN;
    DenseGemm q = p;
    // wide outputs: grouped tile order (the kernel's comment); tuning key dense_group: -1 auto (4 row tiles per group), 0 off, n
    q.group = g_pv_tuning.dense_group >= 0 ? g_pv_tuning.dense_group : (ntn > 16 ? 4 : 0);
    hipLaunchKernelGGL((dense_gemm_kernel<A_KM, B_KN, EPI, SPLIT>), dim3(ntm * ntn, splits), dim3(256), DG_LDS, stream, q);
    PV_LAUNCH_CHECK("dense_gemm_kernel");
    return PV_OK;
}

// the split-fp16 form when the launch carries both operand maxima (and the tuning key dense_fp32 is off), else the exact fp32 form
template <bool A_KM, bool B_KN, int EPI>
int launch_dense_gemm(const DenseGemm& p, int splits, hipStream_t stream) {
    if constexpr (EPI == DG_EPI_STORE || EPI == DG_EPI_ENC || EPI == DG_EPI_DH || EPI == DG_EPI_GENC || EPI == DG_EPI_ENC_T || EPI == DG_EPI_DH_T) {
        if (p.a_max && p.b_max && !g_pv_tuning.dense_fp32) return launch_dense_gemm_form<A_KM, B_KN, EPI, true>(p, splits, stream);
    }
    return launch_dense_gemm_form<A_KM, B_KN, EPI, false>(p, splits, stream);
}

// ---- operand maxima of the split-fp16 form ---------------------------------------------------------------------------------------
// operand maxima: bit patterns of non-negative floats order like the floats, and a maximum does not depend on the order it is taken in
// the three operands that exist before the first GEMM of a dense step -- sae_in, W_enc^T, W_dec -- in ONE launch (blockIdx.y = tensor):
// slot arrays t * DG_AMAX_SLOTS of `slots`, zeroed by the caller at the head of the step
struct AbsmaxArgs { const float* x[3]; int64_t n4[3]; int slot[3]; };
__global__ __launch_bounds__(256) void dense_absmax3_kernel(const AbsmaxArgs a, uint32_t* __restrict__ slots, const uint32_t* __restrict__ gate) {
    if (gate && *gate != 1u) return;
    const int t = blockIdx.y;
    const float* __restrict__ x = a.x[t];
    const int64_t n4 = a.n4[t];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0)
        atomicMax(slots + a.slot[t] * DG_AMAX_SLOTS + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (DG_AMAX_SLOTS - 1)), __float_as_uint(m));
}
void dense_absmax3(const float* x0, int64_t n0, int s0, const float* x1, int64_t n1, int s1, const float* x2, int64_t n2, int s2, uint32_t* slots,
                   const uint32_t* gate, hipStream_t stream) {
    AbsmaxArgs a;
    a.x[0] = x0; a.x[1] = x1; a.x[2] = x2; a.n4[0] = n0 / 4; a.n4[1] = n1 / 4; a.n4[2] = n2 / 4; a.slot[0] = s0; a.slot[1] = s1; a.slot[2] = s2;
    hipLaunchKernelGGL(dense_absmax3_kernel, dim3(1024, 3), dim3(256), 0, stream, a, slots, gate);
}

// column partials [nblk][d] -> out[j] (+ firing statistics, + per-workgroup totals for l0)
__global__ __launch_bounds__(256) void dense_colreduce_kernel(const float* __restrict__ part, int nblk, int d, float* __restrict__ out,
                                                              float* __restrict__ out2, float* __restrict__ act_freq,
                                                              float* __restrict__ n_since_fired, int update_stats,
                                                              float* __restrict__ block_tot, const uint32_t* __restrict__ gate = nullptr) {
    __shared__ float red[4];
    if (gate && *gate != 1u) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    if (j < d) {
#pragma unroll 16
        for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * d + j];        // fixed order; coalesced across the workgroup; 16 loads in flight
                                                                             // (one per trip was 64 dependent round trips: 28 us for 6 MB)
        out[j] = s;
        if (out2) out2[j] = s;
        if (update_stats) {                                                  // train_sae.py:356-361
            act_freq[j] += s;
            n_since_fired[j] = s > 0.f ? 0.f : n_since_fired[j] + 1.f;
        }
    }
    if (block_tot) {
        float t = wave_sum(s);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) red[wv] = t;
        __syncthreads();
        if (threadIdx.x == 0) block_tot[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

// decoder output from the split-K partial sums: LN-out (sae.py:89-90), mse partial (sae.py:144-149), dY.  One wave per token.
__global__ __launch_bounds__(256) void dense_finish_kernel(const float* __restrict__ x, const float* __restrict__ kpart, int splits,
                                                           int64_t zstride, const float* __restrict__ b_dec, const float* __restrict__ mu,
                                                           const float* __restrict__ sd, const float* __restrict__ norm,
                                                           float* __restrict__ sae_out, float* __restrict__ dY,
                                                           float* __restrict__ loss_partial, int n_tok, int d, float grad_scale,
                                                           float* __restrict__ err_out, const float* __restrict__ addend,
                                                           const uint32_t* __restrict__ gate = nullptr,
                                                           const float* __restrict__ ghost_x = nullptr, uint32_t* __restrict__ amax = nullptr) {
    // amax (or NULL): partial maxima of |dY| (DG_AMAX_SLOTS words, see DenseGemm.a_max)
    // transcoder (pv_sae_state.tc): x = the TARGET, b_dec = b_dec_out, addend = the skip term or nullptr; ghost_x (a transcoder with
    // ghost gradients): the INPUT activation -- err_out is then sae_out - input, what the ghost term is computed on (transcoder.py:82-86)
    if (gate && *gate != 1u) return;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    const float m = mu[n], sdv = sd[n], nf = norm[n];
    float lsum = 0.f, gmax = 0.f;
    for (int c = 4 * lane; c < d; c += 256) {
        float4 a = *reinterpret_cast<const float4*>(kpart + (int64_t)n * d + c);
        for (int z = 1; z < splits; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(kpart + z * zstride + (int64_t)n * d + c);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        if (addend) {
            const float4 t = *reinterpret_cast<const float4*>(addend + (int64_t)n * d + c);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        const float4 bd = *reinterpret_cast<const float4*>(b_dec + c);
        const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)n * d + c);
        float4 o, e, g;
        o.x = (a.x + bd.x) * sdv + m; o.y = (a.y + bd.y) * sdv + m; o.z = (a.z + bd.z) * sdv + m; o.w = (a.w + bd.w) * sdv + m;
        e.x = o.x - xv.x; e.y = o.y - xv.y; e.z = o.z - xv.z; e.w = o.w - xv.w;
        if (sae_out) *reinterpret_cast<float4*>(sae_out + (int64_t)n * d + c) = o;
        if (err_out) {
            float4 eg = e;
            if (ghost_x) {
                const float4 gx = *reinterpret_cast<const float4*>(ghost_x + (int64_t)n * d + c);
                eg.x = o.x - gx.x; eg.y = o.y - gx.y; eg.z = o.z - gx.z; eg.w = o.w - gx.w;
            }
            *reinterpret_cast<float4*>(err_out + (int64_t)n * d + c) = eg;
        }
        lsum += (e.x * e.x) / nf + (e.y * e.y) / nf + (e.z * e.z) / nf + (e.w * e.w) / nf;
        g.x = grad_scale * e.x / nf * sdv; g.y = grad_scale * e.y / nf * sdv;
        g.z = grad_scale * e.z / nf * sdv; g.w = grad_scale * e.w / nf * sdv;
        *reinterpret_cast<float4*>(dY + (int64_t)n * d + c) = g;
        gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(g.x), fabsf(g.y))), fmaxf(fabsf(g.z), fabsf(g.w)));
    }
    lsum = wave_sum(lsum);
    if (lane == 0) loss_partial[n] = lsum;
    if (amax) {
        gmax = wave_max(gmax);
        if (lane == 0) atomicMax(amax + (n & (DG_AMAX_SLOTS - 1)), __float_as_uint(gmax));
    }
}

// lp_norm > 1 (sae.py:617): per token S = sum_j f^p from the ENC epilogue's partials -> ||f_n||_p = S^(1/p) (the loss term) and
// l1 / N_global * S^(1/p - 1), the per-token factor of d ||f_n||_p / d f_j = S^(1/p - 1) f_j^(p - 1).  One wave per token, fixed order.
__global__ __launch_bounds__(256) void dense_lp_finish_kernel(const float* __restrict__ part, int ncb, float p, float l1_over_n,
                                                              float* __restrict__ lp_tok, float* __restrict__ lp_loss, int n_tok,
                                                              const uint32_t* __restrict__ gate) {
    if (gate && *gate != 1u) return;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    float s = 0.f;
    for (int c = lane; c < ncb; c += 64) s += part[(int64_t)n * ncb + c];
    s = wave_sum(s);
    if (lane == 0) {
        lp_loss[n] = s > 0.f ? powf(s, 1.f / p) : 0.f;
        lp_tok[n] = s > 0.f ? l1_over_n * powf(s, 1.f / p - 1.f) : 0.f;
    }
}

// scalars[0] = mse + l1 (+ ghost) (sae.py:628), one thread
__global__ void dense_loss_kernel(float* __restrict__ scalars, int ghost, const uint32_t* __restrict__ gate = nullptr, uint32_t want = 1u) {
    if (gate && *gate != want) return;
    if (!ghost) scalars[5] = 0.f;
    scalars[0] = scalars[1] + scalars[4] + scalars[5];
}

// ---- ghost gradients (SparseAutoencoder._compute_ghost_residual_loss, sae.py:151-179) ---------------------------------------
// rows of W_dec of the dead features, compacted (zero rows up to the padded count)
__global__ __launch_bounds__(256) void ghost_gather_rows_kernel(const float* __restrict__ W_dec, const int32_t* __restrict__ dead_idx,
                                                                int n_dead, int n_pad, int d, float* __restrict__ out) {
    const int s = blockIdx.x;
    const float* src = s < n_dead ? W_dec + (int64_t)dead_idx[s] * d : nullptr;
    for (int c = threadIdx.x; c < d; c += 256) out[(int64_t)s * d + c] = src ? src[c] : 0.f;
}
// gW_dec[dead_idx[s]] += add[s]
__global__ __launch_bounds__(256) void ghost_scatter_add_rows_kernel(float* __restrict__ gW_dec, const int32_t* __restrict__ dead_idx,
                                                                     int n_dead, int d, const float* __restrict__ add) {
    const int s = blockIdx.x;
    if (s >= n_dead) return;
    float* dst = gW_dec + (int64_t)dead_idx[s] * d;
    for (int c = threadIdx.x; c < d; c += 256) dst[c] += add[(int64_t)s * d + c];
}
// A Transcoder's ghost term rescales by _compute_mse_loss(INPUT, sae_out) (transcoder.py:82-86, sae.py:144-149, 170-173): per token
// sum_i err_i^2 / ||x_n - mean_batch(x)||, err = sae_out - x.  One wave per token.
__global__ __launch_bounds__(256) void tc_ghost_mse_kernel(const float* __restrict__ x, const float* __restrict__ err,
                                                           const float* __restrict__ xmean, float* __restrict__ part, int n_tok, int d) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    float c2 = 0.f, e2 = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float c = x[(int64_t)n * d + i] - xmean[i], e = err[(int64_t)n * d + i];
        c2 += c * c; e2 += e * e;
    }
    c2 = wave_sum(c2); e2 = wave_sum(e2);
    if (lane == 0) part[n] = e2 / sqrtf(c2);
}

// One wave per token.  res = x - sae_out = -err; G0 = exp(hidden_pre[:, dead]) @ W_dec[dead]:
//   s = ||res|| / (1e-6 + 2 ||G0||)  (detached);  G = s G0;  den = || res - mean_batch(res) ||  (detached)
//   mg = (G - res)^2 / den;  r = mse_loss / (mg + 1e-6)  (detached);  ghost loss = mean(r mg)
//   d ghost / d G0 = s r 2 (G - res) / (den N d)
__global__ __launch_bounds__(256) void ghost_rows_kernel(const float* __restrict__ err, const float* __restrict__ G0,
                                                         const float* __restrict__ colmean_err, const float* __restrict__ mse_ptr,
                                                         float* __restrict__ dG0, float* __restrict__ part, int n_tok, int d,
                                                         float inv_count) {
    // colmean_err / mse_ptr / inv_count: over the GLOBAL batch (tokens sharded over ranks: pv_sae_ghost.err_colmean / mse_global /
    // n_global), this call's own batch otherwise
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    const float* e = err + (int64_t)n * d;
    const float* g0 = G0 + (int64_t)n * d;
    float r2 = 0.f, g2 = 0.f, c2 = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float res = -e[i], g = g0[i], rc = res + colmean_err[i];        // res - mean(res) = -(err - mean(err))
        r2 += res * res; g2 += g * g; c2 += rc * rc;
    }
    r2 = wave_sum(r2); g2 = wave_sum(g2); c2 = wave_sum(c2);
    const float s = sqrtf(r2) / (1e-6f + 2.0f * sqrtf(g2));
    const float den = sqrtf(c2);
    const float mse = *mse_ptr;
    float acc = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float res = -e[i], diff = g0[i] * s - res;
        const float mg = diff * diff / den;
        const float r = mse / (mg + 1e-6f);
        acc += r * mg;
        dG0[(int64_t)n * d + i] = s * r * 2.0f * diff / den * inv_count;
    }
    acc = wave_sum(acc);
    if (lane == 0) part[n] = acc;
}


// out[i] = sum_z part[z * zstride + i] (fixed order): the split-K partials of gW_skip
__global__ __launch_bounds__(256) void dense_sumz_kernel(const float* __restrict__ part, int splits, int64_t zstride, int64_t n,
                                                         float* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float4 a = *reinterpret_cast<const float4*>(part + i);
    for (int z = 1; z < splits; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(part + z * zstride + i);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    *reinterpret_cast<float4*>(out + i) = a;
}
}  // namespace

namespace {
struct GhostWs {
    size_t total, dead_act, dhd, wdd, g0, dg0, err, tmpw, colmean, part, colpart, vec_a, vec_b, colpart_p;
    int n_pad;
};
GhostWs ghost_carve(const pv_sae_desc& d, int N, int n_dead) {
    GhostWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (size_t)pv_align_up((int64_t)bytes, 256); return o; };
    w.n_pad = n_dead > 0 ? (n_dead + 7) / 8 * 8 : 0;
    const size_t P = w.n_pad, D = d.d_in, n = N;
    w.dead_act = take(n * P * 4);
    w.dhd = take(n * P * 4);
    w.wdd = take(P * D * 4);
    w.tmpw = take(P * D * 4);
    w.g0 = take(n * D * 4);
    w.dg0 = take(n * D * 4);
    w.err = take(n * D * 4);
    w.colmean = take(D * 4);
    w.part = take(n * 4);
    w.colpart = take((n / 16 + 2) * D * 4);
    w.vec_a = take((P + 8) * 4);                       // pv_sae_topk_ghost: b_enc of the dead features / column sums over them
    w.vec_b = take((P + 8) * 4);
    w.colpart_p = take((n / 16 + 2) * (P + 8) * 4);
    w.total = off + 256;
    return w;
}

// out[s] = src[idx[s]] for s < n, pad_value up to n_pad
__global__ __launch_bounds__(256) void ghost_gather_vec_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int n, int n_pad,
                                                               float pad_value, float* __restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < n_pad) out[s] = s < n ? src[idx[s]] : pad_value;
}
// dst[idx[s]] += add[s]
__global__ __launch_bounds__(256) void ghost_scatter_add_vec_kernel(float* __restrict__ dst, const int32_t* __restrict__ idx, int n,
                                                                    const float* __restrict__ add) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < n) dst[idx[s]] += add[s];
}
// err = sae_out - y
__global__ __launch_bounds__(256) void ghost_err_kernel(const float* __restrict__ sae_out, const float* __restrict__ y, float* __restrict__ err,
                                                        int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = reinterpret_cast<const float4*>(sae_out)[i], b = reinterpret_cast<const float4*>(y)[i];
    reinterpret_cast<float4*>(err)[i] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
// scalars[0] = mse + ghost residual loss (top-k: no L1 term), scalars[4] = 0
__global__ void topk_ghost_loss_kernel(float* __restrict__ scalars) {
    scalars[4] = 0.f;
    scalars[0] = scalars[1] + scalars[5];
}
}  // namespace

// ---- transcoder skip connection (sae/transcoder.py:73-76): two more GEMMs on the same kernel ---------------------------------
int sae_tc_skip_forward(const pv_sae_desc& d, const pv_sae_state* st, const float* x, int N, const float** skip, hipStream_t stream) {
    *skip = nullptr;
    if (!st->tc.W_skip) return PV_OK;
    const int D = d.d_in;
    float* buf = (float*)st->tc.scratch;                    // [N][D]
    DenseGemm g = {};
    g.A = x; g.lda = D; g.B = st->tc.W_skip; g.ldb = D; g.M = N; g.N = D; g.K = D; g.k_chunk = D;      // x @ W_skip^T: B as [N][K]
    g.out = buf; g.ldo = D;
    const int rc = launch_dense_gemm<false, false, DG_EPI_STORE>(g, 1, stream);
    if (rc) return rc;
    *skip = buf;
    return PV_OK;
}

int sae_tc_skip_backward(const pv_sae_desc& d, const pv_sae_state* st, const float* x, const float* dY, int N, hipStream_t stream) {
    if (!st->tc.W_skip) return PV_OK;
    const int D = d.d_in;
    float* part = (float*)st->tc.scratch + (size_t)N * D;   // [S][D][D]
    // gW_skip[o][i] = sum_n dY[n][o] x[n][i]: both operands row-major over the tokens, K = N split over the workgroups
    int S = (N + DG_KSLAB - 1) / DG_KSLAB;
    if (S > PV_SAE_SKIP_SPLITK) S = PV_SAE_SKIP_SPLITK;
    DenseGemm g = {};
    g.A = dY; g.lda = D; g.B = x; g.ldb = D; g.M = D; g.N = D; g.K = N;
    g.k_chunk = ((N + S - 1) / S + DG_KSLAB - 1) / DG_KSLAB * DG_KSLAB;
    g.out = part; g.ldo = D; g.out_zstride = (int64_t)D * D;
    const int rc = launch_dense_gemm<true, true, DG_EPI_STORE>(g, S, stream);
    if (rc) return rc;
    const int64_t n = (int64_t)D * D;
    hipLaunchKernelGGL(dense_sumz_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)part, S, n, n,
                       st->tc.gW_skip);
    PV_LAUNCH_CHECK("dense_sumz_kernel");
    return PV_OK;
}

extern "C" size_t pv_sae_ghost_workspace_bytes(const pv_sae_plan* plan, int32_t n_tokens, int32_t n_dead) {
    if (!plan || n_tokens < 1 || n_dead < 0) return 0;
    return ghost_carve(plan->d, n_tokens, n_dead).total;
}

namespace {
// validation shared by pv_sae_dense_step and pv_sae_relu_step
// scalars[7] = _compute_mse_loss(INPUT, sae_out) of a transcoder's ghost term (err = sae_out - x); scratch: the ghost workspace's
// colmean / colpart / part (re-used afterwards for the residual's own column mean)
int tc_ghost_mse(const float* x, const float* err, int N, int D, unsigned char* gwb, const GhostWs& gw, float* scalars, hipStream_t stream) {
    int rc = sae_colsum(x, N, D, (float*)(gwb + gw.colmean), 1.0f / (float)N, (float*)(gwb + gw.colpart), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(tc_ghost_mse_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, x, err, (const float*)(gwb + gw.colmean),
                       (float*)(gwb + gw.part), N, D);
    PV_LAUNCH_CHECK("tc_ghost_mse_kernel");
    sae_reduce_sum((const float*)(gwb + gw.part), scalars, N, 1.0f / ((float)N * (float)D), 7, -1, stream);
    return PV_OK;
}

int dense_require(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, int32_t n_global, int update_stats, pv_sae_out* out,
                  void* workspace) {
    PV_REQUIRE(plan && st && x && out && workspace, "null argument");
    PV_REQUIRE(out->scalars, "pv_sae_out.scalars");
    PV_REQUIRE(st->W_dec && st->b_enc && st->b_dec && st->gW_enc && st->gW_dec && st->gb_enc && st->gb_dec, "state");
    PV_REQUIRE(st->W_encT, "the transposed encoder copy (pv_sae_state.W_encT) is required");
    PV_REQUIRE(!update_stats || (st->act_freq_scores && st->n_fwd_since_fired), "stats buffers");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(N >= 1 && N <= d.max_tokens, "n_tokens exceeds plan max_tokens");
    PV_REQUIRE(n_global >= N, "n_global must be >= n_tokens");
    PV_REQUIRE(d.d_in % 8 == 0 && d.d_sae % 8 == 0, "the dense step needs d_in and d_sae to be multiples of 8");
    return PV_OK;
}

// renorm (flags), LN-in / sae_in / loss normaliser, the transcoder's target normaliser and skip term: what both forms of the
// ReLU + L1 step start from
int dense_prepare(pv_sae_plan* plan, pv_sae_state* st, const float* x, int N, const float* batch_mean, int flags, bool want_filter_inputs,
                  unsigned char* wsb, const SaeWs& ws, const float** skip, void* stream_) {
    const pv_sae_desc& d = plan->d;
    hipStream_t stream = (hipStream_t)stream_;
    plan->live_offs = nullptr;
    plan->renorm_pending = false;
    // set_decoder_norm_to_unit_norm (train_sae.py:307): the dense GEMMs read W_dec as it lies, so the rows are rewritten here
    if (flags & PV_SAE_RENORM_DECODER) {
        int rc = pv_sae_renorm_decoder(plan, st, stream_);
        if (rc) return rc;
    }
    int rc = sae_prep(d, x, (const float*)st->b_dec, batch_mean, N, want_filter_inputs, wsb, ws, stream, sae_in_width(d, st));
    if (rc) return rc;
    *skip = nullptr;
    if (sae_is_tc(st)) {
        rc = sae_tc_target_norm(d, st, batch_mean, N, wsb, ws, stream);
        if (rc) return rc;
        rc = sae_tc_skip_forward(d, st, x, N, skip, stream);
        if (rc) return rc;
    }
    return PV_OK;
}

// The five GEMMs + their small kernels (see the file header); `gate` as in DenseGemm.
int dense_step_body(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, int32_t n_global, int update_stats,
                    float l1_coefficient, const pv_sae_ghost* ghost, const GhostWs& gw, pv_sae_out* out, unsigned char* wsb,
                    const SaeWs& ws, const float* skip, const uint32_t* gate, hipStream_t stream, bool bias_grads = true) {
    const pv_sae_desc& d = plan->d;
    const int F = d.d_sae, D = d.d_in;
    const bool tc = sae_is_tc(st);
    unsigned char* gwb = ghost ? (unsigned char*)ghost->workspace : nullptr;
    const int nd = ghost ? ghost->n_dead : 0;
    int rc = PV_OK;
    float* f = (float*)(wsb + ws.hidden);                  // [N][F]: f, later dH
    float* sae_in = (float*)(wsb + ws.sae_in);
    float* dY = (float*)(wsb + ws.dY);
    float* colpart = (float*)(wsb + ws.dense_colpart);
    float* rowpart = (float*)(wsb + ws.dense_rowpart);
    float* kpart = (float*)(wsb + ws.dense_kpart);
    const int rblk = (N + 63) / 64, cblk = (F + 63) / 64;
    const int nb_f = (F + 255) / 256;
    float* blk_tot = (float*)(wsb + ws.sqpart);            // 1024 floats of scratch (nb_f <= 256)
    // operand maxima of the split-fp16 GEMMs (see dense_gemm_kernel): sae_in, W_enc^T, f, W_dec, dY, dH.  The three that exist before
    // the first GEMM are taken here, dY behind the finish kernel, f and dH by the epilogues that write them.
    uint32_t* amax = (uint32_t*)(wsb + ws.dense_amax);
    enum { AM_X = 0, AM_WENC = 1, AM_F = 2, AM_WDEC = 3, AM_DY = 4, AM_DH = 5 };
    auto am = [&](int t) { return amax + t * DG_AMAX_SLOTS; };
    const bool split = !g_pv_tuning.dense_fp32 && (D % 4) == 0 && (F % 4) == 0 && (N % 4) == 0;
    const bool lp_on = d.lp_norm != 0.f && d.lp_norm != 1.f;      // the sparsity term is ||f_n||_p, p > 1 (sae.py:617)
    if (split) {
        PV_HIP_CHECK(hipMemsetAsync(amax, 0, (size_t)PV_SAE_AMAX_TENSORS * DG_AMAX_SLOTS * 4, stream));
        dense_absmax3(sae_in, (int64_t)N * D, AM_X, (const float*)st->W_encT, (int64_t)F * D, AM_WENC, (const float*)st->W_dec, (int64_t)F * D, AM_WDEC,
                      amax, gate, stream);
        PV_LAUNCH_CHECK("dense_absmax3_kernel");
    }
    {
        ProfScope prof(PV_PROF_SAE_ENC, stream, 2.0 * N * (double)D * F, ((double)N * D + (double)D * F + (double)N * F) * 4.0);
        // G1: f = relu(sae_in @ W_enc + b_enc)
        DenseGemm g = {};
        // (W_enc is read as its transposed fp32 master W_encT [F][D] -- the copy Adam runs in, always current; the parameter's
        // own layout may be materialised lazily)
        g.A = sae_in; g.lda = D; g.B = st->W_encT; g.ldb = D; g.M = N; g.N = F; g.K = D; g.k_chunk = D;
        g.out = f; g.ldo = F; g.bias = st->b_enc; g.colpart = colpart; g.rowpart = rowpart; g.gate = gate;
        if (split) { g.a_max = am(AM_X); g.b_max = am(AM_WENC); g.amax_out = am(AM_F); }
        g.act = d.activation; g.lp = d.lp_norm;
        if (lp_on) g.lp_part = (float*)(wsb + ws.dense_lp_part);
        if (nd > 0) {
            PV_HIP_CHECK(hipMemsetAsync(gwb + gw.dead_act, 0, (size_t)N * gw.n_pad * 4, stream));       // (padding columns stay zero)
            g.dead_slot = ghost->dead_slot; g.dead_act = (float*)(gwb + gw.dead_act); g.ldd = gw.n_pad;
        }
        rc = sae_plain_relu(d) ? launch_dense_gemm<false, false, DG_EPI_ENC>(g, 1, stream) : launch_dense_gemm<false, false, DG_EPI_ENC_T>(g, 1, stream);
        if (rc) return rc;
        // firing counts, statistics, l0, l1
        hipLaunchKernelGGL(dense_colreduce_kernel, dim3(nb_f), dim3(256), 0, stream, (const float*)colpart, rblk, F,
                           out->fire_count ? out->fire_count : (float*)(wsb + ws.rowsq), (float*)nullptr, st->act_freq_scores,
                           st->n_fwd_since_fired, update_stats, blk_tot, gate);
        PV_LAUNCH_CHECK("dense_colreduce_kernel");
        sae_reduce_sum(blk_tot, out->scalars, nb_f, 1.0f / (float)N, 2, -1, stream, gate, 1u);                       // l0 (train_sae.py:364)
        if (lp_on) {
            // l1_loss = l1 * mean_n ||f_n||_p (sae.py:617-626 with lp_norm = p)
            hipLaunchKernelGGL(dense_lp_finish_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, (const float*)(wsb + ws.dense_lp_part), cblk,
                               d.lp_norm, l1_coefficient / (float)n_global, (float*)(wsb + ws.dense_lp_tok), (float*)(wsb + ws.dense_lp_loss),
                               N, gate);
            PV_LAUNCH_CHECK("dense_lp_finish_kernel");
            sae_reduce_sum((const float*)(wsb + ws.dense_lp_loss), out->scalars, N, l1_coefficient / (float)n_global, 4, -1, stream, gate, 1u);
        } else {
            sae_reduce_sum(rowpart, out->scalars, rblk * cblk, l1_coefficient / (float)n_global, 4, -1, stream, gate, 1u);  // l1_loss (sae.py:617-626)
        }
    }
    {
        ProfScope prof(PV_PROF_SAE_BWD, stream, 8.0 * N * (double)D * F, 0.0);
        // G2: pre = f @ W_dec, split over K
        const int S = PV_SAE_DENSE_SPLITK;
        DenseGemm g = {};
        g.A = f; g.lda = F; g.B = st->W_dec; g.ldb = D; g.M = N; g.N = D; g.K = F;
        g.k_chunk = ((F + S - 1) / S + DG_KSLAB - 1) / DG_KSLAB * DG_KSLAB;
        g.out = kpart; g.ldo = D; g.out_zstride = (int64_t)N * D; g.gate = gate;
        if (split) { g.a_max = am(AM_F); g.b_max = am(AM_WDEC); }
        rc = launch_dense_gemm<false, true, DG_EPI_STORE>(g, S, stream);
        if (rc) return rc;
        const float grad_scale = 2.0f / ((float)n_global * (float)sae_loss_width(d, st));
        hipLaunchKernelGGL(dense_finish_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, tc ? st->tc.target : x, (const float*)kpart, S,
                           (int64_t)N * D, tc ? (const float*)st->tc.b_dec_out : (const float*)st->b_dec, (const float*)(wsb + ws.mu),
                           (const float*)(wsb + ws.sd), (const float*)(wsb + ws.norm), out->sae_out, dY, (float*)(wsb + ws.loss_part), N, D,
                           grad_scale, ghost ? (float*)(gwb + gw.err) : (float*)nullptr, skip, gate, (ghost && tc) ? x : (const float*)nullptr,
                           split ? am(AM_DY) : (uint32_t*)nullptr);
        PV_LAUNCH_CHECK("dense_finish_kernel");
        sae_reduce_sum((const float*)(wsb + ws.loss_part), out->scalars, N, 1.0f / ((float)n_global * (float)sae_loss_width(d, st)), 1, -1,
                       stream, gate, 1u);
        if (ghost) {
            // ghost forward: G0 = exp(hidden_pre[:, dead]) @ W_dec[dead]; loss and d loss / d G0 per token; then the part of
            // d loss / d hidden_pre that reaches the dead columns, dHd = (dG0 @ W_dec[dead]^T) * exp(hidden_pre[:, dead])
            float* g0 = (float*)(gwb + gw.g0);
            float* dg0 = (float*)(gwb + gw.dg0);
            float* err = (float*)(gwb + gw.err);
            float* act = (float*)(gwb + gw.dead_act);
            float* wdd = (float*)(gwb + gw.wdd);
            if (nd > 0) {
                hipLaunchKernelGGL(ghost_gather_rows_kernel, dim3(gw.n_pad), dim3(256), 0, stream, (const float*)st->W_dec, ghost->dead_idx,
                                   nd, gw.n_pad, D, wdd);
                DenseGemm gg = {};
                gg.A = act; gg.lda = gw.n_pad; gg.B = wdd; gg.ldb = D; gg.M = N; gg.N = D; gg.K = gw.n_pad; gg.k_chunk = gw.n_pad;
                gg.out = g0; gg.ldo = D;
                rc = launch_dense_gemm<false, true, DG_EPI_STORE>(gg, 1, stream);
                if (rc) return rc;
            } else {
                PV_HIP_CHECK(hipMemsetAsync(g0, 0, (size_t)N * D * 4, stream));
            }
            if (tc) {
                // a transcoder's ghost term rescales by the mse of (INPUT, sae_out), not by the step's mse against the target
                rc = tc_ghost_mse(x, err, N, D, gwb, gw, out->scalars, stream);
                if (rc) return rc;
            }
            const float* colmean = ghost->err_colmean;
            if (!colmean) {
                rc = sae_colsum(err, N, D, (float*)(gwb + gw.colmean), 1.0f / (float)N, (float*)(gwb + gw.colpart), stream);
                if (rc) return rc;
                colmean = (const float*)(gwb + gw.colmean);
            }
            const float inv_cnt = 1.0f / ((float)n_global * (float)D);
            hipLaunchKernelGGL(ghost_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, (const float*)err, (const float*)g0, colmean,
                               ghost->mse_global ? ghost->mse_global : (const float*)(out->scalars + (tc ? 7 : 1)), dg0, (float*)(gwb + gw.part), N, D,
                               inv_cnt);
            PV_LAUNCH_CHECK("ghost_rows_kernel");
            sae_reduce_sum((const float*)(gwb + gw.part), out->scalars, N, inv_cnt, 5, -1, stream);
            if (nd > 0) {
                DenseGemm gb = {};
                gb.A = dg0; gb.lda = D; gb.B = wdd; gb.ldb = D; gb.M = N; gb.N = gw.n_pad; gb.K = D; gb.k_chunk = D;
                gb.out = (float*)(gwb + gw.dhd); gb.ldo = gw.n_pad; gb.mul = act;
                rc = launch_dense_gemm<false, false, DG_EPI_MUL>(gb, 1, stream);
                if (rc) return rc;
            }
        }
        hipLaunchKernelGGL(dense_loss_kernel, dim3(1), dim3(1), 0, stream, out->scalars, ghost ? 1 : 0, gate, 1u);
        // G4: gW_dec = f^T @ dY
        DenseGemm g4 = {};
        g4.A = f; g4.lda = F; g4.B = dY; g4.ldb = D; g4.M = F; g4.N = D; g4.K = N; g4.k_chunk = N;
        g4.out = st->gW_dec; g4.ldo = D; g4.gate = gate;
        if (split) { g4.a_max = am(AM_F); g4.b_max = am(AM_DY); }
        rc = launch_dense_gemm<true, true, DG_EPI_STORE>(g4, 1, stream);
        if (rc) return rc;
        if (nd > 0) {                                                 // gW_dec[dead] += exp(hidden_pre[:, dead])^T @ dG0
            DenseGemm gt = {};
            gt.A = (float*)(gwb + gw.dead_act); gt.lda = gw.n_pad; gt.B = (float*)(gwb + gw.dg0); gt.ldb = D;
            gt.M = gw.n_pad; gt.N = D; gt.K = N; gt.k_chunk = N; gt.out = (float*)(gwb + gw.tmpw); gt.ldo = D;
            rc = launch_dense_gemm<true, true, DG_EPI_STORE>(gt, 1, stream);
            if (rc) return rc;
            hipLaunchKernelGGL(ghost_scatter_add_rows_kernel, dim3(nd), dim3(256), 0, stream, st->gW_dec, ghost->dead_idx, nd, D,
                               (const float*)(gwb + gw.tmpw));
            PV_LAUNCH_CHECK("ghost_scatter_add_rows_kernel");
        }
        // G3: dH = (dY @ W_dec^T + l1 / N) [f > 0], over f
        DenseGemm g3 = {};
        g3.A = dY; g3.lda = D; g3.B = st->W_dec; g3.ldb = D; g3.M = N; g3.N = F; g3.K = D; g3.k_chunk = D;
        g3.out = f; g3.ldo = F; g3.colpart = colpart; g3.add = l1_coefficient / (float)n_global; g3.gate = gate;
        if (split) { g3.a_max = am(AM_DY); g3.b_max = am(AM_WDEC); g3.amax_out = am(AM_DH); }
        g3.act = d.activation; g3.lp = d.lp_norm;
        if (lp_on) g3.lp_tok = (const float*)(wsb + ws.dense_lp_tok);
        if (nd > 0) { g3.dead_slot = ghost->dead_slot; g3.dead_act = (float*)(gwb + gw.dhd); g3.ldd = gw.n_pad; }      // + the ghost term
        rc = sae_plain_relu(d) ? launch_dense_gemm<false, false, DG_EPI_DH>(g3, 1, stream) : launch_dense_gemm<false, false, DG_EPI_DH_T>(g3, 1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(dense_colreduce_kernel, dim3(nb_f), dim3(256), 0, stream, (const float*)colpart, rblk, F, st->gb_enc,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, gate);
        PV_LAUNCH_CHECK("dense_colreduce_kernel");
        // G5: gW_enc^T = dH^T @ sae_in
        DenseGemm g5 = {};
        g5.A = f; g5.lda = F; g5.B = sae_in; g5.ldb = D; g5.M = F; g5.N = D; g5.K = N; g5.k_chunk = N;
        g5.out = st->gW_enc; g5.ldo = D; g5.gate = gate;
        if (split) { g5.a_max = am(AM_DH); g5.b_max = am(AM_X); }
        rc = launch_dense_gemm<true, true, DG_EPI_STORE>(g5, 1, stream);
        if (rc) return rc;
        if (bias_grads) {                                     // (false: pv_sae_relu_step runs them once behind both of its forms)
            rc = tc ? sae_tc_bias_grads(d, st, dY, N, wsb, ws, stream) : sae_gbdec(d, st, dY, N, wsb, ws, stream);
            if (rc) return rc;
            if (tc) {
                rc = sae_tc_skip_backward(d, st, x, dY, N, stream);
                if (rc) return rc;
            }
        }
    }
    return PV_OK;
}

// scalars[4] = l1_loss of the sparse form, scalars[5] = 0, scalars[0] = mse + l1 -- when the step ran sparse (*mode == 0)
__global__ __launch_bounds__(256) void relu_sparse_scalars_kernel(const float* __restrict__ l1part, int n, float scale,
                                                                  float* __restrict__ scalars, const uint32_t* __restrict__ mode) {
    __shared__ float red[4];
    if (*mode != 0u) return;
    float s = 0.f;
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += 256) s += l1part[i];        // (loads of 8 trips in flight, the sum in order)
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l1 = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
        scalars[4] = l1;
        scalars[5] = 0.f;
        scalars[0] = scalars[1] + l1;
    }
}
__global__ void set_u32_kernel(uint32_t* p, uint32_t v) { *p = v; }

// pv_sae_relu_step with the decoder renorm deferred (PV_SAE_RENORM_DECODER on a plan the sparse form covers): the sparse kernels use
// W_dec[j] * inv[j] on the fly; when the step turns out dense (*gate == 1) the five GEMMs read W_dec as it lies, so the rows are
// rewritten here -- the same products, stored -- and inv[j] = 1 tells pv_sae_apply that nothing is left to scale.  A wave per row.
__global__ __launch_bounds__(256) void relu_dense_renorm_kernel(float* __restrict__ W, float* __restrict__ inv, int rows, int d,
                                                                const uint32_t* __restrict__ gate) {
    if (*gate != 1u) return;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= rows) return;
    const float s = inv[j];
    for (int c = 4 * lane; c < d; c += 256) {
        float4 w = *reinterpret_cast<const float4*>(W + (int64_t)j * d + c);
        w.x *= s; w.y *= s; w.z *= s; w.w *= s;
        *reinterpret_cast<float4*>(W + (int64_t)j * d + c) = w;
    }
    if (lane == 0) inv[j] = 1.0f;
}

// PV_SAE_SPARSE_GRADS on pv_sae_relu_step, behind a step that ran DENSE (gate == NULL: always; else *gate == 1): the gradient buffers
// are complete, so every feature is marked live for pv_sae_apply (offs[j] = j) and the per-feature terms of the clip norm, which the
// sparse backward leaves as a by-product, are taken from the rows here (rowsq[j] = |gW_dec[j]|^2 + |gW_enc^T[j]|^2 + gb_enc[j]^2).
__global__ __launch_bounds__(256) void relu_dense_live_kernel(const float* __restrict__ gW_dec, const float* __restrict__ gW_encT,
                                                              const float* __restrict__ gb_enc, uint32_t* __restrict__ offs,
                                                              float* __restrict__ rowsq, int rows, int d,
                                                              const uint32_t* __restrict__ gate) {
    if (gate && *gate != 1u) return;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= rows) return;
    float sq = 0.f;
    for (int c = 4 * lane; c < d; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(gW_dec + (int64_t)j * d + c);
        const float4 b = *reinterpret_cast<const float4*>(gW_encT + (int64_t)j * d + c);
        sq += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w) + (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
    }
    sq = wave_sum(sq);
    if (lane == 0) {
        const float gb = gb_enc[j];
        rowsq[j] = sq + gb * gb;
        offs[j] = (uint32_t)j;
        if (j == rows - 1) offs[rows] = (uint32_t)rows;
    }
}
}  // namespace

// One train step of the ReLU + L1 SAE on N tokens: forward + backward + statistics; gradients are WRITTEN into st->g*
// (complete buffers: pv_sae_grad_sqnorm and pv_sae_apply follow as usual).  scalars: 0 loss, 1 mse_loss, 2 l0, 4 l1_loss.
extern "C" int pv_sae_dense_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, const float* batch_mean,
                                 int32_t n_global, int32_t flags, float l1_coefficient, const pv_sae_ghost* ghost, pv_sae_out* out,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
    const int update_stats = (flags & PV_SAE_UPDATE_STATS) ? 1 : 0;
    int rc = dense_require(plan, st, x, N, n_global, update_stats, out, workspace);
    if (rc) return rc;
    const pv_sae_desc& d = plan->d;
    // ghost gradients (sae.py:151-179; train_sae.py:337-346): the caller lists the dead features (n_forward_passes_since_fired
    // > dead_feature_window BEFORE this step's statistics) and owns the extra workspace
    GhostWs gw = {};
    const int nd = ghost ? ghost->n_dead : 0;
    if (ghost) {
        PV_REQUIRE(n_global == N || (ghost->err_colmean && ghost->mse_global),
                   "ghost gradients with tokens sharded over ranks need pv_sae_ghost.err_colmean and mse_global (the global batch's)");
        PV_REQUIRE(nd >= 0 && nd <= d.d_sae && ghost->workspace && (nd == 0 || (ghost->dead_idx && ghost->dead_slot)), "pv_sae_ghost");
        gw = ghost_carve(d, N, nd);
        PV_REQUIRE(ghost->workspace_bytes >= gw.total && ((uintptr_t)ghost->workspace & 255) == 0, "ghost workspace too small / misaligned");
    }
    const SaeWs ws = sae_carve(d);
    PV_REQUIRE(workspace_bytes >= ws.total, "workspace too small");
    PV_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace alignment");
    unsigned char* wsb = (unsigned char*)workspace;
    if (sae_is_tc(st)) {
        // ghost gradients on a transcoder (round 6; transcoder.py:82-86): equal widths (the ghost residual is INPUT - sae_out), one process
        PV_REQUIRE(!ghost || (st->tc.d_in_true <= 0 && st->tc.d_out_true <= 0 && n_global == N),
                   "transcoder + ghost gradients: d_out == d_in and the whole batch in one process");
        const int rq = sae_tc_require(d, st, N);
        if (rq) return rq;
    }
    const float* skip = nullptr;
    rc = dense_prepare(plan, st, x, N, batch_mean, flags, false, wsb, ws, &skip, stream_);
    if (rc) return rc;
    return dense_step_body(plan, st, x, N, n_global, update_stats, l1_coefficient, ghost, gw, out, wsb, ws, skip, nullptr, (hipStream_t)stream_);
}

// The same step, sparse where the batch allows it ("ReLU is top-k with threshold 0 and a variable k", sae_enc.hip: relu_select_kernel):
// ONE product over all features -- the fp16 filter with the threshold -B_n and the exact fp32 re-scoring of its survivors -- gives
// every token's positive activations as a list of at most sp->cap pairs; decode, CSR, the sparse backward and the bias gradients
// then run on the k-sparse kernels of sae.hip with k = cap (the L1 term: sum of the kept values; its gradient l1 / N on every kept
// pair).  A batch some token of which cannot be held (more positives than cap, a candidate slot overflow: the early, dense phase of
// training; the published x64 SAEs with L0 ~ 600-2000) raises the step's device-side mode word, the sparse kernels leave at once
// and the five dense GEMMs run instead -- decided on the GPU, no host round trip, same results either way (exact fp32 values on
// both paths).  sp == NULL or a plan the filter does not cover: the dense step.
extern "C" size_t pv_sae_relu_workspace_bytes(const pv_sae_plan* plan, int32_t n_tokens, int32_t cap) {
    if (!plan || n_tokens < 1 || cap < 4 || cap > PV_SAE_RELU_CAP_MAX || cap % 4) return 0;
    return relu_carve(plan->d, n_tokens, cap).total;
}

extern "C" size_t pv_debug_sae_relu_offset(const pv_sae_plan* plan, int32_t n_tokens, int32_t cap, const char* name) {
    if (!plan || !name || n_tokens < 1 || cap < 4 || cap > PV_SAE_RELU_CAP_MAX || cap % 4) return (size_t)-1;
    const ReluWs w = relu_carve(plan->d, n_tokens, cap);
    if (!strcmp(name, "mode")) return w.mode;
    if (!strcmp(name, "idx")) return w.idx;
    if (!strcmp(name, "val")) return w.val;
    if (!strcmp(name, "tok_cnt")) return w.tok_cnt;
    return (size_t)-1;
}

extern "C" int pv_sae_relu_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, const float* batch_mean,
                                int32_t n_global, int32_t flags, float l1_coefficient, const pv_sae_relu_sparse* sp, pv_sae_out* out,
                                void* workspace, size_t workspace_bytes, void* stream_) {
    const int update_stats = (flags & PV_SAE_UPDATE_STATS) ? 1 : 0;
    int rc = dense_require(plan, st, x, N, n_global, update_stats, out, workspace);
    if (rc) return rc;
    const pv_sae_desc& d = plan->d;
    const SaeWs ws = sae_carve(d);
    PV_REQUIRE(workspace_bytes >= ws.total, "workspace too small");
    PV_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace alignment");
    unsigned char* wsb = (unsigned char*)workspace;
    hipStream_t stream = (hipStream_t)stream_;
    const bool tc = sae_is_tc(st);
    if (tc) {
        const int rq = sae_tc_require(d, st, N);
        if (rq) return rq;
    }
    const bool sparse = sp && pv_sae_relu_sparse_ok(d) && sae_plain_relu(d) && st->W_enc16T && st->enc_colsq;      // (tanh-relu / lp_norm > 1: dense)
    // PV_SAE_SPARSE_GRADS (single-process training): a step that ran sparse leaves the rows of features no token kept unwritten, as
    // pv_sae_step does; a step that ran dense marks every feature live.  Either way pv_sae_grad_sqnorm_step / pv_sae_apply follow.
    const bool sparse_grads = (flags & PV_SAE_SPARSE_GRADS) != 0;
    // PV_SAE_RENORM_DECODER: deferred as in pv_sae_step where the sparse form applies (inverse norms now, the rows rewritten by
    // pv_sae_apply -- or here, in place, when the step turns out dense); otherwise W_dec is renormalised in place first
    const bool defer = sparse && (flags & PV_SAE_RENORM_DECODER) && st->dec_inv_norm;
    ReluWs rw = {};
    unsigned char* rwb = nullptr;
    if (sp) {
        PV_REQUIRE(sp->cap >= 4 && sp->cap <= PV_SAE_RELU_CAP_MAX && sp->cap % 4 == 0, "pv_sae_relu_sparse.cap: a multiple of 4 in [4, 256]");
        rw = relu_carve(d, N, sp->cap);
        PV_REQUIRE(sp->workspace && sp->workspace_bytes >= rw.total && ((uintptr_t)sp->workspace & 255) == 0,
                   "pv_sae_relu_sparse workspace too small / misaligned (pv_sae_relu_workspace_bytes)");
        rwb = (unsigned char*)sp->workspace;
    }
    const float* skip = nullptr;
    rc = dense_prepare(plan, st, x, N, batch_mean, defer ? (flags & ~PV_SAE_RENORM_DECODER) : flags, sparse, wsb, ws, &skip, stream_);
    if (rc) return rc;
    uint32_t* mode = sp ? (uint32_t*)(rwb + rw.mode) : nullptr;
    if (defer) {
        if (!(flags & PV_SAE_INV_NORM_VALID)) {
            rc = sae_dec_inv_norm(d, st, stream);
            if (rc) return rc;
        }
        plan->renorm_pending = true;
    }
    if (sparse) {
        const int cap = sp->cap;
        int32_t* idx = (int32_t*)(rwb + rw.idx);
        float* val = (float*)(rwb + rw.val);
        uint32_t* tok_cnt = (uint32_t*)(rwb + rw.tok_cnt);
        float* l1part = (float*)(rwb + rw.l1part);
        {
            ProfScope prof(PV_PROF_SAE_ENC, stream, 2.0 * N * (double)d.d_in * d.d_sae, ((double)N * d.d_in + (double)d.d_in * d.d_sae) * 2.0);
            rc = sae_encode_relu(d, st, N, cap, idx, val, tok_cnt, l1part, (uint32_t*)(rwb + rw.cand_cnt), rwb + rw.cand,
                                 (uint32_t*)(wsb + ws.cnt), (uint32_t*)(rwb + rw.wpos), mode, (const float*)out->scalars, wsb, ws, stream);
            if (rc) return rc;
        }
        SaeTail tb;
        tb.dh = (float*)(rwb + rw.dh); tb.chunk_start = (uint32_t*)(rwb + rw.cursor); tb.wpos = (uint32_t*)(rwb + rw.wpos);
        tb.seg_range = (uint32_t*)(rwb + rw.seg_range); tb.seg_rows = (float*)(rwb + rw.seg_rows); tb.seg_b = (float*)(rwb + rw.seg_b);
        tb.pairs = (int32_t*)(rwb + rw.pairs); tb.max_segs = rw.max_segs;
        rc = sae_sparse_tail(plan, st, x, N, n_global, cap, idx, val, out->sae_out, out->scalars, out->fire_count, update_stats,
                             sparse_grads, defer ? (const float*)st->dec_inv_norm : (const float*)nullptr, tb, wsb, ws,
                             tc ? (const float*)st->tc.target : x, tc ? (const float*)st->tc.b_dec_out : (const float*)st->b_dec, skip, tc,
                             l1_coefficient / (float)n_global, tok_cnt, mode, stream, /*bias_grads*/ false);
        if (rc) return rc;
        hipLaunchKernelGGL(relu_sparse_scalars_kernel, dim3(1), dim3(256), 0, stream, (const float*)l1part, N,
                           l1_coefficient / (float)n_global, out->scalars, (const uint32_t*)mode);
        PV_LAUNCH_CHECK("relu_sparse_scalars_kernel");
    } else if (mode) {
        hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, stream, mode, 1u);
    }
    if (defer) {
        hipLaunchKernelGGL(relu_dense_renorm_kernel, dim3((d.d_sae + 3) / 4), dim3(256), 0, stream, st->W_dec, st->dec_inv_norm, d.d_sae,
                           d.d_in, (const uint32_t*)mode);
        PV_LAUNCH_CHECK("relu_dense_renorm_kernel");
    }
    GhostWs gw = {};
    rc = dense_step_body(plan, st, x, N, n_global, update_stats, l1_coefficient, nullptr, gw, out, wsb, ws, skip, mode, stream,
                         /*bias_grads*/ false);
    if (rc) return rc;
    // gb_dec = colsum(dY) - W_enc gb_enc (transcoder: its two bias gradients, gW_skip): from the dY and gb_enc of whichever form ran
    const float* dY = (const float*)(wsb + ws.dY);
    rc = tc ? sae_tc_bias_grads(d, st, dY, N, wsb, ws, stream) : sae_gbdec(d, st, dY, N, wsb, ws, stream);
    if (rc) return rc;
    if (tc) {
        rc = sae_tc_skip_backward(d, st, x, dY, N, stream);
        if (rc) return rc;
    }
    if (sparse_grads) {
        uint32_t* offs = (uint32_t*)(wsb + ws.offs);
        hipLaunchKernelGGL(relu_dense_live_kernel, dim3((d.d_sae + 3) / 4), dim3(256), 0, stream, (const float*)st->gW_dec,
                           (const float*)st->gW_enc, (const float*)st->gb_enc, offs, (float*)(wsb + ws.rowsq), d.d_sae, d.d_in,
                           sparse ? (const uint32_t*)mode : (const uint32_t*)nullptr);
        PV_LAUNCH_CHECK("relu_dense_live_kernel");
        plan->live_offs = offs;
    }
    return PV_OK;
}

// Ghost gradients on a TOP-K SAE (use_ghost_grads with activation_fn_str = "topk": SparseAutoencoder._compute_ghost_residual_loss,
// sae.py:151-179, behind TopK :795-810; the dead mask of train_sae.py:330-332).  Runs AFTER pv_sae_step on the same batch (complete
// gradient buffers: no PV_SAE_SPARSE_GRADS; the decoder renormalised in place beforehand: pv_sae_renorm_decoder, no
// PV_SAE_RENORM_DECODER) and adds what the ghost term contributes: exp(hidden_pre) of the dead features is ONE small GEMM
// (sae_in against the gathered rows of W_encT, exp in its epilogue -- the k-sparse encoder never materialises hidden_pre), the
// ghost reconstruction, its loss (scalars[5]; scalars[0] = mse + ghost) and gradient as in pv_sae_dense_step, and the gradient
// reaches the dead features' rows of gW_dec, gW_enc^T, gb_enc (row scatter-adds) and gb_dec (recomputed).  The per-feature
// norm terms of the step are stale afterwards: follow with pv_sae_grad_sqnorm over the whole flat buffer, then pv_sae_apply.
// out->sae_out must be the reconstruction pv_sae_step wrote; ghost as for pv_sae_dense_step.  Single process, no transcoder.
extern "C" int pv_sae_topk_ghost(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, const pv_sae_ghost* ghost, pv_sae_out* out,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
    PV_REQUIRE(plan && st && x && ghost && out && workspace, "null argument");
    PV_REQUIRE(out->sae_out && out->scalars, "pv_sae_out: sae_out (as pv_sae_step wrote it) and scalars are required");
    PV_REQUIRE(st->W_dec && st->W_encT && st->b_enc && st->gW_enc && st->gW_dec && st->gb_enc && st->gb_dec, "state");
    PV_REQUIRE(!sae_is_gated(st), "top-k ghost gradients: not for a gated SAE");
    const bool tc = sae_is_tc(st);
    // a Transcoder (round 6): x = the INPUT activation -- the ghost residual is input - sae_out (transcoder.py:82-86); equal widths, one process
    PV_REQUIRE(!tc || (st->tc.d_in_true <= 0 && st->tc.d_out_true <= 0 && ghost->n_global <= N && !ghost->mse_global),
               "transcoder + ghost gradients: d_out == d_in and the whole batch in one process");
    PV_REQUIRE(!plan->renorm_pending, "top-k ghost gradients need the decoder renormalised IN PLACE before the step (pv_sae_renorm_decoder), "
                                      "not deferred (PV_SAE_RENORM_DECODER)");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(N >= 1 && N <= d.max_tokens, "n_tokens exceeds plan max_tokens");
    PV_REQUIRE(d.d_in % 8 == 0, "d_in must be a multiple of 8");
    const int nd = ghost->n_dead, D = d.d_in;
    PV_REQUIRE(nd >= 0 && nd <= d.d_sae && ghost->workspace && (nd == 0 || (ghost->dead_idx && ghost->dead_slot)), "pv_sae_ghost");
    const GhostWs gw = ghost_carve(d, N, nd);
    PV_REQUIRE(ghost->workspace_bytes >= gw.total && ((uintptr_t)ghost->workspace & 255) == 0, "ghost workspace too small / misaligned");
    const SaeWs ws = sae_carve(d);
    PV_REQUIRE(workspace_bytes >= ws.total && ((uintptr_t)workspace & 255) == 0, "workspace too small / misaligned");
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* wsb = (unsigned char*)workspace;
    unsigned char* gwb = (unsigned char*)ghost->workspace;
    const float* sae_in = (const float*)(wsb + ws.sae_in);
    const float* dY = (const float*)(wsb + ws.dY);
    float* err = (float*)(gwb + gw.err);
    float* g0 = (float*)(gwb + gw.g0);
    float* dg0 = (float*)(gwb + gw.dg0);
    float* act = (float*)(gwb + gw.dead_act);
    float* wdd = (float*)(gwb + gw.wdd);
    float* tmpw = (float*)(gwb + gw.tmpw);
    float* dhd = (float*)(gwb + gw.dhd);
    const int P = gw.n_pad;
    int rc = PV_OK;
    hipLaunchKernelGGL(ghost_err_kernel, dim3((unsigned)(((int64_t)N * D / 4 + 255) / 256)), dim3(256), 0, stream,
                       (const float*)out->sae_out, x, err, (int64_t)N * D / 4);
    if (nd > 0) {
        // E = exp(sae_in @ W_enc[:, dead] + b_enc[dead]): the rows of W_encT of the dead features, gathered (tmpw holds them until it is
        // needed for the gradients); padding columns get a bias of -1e30 (E = 0 there)
        hipLaunchKernelGGL(ghost_gather_rows_kernel, dim3(P), dim3(256), 0, stream, (const float*)st->W_encT, ghost->dead_idx, nd, P, D, tmpw);
        hipLaunchKernelGGL(ghost_gather_vec_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, (const float*)st->b_enc, ghost->dead_idx, nd, P,
                           -1e30f, (float*)(gwb + gw.vec_a));
        DenseGemm ge = {};
        ge.A = sae_in; ge.lda = D; ge.B = tmpw; ge.ldb = D; ge.M = N; ge.N = P; ge.K = D; ge.k_chunk = D;
        ge.out = act; ge.ldo = P; ge.bias = (const float*)(gwb + gw.vec_a);
        rc = launch_dense_gemm<false, false, DG_EPI_EXPB>(ge, 1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(ghost_gather_rows_kernel, dim3(P), dim3(256), 0, stream, (const float*)st->W_dec, ghost->dead_idx, nd, P, D, wdd);
        DenseGemm gg = {};
        gg.A = act; gg.lda = P; gg.B = wdd; gg.ldb = D; gg.M = N; gg.N = D; gg.K = P; gg.k_chunk = P;
        gg.out = g0; gg.ldo = D;
        rc = launch_dense_gemm<false, true, DG_EPI_STORE>(gg, 1, stream);
        if (rc) return rc;
    } else {
        PV_HIP_CHECK(hipMemsetAsync(g0, 0, (size_t)N * D * 4, stream));
    }
    if (tc) {
        rc = tc_ghost_mse(x, err, N, D, gwb, gw, out->scalars, stream);
        if (rc) return rc;
    }
    const float* colmean = ghost->err_colmean;
    if (!colmean) {
        rc = sae_colsum(err, N, D, (float*)(gwb + gw.colmean), 1.0f / (float)N, (float*)(gwb + gw.colpart), stream);
        if (rc) return rc;
        colmean = (const float*)(gwb + gw.colmean);
    }
    const int ng = ghost->n_global > 0 ? ghost->n_global : N;
    PV_REQUIRE(ng >= N && (ng == N || (ghost->err_colmean && ghost->mse_global)),
               "pv_sae_ghost.n_global > n_tokens needs err_colmean and mse_global (the global batch's)");
    const float inv_cnt = 1.0f / ((float)ng * (float)D);
    hipLaunchKernelGGL(ghost_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, (const float*)err, (const float*)g0, colmean,
                       ghost->mse_global ? ghost->mse_global : (const float*)(out->scalars + (tc ? 7 : 1)), dg0, (float*)(gwb + gw.part), N, D, inv_cnt);
    PV_LAUNCH_CHECK("ghost_rows_kernel");
    sae_reduce_sum((const float*)(gwb + gw.part), out->scalars, N, inv_cnt, 5, -1, stream);
    hipLaunchKernelGGL(topk_ghost_loss_kernel, dim3(1), dim3(1), 0, stream, out->scalars);
    if (nd > 0) {
        // dHd = (dG0 @ W_dec[dead]^T) * E: what reaches hidden_pre of the dead features, for every token
        DenseGemm gb = {};
        gb.A = dg0; gb.lda = D; gb.B = wdd; gb.ldb = D; gb.M = N; gb.N = P; gb.K = D; gb.k_chunk = D;
        gb.out = dhd; gb.ldo = P; gb.mul = act;
        rc = launch_dense_gemm<false, false, DG_EPI_MUL>(gb, 1, stream);
        if (rc) return rc;
        // gW_dec[dead] += E^T @ dG0
        DenseGemm gt = {};
        gt.A = act; gt.lda = P; gt.B = dg0; gt.ldb = D; gt.M = P; gt.N = D; gt.K = N; gt.k_chunk = N; gt.out = tmpw; gt.ldo = D;
        rc = launch_dense_gemm<true, true, DG_EPI_STORE>(gt, 1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(ghost_scatter_add_rows_kernel, dim3(nd), dim3(256), 0, stream, st->gW_dec, ghost->dead_idx, nd, D, (const float*)tmpw);
        // gW_enc^T[dead] += dHd^T @ sae_in
        DenseGemm gu = {};
        gu.A = dhd; gu.lda = P; gu.B = sae_in; gu.ldb = D; gu.M = P; gu.N = D; gu.K = N; gu.k_chunk = N; gu.out = tmpw; gu.ldo = D;
        rc = launch_dense_gemm<true, true, DG_EPI_STORE>(gu, 1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(ghost_scatter_add_rows_kernel, dim3(nd), dim3(256), 0, stream, st->gW_enc, ghost->dead_idx, nd, D, (const float*)tmpw);
        // gb_enc[dead] += colsum(dHd); gb_dec = colsum(dY) - W_enc gb_enc again
        rc = sae_colsum(dhd, N, P, (float*)(gwb + gw.vec_b), 1.0f, (float*)(gwb + gw.colpart_p), stream);
        if (rc) return rc;
        hipLaunchKernelGGL(ghost_scatter_add_vec_kernel, dim3((nd + 255) / 256), dim3(256), 0, stream, st->gb_enc, ghost->dead_idx, nd,
                           (const float*)(gwb + gw.vec_b));
        PV_LAUNCH_CHECK("ghost scatter kernels");
        rc = tc ? sae_tc_bias_grads(d, st, dY, N, wsb, ws, stream) : sae_gbdec(d, st, dY, N, wsb, ws, stream);
        if (rc) return rc;
    }
    plan->live_offs = nullptr;
    return PV_OK;
}

// ------------------------------------------------------------------------------------------------
// Gated SAE step (GatedSparseAutoencoder, sae.py:648-792; pv_sae_gated).  hs [2N][F]: rows [0, N) feature_acts (later dM, then
// dP), rows [N, 2N) relu(gate pre-activation) (later dG); dYs [2N][D]: rows [0, N) dY of the reconstruction, rows [N, 2N)
// d aux / d (reconstruction through the gate).  Stacking the two row sets makes the two decoder products one GEMM, and
// gW_dec = f^T dY + relu(gate)^T dVia one GEMM over K = 2N.
// ------------------------------------------------------------------------------------------------
namespace {
struct GatedWs {
    size_t total, hs, dys, kpart, colpart2, auxpart, vec0, vec1, vec2, tmpd, cspart;
};
GatedWs gated_carve(const pv_sae_desc& d, int N) {
    GatedWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (size_t)pv_align_up((int64_t)bytes, 256); return o; };
    const size_t F = d.d_sae, D = d.d_in, n = N;
    w.hs = take(2 * n * F * 4);
    w.dys = take(2 * n * D * 4);
    w.kpart = take((size_t)PV_SAE_DENSE_SPLITK * 2 * n * D * 4);
    w.colpart2 = take(((n + 63) / 64) * F * 4);
    w.auxpart = take(n * 4);
    w.vec0 = take(F * 4);
    w.vec1 = take(F * 4);
    w.vec2 = take(F * 4);
    w.tmpd = take(D * 4);
    w.cspart = take((n / 16 + 2) * D * 4);
    w.total = off + 256;
    return w;
}

// rows [0, N): decoder output -> LN-out, mse partial, dY (as dense_finish_kernel); rows [N, 2N): reconstruction through the gate
// -> auxiliary loss partial and its gradient (sae.py:786-792: sum_i (via - sae_in)^2, mean over the batch).  One wave per row.
__global__ __launch_bounds__(256) void gated_finish_kernel(const float* __restrict__ x, const float* __restrict__ sae_in,
                                                           const float* __restrict__ kpart, int splits, int64_t zstride,
                                                           const float* __restrict__ b_dec, const float* __restrict__ mu,
                                                           const float* __restrict__ sd, const float* __restrict__ norm,
                                                           float* __restrict__ sae_out, float* __restrict__ dYs,
                                                           float* __restrict__ loss_partial, float* __restrict__ aux_partial, int n_tok,
                                                           int d, float grad_scale, float aux_scale /* 2 / N */,
                                                           const uint32_t* __restrict__ gate, uint32_t* __restrict__ amax = nullptr) {
    // amax (or NULL): partial maxima of |dYs| (DG_AMAX_SLOTS words, see DenseGemm.a_max)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gate && *gate != 1u) return;                           // (gate: as in DenseGemm)
    if (r >= 2 * n_tok) return;
    float gmax = 0.f;
    const bool second = r >= n_tok;
    const int n = second ? r - n_tok : r;
    const float m = mu[n], sdv = sd[n], nf = norm[n];
    float lsum = 0.f;
    for (int c = 4 * lane; c < d; c += 256) {
        float4 a = *reinterpret_cast<const float4*>(kpart + (int64_t)r * d + c);
        for (int z = 1; z < splits; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(kpart + z * zstride + (int64_t)r * d + c);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        const float4 bd = *reinterpret_cast<const float4*>(b_dec + c);
        float4 e, g;
        if (!second) {
            const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)n * d + c);
            float4 o;
            o.x = (a.x + bd.x) * sdv + m; o.y = (a.y + bd.y) * sdv + m; o.z = (a.z + bd.z) * sdv + m; o.w = (a.w + bd.w) * sdv + m;
            e.x = o.x - xv.x; e.y = o.y - xv.y; e.z = o.z - xv.z; e.w = o.w - xv.w;
            if (sae_out) *reinterpret_cast<float4*>(sae_out + (int64_t)n * d + c) = o;
            lsum += (e.x * e.x) / nf + (e.y * e.y) / nf + (e.z * e.z) / nf + (e.w * e.w) / nf;
            g.x = grad_scale * e.x / nf * sdv; g.y = grad_scale * e.y / nf * sdv;
            g.z = grad_scale * e.z / nf * sdv; g.w = grad_scale * e.w / nf * sdv;
        } else {
            const float4 sv = *reinterpret_cast<const float4*>(sae_in + (int64_t)n * d + c);
            e.x = a.x + bd.x - sv.x; e.y = a.y + bd.y - sv.y; e.z = a.z + bd.z - sv.z; e.w = a.w + bd.w - sv.w;
            lsum += e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w;
            g.x = aux_scale * e.x; g.y = aux_scale * e.y; g.z = aux_scale * e.z; g.w = aux_scale * e.w;
        }
        *reinterpret_cast<float4*>(dYs + (int64_t)r * d + c) = g;
        gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(g.x), fabsf(g.y))), fmaxf(fabsf(g.z), fabsf(g.w)));
    }
    lsum = wave_sum(lsum);
    if (lane == 0) (second ? aux_partial : loss_partial)[n] = lsum;
    if (amax) {
        gmax = wave_max(gmax);
        if (lane == 0) atomicMax(amax + (r & (DG_AMAX_SLOTS - 1)), __float_as_uint(gmax));
    }
}

// scalars[0] = mse + l1 + aux (sae.py:748), one thread
__global__ void gated_loss_kernel(float* __restrict__ scalars, const uint32_t* __restrict__ gate) {
    if (gate && *gate != 1u) return;
    scalars[0] = scalars[1] + scalars[4] + scalars[6];
}

// dP = dM e^r + dG, in place over dM (rows [0, N) of hs; dG = rows [N, 2N))
// amax (or NULL): partial maxima of |dP| (DG_AMAX_SLOTS words, see DenseGemm.a_max)
__global__ __launch_bounds__(256) void gated_dp_kernel(float* __restrict__ hs, const float* __restrict__ r_mag, int64_t n_rows, int F,
                                                       const uint32_t* __restrict__ gate, uint32_t* __restrict__ amax) {
    if (gate && *gate != 1u) return;
    float m = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n_rows * F; i += (int64_t)gridDim.x * 1024) {
        const int c = (int)(i % F);
        float4 a = *reinterpret_cast<const float4*>(hs + i);
        const float4 g = *reinterpret_cast<const float4*>(hs + n_rows * F + i);
        const float4 r = *reinterpret_cast<const float4*>(r_mag + c);
        a.x = a.x * expf(r.x) + g.x; a.y = a.y * expf(r.y) + g.y; a.z = a.z * expf(r.z) + g.z; a.w = a.w * expf(r.w) + g.w;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
        *reinterpret_cast<float4*>(hs + i) = a;
    }
    if (amax) {
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) atomicMax(amax + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (DG_AMAX_SLOTS - 1)), __float_as_uint(m));
    }
}

// per feature: gr_mag = sum_n dM (mag_pre - b_mag) = sum_n dM f - b_mag gb_mag (dM != 0 only where f = mag_pre > 0);
// colsum(dP) = e^r gb_mag + gb_gate -> dpsum (the encoder-input term of gb_dec)
__global__ __launch_bounds__(256) void gated_vec_kernel(const float* __restrict__ sdmf, const float* __restrict__ gb_mag,
                                                        const float* __restrict__ gb_gate, const float* __restrict__ b_mag,
                                                        const float* __restrict__ r_mag, float* __restrict__ gr_mag,
                                                        float* __restrict__ dpsum, int F, const uint32_t* __restrict__ gate) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (gate && *gate != 1u) return;
    if (j >= F) return;
    gr_mag[j] = sdmf[j] - b_mag[j] * gb_mag[j];
    dpsum[j] = expf(r_mag[j]) * gb_mag[j] + gb_gate[j];
}

// gW_dec[j] += coef * pgsum[j] * W_dec[j] / ||W_dec[j]||: the decoder-norm factor of the L1 term (sae.py:780-784); one wave per row
__global__ __launch_bounds__(256) void gated_l1_rows_kernel(float* __restrict__ gW_dec, const float* __restrict__ W_dec,
                                                            const float* __restrict__ pgsum, float coef, int F, int d) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= F) return;
    if (pgsum[j] == 0.f) return;                               // (no token opened this gate: a zero term; most rows of a sparse step)
    // (rows of up to 64 * L1_NE elements: the row of W_dec and of gW_dec in registers, all loads in flight -- two loops of one dependent
    // load per trip were 24 round trips to memory per row; same lane mapping, same order)
    constexpr int L1_NE = 20;
    if (d <= 64 * L1_NE) {
        float wv[L1_NE], gv[L1_NE];
#pragma unroll
        for (int u = 0; u < L1_NE; ++u) {
            const int c = lane + 64 * u;
            wv[u] = c < d ? W_dec[(int64_t)j * d + c] : 0.f;
            gv[u] = c < d ? gW_dec[(int64_t)j * d + c] : 0.f;
        }
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < L1_NE; ++u)
            if (lane + 64 * u < d) sq += wv[u] * wv[u];
        sq = wave_sum(sq);
        const float s = coef * pgsum[j] / sqrtf(sq);
#pragma unroll
        for (int u = 0; u < L1_NE; ++u) {
            const int c = lane + 64 * u;
            if (c < d) gW_dec[(int64_t)j * d + c] = gv[u] + s * wv[u];
        }
        return;
    }
    float sq = 0.f;
    for (int c = lane; c < d; c += 64) { const float w = W_dec[(int64_t)j * d + c]; sq += w * w; }
    sq = wave_sum(sq);
    const float s = coef * pgsum[j] / sqrtf(sq);
    for (int c = lane; c < d; c += 64) gW_dec[(int64_t)j * d + c] += s * W_dec[(int64_t)j * d + c];
}

__global__ __launch_bounds__(256) void gated_axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += x[i];
}
}  // namespace

extern "C" size_t pv_sae_gated_scratch_bytes(const pv_sae_plan* plan, int32_t n_tokens) {
    if (!plan || n_tokens < 1) return 0;
    return gated_carve(plan->d, n_tokens).total;
}

// sp != NULL (pv_sae_gated_step_sparse): the step runs sparse where the batch allows it, as pv_sae_relu_step does -- the open gates of
// every token as a list of at most sp->cap pairs (sae_gated_sparse, sae.hip); a batch that cannot be held raises the device-side mode
// word and the dense GEMMs below run instead (every kernel of either form leaves at once in the other's mode; same results either way).
static int gated_step_impl(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, const float* batch_mean, int32_t n_global,
                           int32_t flags, float l1_coefficient, const pv_sae_relu_sparse* sp, pv_sae_out* out, void* workspace,
                           size_t workspace_bytes, void* stream_) {
    const int update_stats = (flags & PV_SAE_UPDATE_STATS) ? 1 : 0;
    PV_REQUIRE(plan && st && x && out && workspace && out->scalars, "null argument");
    PV_REQUIRE(sae_is_gated(st) && !sae_is_tc(st), "pv_sae_gated_step needs a gated state (pv_sae_state.gt) and no transcoder");
    const pv_sae_gated& t = st->gt;
    PV_REQUIRE(t.r_mag && t.b_mag && t.gb_gate && t.gr_mag && t.gb_mag && t.scratch, "gated state");
    PV_REQUIRE(st->W_dec && st->b_dec && st->gW_enc && st->gW_dec && st->gb_enc && st->gb_dec && st->W_encT, "state");
    PV_REQUIRE(!update_stats || (st->act_freq_scores && st->n_fwd_since_fired), "stats buffers");
    PV_REQUIRE(flags & PV_SAE_RENORM_DECODER, "pv_sae_gated_step: PV_SAE_RENORM_DECODER is required (unit decoder rows in the L1 term)");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(N >= 1 && N <= d.max_tokens, "n_tokens exceeds plan max_tokens");
    PV_REQUIRE(n_global >= N, "n_global must be >= n_tokens");
    PV_REQUIRE(d.d_in % 8 == 0 && d.d_sae % 8 == 0, "the gated step needs d_in and d_sae to be multiples of 8");
    const SaeWs ws = sae_carve(d);
    PV_REQUIRE(workspace_bytes >= ws.total && ((uintptr_t)workspace & 255) == 0, "workspace too small / misaligned");
    const GatedWs gw = gated_carve(d, N);
    PV_REQUIRE(t.scratch_bytes >= gw.total && ((uintptr_t)t.scratch & 255) == 0, "gated scratch too small / misaligned");
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* wsb = (unsigned char*)workspace;
    unsigned char* gb = (unsigned char*)t.scratch;
    const bool sparse = sp && pv_sae_relu_sparse_ok(d) && st->W_enc16T && st->enc_colsq;
    GatedSparseWs gs = {};
    unsigned char* rwb = nullptr;
    if (sp) {
        PV_REQUIRE(sp->cap >= 4 && sp->cap <= PV_SAE_RELU_CAP_MAX && sp->cap % 4 == 0, "pv_sae_relu_sparse.cap: a multiple of 4 in [4, 256]");
        gs = gated_sparse_carve(d, N, sp->cap);
        PV_REQUIRE(sp->workspace && sp->workspace_bytes >= gs.total && ((uintptr_t)sp->workspace & 255) == 0,
                   "pv_sae_relu_sparse workspace too small / misaligned (pv_sae_gated_sparse_workspace_bytes)");
        rwb = (unsigned char*)sp->workspace;
    }
    const uint32_t* mode = sp ? (const uint32_t*)(rwb + gs.rw.mode) : nullptr;
    plan->live_offs = nullptr;
    plan->renorm_pending = false;
    const int F = d.d_sae, D = d.d_in;
    int rc = pv_sae_renorm_decoder(plan, st, stream_);                   // train_sae.py:307
    if (rc) return rc;
    rc = sae_prep(d, x, (const float*)st->b_dec, batch_mean, N, sparse, wsb, ws, stream);
    const float ng = (float)n_global;
    if (rc) return rc;
    float* hs = (float*)(gb + gw.hs);
    float* dYs = (float*)(gb + gw.dys);
    float* kpart = (float*)(gb + gw.kpart);
    float* sae_in = (float*)(wsb + ws.sae_in);
    float* colpart = (float*)(wsb + ws.dense_colpart);
    float* colpart2 = (float*)(gb + gw.colpart2);
    float* rowpart = (float*)(wsb + ws.dense_rowpart);
    float* pgsum = (float*)(gb + gw.vec0);
    float* sdmf = (float*)(gb + gw.vec1);
    const int rblk = (N + 63) / 64, cblk = (F + 63) / 64, nb_f = (F + 255) / 256;
    float* blk_tot = (float*)(wsb + ws.sqpart);
    // operand maxima of the split-fp16 GEMMs (see dense_gemm_kernel): sae_in, W_enc^T, hs = [f; relu(gate)], W_dec, dYs = [dY; dVia],
    // dM, dG, dP
    uint32_t* amax = (uint32_t*)(wsb + ws.dense_amax);
    enum { AM_X = 0, AM_WENC = 1, AM_HS = 2, AM_WDEC = 3, AM_DY = 4, AM_DM = 5, AM_DG = 6, AM_DP = 7 };
    auto am = [&](int t) { return amax + t * DG_AMAX_SLOTS; };
    const bool split = !g_pv_tuning.dense_fp32 && (D % 4) == 0 && (F % 4) == 0 && (N % 4) == 0;
    if (sparse) {
        rc = sae_gated_sparse(plan, st, x, N, n_global, sp->cap, l1_coefficient, update_stats, rwb, gs, out, wsb, ws, dYs,
                              (float*)(gb + gw.auxpart), pgsum, stream);
        if (rc) return rc;
    } else if (mode) {
        hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, stream, (uint32_t*)(rwb + gs.rw.mode), 1u);
    }
    if (split) {
        PV_HIP_CHECK(hipMemsetAsync(amax, 0, (size_t)PV_SAE_AMAX_TENSORS * DG_AMAX_SLOTS * 4, stream));
        dense_absmax3(sae_in, (int64_t)N * D, AM_X, (const float*)st->W_encT, (int64_t)F * D, AM_WENC, (const float*)st->W_dec, (int64_t)F * D, AM_WDEC,
                      amax, mode, stream);
        PV_LAUNCH_CHECK("dense_absmax3_kernel");
    }
    {
        ProfScope prof(PV_PROF_SAE_ENC, stream, 2.0 * N * (double)D * F, ((double)N * D + (double)D * F + 2.0 * N * F) * 4.0);
        // G1: p = sae_in @ W_enc once; both paths in the epilogue
        DenseGemm g = {};
        g.A = sae_in; g.lda = D; g.B = st->W_encT; g.ldb = D; g.M = N; g.N = F; g.K = D; g.k_chunk = D;
        g.out = hs; g.ldo = F; g.out2 = hs + (size_t)N * F; g.bias = t.b_gate; g.bias2 = t.b_mag; g.cscale = t.r_mag;
        g.colpart = colpart; g.colpart2 = colpart2; g.rowpart = rowpart; g.gate = mode;
        if (split) { g.a_max = am(AM_X); g.b_max = am(AM_WENC); g.amax_out = am(AM_HS); }
        rc = launch_dense_gemm<false, false, DG_EPI_GENC>(g, 1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(dense_colreduce_kernel, dim3(nb_f), dim3(256), 0, stream, (const float*)colpart, rblk, F,
                           out->fire_count ? out->fire_count : (float*)(wsb + ws.rowsq), (float*)nullptr, st->act_freq_scores,
                           st->n_fwd_since_fired, update_stats, blk_tot, mode);
        hipLaunchKernelGGL(dense_colreduce_kernel, dim3(nb_f), dim3(256), 0, stream, (const float*)colpart2, rblk, F, pgsum,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, mode);
        PV_LAUNCH_CHECK("dense_colreduce_kernel");
        sae_reduce_sum(blk_tot, out->scalars, nb_f, 1.0f / (float)N, 2, -1, stream, mode, 1u);                        // l0
        sae_reduce_sum(rowpart, out->scalars, rblk * cblk, l1_coefficient / ng, 4, -1, stream, mode, 1u);             // l1 (unit decoder rows)
    }
    {
        ProfScope prof(PV_PROF_SAE_BWD, stream, 14.0 * N * (double)D * F, 0.0);
        // G2: [f; relu(gate)] @ W_dec, split over K
        const int S = PV_SAE_DENSE_SPLITK;
        DenseGemm g = {};
        g.A = hs; g.lda = F; g.B = st->W_dec; g.ldb = D; g.M = 2 * N; g.N = D; g.K = F;
        g.k_chunk = ((F + S - 1) / S + DG_KSLAB - 1) / DG_KSLAB * DG_KSLAB;
        g.out = kpart; g.ldo = D; g.out_zstride = (int64_t)2 * N * D; g.gate = mode;
        if (split) { g.a_max = am(AM_HS); g.b_max = am(AM_WDEC); }
        rc = launch_dense_gemm<false, true, DG_EPI_STORE>(g, S, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(gated_finish_kernel, dim3((2 * N + 3) / 4), dim3(256), 0, stream, x, (const float*)sae_in, (const float*)kpart, S,
                           (int64_t)2 * N * D, (const float*)st->b_dec, (const float*)(wsb + ws.mu), (const float*)(wsb + ws.sd),
                           (const float*)(wsb + ws.norm), out->sae_out, dYs, (float*)(wsb + ws.loss_part), (float*)(gb + gw.auxpart), N, D,
                           2.0f / (ng * (float)D), 2.0f / ng, mode, split ? am(AM_DY) : (uint32_t*)nullptr);
        PV_LAUNCH_CHECK("gated_finish_kernel");
        sae_reduce_sum((const float*)(wsb + ws.loss_part), out->scalars, N, 1.0f / (ng * (float)D), 1, -1, stream, mode, 1u);
        sae_reduce_sum((const float*)(gb + gw.auxpart), out->scalars, N, 1.0f / ng, 6, -1, stream, mode, 1u);
        hipLaunchKernelGGL(gated_loss_kernel, dim3(1), dim3(1), 0, stream, out->scalars, mode);
        // G4: gW_dec = [f; relu(gate)]^T @ [dY; dVia] (K = 2N); the decoder-norm factor of the L1 term is added behind both forms
        DenseGemm g4 = {};
        g4.A = hs; g4.lda = F; g4.B = dYs; g4.ldb = D; g4.M = F; g4.N = D; g4.K = 2 * N; g4.k_chunk = 2 * N;
        g4.out = st->gW_dec; g4.ldo = D; g4.gate = mode;
        if (split) { g4.a_max = am(AM_HS); g4.b_max = am(AM_DY); }
        rc = launch_dense_gemm<true, true, DG_EPI_STORE>(g4, 1, stream);
        if (rc) return rc;
        // G3a: dM = (dY @ W_dec^T) [f > 0] over f; column sums = gb_mag, of dM * f = the raw term of gr_mag
        DenseGemm g3 = {};
        g3.A = dYs; g3.lda = D; g3.B = st->W_dec; g3.ldb = D; g3.M = N; g3.N = F; g3.K = D; g3.k_chunk = D;
        g3.out = hs; g3.ldo = F; g3.colpart = colpart; g3.colpart2 = colpart2; g3.add = 0.f; g3.gate = mode;
        if (split) { g3.a_max = am(AM_DY); g3.b_max = am(AM_WDEC); g3.amax_out = am(AM_DM); }
        rc = launch_dense_gemm<false, false, DG_EPI_DH>(g3, 1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(dense_colreduce_kernel, dim3(nb_f), dim3(256), 0, stream, (const float*)colpart, rblk, F, t.gb_mag,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, mode);
        hipLaunchKernelGGL(dense_colreduce_kernel, dim3(nb_f), dim3(256), 0, stream, (const float*)colpart2, rblk, F, sdmf,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, mode);
        // G3b: dG = (dVia @ W_dec^T + l1 / N) [gate > 0] over relu(gate); column sums = gb_gate
        g3.A = dYs + (size_t)N * D; g3.out = hs + (size_t)N * F; g3.colpart2 = nullptr; g3.add = l1_coefficient / ng;
        if (split) g3.amax_out = am(AM_DG);
        rc = launch_dense_gemm<false, false, DG_EPI_DH>(g3, 1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(dense_colreduce_kernel, dim3(nb_f), dim3(256), 0, stream, (const float*)colpart, rblk, F, t.gb_gate,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, mode);
        // gr_mag; colsum(dP) parked in gb_enc for the bias-gradient kernels below
        hipLaunchKernelGGL(gated_vec_kernel, dim3(nb_f), dim3(256), 0, stream, (const float*)sdmf, (const float*)t.gb_mag,
                           (const float*)t.gb_gate, (const float*)t.b_mag, (const float*)t.r_mag, t.gr_mag, st->gb_enc, F, mode);
        // dP = dM e^r + dG, then G5: gW_enc^T = dP^T @ sae_in
        // (a bounded grid walking the rows: in the sparse form's mode the launch is empty, and ~100k empty workgroups are 20 us)
        hipLaunchKernelGGL(gated_dp_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)N * F / 4 + 255) / 256, 8192)), dim3(256), 0, stream, hs,
                           (const float*)t.r_mag, (int64_t)N, F, mode, split ? am(AM_DP) : (uint32_t*)nullptr);
        PV_LAUNCH_CHECK("gated elementwise kernels");
        DenseGemm g5 = {};
        g5.A = hs; g5.lda = F; g5.B = sae_in; g5.ldb = D; g5.M = F; g5.N = D; g5.K = N; g5.k_chunk = N;
        g5.out = st->gW_enc; g5.ldo = D; g5.gate = mode;
        if (split) { g5.a_max = am(AM_DP); g5.b_max = am(AM_X); }
        rc = launch_dense_gemm<true, true, DG_EPI_STORE>(g5, 1, stream);
        if (rc) return rc;
    }
    // behind either form: dYs = [dY; dVia], pgsum = colsum(relu(gate_pre)), colsum(dP) in gb_enc, gW_dec without the L1 term
    hipLaunchKernelGGL(gated_l1_rows_kernel, dim3((F + 3) / 4), dim3(256), 0, stream, st->gW_dec, (const float*)st->W_dec,
                       (const float*)pgsum, l1_coefficient / ng, F, D);
    // gb_dec = colsum(dY) + 2 colsum(dVia) - W_enc colsum(dP): b_dec sits in the decoder twice and in sae_in, which is also
    // the auxiliary loss's target
    rc = sae_gbdec(d, st, dYs, N, wsb, ws, stream);                   // colsum(dY) - W_enc gb_enc (= colsum(dP) for now)
    if (rc) return rc;
    rc = sae_colsum(dYs + (size_t)N * D, N, D, (float*)(gb + gw.tmpd), 2.0f, (float*)(gb + gw.cspart), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(gated_axpy_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, st->gb_dec, (const float*)(gb + gw.tmpd), D);
    PV_LAUNCH_CHECK("gated_axpy_kernel");
    PV_HIP_CHECK(hipMemsetAsync(st->gb_enc, 0, (size_t)F * 4, stream));      // b_enc takes no part in a gated SAE
    return PV_OK;
}

extern "C" int pv_sae_gated_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, const float* batch_mean, int32_t n_global,
                                 int32_t flags, float l1_coefficient, pv_sae_out* out, void* workspace, size_t workspace_bytes,
                                 void* stream_) {
    return gated_step_impl(plan, st, x, N, batch_mean, n_global, flags, l1_coefficient, nullptr, out, workspace, workspace_bytes, stream_);
}

extern "C" size_t pv_sae_gated_sparse_workspace_bytes(const pv_sae_plan* plan, int32_t n_tokens, int32_t cap) {
    if (!plan || n_tokens < 1 || cap < 4 || cap > PV_SAE_RELU_CAP_MAX || cap % 4) return 0;
    return gated_sparse_carve(plan->d, n_tokens, cap).total;
}

extern "C" int pv_sae_gated_step_sparse(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, const float* batch_mean,
                                        int32_t n_global, int32_t flags, float l1_coefficient, const pv_sae_relu_sparse* sp,
                                        pv_sae_out* out, void* workspace, size_t workspace_bytes, void* stream_) {
    PV_REQUIRE(sp, "pv_sae_gated_step_sparse: pv_sae_relu_sparse is required (pv_sae_gated_step is the dense form)");
    return gated_step_impl(plan, st, x, N, batch_mean, n_global, flags, l1_coefficient, sp, out, workspace, workspace_bytes, stream_);
}
