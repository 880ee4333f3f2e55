// Shared device/host helpers for the gfx950 kernels.  gfx950 only: wave = 64 lanes, MFMA
// 32x32x16 bf16 / 32x32x2 f32, 160 KiB LDS per CU.  No CUDA/HIP dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/pv_native.h"

typedef uint16_t bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define PV_WAVE 64

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
void pv_set_error(const std::string& msg);
#define PV_HIP_CHECK(expr)                                                                \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            pv_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));              \
            return PV_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)
#define PV_LAUNCH_CHECK(name)                                                             \
    do {                                                                                  \
        hipError_t _e = hipGetLastError();                                                \
        if (_e != hipSuccess) {                                                           \
            pv_set_error(std::string("launch ") + name + ": " + hipGetErrorString(_e));   \
            return PV_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)
#define PV_REQUIRE(cond, msg)                                                             \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            pv_set_error(std::string("invalid: ") + msg + " [" #cond "]");                \
            return PV_ERR_INVALID;                                                        \
        }                                                                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------
// scalar conversions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// float -> bf16, round-to-nearest-even with NaN preserved (== torch's conversion): a plain cast lets hipcc
// emit the gfx950 hardware instruction v_cvt_pk_bf16_f32 (1 VALU per 2 values instead of ~6 per value)
typedef __bf16 pv_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pv_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const pv_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pv_bf16x2));
}

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int kBytes = 4;
    static constexpr int kPerChunk = 4;   // elements per 16-byte chunk
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct DT<bf16_t> {
    static constexpr int kBytes = 2;
    static constexpr int kPerChunk = 8;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    __device__ static __forceinline__ float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// load / store 8 consecutive elements (as floats); pointer must be 16-byte aligned for the vector
// forms (callers guarantee it, scalar fallbacks otherwise)
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    uint4 a;
    a.x = pack_bf16x2(v[0], v[1]); a.y = pack_bf16x2(v[2], v[3]);
    a.z = pack_bf16x2(v[4], v[5]); a.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = a;
}

// 16 bytes to a TAP-ONLY destination (a cache entry nothing on the device reads back: the pre-activation / attn_out / mlp_out taps
// of the GEMM epilogues, LayerNorm's fp32 hook_normalized, the attention scores / pattern): a nontemporal store, so that the tap
// stream does not push the next kernel's operands (the tensor written beside it) out of the L2 / MALL.  Measured on the B/32
// forward: MLP-2 131 -> 115 us with its A operand (mlp.hook_post) no longer evicted by the mlp.hook_pre tap (profiles/r04_notes.md).
// V: one of the vector types below (the destination's alignment).  -DPV_NO_NT (A/B builds): plain stores.
typedef uint32_t pv_u32x4_a16 __attribute__((ext_vector_type(4)));
typedef uint32_t pv_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t pv_u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
template <typename V>
__device__ __forceinline__ void pv_store16_stream(void* p, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
#ifdef PV_NO_NT
    *reinterpret_cast<V*>(p) = V{x, y, z, w};
#else
    __builtin_nontemporal_store(V{x, y, z, w}, reinterpret_cast<V*>(p));
#endif
}
__device__ __forceinline__ void store8_stream(float* p, const float (&v)[8]) {
    pv_store16_stream<pv_u32x4_a16>(p, __float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    pv_store16_stream<pv_u32x4_a16>(p + 4, __float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
}

// ---------------------------------------------------------------------------------------------
// wave-level reductions (64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): 1 v_exp + 1 v_rcp + 6 FMA instead of ocml's
// erff (~35 VALU); used where the result is rounded to bf16 anyway (2^-9 relative)
__device__ __forceinline__ float pv_erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}

// activation functions of models/layers/mlp.py:41-63 that the fast path supports
template <bool FAST = false>
__device__ __forceinline__ float pv_act(float x, int act) {
    if (act == PV_ACT_GELU) {
        if constexpr (FAST) return 0.5f * x * (1.0f + pv_erf_fast(x * 0.70710678118654752440f));
        else return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    }
    if (act == PV_ACT_QUICK_GELU) return x / (1.0f + __expf(-1.702f * x));
    return fmaxf(x, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// kernel-choice overrides for tests / A-B measurements (pv_debug_set_tuning in pv_native.h).  The launch path reads
// these process-global ints -- never the environment; every field's default is "let the library decide", and
// bench.py records pv_debug_get_tuning("any") in its JSON line and refuses to run when it is non-zero.
// ---------------------------------------------------------------------------------------------
struct PvTuning {
    int gemm_tile = -1;      // -1 auto; 0 = 128 x 128 kernel (v4); 4 / 5 = one-workgroup-per-CU kernel with a 256 / 320 x 256 tile
    int gemm_v1 = 0;         // 1: register-staged 128 x 128 kernel for everything
    int gemm_v1patch = 0;    // 1: register-staged kernel for the patch embedding only
    int attn_wg = 0;         // 1: workgroup-per-(image, head, query block) attention kernel for every shape
    int attn_direct = 0;     // 1: the one-wave-per-head attention kernel loads its Q / K fragments straight from global (no whole-row staging)
    int prof_markers = 0;    // 1: time v7 launches with hipEventRecord markers instead of dispatch-packet events
    int sae_exact = 0;       // 1: SAE encoder on the exact-fp32 MFMA GEMM + streaming top-k (the small-shape / fallback path)
    int sae_fold = 1;        // 0: pv_sae_step launches its small kernels one by one (the A/B of the fused pre-pass, SaePre, the scan as a role of
                             // the decode launch, ScanRole, the merged list sorts and the Adam pair's tail roles)
    int dense_group = -1;    // tile order of the dense SAE GEMMs (sae_dense.hip): -1 auto (wide outputs: 4 row tiles per group), 0 = N fastest
                             // everywhere (the order up to round 6's PMC pass), n = n row tiles per group
    int sae_inline_fb = 0;   // 1: the folded SAE step recomputes a token the filter cannot decide inside the select kernel (no fallback launches:
                             // - 14 us per step) -- off: ONE such token costs the step 0.3 ms of latency (its workgroup walks all features
                             // alone), and harvested activations have a few per step (MEASURED.md); the two fallback launches spread a token
                             // over 24 workgroups
    int enc_tm256 = 0;       // 1: the SAE sample pass keeps 256-row tiles where 128-row ones would fill more of the chip (A/B of sae_enc_gemm_kernel's MB_)
    int enc_rounds = 0;      // 1: the SAE filter GEMM compacts its hits in a round per 32-row block whatever the shape (A/B of the one-round epilogue)
    int gemm_dbg = 0;        // K-loop / epilogue ablations; honoured only by -DPV_TUNING builds
    int gemm_loop = -1;      // K loop of the one-workgroup-per-CU kernels (ViT GEMMs, SAE filter GEMM): -1 auto = the full-line form (128-byte
                             // slabs, two slots) where K is a whole number of 128-byte slabs, else the pipelined 64-byte-slab form,
                             // else (patch gather, K tails) the barrier-then-fetch loop; 0 = barrier-then-fetch everywhere;
                             // 1 = pipelined 64-byte slabs where legal; 2 = as auto
    int gemm_persist = -1;   // persistent form of the full-line kernel (one workgroup per CU walks its tiles, the K slabs of consecutive tiles one
                             // stream): -1 auto = where the launch has more tiles than CUs, 0 = never, 1 = wherever it is legal
    int gemm_stagger = 0;    // A/B knob of the persistent form: start delay step between workgroups (units of 127 x 64 clocks), 0 = none
    int dense_fp32 = 0;      // 1: the dense SAE steps (ReLU + L1, gated) on the exact fp32 matrix instruction (round 5's form) instead of the
                             // split-fp16 one (three fp16 products per fp32 product, fp32 accumulation): the A/B of sae_dense.hip
    int gemm_cus = 0;        // CUs a GEMM launch may count on (grid of the persistent form, rounds of the tile model): 0 = the device's; set by
                             // probes that run forwards on CU-masked streams (tools/cu_mask_forward_probe.py); a plan with a pipeline passes its own
};
extern PvTuning g_pv_tuning;

static inline int64_t pv_align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline bool pv_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
