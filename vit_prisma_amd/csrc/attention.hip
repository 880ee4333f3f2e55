// Attention core for one (image, head, query block) per workgroup on gfx950:
//     scores  = q k^T / attn_scale          -> hook_attn_scores tap   (attention.py:246-265)
//     pattern = softmax(scores), NaN -> 0   -> hook_pattern tap       (attention.py:148-150)
//     z       = pattern v                   -> hook_z                 (attention.py:267-281)
// Because attn_scores and pattern are OUTPUTS of run_with_cache, the full T x T matrices are
// materialised (no flash-style fusion is allowed); each is written exactly
// once, row-contiguous, straight from the LDS copy that also feeds the PV MFMAs.
//
//   phase 1  QK^T on MFMA 32x32 tiles, operands loaded as 16-byte fragments straight from global
//            (q/k rows of one head are 64..256-byte segments, no cross-wave reuse -> no LDS staging),
//            scaled scores (rounded to the storage dtype, like the reference's bf16 tensor) -> LDS
//            fp32 [QB][Tpad] (row stride Tpad*4+16 bytes: conflict-free b128 fragment reads later)
//   phase 2  one wave per row: max / exp / sum / normalise in registers; stores both taps with
//            coalesced row writes; writes P back to LDS in the storage dtype (bf16 in place)
//   phase 3  z = P V on MFMA; bf16: V^T blocks staged through LDS so the B fragment is k-contiguous,
//            fp32: V rows read directly (the f32 MFMA takes one float per lane)
#include "attention.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "prof.hpp"

namespace {

constexpr int VT_ROW = 272;     // bytes: 128 keys * 2 + 16 pad
constexpr int KBLK = 128;       // keys per V^T block (bf16)

template <typename T, int QB, int DH, int MAXC>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int EB = DT<T>::kBytes;
    constexpr int NJ = DH * EB / 32;       // 16-byte fragment pairs along d_head
    constexpr int NTQ = QB / 32;
    constexpr int NTN = DH / 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int T_ = p.T, H = p.H;
    const int q0 = qb * QB;
    const int Tpad = p.Tpad;
    // S element: fp32 in the whole-head (FLAT) variant, the storage dtype otherwise -- the scores are rounded to
    // it anyway before the softmax (like the reference's bf16 tensor), and at T = 577 a bf16 S block is 41 KB
    // instead of 82 KB: two workgroups per CU instead of one
    constexpr int SB = QB == 64 ? 4 : EB;
    const int srow = Tpad * SB + 16;                // bytes
    // FLAT (QB == 64: the whole T x T matrix of this head lives in the workgroup): the fp32 scores stay
    // intact in S, P goes to its own region, and both taps are stored afterwards as ONE flat contiguous
    // range per (image, head) (256 B per wave-instruction instead of one <= 100-byte row per instruction)
    constexpr bool FLAT = QB == 64;
    const int prow = FLAT ? Tpad * EB + 16 : srow;  // bytes, row stride of P
    unsigned char* S = smem;                        // [QB][srow]
    unsigned char* Pb = FLAT ? smem + QB * srow : smem;           // [QB][prow] (in place over S otherwise)
    unsigned char* Vt = Pb + (FLAT ? QB * prow : QB * srow);      // bf16 only: [DH][VT_ROW]

    const int64_t tok_stride = (int64_t)H * DH;     // elements between tokens of one head
    const T* qbase = reinterpret_cast<const T*>(p.q) + ((int64_t)b * T_ * H + h) * DH;
    const T* kbase = reinterpret_cast<const T*>(p.k) + ((int64_t)b * T_ * H + h) * DH;
    const T* vbase = reinterpret_cast<const T*>(p.v) + ((int64_t)b * T_ * H + h) * DH;

    // V^T staging (bf16): item -> (key pair kp, 8-wide d_head chunk ch); lanes 0-15 take 16 key pairs of one
    // chunk (conflict-free ds_write_b32 rows), the next 16 lanes the next chunk
    auto stage_v = [&](int kb0, int nkeys) {
        constexpr int NCH = DH / 8;
        for (int item = tid; item < (nkeys / 2) * NCH; item += 256) {
            const int kp = (item & 15) | ((item / (16 * NCH)) << 4);
            const int ch = (item >> 4) % NCH;
            const int k0 = kb0 + 2 * kp, k1 = k0 + 1;
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
            if (k0 < T_) v0 = *reinterpret_cast<const uint4*>(vbase + k0 * tok_stride + ch * 8);
            if (k1 < T_) v1 = *reinterpret_cast<const uint4*>(vbase + k1 * tok_stride + ch * 8);
            const uint32_t a0[4] = {v0.x, v0.y, v0.z, v0.w};
            const uint32_t a1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t lo = (a0[e] & 0xffffu) | (a1[e] << 16);            // element 2e
                const uint32_t hi = (a0[e] >> 16) | (a1[e] & 0xffff0000u);        // element 2e+1
                *reinterpret_cast<uint32_t*>(Vt + (ch * 8 + 2 * e) * VT_ROW + kp * 4) = lo;
                *reinterpret_cast<uint32_t*>(Vt + (ch * 8 + 2 * e + 1) * VT_ROW + kp * 4) = hi;
            }
        }
    };
    // whole-head variant: V^T has its own LDS region, so its global loads are issued FIRST and their latency
    // overlaps the Q / K fragment loads and the QK^T phase (one exposed HBM latency instead of two)
    if constexpr (FLAT && EB == 2) stage_v(0, Tpad);

    // ------------------------------------------------------------------ phase 1: scores
    const int ntk = Tpad / 32;
    for (int id = wave; id < NTQ * ntk; id += 4) {
        const int tq = id % NTQ, tk = id / NTQ;
        const int qi = q0 + tq * 32 + l31;
        const int ki = tk * 32 + l31;
        uint4 qa[NJ], kb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            qa[j] = (qi < T_) ? *reinterpret_cast<const uint4*>(
                                    reinterpret_cast<const unsigned char*>(qbase + qi * tok_stride) + (2 * j + half) * 16)
                              : make_uint4(0, 0, 0, 0);
            kb[j] = (ki < T_) ? *reinterpret_cast<const uint4*>(
                                    reinterpret_cast<const unsigned char*>(kbase + ki * tok_stride) + (2 * j + half) * 16)
                              : make_uint4(0, 0, 0, 0);
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (EB == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[j]),
                                                              __builtin_bit_cast(bf16x8, kb[j]), acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(qa[j].x), __uint_as_float(kb[j].x), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(qa[j].y), __uint_as_float(kb[j].y), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(qa[j].z), __uint_as_float(kb[j].z), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(qa[j].w), __uint_as_float(kb[j].w), acc, 0, 0, 0);
            }
        }
        // C layout: col = lane & 31 (key), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (query)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = tq * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            const float s = DT<T>::round(acc[e] / p.attn_scale);
            if constexpr (SB == 4) *reinterpret_cast<float*>(S + row * srow + (tk * 32 + l31) * 4) = s;
            else DT<T>::store(reinterpret_cast<T*>(S + row * srow) + tk * 32 + l31, s);
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 2: softmax rows
    for (int r = wave; r < QB; r += 4) {
        const int qi = q0 + r;
        if (qi >= T_) continue;                      // pad rows: never stored, never normalised
        const unsigned char* srowp = S + r * srow;
        float v[MAXC];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = lane + 64 * c;
            if constexpr (SB == 4) v[c] = (col < Tpad) ? reinterpret_cast<const float*>(srowp)[col] : 0.f;
            else v[c] = (col < Tpad) ? DT<T>::load(reinterpret_cast<const T*>(srowp) + col) : 0.f;
            if (col < T_) m = fmaxf(m, v[c]);
        }
        m = wave_max(m);
        float sum = 0.f;
        float e_[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = lane + 64 * c;
            e_[c] = (col < T_) ? expf(v[c] - m) : 0.f;
            sum += e_[c];
        }
        sum = wave_sum(sum);
        const int64_t grow = (((int64_t)b * H + h) * T_ + qi) * T_;
        T* sc_out = (!FLAT && p.scores) ? reinterpret_cast<T*>(p.scores) + grow : nullptr;
        T* pt_out = (!FLAT && p.pattern) ? reinterpret_cast<T*>(p.pattern) + grow : nullptr;
        unsigned char* prowp = Pb + r * prow;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = lane + 64 * c;
            if (col < Tpad) {
                float pr = e_[c] / sum;
                if (pr != pr) pr = 0.f;              // attention.py:149
                pr = DT<T>::round(pr);               // attention.py:152 pattern.to(cfg.dtype)
                if (col < T_) {
                    if (sc_out) DT<T>::store(sc_out + col, v[c]);
                    if (pt_out) DT<T>::store(pt_out + col, pr);
                }
                // P for the PV product, in the storage dtype, in place (all of this row's fp32
                // scores were read into registers above; the row belongs to this wave only)
                if constexpr (EB == 2) reinterpret_cast<bf16_t*>(prowp)[col] = f32_to_bf16(pr);
                else reinterpret_cast<float*>(prowp)[col] = pr;
            }
        }
    }

    if constexpr (FLAT) {
        if (p.scores || p.pattern) {
            __syncthreads();
            const int TT = T_ * T_;
            const int64_t gbase = ((int64_t)b * H + h) * TT;
            T* sc = p.scores ? reinterpret_cast<T*>(p.scores) + gbase : nullptr;
            T* pt = p.pattern ? reinterpret_cast<T*>(p.pattern) + gbase : nullptr;
            if (EB == 2 && (T_ & 1) == 0) {
                // pairs of bf16: T even -> a pair never straddles rows and every head starts 4-byte aligned
                for (int f2 = tid; f2 < (TT >> 1); f2 += 256) {
                    const int f = f2 * 2, i = f / T_, j = f - i * T_;
                    if (sc) {
                        const float* sr = reinterpret_cast<const float*>(S + i * srow) + j;
                        reinterpret_cast<uint32_t*>(sc)[f2] = pack_bf16x2(sr[0], sr[1]);
                    }
                    if (pt) reinterpret_cast<uint32_t*>(pt)[f2] = *reinterpret_cast<const uint32_t*>(Pb + i * prow + j * 2);
                }
            } else {
                for (int f = tid; f < TT; f += 256) {
                    const int i = f / T_, j = f - i * T_;
                    if (sc) DT<T>::store(sc + f, reinterpret_cast<const float*>(S + i * srow)[j]);
                    if (pt) pt[f] = reinterpret_cast<const T*>(Pb + i * prow)[j];
                }
            }
        }
    }

    // ------------------------------------------------------------------ phase 3: z = P V
    f32x16 zacc[(NTQ * NTN + 3) / 4];
#pragma unroll
    for (int t = 0; t < (NTQ * NTN + 3) / 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) zacc[t][e] = 0.f;

    if constexpr (EB == 2) {
        for (int kb0 = 0; kb0 < Tpad; kb0 += KBLK) {
            __syncthreads();   // P rows visible (first block) / previous V^T block fully consumed
            if constexpr (!FLAT) {
                stage_v(kb0, KBLK);
                __syncthreads();
            }
            const int ksteps = min(KBLK, Tpad - kb0) / 16;
            int t = 0;
            for (int id = wave; id < NTQ * NTN; id += 4, ++t) {
                const int tq = id % NTQ, tn = id / NTQ;
                const unsigned char* pfrag = Pb + (tq * 32 + l31) * prow + (kb0 + half * 8) * 2;
                const unsigned char* vrow = Vt + (tn * 32 + l31) * VT_ROW + half * 16;
                for (int ks = 0; ks < ksteps; ++ks) {
                    const uint4 a = *reinterpret_cast<const uint4*>(pfrag + ks * 32);
                    const uint4 bb = *reinterpret_cast<const uint4*>(vrow + ks * 32);
                    zacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                                      __builtin_bit_cast(bf16x8, bb), zacc[t], 0, 0, 0);
                }
            }
        }
    } else {
        __syncthreads();       // P rows visible to every wave
        int t = 0;
        for (int id = wave; id < NTQ * NTN; id += 4, ++t) {
            const int tq = id % NTQ, tn = id / NTQ;
            const float* pfrag = reinterpret_cast<const float*>(Pb + (tq * 32 + l31) * prow);
            const T* vcol = vbase + tn * 32 + l31;
            for (int kb0 = 0; kb0 < Tpad; kb0 += 8) {
                const int kk = kb0 + half * 4;
                const float4 a = *reinterpret_cast<const float4*>(pfrag + kk);
                float bv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (kk + e < T_) ? DT<T>::load(vcol + (kk + e) * tok_stride) : 0.f;
                zacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv[0], zacc[t], 0, 0, 0);
                zacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv[1], zacc[t], 0, 0, 0);
                zacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv[2], zacc[t], 0, 0, 0);
                zacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv[3], zacc[t], 0, 0, 0);
            }
        }
    }

    // ------------------------------------------------------------------ store z [B, T, H, dh]
    {
        T* zbase = reinterpret_cast<T*>(p.z) + ((int64_t)b * T_ * H + h) * DH;
        int t = 0;
        for (int id = wave; id < NTQ * NTN; id += 4, ++t) {
            const int tq = id % NTQ, tn = id / NTQ;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qi = q0 + tq * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (qi < T_) DT<T>::store(zbase + qi * tok_stride + tn * 32 + l31, zacc[t][e]);
            }
        }
    }
}

template <typename T, int QB, int DH, int MAXC>
int launch_attn(const AttnParams& p, hipStream_t stream) {
    constexpr int EB = DT<T>::kBytes;
    const int srow = p.Tpad * (QB == 64 ? 4 : EB) + 16;
    const int lds = QB * srow + (QB == 64 ? QB * (p.Tpad * EB + 16) : 0) + (EB == 2 ? DH * VT_ROW : 0);
    PV_REQUIRE(lds <= 160 * 1024, "attention LDS footprint exceeds 160 KiB");
    static int max_set = 0;
    if (lds > max_set) {
        PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<T, QB, DH, MAXC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        max_set = lds;
    }
    const dim3 grid((p.T + QB - 1) / QB, p.H, p.B), block(256);
    {
        const double bh = (double)p.B * p.H, tt = (double)p.T * p.T;
        const double bytes = (4.0 * bh * p.T * DH + ((p.scores ? 1.0 : 0.0) + (p.pattern ? 1.0 : 0.0)) * bh * tt) * EB;
        ProfScope prof(PV_PROF_ATTN, stream, 4.0 * bh * tt * DH, bytes);
        hipLaunchKernelGGL((attn_kernel<T, QB, DH, MAXC>), grid, block, lds, stream, p);
    }
    PV_LAUNCH_CHECK("attn_kernel");
    return PV_OK;
}

// ---------------------------------------------------------------------------------------------------
// Whole-head-per-WAVE variant (bf16, T <= 64, T even): the B/32 shape (T = 50, 6144 heads at bs = 512).
//   The workgroup kernel above spends its time in barriers between four waves that share one 50 x 50
//   problem (112 us per layer at bs = 512 against ~45 us of HBM time).  Here a wave owns one (image, head):
//   no workgroup barrier anywhere, 9 KB of LDS per wave (16 waves per CU), every phase ordered by the wave's
//   own in-order LDS queue:
//     1  K, Q fragments straight from global (buffer loads clipped to the head's T rows: pad rows read 0)
//     2  S = Q K^T on MFMA, scaled + rounded to bf16 -> LDS [64][72] (144-byte rows)
//     3  hook_attn_scores: the head's T*T bf16 block leaves as flat 16-byte chunks (it is contiguous in HBM)
//     4  softmax with lane r owning row r (8 ds_read_b128, all math in registers), P written back in place
//     5  hook_pattern: flat copy as in 3
//     6  z = P V on MFMA: P fragments from LDS, V fragments as 2-byte buffer loads (a lane needs 8 keys of
//        ONE d_head column: k-strided in HBM, but 32 lanes cover 64 contiguous bytes of each key row)
//     7  z staged through the same LDS rows -> 16-byte row stores
// ---------------------------------------------------------------------------------------------------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int AW_ROW = 144;                 // bytes per LDS row: 64 bf16 + 16 pad (conflict-free b128 rows)

template <int DH, bool STAGE = true>
__global__ __launch_bounds__(256) void attn_wave_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][64 * AW_ROW];
    constexpr int NKS = DH / 16;            // k16 steps of Q K^T
    constexpr int NTN = DH / 32;            // 32-wide d_head tiles of P V
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x * 4 + wave;    // (image, head)
    if (g >= p.B * p.H) return;             // the whole wave leaves; nothing below synchronises across waves
    const int T_ = p.T, H = p.H;
    const int b = g / H, h = g - b * H;
    const int half = lane >> 5, l31 = lane & 31;
    unsigned char* SP = smem[wave];
    const unsigned tokb = (unsigned)H * DH * 2u;                          // bytes between tokens of one head
    const int64_t head_off = ((int64_t)b * T_ * H + h) * DH;              // elements
    const int span = (int)((unsigned)(T_ - 1) * tokb + DH * 2u);          // this head's rows; beyond -> 0
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.q) + head_off), 0, span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.k) + head_off), 0, span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.v) + head_off), 0, span, 0x00020000);

    // ---- 1, 2: scores
    // STAGE: K and Q arrive as WHOLE rows (a lane takes one 16-byte chunk of a row; 64 / CPR rows = whole cache lines per
    // instruction), are parked in the wave's LDS rows and read back as MFMA fragments.  The direct form asks for the
    // fragment layout straight from global: 32 rows x 32 bytes per instruction, i.e. every 128-byte head row is looked
    // up by four instructions -- the texture addresser's line lookups, not the bytes, are what that costs.
    constexpr int CPR = DH / 8, RPI = 64 / CPR, NLD = 64 / RPI;       // chunks per row, rows per instruction, instructions
    const int s_row = lane / CPR, s_ch = lane % CPR;
    u32x4_t kf[2][NKS];
    u32x4_t qrow[STAGE ? NLD : 1];
    if constexpr (STAGE) {
        u32x4_t krow[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) krow[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, (unsigned)(i * RPI + s_row) * tokb + s_ch * 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) qrow[i] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (unsigned)(i * RPI + s_row) * tokb + s_ch * 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4_t*>(SP + (i * RPI + s_row) * AW_ROW + s_ch * 16) = krow[i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                kf[ki][ks] = *reinterpret_cast<const u32x4_t*>(SP + (ki * 32 + l31) * AW_ROW + (2 * ks + half) * 16);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4_t*>(SP + (i * RPI + s_row) * AW_ROW + s_ch * 16) = qrow[i];
        __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                kf[ki][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsK, (unsigned)(ki * 32 + l31) * tokb + (2 * ks + half) * 16, 0, 0);
    }
    // attn_scale = sqrt(d_head) is a power of two for d_head 64 (and 16, 256): x * (1 / scale) is then x / scale exactly
    const float inv_scale = 1.0f / p.attn_scale;
    const bool scale_pow2 = (__float_as_uint(p.attn_scale) & 0x007fffffu) == 0u && p.attn_scale > 0.f;
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
        // (STAGE: the score rows tq * 32 .. + 31 written below land on the Q rows this iteration has just read; the Q rows of
        // the other iteration sit in the other half of the LDS block)
        u32x4_t qf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if constexpr (STAGE) qf[ks] = *reinterpret_cast<const u32x4_t*>(SP + (tq * 32 + l31) * AW_ROW + (2 * ks + half) * 16);
            else qf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (unsigned)(tq * 32 + l31) * tokb + (2 * ks + half) * 16, 0, 0);
        }
        f32x16 acc[2];
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ki][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int ki = 0; ki < 2; ++ki)        // S^T = K Q^T: a lane ends up with 4 x 4 consecutive keys of ONE query
                acc[ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ki][ks]),
                                                                  __builtin_bit_cast(bf16x8, qf[ks]), acc[ki], 0, 0, 0);
        // C layout: col = lane & 31 (query), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (key): four consecutive keys ->
        // one 8-byte LDS write into the query's row (64 two-byte writes per lane in the Q K^T orientation)
        unsigned char* srow = SP + (tq * 32 + l31) * AW_ROW;
        auto put_scores = [&](auto exact_recip) {
#pragma unroll
            for (int ki = 0; ki < 2; ++ki)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (decltype(exact_recip)::value) v[i] = acc[ki][4 * j + i] * inv_scale;
                        else v[i] = acc[ki][4 * j + i] / p.attn_scale;
                    }
                    *reinterpret_cast<uint2*>(srow + (ki * 32 + 8 * j + 4 * half) * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
        };
        if (scale_pow2) put_scores(std::true_type{});       // (wave-uniform branch: the divisions are not executed)
        else put_scores(std::false_type{});
    }
    __builtin_amdgcn_wave_barrier();

    // flat copy of the head's [T][T] bf16 block out of the LDS rows: chunk c = elements 8c .. 8c+7 (T even: a pair
    // of elements never straddles a row); the last chunk may be short
    const int TT = T_ * T_;
    const float inv_T = 1.0f / (float)T_;
    auto flat_store = [&](void* dst_base) {
        unsigned char* dst = reinterpret_cast<unsigned char*>(dst_base) + (int64_t)g * TT * 2;
        for (int c = lane; c * 8 < TT; c += 64) {
            uint32_t w[4];
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const int f = c * 8 + q2 * 2;
                const int i = (int)(((float)f + 0.5f) * inv_T);
                const int j = f - i * T_;
                w[q2] = (f < TT) ? *reinterpret_cast<const uint32_t*>(SP + i * AW_ROW + j * 2) : 0u;
            }
            unsigned char* d = dst + (int64_t)c * 16;
            if (c * 8 + 8 <= TT) {
                pv_store16_stream<pv_u32x4_a4>(d, w[0], w[1], w[2], w[3]);              // 4-byte aligned is enough; tap-only
            } else {
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2)
                    if (c * 8 + q2 * 2 < TT) *reinterpret_cast<uint32_t*>(d + q2 * 4) = w[q2];
            }
        }
    };
    if (p.scores) flat_store(p.scores);
    __builtin_amdgcn_wave_barrier();

    // ---- 4: softmax, lane r = query row r (attention.py:148-152: softmax, NaN -> 0, cast to the model dtype)
    {
        uint4 raw[8];
        unsigned char* rowp = SP + lane * AW_ROW;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) raw[c8] = *reinterpret_cast<const uint4*>(rowp + c8 * 16);
        float e_[64];
        float m = -INFINITY;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const uint32_t w[4] = {raw[c8].x, raw[c8].y, raw[c8].z, raw[c8].w};
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const int j = c8 * 8 + q2 * 2;
                e_[j] = __uint_as_float(w[q2] << 16);
                e_[j + 1] = __uint_as_float(w[q2] & 0xffff0000u);
                if (j < T_) m = fmaxf(m, e_[j]);
                if (j + 1 < T_) m = fmaxf(m, e_[j + 1]);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            e_[j] = (j < T_) ? __expf(e_[j] - m) : 0.f;
            sum += e_[j];
        }
        const float rs = 1.0f / sum;
        // NaN -> 0 (attention.py:149): a softmax row is NaN throughout or nowhere (exp(s - max) <= 1; a NaN / +inf score or an
        // all -inf row makes the SUM NaN), so the where() is one test per row
        const bool keep = lane < T_ && rs == rs;                          // pad rows: finite zeros for the MFMA
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            uint32_t w[4];
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const float p0 = e_[c8 * 8 + q2 * 2] * rs, p1 = e_[c8 * 8 + q2 * 2 + 1] * rs;
                w[q2] = keep ? pack_bf16x2(p0, p1) : 0u;
            }
            *reinterpret_cast<uint4*>(rowp + c8 * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (p.pattern) flat_store(p.pattern);

    // ---- 6: z = P V
    f32x16 zacc[2][NTN];
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
            for (int e = 0; e < 16; ++e) zacc[tq][tn][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks * 16 >= T_) break;                                        // keys beyond T: P and V are zero there
        u32x4_t vf[NTN];
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn) {
            uint32_t w[4];
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const unsigned key = ks * 16 + half * 8 + q2 * 2;
                const unsigned o = key * tokb + (tn * 32 + l31) * 2;
                const uint32_t lo = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsV, o, 0, 0);
                const uint32_t hi = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsV, o + tokb, 0, 0);
                w[q2] = lo | (hi << 16);
            }
            vf[tn] = u32x4_t{w[0], w[1], w[2], w[3]};
        }
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const uint4 pa = *reinterpret_cast<const uint4*>(SP + (tq * 32 + l31) * AW_ROW + ks * 32 + half * 16);
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn)
                zacc[tq][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[tn]),      // z^T = V^T P^T
                                                                       __builtin_bit_cast(bf16x8, pa), zacc[tq][tn], 0, 0, 0);
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- 7: z -> LDS rows [64][DH] -> 16-byte row stores into [B, T, H, dh]
    // (C layout of z^T: col = lane & 31 = query, rows = four consecutive d per group -> 8-byte writes into the query's row)
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<uint2*>(SP + (tq * 32 + l31) * AW_ROW + (tn * 32 + 8 * j + 4 * half) * 2) =
                    make_uint2(pack_bf16x2(zacc[tq][tn][4 * j], zacc[tq][tn][4 * j + 1]),
                               pack_bf16x2(zacc[tq][tn][4 * j + 2], zacc[tq][tn][4 * j + 3]));
    __builtin_amdgcn_wave_barrier();
    {
        bf16_t* zb = reinterpret_cast<bf16_t*>(p.z) + head_off;
        for (int c = lane; c < T_ * CPR; c += 64) {
            const int row = c / CPR, ch = c - row * CPR;
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(zb) + (int64_t)row * tokb + ch * 16) =
                *reinterpret_cast<const uint4*>(SP + row * AW_ROW + ch * 16);
        }
    }
}

template <int DH>
int launch_attn_wave(const AttnParams& p, hipStream_t stream) {
    const int heads = p.B * p.H;
    {
        const double bh = (double)heads, tt = (double)p.T * p.T;
        const double bytes = (4.0 * bh * p.T * DH + ((p.scores ? 1.0 : 0.0) + (p.pattern ? 1.0 : 0.0)) * bh * tt) * 2.0;
        ProfScope prof(PV_PROF_ATTN, stream, 4.0 * bh * tt * DH, bytes);
        if (g_pv_tuning.attn_direct) hipLaunchKernelGGL((attn_wave_kernel<DH, false>), dim3((heads + 3) / 4), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((attn_wave_kernel<DH, true>), dim3((heads + 3) / 4), dim3(256), 0, stream, p);
    }
    PV_LAUNCH_CHECK("attn_wave_kernel");
    return PV_OK;
}

// ---------------------------------------------------------------------------------------------------
// Long-sequence variant (bf16, T > 64, d_head 64: L/14@336 has T = 577): a WAVE owns 32 query rows of one (image, head)
// and streams the keys twice, with the product SWAPPED (S^T = K Q^T) so that a lane holds 16 keys of ONE query:
//   pass 1  per 32-key tile: S^T tile on MFMA, scaled + rounded to bf16 like the reference's score tensor, online
//           (max, sum) per lane; the two lanes of a query (halves of the wave) are merged at the end
//           (+ hook_attn_scores, when tapped, leaves through the same window as the pattern)
//   pass 2  the same tiles again: p = exp(s - max) / sum, rounded -> hook_pattern, and z += P V on MFMA
// The four waves of a workgroup (128 consecutive queries of one head) share each K / V tile: fetched cooperatively one
// tile ahead into registers, parked in double-buffered LDS (one barrier per tile).  No [QB][T] score block in LDS.
// The kernel is bound by VALU issue and by the tap's HBM write stream, not by MFMA (profiles/r02_notes.md): round 1's
// form spent about 640 wave instructions per 32 x 32 tile over the two passes; this one about 300:
//   * 1 / attn_scale folded into the Q fragments when it is a power of two (exact in bf16; d_head 64 -> 1/8)
//   * score rounding = v_cvt_pk_bf16_f32 + shift / mask; max as v_max3; exp(s - m) = v_exp(fma(s, log2 e, -m log2 e)) on
//     packed f32 pairs; pass 2 folds 1 / sum into the exponent (p = exp2(s log2 e - (m log2 e + log2 sum)))
//   * NaN rows (attention.py:149): a softmax row is NaN entirely or not at all, so the where() is one AND with a row mask,
//     taken only by waves that hold such a row
//   * P: C layout -> A operand with v_permlane32_swap (4 per tile); V staged TRANSPOSED once per workgroup ([d][key]) so a
//     B fragment is one ds_read_b128 (round 1: 32 two-byte reads per tile)
//   * taps: four 32 x 32 tiles collect in a per-wave LDS window and leave as 4 rows x 256 contiguous bytes per store
//     instruction, 16-byte global stores at the row's own 2-byte alignment (the [T][T] block of a head is only 2-byte
//     aligned when T is odd; gfx950 global memory takes unaligned dwordx4 stores) -- no funnel shifts, no parity branch
//   * workgroups dealt to the XCDs by whole heads (the query blocks of a head share one L2's copy of its K / V)
// A single-pass variant (the wave's 32 x T score strip kept in LDS, QK^T computed once) was built and measured: 37 KB per
// wave = one workgroup per CU = one wave per SIMD, 1.9 ms per layer against 0.66 ms here without taps -- removed.  So was a
// register-strip variant (round 3: 16 queries per wave on 16 x 16 MFMA tiles, exp(s - max) of the whole strip in 160 VGPRs, one
// exponential per score, P handed to the P V product without leaving its registers; parity green): 1.22 ms without taps, 1.48 ms
// with the pattern tap against 0.63 / 0.83 here -- per score it issues MORE instructions, because a wave of 16 queries pays the
// same tile staging, fragment reads and MFMA issue as a wave of 32 (profiles/r03_notes.md 11).  Round 4: a workgroup per 32 queries with
// the whole [32][T] strip in LDS and the key tiles split over its four waves (QK^T and the exponential once per score, the pattern tap
// as ONE flat stream per workgroup; parity green): 1.36 ms without taps, 1.61 ms with the pattern tap -- 62 KB of LDS = 8 waves per CU,
// four serial phases between workgroup barriers, no tile sharing: latency-bound (profiles/r04_l14_attention_strip_in_lds_ab.json).
// ---------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(2))) U4a2 { uint32_t x, y, z, w; };      // a 16-byte store at 2-byte alignment
constexpr int L2_VROW = 80;                 // V^T tile row: 32 keys x 2 B + 16 pad
constexpr float PV_LOG2E = 1.4426950408889634f;

template <int DH, bool PRESCALE>
__global__ __launch_bounds__(256) void attn_lean_kernel(const AttnParams p) {
    static_assert(DH == 64, "d_head 64");
    __shared__ __attribute__((aligned(16))) unsigned char Kst[2][32 * 144];          // key tile [32][d_head] (+16 B pad), double-buffered
    __shared__ __attribute__((aligned(16))) unsigned char Vt[2][DH * L2_VROW];       // value tile transposed [d][32 keys]
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][32 * 256];         // per wave: tap window of 4 tiles (8 KB) / z staging (4.5 KB)
    constexpr int NKS = DH / 16;
    constexpr int NTN = DH / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T_ = p.T, H = p.H;
    const int qblocks = (T_ + 127) / 128;
    // workgroups are dealt to the 8 XCDs round-robin: give every XCD whole heads, so the query blocks of a head share one
    // L2's copy of its K / V (plain order: each of the 5 blocks at T = 577 pulls them through a different L2 -- PMC: 1.2 GB
    // fetched per launch for 0.45 GB of q, k, v)
    int bid = blockIdx.x;
    {
        const int heads = p.B * H, per_xcd = heads / 8;
        if (bid < per_xcd * 8 * qblocks) {
            const int xcd = bid & 7, i = bid >> 3;
            bid = ((i / qblocks) * 8 + xcd) * qblocks + i % qblocks;
        }
    }
    const int g = bid / qblocks;                            // (image, head)
    const int q0 = (bid - g * qblocks) * 128 + wave * 32;
    const bool active = q0 < T_;                            // idle waves of a head's last block still stage tiles and meet the barriers
    const int b = g / H, h = g - b * H;
    const int half = lane >> 5, l31 = lane & 31;
    unsigned char* L = smem[wave];
    const unsigned tokb = (unsigned)H * DH * 2u;
    const int64_t head_off = ((int64_t)b * T_ * H + h) * DH;
    const int span = (int)((unsigned)(T_ - 1) * tokb + DH * 2u);
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.q) + head_off), 0, span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.k) + head_off), 0, span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.v) + head_off), 0, span, 0x00020000);
    const int ntile = (T_ + 31) / 32;
    const float inv_scale = 1.0f / p.attn_scale;
    const f32x2_t inv_scale2 = {inv_scale, inv_scale};
    const f32x2_t log2e2 = {PV_LOG2E, PV_LOG2E};

    // Q as the B operand (columns = this wave's queries): lane (query l31, half) holds d-chunk (2 ks + half)
    u32x4_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        qf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (unsigned)(q0 + l31) * tokb + (2 * ks + half) * 16, 0, 0);
        if (PRESCALE) {
            uint32_t w[4] = {qf[ks].x, qf[ks].y, qf[ks].z, qf[ks].w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[i] = pack_bf16x2(__uint_as_float(w[i] << 16) * inv_scale, __uint_as_float(w[i] & 0xffff0000u) * inv_scale);
            qf[ks] = u32x4_t{w[0], w[1], w[2], w[3]};
        }
    }

    // cooperative tile fetch: thread t moves 16 B (8 d-values, chunk t / 32) of key row t % 32; rows >= T read 0.  A half-wave
    // = the 32 keys of one chunk: its transposed two-byte V^T writes fall into 64 contiguous bytes per d row (key-major
    // lanes would put eight d rows -- 8 x 80 B apart, the same banks -- into every write: measured 8-way conflicts)
    const int t_key = threadIdx.x & 31, t_dc = threadIdx.x >> 5;
    const unsigned tile_off = (unsigned)t_key * tokb + t_dc * 16;
    const int k_lds = t_key * 144 + t_dc * 16;
    auto fetch = [&](const __amdgpu_buffer_rsrc_t& rs, int kt) -> u32x4_t {
        if (kt < ntile) return __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)kt * 32u * tokb + tile_off, 0, 0);
        return u32x4_t{0, 0, 0, 0};
    };
    auto stage_v = [&](unsigned char* dst, const u32x4_t& v) {          // 8 d-values of key t_key -> V^T[d][key]
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<unsigned short*>(dst + (t_dc * 8 + 2 * i) * L2_VROW + t_key * 2) = (unsigned short)(w[i] & 0xffffu);
            *reinterpret_cast<unsigned short*>(dst + (t_dc * 8 + 2 * i + 1) * L2_VROW + t_key * 2) = (unsigned short)(w[i] >> 16);
        }
    };

    // S^T tile out of Kst[buf]: pk[i] = bf16 scores of keys kt*32 + key_of(2 i), key_of(2 i + 1); s[e] the same as floats;
    // key_of(e) = (e & 3) + 8 * (e >> 2) + 4 * half
    auto score_tile = [&](const unsigned char* kb, uint32_t (&pk)[8], float (&sc)[16]) {
        uint4 kf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(kb + l31 * 144 + (2 * ks + half) * 16);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), __builtin_bit_cast(bf16x8, qf[ks]), acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x2_t t = {acc[2 * i], acc[2 * i + 1]};
            if (!PRESCALE) t = t * inv_scale2;
            pk[i] = pack_bf16x2(t.x, t.y);
            sc[2 * i] = __uint_as_float(pk[i] << 16);
            sc[2 * i + 1] = __uint_as_float(pk[i] & 0xffff0000u);
        }
    };

    // P / S tile from the C layout (this lane: keys 4 half + 8 j + 0..3, j = 0..3, as pk[2j], pk[2j+1]) to the A-operand layout
    // (keys ks*16 + half*8 + 0..7 = 16 contiguous bytes of the query's row): the upper half-wave's group 2 ks trades places
    // with the lower half-wave's group 2 ks + 1
    auto to_rows = [&](const uint32_t (&pk)[8], u32x4_t (&pa)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 0], pk[4 * ks + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 1], pk[4 * ks + 3], false, false);
            pa[ks] = u32x4_t{s0[0], s1[0], s0[1], s1[1]};
        }
    };
    // Taps: the tile goes into the wave's LDS window [32 rows][4 tiles x 64 B] (16-byte chunk c of row r at position
    // c ^ (r & 15): the rows of one ds_write_b128 cover every bank); every fourth tile (and after the last) the window leaves
    // as 16-byte stores, 4 rows x 256 contiguous bytes per instruction, at the row's own 2-byte alignment.  (One tile per
    // flush = 64-byte pieces: every piece straddles two 64-byte blocks of HBM and the L2 writes most of them out twice --
    // PMC: 2.2 GB written for 1.36 GB of pattern.)
    const int st_row = lane >> 4, st_ch = lane & 15;
    auto tap_put = [&](int kt, const u32x4_t (&pa)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            *reinterpret_cast<u32x4_t*>(L + l31 * 256 + ((((kt & 3) * 4 + 2 * ks + half) ^ (l31 & 15)) * 16)) = pa[ks];
    };
    auto tap_flush = [&](unsigned char* dst, int kt) {
        if ((kt & 3) != 3 && kt + 1 < ntile) return;
        __builtin_amdgcn_wave_barrier();
        const int k0 = (kt & ~3) * 32;                                       // first key of the window
        const int nb = min(256, (T_ - k0) * 2);                              // valid bytes per row
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int row = st_row + 4 * it;
            if (q0 + row < T_ && st_ch * 16 < nb) {
                const unsigned char* src = L + row * 256 + ((st_ch ^ (row & 15)) * 16);
                unsigned char* d = dst + ((size_t)(uint32_t)((q0 + row) * T_ + k0 + st_ch * 8)) * 2;
                if (st_ch * 16 + 16 <= nb) {
                    const uint4 r = *reinterpret_cast<const uint4*>(src);
                    *reinterpret_cast<U4a2*>(d) = U4a2{r.x, r.y, r.z, r.w};       // (a nontemporal store here measured 1 % slower on the L/14 leg)
                } else {
                    for (int e = 0; e < (nb - st_ch * 16) / 2; ++e)
                        *reinterpret_cast<unsigned short*>(d + 2 * e) = *reinterpret_cast<const unsigned short*>(src + 2 * e);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // ---- pass 1: online max / sum (and the score tap)
    unsigned char* sc_dst = p.scores ? reinterpret_cast<unsigned char*>(p.scores) + (int64_t)g * T_ * T_ * 2 : nullptr;
    float m = -INFINITY, l = 0.f;
    auto pass1_tile = [&](int kt, auto masked) {
        constexpr bool MASK = decltype(masked)::value;
        uint32_t pk[8];
        float sc[16];
        score_tile(Kst[kt & 1], pk, sc);
        if (MASK) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half >= T_) sc[e] = -INFINITY;
        }
        float tm = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) tm = fmaxf(fmaxf(tm, sc[e]), sc[e + 1]);
        tm = fmaxf(tm, sc[15]);
        const float mn = fmaxf(m, tm);
        const float nb = -(mn * PV_LOG2E);
        const f32x2_t nb2 = {nb, nb};
        f32x2_t sum2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2_t a = __builtin_elementwise_fma(f32x2_t{sc[2 * i], sc[2 * i + 1]}, log2e2, nb2);
            sum2 += f32x2_t{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};      // exp2(-inf) = 0 for masked keys
        }
        l = l * __builtin_amdgcn_exp2f((m - mn) * PV_LOG2E) + (sum2.x + sum2.y);
        m = mn;
        if (sc_dst) {
            u32x4_t sa[2];
            to_rows(pk, sa);
            tap_put(kt, sa);
        }
    };
    u32x4_t kn1;
    {
        const u32x4_t k0 = fetch(rsK, 0);
        kn1 = fetch(rsK, 1);
        *reinterpret_cast<u32x4_t*>(Kst[0] + k_lds) = k0;
    }
    // per tile: issue the fetch of tile kt + 2 -> multiply tile kt out of LDS -> park tile kt + 1 (fetched one iteration ago) in
    // the other buffer -> flush the tap window when it is due -> barrier.  Loads and stores retire through ONE in-order counter
    // (vmcnt): a tile that is waited for must have been requested BEFORE the stores of the last flush, or the wave sits out HBM
    // write latency at every flush -- hence two tiles of fetch distance, and the flush behind the park.
    __syncthreads();
    for (int kt = 0; kt < ntile; ++kt) {
        const u32x4_t kn2 = fetch(rsK, kt + 2);
        if (active) {
            if (kt + 1 < ntile) pass1_tile(kt, std::false_type{});
            else pass1_tile(kt, std::true_type{});
        }
        *reinterpret_cast<u32x4_t*>(Kst[(kt + 1) & 1] + k_lds) = kn1;       // free since the barrier that ended tile kt - 1
        if (active && sc_dst) tap_flush(sc_dst, kt);
        __syncthreads();
        kn1 = kn2;
    }
    {   // merge the two lanes of a query
        const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
        const float M = fmaxf(m, mo);
        l = l * __builtin_amdgcn_exp2f((m - M) * PV_LOG2E) + lo * __builtin_amdgcn_exp2f((mo - M) * PV_LOG2E);
        m = M;
    }
    // a row with an infinite / NaN score (or none at all) is NaN throughout in the reference -> zeros (attention.py:149)
    const bool row_ok = active && l > 0.f && l < INFINITY && m > -INFINITY && m < INFINITY;
    const uint32_t row_mask = row_ok ? 0xffffffffu : 0u;
    const bool any_bad = active && __builtin_amdgcn_ballot_w64(!row_ok) != 0;
    const float nbias = row_ok ? -(m * PV_LOG2E + __builtin_amdgcn_logf(l)) : 0.f;
    const f32x2_t nbias2 = {nbias, nbias};

    // ---- pass 2: pattern tap + z
    unsigned char* pt_dst = p.pattern ? reinterpret_cast<unsigned char*>(p.pattern) + (int64_t)g * T_ * T_ * 2 : nullptr;
    f32x16 zacc[NTN];
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) zacc[tn][e] = 0.f;
    auto pass2_tile = [&](int kt, auto masked) {
        constexpr bool MASK = decltype(masked)::value;
        uint32_t pk[8];
        float sc[16];
        score_tile(Kst[kt & 1], pk, sc);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2_t a = __builtin_elementwise_fma(f32x2_t{sc[2 * i], sc[2 * i + 1]}, log2e2, nbias2);
            float x0 = __builtin_amdgcn_exp2f(a.x), x1 = __builtin_amdgcn_exp2f(a.y);
            if (MASK) {
                if (kt * 32 + ((2 * i) & 3) + 8 * ((2 * i) >> 2) + 4 * half >= T_) x0 = 0.f;
                if (kt * 32 + ((2 * i + 1) & 3) + 8 * ((2 * i + 1) >> 2) + 4 * half >= T_) x1 = 0.f;
            }
            pk[i] = pack_bf16x2(x0, x1);
        }
        if (any_bad) {
#pragma unroll
            for (int i = 0; i < 8; ++i) pk[i] &= row_mask;
        }
        u32x4_t pa[2];
        to_rows(pk, pa);
        if (pt_dst) tap_put(kt, pa);
        const unsigned char* vb = Vt[kt & 1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) {
                const uint4 vf = *reinterpret_cast<const uint4*>(vb + (tn * 32 + l31) * L2_VROW + (16 * ks + 8 * half) * 2);
                zacc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[ks]), __builtin_bit_cast(bf16x8, vf), zacc[tn], 0, 0, 0);
            }
    };
    u32x4_t vn1;
    {
        const u32x4_t k0 = fetch(rsK, 0), v0 = fetch(rsV, 0);
        kn1 = fetch(rsK, 1);
        vn1 = fetch(rsV, 1);
        *reinterpret_cast<u32x4_t*>(Kst[0] + k_lds) = k0;      // (everyone left pass 1's last tile, in Kst[(ntile - 1) & 1], through its closing barrier;
        stage_v(Vt[0], v0);                                      //  Kst[0] was last read one barrier earlier still when ntile is even)
    }
    __syncthreads();
    for (int kt = 0; kt < ntile; ++kt) {
        const u32x4_t kn2 = fetch(rsK, kt + 2), vn2 = fetch(rsV, kt + 2);
        if (active) {
            if (kt + 1 < ntile) pass2_tile(kt, std::false_type{});
            else pass2_tile(kt, std::true_type{});
        }
        *reinterpret_cast<u32x4_t*>(Kst[(kt + 1) & 1] + k_lds) = kn1;
        stage_v(Vt[(kt + 1) & 1], vn1);
        if (active && pt_dst) tap_flush(pt_dst, kt);
        __syncthreads();
        kn1 = kn2;
        vn1 = vn2;
    }
    if (!active) return;

    // ---- z: C layout (col = d, rows = queries) -> LDS rows [32][DH] -> 16-byte row stores
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            *reinterpret_cast<bf16_t*>(L + row * 144 + (tn * 32 + l31) * 2) = f32_to_bf16(zacc[tn][e]);
        }
    __builtin_amdgcn_wave_barrier();
    {
        constexpr int CPR = DH / 8;
        bf16_t* zb = reinterpret_cast<bf16_t*>(p.z) + head_off;
        for (int c = lane; c < 32 * CPR; c += 64) {
            const int row = c / CPR, ch = c - row * CPR;
            if (q0 + row < T_)
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(zb) + (int64_t)(q0 + row) * tokb + ch * 16) =
                    *reinterpret_cast<const uint4*>(L + row * 144 + ch * 16);
        }
    }
}

int launch_attn_lean(const AttnParams& p, hipStream_t stream) {
    const int heads = p.B * p.H, qblocks = (p.T + 127) / 128;
    int ex = 0;
    const bool pow2 = p.attn_scale > 0.f && std::frexp(p.attn_scale, &ex) == 0.5f;
    {
        const double bh = (double)heads, tt = (double)p.T * p.T;
        const double bytes = (4.0 * bh * p.T * p.dh + ((p.scores ? 1.0 : 0.0) + (p.pattern ? 1.0 : 0.0)) * bh * tt) * 2.0;
        ProfScope prof(PV_PROF_ATTN, stream, 4.0 * bh * tt * p.dh, bytes);
        if (pow2) hipLaunchKernelGGL((attn_lean_kernel<64, true>), dim3(heads * qblocks), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((attn_lean_kernel<64, false>), dim3(heads * qblocks), dim3(256), 0, stream, p);
    }
    PV_LAUNCH_CHECK("attn_lean_kernel");
    return PV_OK;
}

template <typename T>
int dispatch_attn(AttnParams& p, hipStream_t stream) {
    p.Tpad = (p.T + 31) / 32 * 32;
    if constexpr (sizeof(T) == 2) {
        // one head per wave: needs whole bf16 pairs per row and 4-byte aligned head blocks in the taps
        const bool taps_ok = (reinterpret_cast<uintptr_t>(p.scores) % 4 == 0) && (reinterpret_cast<uintptr_t>(p.pattern) % 4 == 0);
        if (p.T <= 64 && p.T % 2 == 0 && taps_ok && pv_aligned16(p.z) && !g_pv_tuning.attn_wg &&
            (int64_t)p.T * p.H * p.dh * 2 < (1ll << 31)) {
            if (p.dh == 64) return launch_attn_wave<64>(p, stream);
            if (p.dh == 32) return launch_attn_wave<32>(p, stream);
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (p.T > 64 && p.dh == 64 && pv_aligned16(p.z) && !g_pv_tuning.attn_wg &&
            (int64_t)p.T * p.H * p.dh * 2 < (1ll << 31) && (int64_t)p.B * p.H * ((p.T + 127) / 128) < (1ll << 31)) {
            return launch_attn_lean(p, stream);
        }
    }
    if (p.T <= 64) {
        if (p.dh == 64) return launch_attn<T, 64, 64, 1>(p, stream);
        if (p.dh == 32) return launch_attn<T, 64, 32, 1>(p, stream);
    } else if (p.T <= 640) {
        if (p.dh == 64) return launch_attn<T, 32, 64, 10>(p, stream);
        if (p.dh == 32) return launch_attn<T, 32, 32, 10>(p, stream);
    }
    pv_set_error("attention: unsupported (T, d_head); supported: T <= 640, d_head in {32, 64}");
    return PV_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------------
// Resume behind a hooked hook_attn_scores / hook_pattern (see attention.hpp): a wave owns one query row of one (image, head).
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_resume_kernel(const AttnParams p, int from_scores) {
    __shared__ float prob[4][640];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;                 // (b * H + h) * T + t
    const int T_ = p.T, H = p.H, dh = p.dh;
    if (row >= (int64_t)p.B * H * T_) return;
    const int t = (int)(row % T_);
    const int64_t g = row / T_;
    const int b = (int)(g / H), h = (int)(g - (int64_t)b * H);
    float* pr = prob[wave];
    const T* in = reinterpret_cast<const T*>(from_scores ? p.scores : p.pattern) + row * T_;
    if (from_scores) {
        float m = -INFINITY;
        for (int c = lane; c < T_; c += 64) {
            const float s = DT<T>::load(in + c);
            pr[c] = s;
            m = fmaxf(m, s);
        }
        m = wave_max(m);
        float l = 0.f;
        bool bad = false;
        for (int c = lane; c < T_; c += 64) {
            const float s = pr[c];
            bad |= !(s == s);
            const float e = expf(s - m);
            pr[c] = e;
            l += e;
        }
        l = wave_sum(l);
        // a softmax row with a NaN (or nothing but -inf) in it is NaN throughout in the reference -> zeros (attention.py:149)
        const bool row_bad = __builtin_amdgcn_ballot_w64(bad) != 0 || !(l > 0.f) || !(l < INFINITY) || !(m > -INFINITY) || !(m < INFINITY);
        T* tap = p.pattern ? reinterpret_cast<T*>(p.pattern) + row * T_ : nullptr;
        for (int c = lane; c < T_; c += 64) {
            const float q = row_bad ? 0.f : DT<T>::round(pr[c] / l);
            pr[c] = q;
            if (tap) DT<T>::store(tap + c, q);
        }
    } else {
        for (int c = lane; c < T_; c += 64) pr[c] = DT<T>::load(in + c);
    }
    __builtin_amdgcn_wave_barrier();
    // z[t, h, :] = sum_c pattern[c] v[c, h, :]  (attention.py:267-281); lane = d (two passes for d_head > 64 never occur: dh <= 64)
    const T* vb = reinterpret_cast<const T*>(p.v) + ((int64_t)b * T_ * H + h) * dh;
    const int64_t tok = (int64_t)H * dh;
    if (lane < dh) {
        float acc = 0.f;
        for (int c = 0; c < T_; ++c) acc = fmaf(pr[c], DT<T>::load(vb + c * tok + lane), acc);
        DT<T>::store(reinterpret_cast<T*>(p.z) + ((int64_t)b * T_ * H + h) * dh + (int64_t)t * tok + lane, acc);
    }
}

}  // namespace

int pv_attention_supported(int T, int dh) { return (T <= 640 && (dh == 64 || dh == 32)) ? 1 : 0; }

int pv_launch_attention(int dtype, AttnParams p, hipStream_t stream) {
    PV_REQUIRE(p.q && p.k && p.v && p.z, "attention operands must be non-null");
    PV_REQUIRE(pv_aligned16(p.q) && pv_aligned16(p.k) && pv_aligned16(p.v), "attention operands must be 16-byte aligned");
    PV_REQUIRE(p.B <= 65535 && p.H <= 65535, "attention grid limits");
    if (dtype == PV_DTYPE_BF16) return dispatch_attn<bf16_t>(p, stream);
    if (dtype == PV_DTYPE_F32) return dispatch_attn<float>(p, stream);
    pv_set_error("attention: unsupported dtype");
    return PV_ERR_INVALID;
}

int pv_launch_attention_resume(int dtype, AttnParams p, int from_scores, hipStream_t stream) {
    PV_REQUIRE(p.v && p.z && (from_scores ? p.scores != nullptr : p.pattern != nullptr), "attention resume operands must be non-null");
    PV_REQUIRE(p.T >= 1 && p.T <= 640 && p.dh >= 1 && p.dh <= 64, "attention resume: T <= 640, d_head <= 64");
    const int64_t rows = (int64_t)p.B * p.H * p.T;
    PV_REQUIRE((rows + 3) / 4 < (1ll << 31), "attention resume grid");
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == PV_DTYPE_BF16) hipLaunchKernelGGL(attn_resume_kernel<bf16_t>, grid, block, 0, stream, p, from_scores);
    else if (dtype == PV_DTYPE_F32) hipLaunchKernelGGL(attn_resume_kernel<float>, grid, block, 0, stream, p, from_scores);
    else { pv_set_error("attention resume: unsupported dtype"); return PV_ERR_INVALID; }
    PV_LAUNCH_CHECK("attn_resume_kernel");
    return PV_OK;
}
