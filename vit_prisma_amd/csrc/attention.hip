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

template <int DH>
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
    u32x4_t kf[2][NKS];
#pragma unroll
    for (int ki = 0; ki < 2; ++ki)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            kf[ki][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsK, (unsigned)(ki * 32 + l31) * tokb + (2 * ks + half) * 16, 0, 0);
    const float inv_scale = 1.0f / p.attn_scale;
    (void)inv_scale;
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
        u32x4_t qf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (unsigned)(tq * 32 + l31) * tokb + (2 * ks + half) * 16, 0, 0);
        f32x16 acc[2];
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ki][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int ki = 0; ki < 2; ++ki)
                acc[ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qf[ks]),
                                                                  __builtin_bit_cast(bf16x8, kf[ki][ks]), acc[ki], 0, 0, 0);
        // C layout: col = lane & 31 (key), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (query)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = tq * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                *reinterpret_cast<bf16_t*>(SP + row * AW_ROW + (ki * 32 + l31) * 2) = f32_to_bf16(acc[ki][e] / p.attn_scale);
            }
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
                *reinterpret_cast<uint4*>(d) = make_uint4(w[0], w[1], w[2], w[3]);      // 4-byte aligned is enough
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
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            uint32_t w[4];
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                float p0 = e_[c8 * 8 + q2 * 2] * rs, p1 = e_[c8 * 8 + q2 * 2 + 1] * rs;
                if (p0 != p0) p0 = 0.f;
                if (p1 != p1) p1 = 0.f;
                w[q2] = (lane < T_) ? pack_bf16x2(p0, p1) : 0u;          // pad rows: finite zeros for the MFMA
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
                zacc[tq][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa),
                                                                       __builtin_bit_cast(bf16x8, vf[tn]), zacc[tq][tn], 0, 0, 0);
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- 7: z -> LDS rows [64][DH] -> 16-byte row stores into [B, T, H, dh]
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = tq * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                *reinterpret_cast<bf16_t*>(SP + row * AW_ROW + (tn * 32 + l31) * 2) = f32_to_bf16(zacc[tq][tn][e]);
            }
    __builtin_amdgcn_wave_barrier();
    {
        constexpr int CPR = DH / 8;                                      // 16-byte chunks per row
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
        hipLaunchKernelGGL((attn_wave_kernel<DH>), dim3((heads + 3) / 4), dim3(256), 0, stream, p);
    }
    PV_LAUNCH_CHECK("attn_wave_kernel");
    return PV_OK;
}

// ---------------------------------------------------------------------------------------------------
// Long-sequence variant (bf16, T > 64: L/14@336 has T = 577): a WAVE owns 32 query rows of one (image, head) and
// streams the keys twice, with the product SWAPPED (S^T = K Q^T) so that a lane holds 16 keys of ONE query:
//   pass 1  per 32-key tile: S^T tile on MFMA (K rows and the wave's Q rows straight from global, pad rows read 0),
//           scaled + rounded to bf16 like the reference's score tensor, online (max, sum) per lane; the two lanes of
//           a query (halves of the wave) are merged at the end
//   pass 2  the same tiles again: p = exp(s - max) / sum, rounded; (a) hook_pattern: the 32 x 32 tile goes through
//           2.5 KB of LDS to become 32-byte row pieces -> dword stores (the [T][T] block of a head is only 2-byte
//           aligned when T is odd: odd-address rows are shifted one element with funnel shifts);
//           (b) z += P V on MFMA -- P moves from the C layout to the A-operand layout with one cross-half
//           exchange per k16 step, V fragments are 2-byte buffer loads (32 lanes = 64 contiguous bytes of a key row)
//   The four waves of a workgroup (128 consecutive queries of one head) share each 32-key K (and V) tile: fetched
//   cooperatively as full 128-byte rows into registers one tile ahead, parked in 4.5 KB of LDS between two barriers.
//   No [QB][T] score block in LDS (the workgroup kernel above needs 41 KB of it and runs 8 waves per CU);
//   hook_attn_scores, when tapped, is written from pass 1 the same way as the pattern.
// ---------------------------------------------------------------------------------------------------
constexpr int AS_PROW = 80;                 // bytes per LDS row of a 32 x 32 bf16 tile (64 + 16 pad)

template <int DH>
__global__ __launch_bounds__(256) void attn_stream_kernel(const AttnParams p) {
    static_assert(DH == 64, "d_head 64");
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][32 * 144];      // per wave: P tile (2.5 KB) / z staging (4.5 KB)
    __shared__ __attribute__((aligned(16))) unsigned char Kst[32 * 144];          // the workgroup's current key tile   [32][d_head] (+16 B pad)
    __shared__ __attribute__((aligned(16))) unsigned char Vst[32 * 144];          // ... and value tile (pass 2)
    constexpr int NKS = DH / 16;
    constexpr int NTN = DH / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T_ = p.T, H = p.H;
    const int qblocks = (T_ + 127) / 128;
    const int g = blockIdx.x / qblocks;                     // (image, head)
    const int q0 = (blockIdx.x - g * qblocks) * 128 + wave * 32;
    const bool active = q0 < T_;                            // idle waves of a head's last block still load tiles and meet the barriers
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
    const float inv_scale = 1.0f / p.attn_scale;     // (a power of two for every d_head in use; the product is rounded to bf16 next)

    // Q as the B operand (columns = this wave's queries): lane (query l31, half) holds d-chunk (2 ks + half)
    u32x4_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        qf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (unsigned)(q0 + l31) * tokb + (2 * ks + half) * 16, 0, 0);

    // S^T tile kt: acc[e] = score(query q0 + l31, key kt*32 + (e & 3) + 8 * (e >> 2) + 4 * half), bf16-rounded
    // cooperative tile fetch: thread t moves 16 B of key row t / 8 (one full 128-byte row per 8 lanes; rows >= T read 0)
    const unsigned tile_off = (unsigned)(threadIdx.x >> 3) * tokb + (threadIdx.x & 7) * 16;
    const int tile_lds = (threadIdx.x >> 3) * 144 + (threadIdx.x & 7) * 16;
    auto score_tile = [&](int kt, float (&sc)[16]) {
        uint4 kf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(Kst + l31 * 144 + (2 * ks + half) * 16);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), __builtin_bit_cast(bf16x8, qf[ks]), acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = bf16_to_f32(f32_to_bf16(acc[e] * inv_scale));
    };
    // a 32 x 32 bf16 tile held as 8 packed dwords per lane (pk[i] = keys (4 i' .. ) see above) -> LDS -> row-piece stores
    // into the head's [T][T] block `dst` (element (q, key)); lane handles row lane >> 1, 16 elements from key (lane & 1) * 16
    auto store_tile = [&](bf16_t* dst, int kt, const uint32_t (&pk)[8]) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)          // keys 8 gq + 4 half + 0..3 of query l31: 8 contiguous bytes
            *reinterpret_cast<uint2*>(L + l31 * AS_PROW + (8 * gq + 4 * half) * 2) = make_uint2(pk[2 * gq], pk[2 * gq + 1]);
        __builtin_amdgcn_wave_barrier();
        const int row = lane >> 1, k0 = kt * 32 + (lane & 1) * 16;
        const uint4 r0 = *reinterpret_cast<const uint4*>(L + row * AS_PROW + (lane & 1) * 32);
        const uint4 r1 = *reinterpret_cast<const uint4*>(L + row * AS_PROW + (lane & 1) * 32 + 16);
        __builtin_amdgcn_wave_barrier();
        const int q = q0 + row;
        const int nv = min(16, T_ - k0);
        if (q < T_ && nv > 0) {
            const int64_t gidx = (int64_t)q * T_ + k0;                 // element index inside the head's block
            unsigned char* d = reinterpret_cast<unsigned char*>(dst) + gidx * 2;
            const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const bool odd = (reinterpret_cast<uintptr_t>(d) & 2) != 0;
            if (nv == 16 && !odd) {
                *reinterpret_cast<uint4*>(d) = r0;                     // dword-aligned 16-byte stores
                *reinterpret_cast<uint4*>(d + 16) = r1;
            } else if (nv == 16) {
                *reinterpret_cast<unsigned short*>(d) = (unsigned short)(w[0] & 0xffffu);
                uint32_t o[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) o[i] = (w[i] >> 16) | (w[i + 1] << 16);
                *reinterpret_cast<uint4*>(d + 2) = make_uint4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<uint2*>(d + 18) = make_uint2(o[4], o[5]);
                *reinterpret_cast<uint32_t*>(d + 26) = o[6];
                *reinterpret_cast<unsigned short*>(d + 30) = (unsigned short)(w[7] >> 16);
            } else {
                for (int i = 0; i < nv; ++i)
                    *reinterpret_cast<unsigned short*>(d + 2 * i) = (unsigned short)((i & 1) ? (w[i >> 1] >> 16) : (w[i >> 1] & 0xffffu));
            }
        }
    };

    // ---- pass 1: online max / sum (and the score tap)
    bf16_t* sc_dst = p.scores ? reinterpret_cast<bf16_t*>(p.scores) + (int64_t)g * T_ * T_ : nullptr;
    float m = -INFINITY, l = 0.f;
    auto pass1_tile = [&](int kt, auto masked) {
        constexpr bool MASK = decltype(masked)::value;
        float sc[16];
        score_tile(kt, sc);
        float tm = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (!MASK || key < T_) tm = fmaxf(tm, sc[e]);
        }
        const float mn = fmaxf(m, tm);
        float add = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (!MASK || key < T_) add += __expf(sc[e] - mn);
        }
        l = (mn == -INFINITY) ? 0.f : l * __expf(m - mn) + add;
        m = mn;
        if (sc_dst) {
            uint32_t pk[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pk[i] = pack_bf16x2(sc[2 * i], sc[2 * i + 1]);
            store_tile(sc_dst, kt, pk);
        }
    };
    // tile kt+1 is fetched into registers while tile kt is consumed out of LDS (only the last tile can hold keys >= T)
    *reinterpret_cast<u32x4_t*>(Kst + tile_lds) = __builtin_amdgcn_raw_buffer_load_b128(rsK, tile_off, 0, 0);
    __syncthreads();
    for (int kt = 0; kt < ntile; ++kt) {
        u32x4_t kn = {0, 0, 0, 0};
        if (kt + 1 < ntile) kn = __builtin_amdgcn_raw_buffer_load_b128(rsK, (unsigned)(kt + 1) * 32u * tokb + tile_off, 0, 0);
        if (active) {
            if (kt + 1 < ntile) pass1_tile(kt, std::false_type{});
            else pass1_tile(kt, std::true_type{});
        }
        __syncthreads();
        *reinterpret_cast<u32x4_t*>(Kst + tile_lds) = kn;
        __syncthreads();
    }
    {   // merge the two lanes of a query
        const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
        const float M = fmaxf(m, mo);
        l = (M == -INFINITY) ? 0.f : l * __expf(m - M) + lo * __expf(mo - M);
        m = M;
    }
    const float rl = active ? 1.0f / l : 0.f;

    // ---- pass 2: pattern tap + z
    bf16_t* pt_dst = p.pattern ? reinterpret_cast<bf16_t*>(p.pattern) + (int64_t)g * T_ * T_ : nullptr;
    f32x16 zacc[NTN];
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) zacc[tn][e] = 0.f;
    auto pass2_tile = [&](int kt, auto masked) {
        constexpr bool MASK = decltype(masked)::value;
        float sc[16];
        score_tile(kt, sc);
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float pv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = 2 * i + u;
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                float x = (!MASK || key < T_) ? __expf(sc[e] - m) * rl : 0.f;
                if (x != x) x = 0.f;                                     // attention.py:149
                pv[u] = x;
            }
            pk[i] = pack_bf16x2(pv[0], pv[1]);
        }
        if (pt_dst) store_tile(pt_dst, kt, pk);
        // P from the C layout (this lane: keys 4 half + 8 j + 0..3, j = 0..3, as pk[2j], pk[2j+1]) to the A operand
        // (lane needs keys ks*16 + half*8 + 0..7): per k16 step the halves swap one group of four keys
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // half 0 keeps group 2ks, sends group 2ks+1, receives the partner's group 2ks; half 1 the mirror image
            const uint32_t s0 = half ? pk[4 * ks + 0] : pk[4 * ks + 2];
            const uint32_t s1 = half ? pk[4 * ks + 1] : pk[4 * ks + 3];
            const uint32_t r0 = __shfl_xor(s0, 32, 64), r1 = __shfl_xor(s1, 32, 64);
            u32x4_t pa;
            if (half) pa = u32x4_t{r0, r1, pk[4 * ks + 2], pk[4 * ks + 3]};
            else pa = u32x4_t{pk[4 * ks + 0], pk[4 * ks + 1], r0, r1};
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) {
                uint32_t w[4];
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2) {
                    const unsigned char* vp = Vst + (ks * 16 + half * 8 + q2 * 2) * 144 + (tn * 32 + l31) * 2;
                    const uint32_t lo16 = *reinterpret_cast<const unsigned short*>(vp);
                    const uint32_t hi16 = *reinterpret_cast<const unsigned short*>(vp + 144);
                    w[q2] = lo16 | (hi16 << 16);
                }
                zacc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa),
                                                                   __builtin_bit_cast(bf16x8, u32x4_t{w[0], w[1], w[2], w[3]}), zacc[tn], 0, 0, 0);
            }
        }
    };
    *reinterpret_cast<u32x4_t*>(Kst + tile_lds) = __builtin_amdgcn_raw_buffer_load_b128(rsK, tile_off, 0, 0);
    *reinterpret_cast<u32x4_t*>(Vst + tile_lds) = __builtin_amdgcn_raw_buffer_load_b128(rsV, tile_off, 0, 0);
    __syncthreads();
    for (int kt = 0; kt < ntile; ++kt) {
        u32x4_t kn = {0, 0, 0, 0}, vn = {0, 0, 0, 0};
        if (kt + 1 < ntile) {
            kn = __builtin_amdgcn_raw_buffer_load_b128(rsK, (unsigned)(kt + 1) * 32u * tokb + tile_off, 0, 0);
            vn = __builtin_amdgcn_raw_buffer_load_b128(rsV, (unsigned)(kt + 1) * 32u * tokb + tile_off, 0, 0);
        }
        if (active) {
            if (kt + 1 < ntile) pass2_tile(kt, std::false_type{});
            else pass2_tile(kt, std::true_type{});
        }
        __syncthreads();
        *reinterpret_cast<u32x4_t*>(Kst + tile_lds) = kn;
        *reinterpret_cast<u32x4_t*>(Vst + tile_lds) = vn;
        __syncthreads();
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

int launch_attn_stream(const AttnParams& p, hipStream_t stream) {
    const int heads = p.B * p.H, qblocks = (p.T + 127) / 128;
    {
        const double bh = (double)heads, tt = (double)p.T * p.T;
        const double bytes = (4.0 * bh * p.T * p.dh + ((p.scores ? 1.0 : 0.0) + (p.pattern ? 1.0 : 0.0)) * bh * tt) * 2.0;
        ProfScope prof(PV_PROF_ATTN, stream, 4.0 * bh * tt * p.dh, bytes);
        hipLaunchKernelGGL((attn_stream_kernel<64>), dim3(heads * qblocks), dim3(256), 0, stream, p);
    }
    PV_LAUNCH_CHECK("attn_stream_kernel");
    return PV_OK;
}

// ---------------------------------------------------------------------------------------------------
// Long-sequence variant, single pass over the keys (bf16, 64 < T <= 600, d_head 64): the wave's 32-query x T score strip
// lives in LDS (32 x roundup16(2 T) bytes = 36.5 KB at T = 577; four strips per workgroup = one workgroup per CU), so
//   pass A  K streamed ONCE: S^T = K Q^T tile on MFMA, scaled + rounded to bf16 like the reference's score tensor,
//           online (max, sum) per lane, the tile parked in the strip (8-byte pieces in the MFMA C layout)
//   [tap]   hook_attn_scores = the strip, flushed row by row (below)
//   pass B  V streamed once: the tile comes back out of the strip, p = exp(s - max) / sum rounded to bf16, moves to the
//           A-operand layout (one cross-half exchange per k16 step), goes back into the strip as 16-byte pieces and
//           feeds z += P V on MFMA; V is staged TRANSPOSED ([d][key]) so a B fragment is one ds_read_b128
//   [tap]   hook_pattern = the strip, flushed
// Flush: a query row is 2 T contiguous bytes of the head's [T][T] block, but the block is only 2-byte aligned when T is
// odd -- the row is written as 16-byte-aligned global chunks whose LDS source is realigned ONCE per chunk with
// v_alignbyte (the strip rows are 16-byte aligned), plus <= 7 two-byte stores at each end.  QK^T is computed once
// (attn_stream_kernel: twice), the taps leave as full 16-byte stores (there: 32-byte pieces with funnel shifts per piece).
// ---------------------------------------------------------------------------------------------------
constexpr int ST_KROW = 144;                // K tile row: 128 B + 16 pad
constexpr int ST_VROW = 80;                 // V^T tile row: 32 keys x 2 B + 16 pad

__device__ __forceinline__ int st_strip_bytes(int T) { return 32 * ((2 * T + 15) / 16 * 16) + 32; }

template <int DH>
__global__ __launch_bounds__(256) void attn_strip_kernel(const AttnParams p) {
    static_assert(DH == 64, "d_head 64");
    extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];
    constexpr int NKS = DH / 16;
    constexpr int NTN = DH / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T_ = p.T, H = p.H;
    const int RS = (2 * T_ + 15) / 16 * 16;
    unsigned char* Kst = st_smem;                               // [32][ST_KROW]
    unsigned char* Vt = st_smem + 32 * ST_KROW;                 // [DH][ST_VROW]
    unsigned char* S = Vt + DH * ST_VROW + wave * st_strip_bytes(T_);          // this wave's strip [32][RS]
    const int qblocks = (T_ + 127) / 128;
    const int g = blockIdx.x / qblocks;                         // (image, head)
    const int q0 = (blockIdx.x - g * qblocks) * 128 + wave * 32;
    const bool active = q0 < T_;                                // idle waves of a head's last block still load tiles and meet the barriers
    const int b = g / H, h = g - b * H;
    const int half = lane >> 5, l31 = lane & 31;
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

    u32x4_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        qf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (unsigned)(q0 + l31) * tokb + (2 * ks + half) * 16, 0, 0);
    // cooperative tile fetch: thread t moves 16 B (8 d-values) of key row t / 8
    const int t_key = threadIdx.x >> 3, t_dc = threadIdx.x & 7;
    const unsigned tile_off = (unsigned)t_key * tokb + t_dc * 16;
    unsigned char* my_row = S + l31 * RS;

    // flush the strip's rows (queries q0 .. ) into the head's [T][T] block `dst`
    auto flush = [&](bf16_t* dst) {
        __builtin_amdgcn_wave_barrier();
        const int rows = min(32, T_ - q0);
        for (int r = 0; r < rows; ++r) {
            unsigned char* d = reinterpret_cast<unsigned char*>(dst) + ((int64_t)(q0 + r) * T_) * 2;
            const unsigned char* src = S + r * RS;
            const int n_head = (int)((16u - (unsigned)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u);       // bytes up to the first aligned chunk
            const int body = (2 * T_ - n_head) / 16;                                                   // full 16-byte chunks
            const int tail0 = n_head + 16 * body;                                                      // first byte of the tail
            const int sh = n_head & 3;                                                                 // 0 or 2
            for (int c = lane; c < body; c += 64) {
                const unsigned char* s4 = src + ((n_head + 16 * c) & ~3);
                const uint32_t w0 = *reinterpret_cast<const uint32_t*>(s4), w1 = *reinterpret_cast<const uint32_t*>(s4 + 4),
                               w2 = *reinterpret_cast<const uint32_t*>(s4 + 8), w3 = *reinterpret_cast<const uint32_t*>(s4 + 12),
                               w4 = *reinterpret_cast<const uint32_t*>(s4 + 16);
                uint4 o;
                if (sh == 0) o = make_uint4(w0, w1, w2, w3);
                else o = make_uint4(__builtin_amdgcn_alignbyte(w1, w0, 2), __builtin_amdgcn_alignbyte(w2, w1, 2),
                                    __builtin_amdgcn_alignbyte(w3, w2, 2), __builtin_amdgcn_alignbyte(w4, w3, 2));
                *reinterpret_cast<uint4*>(d + n_head + 16 * c) = o;
            }
            // head and tail: two-byte stores (at most 7 each)
            if (lane < 8) {
                if (2 * lane < n_head) *reinterpret_cast<unsigned short*>(d + 2 * lane) = *reinterpret_cast<const unsigned short*>(src + 2 * lane);
            } else if (lane < 16) {
                const int o2 = tail0 + 2 * (lane - 8);
                if (o2 < 2 * T_) *reinterpret_cast<unsigned short*>(d + o2) = *reinterpret_cast<const unsigned short*>(src + o2);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // ---- pass A: scores -> strip, online max / sum
    float m = -INFINITY, l = 0.f;
    {
        u32x4_t kn = __builtin_amdgcn_raw_buffer_load_b128(rsK, tile_off, 0, 0);
        *reinterpret_cast<u32x4_t*>(Kst + t_key * ST_KROW + t_dc * 16) = kn;
    }
    __syncthreads();
    for (int kt = 0; kt < ntile; ++kt) {
        u32x4_t kn = {0, 0, 0, 0};
        if (kt + 1 < ntile) kn = __builtin_amdgcn_raw_buffer_load_b128(rsK, (unsigned)(kt + 1) * 32u * tokb + tile_off, 0, 0);
        if (active) {
            uint4 kf[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(Kst + l31 * ST_KROW + (2 * ks + half) * 16);
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), __builtin_bit_cast(bf16x8, qf[ks]), acc, 0, 0, 0);
            // acc[e] = score(query q0 + l31, key kt*32 + (e & 3) + 8 * (e >> 2) + 4 * half)
            float sc[16];
            float tm = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                sc[e] = bf16_to_f32(f32_to_bf16(acc[e] * inv_scale));
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (key < T_) tm = fmaxf(tm, sc[e]);
            }
            const float mn = fmaxf(m, tm);
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (key < T_) add += __expf(sc[e] - mn);
            }
            l = (mn == -INFINITY) ? 0.f : l * __expf(m - mn) + add;
            m = mn;
#pragma unroll
            for (int j = 0; j < 4; ++j) {                       // keys 8 j + 4 half + 0..3: 8 bytes
                const int kb = (kt * 32 + 8 * j + 4 * half) * 2;
                if (kb + 8 <= RS)
                    *reinterpret_cast<uint2*>(my_row + kb) = make_uint2(pack_bf16x2(sc[4 * j], sc[4 * j + 1]), pack_bf16x2(sc[4 * j + 2], sc[4 * j + 3]));
            }
        }
        __syncthreads();
        *reinterpret_cast<u32x4_t*>(Kst + t_key * ST_KROW + t_dc * 16) = kn;
        __syncthreads();
    }
    {   // merge the two lanes of a query
        const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
        const float M = fmaxf(m, mo);
        l = (M == -INFINITY) ? 0.f : l * __expf(m - M) + lo * __expf(mo - M);
        m = M;
    }
    const float rl = active ? 1.0f / l : 0.f;
    if (active && p.scores) flush(reinterpret_cast<bf16_t*>(p.scores) + (int64_t)g * T_ * T_);

    // ---- pass B: pattern -> strip, z
    f32x16 zacc[NTN];
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) zacc[tn][e] = 0.f;
    auto stage_v = [&](const u32x4_t& v) {                      // 8 d-values of key t_key -> V^T[d][key]
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<unsigned short*>(Vt + (t_dc * 8 + 2 * i) * ST_VROW + t_key * 2) = (unsigned short)(w[i] & 0xffffu);
            *reinterpret_cast<unsigned short*>(Vt + (t_dc * 8 + 2 * i + 1) * ST_VROW + t_key * 2) = (unsigned short)(w[i] >> 16);
        }
    };
    stage_v(__builtin_amdgcn_raw_buffer_load_b128(rsV, tile_off, 0, 0));
    __syncthreads();
    for (int kt = 0; kt < ntile; ++kt) {
        u32x4_t vn = {0, 0, 0, 0};
        if (kt + 1 < ntile) vn = __builtin_amdgcn_raw_buffer_load_b128(rsV, (unsigned)(kt + 1) * 32u * tokb + tile_off, 0, 0);
        if (active) {
            uint32_t pk[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kb = (kt * 32 + 8 * j + 4 * half) * 2;
                uint2 w = make_uint2(0u, 0u);
                if (kb + 8 <= RS) w = *reinterpret_cast<const uint2*>(my_row + kb);
                const uint32_t ww[2] = {w.x, w.y};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int key = kt * 32 + 8 * j + 4 * half + 2 * u;
                    float x0 = key < T_ ? __expf(__uint_as_float(ww[u] << 16) - m) * rl : 0.f;
                    float x1 = key + 1 < T_ ? __expf(__uint_as_float(ww[u] & 0xffff0000u) - m) * rl : 0.f;
                    if (x0 != x0) x0 = 0.f;                                 // attention.py:149
                    if (x1 != x1) x1 = 0.f;
                    pk[2 * j + u] = pack_bf16x2(x0, x1);
                }
            }
            // C layout (keys 8 j + 4 half + 0..3 as pk[2j], pk[2j+1]) -> A operand (keys 16 ks + 8 half + 0..7): per k16 step
            // the halves swap one group of four keys
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t s0 = half ? pk[4 * ks + 0] : pk[4 * ks + 2];
                const uint32_t s1 = half ? pk[4 * ks + 1] : pk[4 * ks + 3];
                const uint32_t r0 = __shfl_xor(s0, 32, 64), r1 = __shfl_xor(s1, 32, 64);
                u32x4_t pa;
                if (half) pa = u32x4_t{r0, r1, pk[4 * ks + 2], pk[4 * ks + 3]};
                else pa = u32x4_t{pk[4 * ks + 0], pk[4 * ks + 1], r0, r1};
                const int kb = (kt * 32 + 16 * ks + 8 * half) * 2;
                if (kb + 16 <= RS) *reinterpret_cast<u32x4_t*>(my_row + kb) = pa;
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn) {
                    const uint4 vf = *reinterpret_cast<const uint4*>(Vt + (tn * 32 + l31) * ST_VROW + (16 * ks + 8 * half) * 2);
                    zacc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, vf), zacc[tn], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        stage_v(vn);
        __syncthreads();
    }
    if (!active) return;
    if (p.pattern) flush(reinterpret_cast<bf16_t*>(p.pattern) + (int64_t)g * T_ * T_);

    // ---- z: C layout (col = d, rows = queries) -> strip rows [32][DH] -> 16-byte row stores
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            *reinterpret_cast<bf16_t*>(S + row * 144 + (tn * 32 + l31) * 2) = f32_to_bf16(zacc[tn][e]);
        }
    __builtin_amdgcn_wave_barrier();
    {
        constexpr int CPR = DH / 8;
        bf16_t* zb = reinterpret_cast<bf16_t*>(p.z) + head_off;
        for (int c = lane; c < 32 * CPR; c += 64) {
            const int row = c / CPR, ch = c - row * CPR;
            if (q0 + row < T_)
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(zb) + (int64_t)(q0 + row) * tokb + ch * 16) =
                    *reinterpret_cast<const uint4*>(S + row * 144 + ch * 16);
        }
    }
}

int launch_attn_strip(const AttnParams& p, hipStream_t stream) {
    const int heads = p.B * p.H, qblocks = (p.T + 127) / 128;
    const int RS = (2 * p.T + 15) / 16 * 16;
    const size_t lds = 32 * ST_KROW + 64 * ST_VROW + 4 * (size_t)(32 * RS + 32);
    static size_t attr_done = 0;
    if (lds > attr_done) {
        PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_strip_kernel<64>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = lds;
    }
    {
        const double bh = (double)heads, tt = (double)p.T * p.T;
        const double bytes = (4.0 * bh * p.T * p.dh + ((p.scores ? 1.0 : 0.0) + (p.pattern ? 1.0 : 0.0)) * bh * tt) * 2.0;
        ProfScope prof(PV_PROF_ATTN, stream, 4.0 * bh * tt * p.dh, bytes);
        hipLaunchKernelGGL((attn_strip_kernel<64>), dim3(heads * qblocks), dim3(256), lds, stream, p);
    }
    PV_LAUNCH_CHECK("attn_strip_kernel");
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
            // single pass with the score strip in LDS while four strips fit beside the K / V tiles; two passes beyond
            const bool taps_ok = (reinterpret_cast<uintptr_t>(p.scores) % 2 == 0) && (reinterpret_cast<uintptr_t>(p.pattern) % 2 == 0);
            if (p.T <= 600 && taps_ok && g_pv_tuning.attn_stream != 1) return launch_attn_strip(p, stream);
            return launch_attn_stream(p, stream);
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
