// SAE encoder + TopK without materialising hidden_pre (gfx950).
//
// Reference semantics: hidden_pre = sae_in @ W_enc + b_enc (sae/sae.py:567-574), TopK keeps the k largest per token
// (sae/sae.py:795-810).  Only k = 32 of the d_sae = 24 576 pre-activations of a token survive, so the [N, d_sae] fp32
// matrix (403 MB at N = 4096) is never written.  Instead:
//
//   pass 0   filter GEMM on every 16th feature (fp16 MFMA, 1/16 of the work) -> sample [N, d_sae/16]
//   thr      per token: s_q = q-th largest sampled value (q = 12 for k = 32).  Any q values bound the k-th largest of
//            the full row from below unless q of the true top-k happen to sit in the sample (2e-7 per token), so
//            T_n = s_q - 2 B_n is a threshold that ~16 q = 190 of the 24 576 values pass
//   filter   the full encoder product on v_mfma_f32_32x32x16_f16 (operands rounded to fp16: a_j = approx hidden_pre,
//            |a_j - exact| <= B_n, see below); the epilogue keeps no tile -- it appends (j, a_j) for a_j >= T_n to the
//            token's candidate list (one atomic per hit, ~0.8 % of the elements)
//   select   per token: t = k-th largest a; if t >= s_q every feature with a_j >= t - 2 B_n is in the list, and that set
//            contains the exact top-k.  Those (~k + 7) candidates are re-scored EXACTLY in fp32 against a transposed
//            fp32 copy of W_enc (coalesced 3 KB rows) and ranked by (value desc, index asc): index sets and values are
//            those of the exact-fp32 path
//   fallback tokens the filter cannot decide (t < s_q, list overflow = massive ties, > 192 candidates in the band,
//            fp16 overflow) are recomputed exactly (fp32 row into the hidden scratch + the streaming / radix top-k of
//            sae.hip).  Expected: none on ordinary data; everything on adversarial data; never a wrong answer.
//
// Error bound of the filter (x = sae_in row, w = W_enc column, K = d_in; fp16 has an 11-bit significand, u = 2^-11):
//   |x.w - fl16(x).fl16(w)| <= (2u + u^2) sum|x_i w_i| + 2^-25 (sum|x_i| + sum|w_i|)      (operand rounding, subnormals)
//   fp32 accumulation of exact products on the matrix core, and the fp32 re-scoring it is compared with:
//                           <= 4 K 2^-24 sum|x_i w_i| + K 2^-24 sum|x_i w_i|
//   => B_n = ||x_n||_2 (C1 Wmax + C2) + C2 Wmax,  C1 = max(1.25e-3, 2^-10 (1 + 2^-12) + 5 K 2^-24) (enc_c1(K): 1.25e-3 up to K = 916,
//      1.36e-3 at K = 1280),
//      C2 = 2^-25 sqrt(K),  Wmax = max_j ||W_enc[:, j]||_2 (maintained by the Adam kernel as enc_colsq).
// Rows with |x| beyond the fp16 range get B = inf (-> fallback); a weight beyond it makes its column norm inf (-> every
// row falls back): slow, never wrong.
#include <atomic>

#include "sae.hpp"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct EncParams {
    const void* A;            // [M][lda] fp16
    const void* B;            // fp16 rows; row of output column c starts at B + c * ldb_bytes
    const float* bias;        // bias of column c = bias[c * bias_stride]
    int32_t M, N, K, lda;
    uint32_t ldb_bytes, b_span;
    int32_t bias_stride;
    float* out;               // MODE 0: [M][ldo] fp32 (sample)
    int32_t ldo;
    const float* thr;         // MODE 1: [M] thresholds
    uint32_t* cnt;            //         [M][ntn] hits of (token, N-tile); 0xffffffff = more than `slots`
    int2* cand;               //         [M][ntn][slots] (feature, bits of a)
    int32_t slots;            //         pv_sae_tile_slots(plan)
    uint64_t* trace;          // debug build (-DPV_ENC_TRACE): per-workgroup wall-clock stamps {start, K loop done, end, hw id}, else unused
    uint32_t* mode;           // ReLU filter (pv_sae_relu_step) or NULL: the step's mode word -- a row with more hits than slots raises it
                              // (the step then runs dense), and a tile that finds it raised skips its hit lists
};

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// ---------------------------------------------------------------------------------------------------
// filter GEMM: the v7 mainloop of gemm.hip (one 8-wave workgroup per CU, 256 x 256 tile, 64-byte K slabs DMA'd into a
// 4-slot LDS ring three slabs ahead, counted vmcnt across a raw s_barrier) on fp16 operands, with an epilogue that
// stores nothing but the hits.  MODE 0: plain fp32 store of acc + bias (the sample of pass 0).
// ---------------------------------------------------------------------------------------------------
// MB_: 32-row blocks per wave along M -- 4 (256 x 256 tiles) everywhere but the sample pass at the bench shape, whose 4096 x 1536 output
// is 96 such tiles on 256 CUs: 2 (128 x 256 tiles, the full-line loop only) makes it 192 half-sized ones, still one round
template <int MODE, int LP = 0, bool ONE = false, int MB_ = 4>
__global__ __launch_bounds__(512, 2) void sae_enc_gemm_kernel(const EncParams p) {
    constexpr int MB = MB_, TM = 64 * MB, TN = 256;
    static_assert(MB == 4 || (MB == 2 && LP == 2 && MODE == 0), "the 128-row tile exists in the full-line form of the sample pass only");
    constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, SLOT = A_BYTES + B_BYTES;
    // LP == 2 (full-line K slabs): two slots of 128-byte rows in ring0 / ring1, ring2 / ring3 shrink to stubs
    __shared__ __attribute__((aligned(16))) unsigned char ring0[LP == 2 ? 2 * SLOT : SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char ring1[LP == 2 ? 2 * SLOT : SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char ring2[LP == 2 ? 16 : SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char ring3[LP == 2 ? 16 : SLOT];
    __shared__ __attribute__((aligned(16))) float trow[512];
    __shared__ uint32_t rowcnt[256];
    __shared__ uint32_t hit_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // tile order as in gemm_kernel_v7: column blocks of <= 8 N-tiles, M-major inside (a block's weight panel -- 8 x 256 rows
    // x K fp16 = 3 MB at K = 768 -- stays in the XCD's L2 while the token rows stream past)
    const int ntn = (p.N + TN - 1) / TN, ntm = (p.M + TM - 1) / TM;
    const int nblk = (ntn + 7) / 8;
    const int wblk = (ntn + nblk - 1) / nblk;
    const int blk = swz / (ntm * wblk);
    const int rem = swz - blk * (ntm * wblk);
    const int wcur = min(wblk, ntn - blk * wblk);
    const int tile_m = rem / wcur, tile_n = blk * wblk + (rem - tile_m * wcur);
    const int m0 = tile_m * TM, n0 = tile_n * TN;

    if constexpr (MODE == 1) {
        if (p.mode) {                                             // (ReLU filter) the step is dense already: nothing to filter
            if (tid == 0) hit_n = __hip_atomic_load(p.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool skip = hit_n != 0u;
            __syncthreads();
            if (skip) return;
        }
    }
#ifdef PV_ENC_TRACE
#define ENC_STAMP(slot)                                                                                                    \
    do {                                                                                                                   \
        if (MODE == 1 && p.trace && tid == 0) {                                                                            \
            p.trace[(int64_t)blockIdx.x * 4 + (slot)] = wall_clock64();                                                    \
            if ((slot) == 0) p.trace[(int64_t)blockIdx.x * 4 + 3] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | \
                                                                    (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);    \
        }                                                                                                                  \
    } while (0)
#else
#define ENC_STAMP(slot) do { } while (0)
#endif
    ENC_STAMP(0);
    const unsigned Kb = (unsigned)p.K * 2u;
    const int nk = (int)((Kb + 63) / 64);
    const bool ktail = (Kb % 64) != 0;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.A), 0, (int)((unsigned)p.M * (unsigned)p.lda * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)p.b_span, 0x00020000);

    unsigned offA[2], kcA[2], offB[2], kcB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 8 + wave) * 16 + (lane >> 2);
        const int kc = (lane & 3) ^ ((row >> 2) & 3);
        kcA[j] = kcB[j] = kc * 16;
        offA[j] = (unsigned)(m0 + row) * (unsigned)p.lda * 2u + kc * 16;       // rows >= M land past the descriptor: zero fill
        offB[j] = (n0 + row < p.N) ? (unsigned)(n0 + row) * p.ldb_bytes + kc * 16 : 0xffffff00u;
    }
    auto issue = [&](int kt, unsigned char* slot) {
        const unsigned kbase = (unsigned)kt * 64;
        const bool dead = kt >= nk;
        // (selects only: a branch around an LDS-DMA makes hipcc drain the queue before the next ds_read)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned o = offA[j] + kbase;
            o = (dead | (ktail & (kbase + kcA[j] >= Kb))) ? 0xffffff00u : o;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(slot + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned o = offB[j] + kbase;
            o = (dead | (offB[j] == 0xffffff00u) | (ktail & (kbase + kcB[j] >= Kb))) ? 0xffffff00u : o;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(slot + A_BYTES + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
        }
    };

    f32x16 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int l31 = lane & 31, half = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    const int co0 = ((0 + half) ^ sw) * 16, co1 = ((2 + half) ^ sw) * 16;
    const int a_row = (wm * 32 * MB + l31) * 64;
    const int b_row = A_BYTES + (wn * 64 + l31) * 64;
    auto compute = [&](const unsigned char* slot) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = j == 0 ? co0 : co1;
            uint4 a[MB], b[2];
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) a[mi] = *reinterpret_cast<const uint4*>(slot + a_row + mi * 2048 + co);
            b[0] = *reinterpret_cast<const uint4*>(slot + b_row + co);
            b[1] = *reinterpret_cast<const uint4*>(slot + b_row + 2048 + co);
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, a[mi]), __builtin_bit_cast(f16x8, b[ni]), acc[mi][ni], 0, 0, 0);
        }
    };

    // epilogue operands in flight before the K loop: this lane's two bias values and (MODE 1) the tile's 256 thresholds,
    // DMA'd into LDS (they are the oldest entries of the vmcnt queue, so the first counted wait retires them)
    const int colb = n0 + wn * 64 + l31;
    float bias[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bias[ni] = (colb + ni * 32 < p.N) ? p.bias[(int64_t)(colb + ni * 32) * p.bias_stride] : 0.0f;
    if constexpr (MODE == 1) {
        const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.thr), 0, p.M * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, (lds_ptr_t)(trow + wave * 64), 4, (unsigned)(m0 + wave * 64 + lane) * 4u, 0, 0, 0);
    }

    if constexpr (LP == 0) {
    // step kt: slab kt must have landed -- the 8 DMA instructions of slabs kt+1, kt+2 may stay in flight
#define PV_ENC_STEP(KT, CUR, NXT3)                 \
    __builtin_amdgcn_s_waitcnt(0x0F70 | 8);        \
    __builtin_amdgcn_s_barrier();                  \
    issue((KT) + 3, NXT3);                         \
    compute(CUR);

    issue(0, ring0);
    issue(1, ring1);
    issue(2, ring2);
    int kt = 0;
    for (; kt + 4 <= nk; kt += 4) {
        PV_ENC_STEP(kt, ring0, ring3)
        PV_ENC_STEP(kt + 1, ring1, ring0)
        PV_ENC_STEP(kt + 2, ring2, ring1)
        PV_ENC_STEP(kt + 3, ring3, ring2)
    }
    if (kt < nk) { PV_ENC_STEP(kt, ring0, ring3) }
    if (kt + 1 < nk) { PV_ENC_STEP(kt + 1, ring1, ring0) }
    if (kt + 2 < nk) { PV_ENC_STEP(kt + 2, ring2, ring1) }
#undef PV_ENC_STEP
    } else if constexpr (LP == 1) {
        // software-pipelined form of the loop (gemm.hip, gemm_kernel_v7<..., LP = 1>): fragments of a half-slab refilled
        // right behind the MFMA pair that consumed them, the slab's barrier between its two halves, the DMA pieces of
        // slab s+4 one per MFMA pair in the second half.  Whole 64-byte slabs only (the launcher checks K % 32 == 0).
        constexpr int NPIECE = 4;
        const unsigned pA0 = (unsigned)(m0 + wave * 16 + (lane >> 2)) * (unsigned)p.lda * 2u + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
        const unsigned strideA = 128u * (unsigned)p.lda * 2u;
        auto issue_piece = [&](int kt, unsigned char* slot, int j) {
            const unsigned kbase = (unsigned)kt * 64;
            const bool dead = kt >= nk;
            if (j < 2) {
                const unsigned o = dead ? 0xffffff00u : pA0 + ((unsigned)j * strideA + kbase);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(slot + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
            } else {
                const int jb = j - 2;
                const unsigned o = (dead | (offB[jb] == 0xffffff00u)) ? 0xffffff00u : offB[jb] + kbase;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(slot + A_BYTES + (jb * 8 + wave) * 1024), 16, o, 0, 0, 0);
            }
        };
        auto issue_all = [&](int kt, unsigned char* slot) {
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) issue_piece(kt, slot, j);
        };
        auto rdA = [&](const unsigned char* slot, int h, int mi) {
            return *reinterpret_cast<const uint4*>(slot + a_row + mi * 2048 + (h == 0 ? co0 : co1));
        };
        auto rdB = [&](const unsigned char* slot, int h, int ni) {
            return *reinterpret_cast<const uint4*>(slot + b_row + ni * 2048 + (h == 0 ? co0 : co1));
        };
        uint4 fa[MB], fb0[2], fb1[2];
#define PV_ENC_PAIR(MI, FB)                                                                                  \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                     \
            acc[MI][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                            \
                __builtin_bit_cast(f16x8, fa[MI]), __builtin_bit_cast(f16x8, FB[ni]), acc[MI][ni], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);
#define PV_ENC_PSTEP(KT, CUR, NXT)                                                                           \
        _Pragma("unroll") for (int mi = 0; mi < MB; ++mi) {                                                  \
            PV_ENC_PAIR(mi, fb0)                                                                             \
            fa[mi] = rdA(CUR, 1, mi);                                                                        \
            if (mi == 1) { fb1[0] = rdB(CUR, 1, 0); fb1[1] = rdB(CUR, 1, 1); }                               \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
        __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NPIECE));                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        _Pragma("unroll") for (int mi = 0; mi < MB; ++mi) {                                                  \
            PV_ENC_PAIR(mi, fb1)                                                                             \
            fa[mi] = rdA(NXT, 0, mi);                                                                        \
            if (mi == 1) { fb0[0] = rdB(NXT, 0, 0); fb0[1] = rdB(NXT, 0, 1); }                               \
            issue_piece((KT) + 4, CUR, mi);                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }
        issue_all(0, ring0);
        issue_all(1, ring1);
        issue_all(2, ring2);
        issue_all(3, ring3);
        __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * NPIECE));
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) fa[mi] = rdA(ring0, 0, mi);
        fb0[0] = rdB(ring0, 0, 0); fb0[1] = rdB(ring0, 0, 1);
        int kt = 0;
        for (; kt + 4 <= nk; kt += 4) {
            PV_ENC_PSTEP(kt, ring0, ring1)
            PV_ENC_PSTEP(kt + 1, ring1, ring2)
            PV_ENC_PSTEP(kt + 2, ring2, ring3)
            PV_ENC_PSTEP(kt + 3, ring3, ring0)
        }
        if (kt < nk) { PV_ENC_PSTEP(kt, ring0, ring1) }
        if (kt + 1 < nk) { PV_ENC_PSTEP(kt + 1, ring1, ring2) }
        if (kt + 2 < nk) { PV_ENC_PSTEP(kt + 2, ring2, ring3) }
#undef PV_ENC_PSTEP
#undef PV_ENC_PAIR
    } else {
        // full-line form (gemm.hip, gemm_kernel_v7<..., LP = 2>): 128-byte K slabs = whole cache lines per DMA piece (8 rows x
        // 128 B), two 64 KB slots, four 16-element k-steps per slab, the next slab's barrier before the last k-step, vmcnt(0)
        // there (one slab of prefetch distance).  Whole 128-byte slabs only (the launcher checks K % 64 == 0).
        constexpr int A2 = TM * 128, NP2 = MB + 4;
        const int prow = lane >> 3;
        const int psw = ((lane >> 4) + 4 * (wave & 1)) & 7;
        const unsigned pcol = (unsigned)(((lane & 7) ^ psw) * 16);
        const unsigned pA0 = (unsigned)(m0 + wave * 8 + prow) * (unsigned)p.lda * 2u + pcol;
        const int brow0 = n0 + wave * 8 + prow;
        const unsigned pB0 = (unsigned)brow0 * p.ldb_bytes + pcol;
        const unsigned strideA = 64u * (unsigned)p.lda * 2u, strideB = 64u * p.ldb_bytes;
        const int nk2 = (int)(Kb / 128);
        auto issue_piece2 = [&](int kt, unsigned char* slot, int j) {
            const unsigned kbase = (unsigned)kt * 128;
            const bool dead = kt >= nk2;
            if (j < MB) {
                const unsigned o = dead ? 0xffffff00u : pA0 + ((unsigned)j * strideA + kbase);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(slot + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
            } else {
                const int jb = j - MB;
                const unsigned o = (dead | (brow0 + jb * 64 >= p.N)) ? 0xffffff00u : pB0 + ((unsigned)jb * strideB + kbase);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(slot + A2 + (jb * 8 + wave) * 1024), 16, o, 0, 0, 0);
            }
        };
        const int fsw = (l31 >> 1) & 7;
        const int a_row2 = (wm * 32 * MB + l31) * 128, b_row2 = A2 + (wn * 64 + l31) * 128;
        int fco[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) fco[h] = ((2 * h + half) ^ fsw) * 16;
        auto rdA2 = [&](const unsigned char* slot, int h, int mi) {
            return *reinterpret_cast<const uint4*>(slot + a_row2 + mi * 4096 + fco[h]);
        };
        auto rdB2 = [&](const unsigned char* slot, int h, int ni) {
            return *reinterpret_cast<const uint4*>(slot + b_row2 + ni * 4096 + fco[h]);
        };
        uint4 fa[MB], fb[2][2];
#define PV_ENC_PAIR2(MI, H)                                                                                   \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                      \
            acc[MI][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                             \
                __builtin_bit_cast(f16x8, fa[MI]), __builtin_bit_cast(f16x8, fb[(H) & 1][ni]), acc[MI][ni], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);
#define PV_ENC_FSTEP(KT, CUR, NXT)                                                                            \
        _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                                       \
            if (h == 3) {                                                                                     \
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                           \
                __builtin_amdgcn_s_barrier();                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                            \
            }                                                                                                 \
            _Pragma("unroll") for (int mi = 0; mi < MB; ++mi) {                                               \
                PV_ENC_PAIR2(mi, h)                                                                           \
                fa[mi] = h < 3 ? rdA2(CUR, h < 3 ? h + 1 : 0, mi) : rdA2(NXT, 0, mi);                         \
                if (mi == 1) {                                                                                \
                    fb[(h + 1) & 1][0] = h < 3 ? rdB2(CUR, h < 3 ? h + 1 : 0, 0) : rdB2(NXT, 0, 0);           \
                    fb[(h + 1) & 1][1] = h < 3 ? rdB2(CUR, h < 3 ? h + 1 : 0, 1) : rdB2(NXT, 0, 1);           \
                }                                                                                             \
                if (h == 3) issue_piece2((KT) + 2, CUR, mi);                                                  \
                if (h == 0) {                                                                                 \
                    _Pragma("unroll") for (int jb = mi; jb < 4; jb += MB) issue_piece2((KT) + 1, NXT, MB + jb); \
                }                                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                            \
            }                                                                                                 \
        }
#pragma unroll
        for (int j = 0; j < NP2; ++j) issue_piece2(0, ring0, j);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < MB; ++j) issue_piece2(1, ring1, j);
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) fa[mi] = rdA2(ring0, 0, mi);
        fb[0][0] = rdB2(ring0, 0, 0); fb[0][1] = rdB2(ring0, 0, 1);
        int kt = 0;
        for (; kt + 2 <= nk2; kt += 2) {
            PV_ENC_FSTEP(kt, ring0, ring1)
            PV_ENC_FSTEP(kt + 1, ring1, ring0)
        }
        if (kt < nk2) { PV_ENC_FSTEP(kt, ring0, ring1) }
#undef PV_ENC_FSTEP
#undef PV_ENC_PAIR2
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): drain the off-the-end prefetches (and the threshold DMA)
    __syncthreads();
    ENC_STAMP(2);

    const int rows_left = p.M - (m0 + wm * 32 * MB);       // local row r of this wave's block is real iff r < rows_left
    if constexpr (MODE == 0) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (row < rows_left) {
                    float* o = p.out + (int64_t)(m0 + wm * 32 * MB + row) * p.ldo + colb;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        if (colb + ni * 32 < p.N) o[ni * 32] = acc[mi][ni][e] + bias[ni];
                }
            }
    } else {
        // Hits are ~0.8 % of the tile (~2 per token row).  Nothing global is contended for them: every (token, N-tile) pair
        // owns pv_sae_tile_slots() candidate slots and a count word that only this workgroup writes (a device-scope atomic
        // per hit on per-token counters cost more than the whole K loop: ~0.8 M atomics per step queue up behind a few dozen
        // memory channels).  The hits are compacted into LDS with no round trip -- a lane builds the bitmaps of its
        // accumulators, a wave scan + ONE LDS atomic per wave hands out list positions -- then every
        // list entry draws its slot from an LDS per-row counter and is stored.  Rows with more hits than slots (or a tile
        // with more than HCAP hits: massive ties) are marked overflowed: those tokens take the exact path.
        constexpr int HCAP = (LP == 2 ? 2 * SLOT : SLOT) / 8;     // (ring0: SLOT bytes, 2 SLOT in the full-line form)
        uint2* hlist = reinterpret_cast<uint2*>(ring0);
        if (tid == 0) hit_n = p.mode ? __hip_atomic_load(p.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        __syncthreads();                                           // (ONE read of the word per workgroup)
        const bool dense_already = hit_n != 0u;
        __syncthreads();
        if (dense_already) {
            // the step is dense already -- half of all features positive, as at the start of a run: compacting ~32 K hits per tile
            // would cost more than the K loop
            if (tid < 256 && m0 + tid < p.M) p.cnt[(int64_t)(m0 + tid) * ntn + tile_n] = 0xffffffffu;
            return;
        }
        if (tid == 0) hit_n = 0u;
        if (tid < 256) rowcnt[tid] = 0u;
        __syncthreads();
        const float* trw = trow + wm * 32 * MB;
        // bitmap of the hits among this lane's 32 accumulators of block mi: bit (g * 8 + s * 2 + ni)
        auto block_mask = [&](int mi) {
            uint32_t m = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r0 = mi * 32 + 8 * g + 4 * half;
                const float4 t4 = *reinterpret_cast<const float4*>(trw + r0);
                const float tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const bool hit = (r0 + s < rows_left) && (acc[mi][ni][4 * g + s] + bias[ni] >= tt[s]);
                        m |= (hit ? 1u : 0u) << (g * 8 + s * 2 + ni);
                    }
            }
            return m;
        };
        // the hits of block mi into the list from position pos on (returns the next position)
        // (eb: the bias values again, behind an asm barrier in the one-round form -- otherwise hipcc keeps all 128 sums acc + bias
        // of the bitmap pass alive for this one and spills)
        float eb[2] = {bias[0], bias[1]};
        auto block_emit = [&](int mi, uint32_t m, uint32_t pos) {
            if (m != 0u) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            if (m & (1u << (g * 8 + s * 2 + ni))) {
                                const int lrow = wm * 32 * MB + mi * 32 + 8 * g + 4 * half + s;
                                if (pos < (uint32_t)HCAP)
                                    hlist[pos] = make_uint2((uint32_t)(lrow << 8 | (wn * 64 + ni * 32 + l31)),
                                                            __float_as_uint(acc[mi][ni][4 * g + s] + eb[ni]));
                                else
                                    atomicOr(&rowcnt[lrow], 0x80000000u);
                                ++pos;
                            }
                        }
            }
            return pos;
        };
        // list entries [0, nhit) -> each draws its slot from the LDS per-row counter and is stored
        auto flush = [&](uint32_t nhit) {
            for (uint32_t e = tid; e < nhit; e += 512) {
                const uint2 h = hlist[e];
                const int lrow = (int)(h.x >> 8);
                const uint32_t li = atomicAdd(&rowcnt[lrow], 1u) & 0x7fffffffu;
                if (li < (uint32_t)p.slots)
                    p.cand[((int64_t)(m0 + lrow) * ntn + tile_n) * p.slots + li] = make_int2(n0 + (int)(h.x & 255u), (int)h.y);
            }
        };
        // ONE (the launcher: expected hits per tile = 256 rows x slots / 4 -- the slot count is 4 x the mean + 6 -- fit the list
        // twice over: the bench shape, the ReLU filter): one compaction round for the whole tile -- all bitmaps, one wave scan, one
        // LDS atomic per wave, one flush.  Otherwise (small d_sae, large k) a round per 32-row block, each with its own flush.
        if constexpr (ONE) {
            uint32_t mk[MB];
            int nh = 0;
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
                mk[mi] = block_mask(mi);
                nh += __popc(mk[mi]);
            }
            asm volatile("" : "+v"(eb[0]), "+v"(eb[1]));
            if (__ballot(nh != 0) != 0ull) {                       // (wave-uniform)
                int incl = nh;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += up;
                }
                uint32_t base = 0;
                if (lane == 63) base = atomicAdd(&hit_n, (uint32_t)incl);
                base = __shfl(base, 63, 64);
                uint32_t pos = base + (uint32_t)(incl - nh);
#pragma unroll
                for (int mi = 0; mi < MB; ++mi) pos = block_emit(mi, mk[mi], pos);
            }
            __syncthreads();
            flush(min(hit_n, (uint32_t)HCAP));
            __syncthreads();
        } else {
            uint32_t flushed = 0;                                  // list positions below this were flushed by earlier blocks
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {                      // one 32-row block (64 rows of the tile) per round
                const uint32_t m = block_mask(mi);
                if (__ballot(m != 0u) != 0ull) {                   // (wave-uniform)
                    const int nh = __popc(m);
                    int incl = nh;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int up = __shfl_up(incl, o, 64);
                        if (lane >= o) incl += up;
                    }
                    uint32_t base = 0;
                    if (lane == 63) base = atomicAdd(&hit_n, (uint32_t)incl);
                    base = __shfl(base, 63, 64);
                    block_emit(mi, m, base + (uint32_t)(incl - nh) - flushed);
                }
                // flush this block's hits (beyond HCAP of the block's 64 x 256 elements the rows are marked overflowed above).
                // hit_n keeps counting; `flushed` rebases the positions of the next block.
                __syncthreads();
                const uint32_t total = hit_n;
                flush(min(total - flushed, (uint32_t)HCAP));
                flushed = total;
                __syncthreads();
            }
        }
        ENC_STAMP(1);
        if (tid < 256 && m0 + tid < p.M) {
            const uint32_t c = rowcnt[tid];
            p.cnt[(int64_t)(m0 + tid) * ntn + tile_n] = c > (uint32_t)p.slots ? 0xffffffffu : c;
            const bool over = p.mode && c > (uint32_t)p.slots;
            const unsigned long long who = __ballot(over);         // one atomic per wave, not one per overflowed row: on a dense batch
            if (who != 0ull && lane == __ffsll((long long)who) - 1) atomicOr(p.mode, 1u);      // every row of the first 256 tiles overflows
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// thr: q-th largest of a token's sampled values, band and threshold.  One wave per token.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float enc_c1(int K) { return fmaxf(1.25e-3f, 9.7680092e-4f + 5.0f * 5.9604645e-8f * (float)K); }

template <int VPL>                                            // sampled values per lane: ns <= 64 VPL
__global__ __launch_bounds__(256) void sae_thr_kernel(const float* __restrict__ sample, int ns, const float* __restrict__ xnorm,
                                                      const float* __restrict__ wmax_sq, int qsel, int d_in, float* __restrict__ thr,
                                                      float* __restrict__ sq_out, float* __restrict__ band, int n_tok,
                                                      int nb_thr, const float* __restrict__ cs_partial, float* __restrict__ cs_out,
                                                      int cs_nblk, float cs_scale) {
    // workgroups beyond nb_thr (the fused pre-pass, SaePre): the batch mean's second stage -- 16 columns each
    if ((int)blockIdx.x >= nb_thr) {
        colsum_final_body_256(blockIdx.x - nb_thr, cs_partial, cs_out, cs_nblk, d_in, cs_scale);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    const float* s = sample + (int64_t)n * ns;
    float v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < ns ? s[c] : -INFINITY;
        v[i] = (v[i] == v[i]) ? v[i] : -INFINITY;             // NaN never bounds anything
    }
    // The q-th largest of a SUBSET of the row's values bounds the k-th largest of the row from below like the q-th largest of the whole
    // sample does (file header), only a hair lower where the subset misses one of the sample's top q.  The subset: every lane's three
    // largest (192 values; a lane holds four of the sample's top 12 about once in 10^4 tokens) -- kept sorted in three registers while
    // the lane's values stream by, so that a selection round is a wave maximum and a pop instead of a pass over all VPL registers
    // (the exact form took 12 rounds x (VPL maxima + VPL compare-and-clear): 20 us of the step; this one 9).  Multiplicity counts.
    float t0 = -INFINITY, t1 = -INFINITY, t2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float x = v[i];
        const float a = fmaxf(t0, x), b = fminf(t0, x);        // (t0, t1, t2, x) -> the three largest, in order
        const float c = fmaxf(t1, b), e = fminf(t1, b);
        t0 = a; t1 = c; t2 = fmaxf(t2, e);
    }
    float m = -INFINITY;
    for (int rnd = 0; rnd < qsel; ++rnd) {
        m = wave_max(t0);
        const unsigned long long owners = __ballot(t0 == m);
        const int owner = __ffsll((long long)owners) - 1;
        if (lane == owner) { t0 = t1; t1 = t2; t2 = -INFINITY; }      // remove ONE instance
    }
    if (lane == 0) {
        const float wmx = sqrtf(*wmax_sq);
        const float c2 = 2.98023224e-8f * sqrtf((float)d_in);
        const float B = xnorm[n] * (enc_c1(d_in) * wmx + c2) + c2 * wmx + 4.8e-7f * fabsf(m);
        sq_out[n] = m;
        band[n] = 2.0f * B;
        thr[n] = m - 2.0f * B;
    }
}

__global__ __launch_bounds__(1024) void sae_wmax_kernel(const float* __restrict__ colsq, int d_sae, float* __restrict__ out,
                                                        uint32_t* __restrict__ fb_count, uint32_t* __restrict__ feat_cnt) {
    __shared__ float red[16];
    // (also the zeroing of the per-feature pair counters the select kernel draws list positions from)
    if (feat_cnt)
        for (int j = threadIdx.x; j < d_sae; j += 1024) feat_cnt[j] = 0u;
    float m = 0.f;
    for (int j0 = threadIdx.x; j0 < d_sae; j0 += 8 * 1024) {  // 8 independent loads in flight per thread
        float c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = j0 + u * 1024 < d_sae ? colsq[j0 + u * 1024] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) m = (c[u] == c[u]) ? fmaxf(m, c[u]) : INFINITY;      // a NaN column norm poisons the bound (-> exact fallback)
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = red[0];
        for (int w = 1; w < 16; ++w) r = fmaxf(r, red[w]);
        *out = r;
        *fb_count = 0u;
    }
}

// ---------------------------------------------------------------------------------------------------
// select: candidates -> exact top-k.  One workgroup per token.
// ---------------------------------------------------------------------------------------------------
#ifndef PV_SEL_WAVES
#define PV_SEL_WAVES 8        // waves per SIMD the select kernel is compiled for (see the kernel)
#endif
// The k-th largest (with multiplicity) of nc <= 64 KPL ordered keys in LDS, by ONE wave: KPL keys per lane in registers, a radix select
// from the highest bit in which the keys differ -- a ballot + popcount per key and bit -- that stops as soon as the survivors are exactly
// the ones still wanted (~10 bits on ordinary data).  Every lane returns the key.
template <int KPL>
__device__ __forceinline__ uint32_t sel_radix_kth(const uint32_t* ckey, uint32_t nc, uint32_t k, int lane) {
    uint32_t key[KPL];
    uint32_t alive = 0u;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const uint32_t c = (uint32_t)lane + 64u * i;
        key[i] = c < nc ? ckey[c] : 0u;
        alive |= c < nc ? (1u << i) : 0u;
    }
    auto alive_minmax = [&](uint32_t& mn, uint32_t& mx) {
        mn = 0xffffffffu; mx = 0u;
#pragma unroll
        for (int i = 0; i < KPL; ++i)
            if (alive & (1u << i)) { mn = min(mn, key[i]); mx = max(mx, key[i]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        }
    };
    uint32_t kmin, kmax;
    alive_minmax(kmin, kmax);
    uint32_t krem = k, n_alive = nc;
    if (kmin != kmax) {
        for (int b = 31 - __clz((int)(kmin ^ kmax)); b >= 0; --b) {       // (everything below is wave-uniform)
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < KPL; ++i)
                cnt += (uint32_t)__popcll(__ballot(((alive >> i) & 1u) != 0u && ((key[i] >> b) & 1u) != 0u));
            const bool take1 = cnt >= krem;                               // the k-th largest has this bit set
#pragma unroll
            for (int i = 0; i < KPL; ++i)
                if ((((key[i] >> b) & 1u) != 0u) != take1) alive &= ~(1u << i);
            if (take1) n_alive = cnt;
            else { krem -= cnt; n_alive -= cnt; }
            if (n_alive == krem || krem == 1u) break;
        }
        alive_minmax(kmin, kmax);
    }
    // survivors == wanted: the smallest of them; one wanted (or all survivors equal): the largest
    return n_alive == krem ? kmin : kmax;
}

// the same for any nc <= 1024 (PV_SAE_CAND_CAP): the keys stay in LDS, a 16-bit mask per lane says which of its (up to 16) keys are alive
__device__ __forceinline__ uint32_t sel_radix_kth_lds(const uint32_t* ckey, uint32_t nc, uint32_t k, int lane) {
    const int nk = (int)((nc + 63u) >> 6);                     // keys per lane (wave-uniform)
    uint32_t alive = 0u;
    for (int i = 0; i < nk; ++i) alive |= ((uint32_t)lane + 64u * i) < nc ? (1u << i) : 0u;
    auto alive_minmax = [&](uint32_t& mn, uint32_t& mx) {
        mn = 0xffffffffu; mx = 0u;
        for (int i = 0; i < nk; ++i)
            if (alive & (1u << i)) { const uint32_t key = ckey[lane + 64 * i]; mn = min(mn, key); mx = max(mx, key); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        }
    };
    uint32_t kmin, kmax;
    alive_minmax(kmin, kmax);
    uint32_t krem = k, n_alive = nc;
    if (kmin != kmax) {
        for (int b = 31 - __clz((int)(kmin ^ kmax)); b >= 0; --b) {       // (everything below is wave-uniform)
            uint32_t cnt = 0, ones = 0u;
            for (int i = 0; i < nk; ++i) {
                const bool one = ((alive >> i) & 1u) != 0u && ((ckey[lane + 64 * i] >> b) & 1u) != 0u;
                cnt += (uint32_t)__popcll(__ballot(one));
                ones |= one ? (1u << i) : 0u;
            }
            const bool take1 = cnt >= krem;                               // the k-th largest has this bit set
            alive = take1 ? ones : (alive & ~ones);
            if (take1) n_alive = cnt;
            else { krem -= cnt; n_alive -= cnt; }
            if (n_alive == krem || krem == 1u) break;
        }
        alive_minmax(kmin, kmax);
    }
    return n_alive == krem ? kmin : kmax;
}

#ifdef PV_SEL_TRACE       // (debug build, tools/build_variant.sh: per-workgroup time stamps of the select kernel into the hidden scratch -- 100 MHz wall clock)
#define SEL_STAMP(slot)                                                                                                   \
    do {                                                                                                                  \
        if (hidden && threadIdx.x == 0) {                                                                                 \
            uint64_t* tr = reinterpret_cast<uint64_t*>(hidden);                                                           \
            tr[(int64_t)blockIdx.x * 4 + (slot)] = wall_clock64();                                                        \
            if ((slot) == 0) tr[(int64_t)blockIdx.x * 4 + 3] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | \
                                                              (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);        \
        }                                                                                                                 \
    } while (0)
#else
#define SEL_STAMP(slot) do { } while (0)
#endif
// INLINE_FB (the folded training step): a token the filter cannot decide is recomputed exactly by ITS OWN workgroup, right here -- the
// row of exact pre-activations in sae_fb_hidden_kernel's arithmetic (a wave per feature, the same fma chain, wave_sum), then
// sae_topk_row_lds on the LDS of the candidate pass -- instead of being listed for the two fallback launches, which cost 9 us + two
// dispatch gaps per step when their list is empty, i.e. on every step of ordinary data.  A workgroup that takes the exact path runs
// ~0.3 ms (24 576 x 768 fp32 MACs on four waves); a batch in which EVERY token does (adversarial data) takes about as long as the
// 32 x 24 workgroups of the listed form did.
template <int V4, bool INLINE_FB = false>
__global__ __launch_bounds__(256, PV_SEL_WAVES) void sae_select_kernel(
    const float* __restrict__ sae_in, const float* __restrict__ W_encT, const float* __restrict__ b_enc,
    const uint32_t* __restrict__ tile_cnt, const int2* __restrict__ cand, const float* __restrict__ sq,
    const float* __restrict__ band, int32_t* __restrict__ idx_out, float* __restrict__ val_out, int32_t* __restrict__ fb_list,
    uint32_t* __restrict__ fb_count, uint32_t* __restrict__ feat_cnt, uint32_t* __restrict__ wpos, int d, int k, int ntn, int slots,
    const float* __restrict__ xn = nullptr, const float* __restrict__ batch_mean = nullptr, float* __restrict__ norm_out = nullptr,
    int d_true = 0, float* __restrict__ hidden = nullptr) {
#ifdef PV_SEL_CAP_TEST        // (timing experiment, tools/build_variant.sh: a smaller candidate list = less LDS per workgroup = more workgroups per CU?)
    constexpr int SEL_CAP = PV_SEL_CAP_TEST;
#else
    constexpr int SEL_CAP = PV_SAE_CAND_CAP;
#endif
    __shared__ uint32_t ckey[SEL_CAP];
    __shared__ int32_t cidx[SEL_CAP];
    __shared__ int32_t ridx[PV_SAE_RESCORE_MAX];
    __shared__ float rval[PV_SAE_RESCORE_MAX];
    __shared__ uint32_t tcnt[256];
    __shared__ uint32_t sh_t, sh_nr, sh_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    SEL_STAMP(0);
    // (the fused pre-pass, SaePre) the loss normaliser ||x_n - mean_batch(x)||_2 that prep left out because the mean was not there yet:
    // sae_prep_kernel's loop in its order, on the wave that has nothing to do while the candidate lists are gathered.  The loads leave
    // here; they are used behind the candidate pass (the first barrier of the kernel would otherwise wait for this wave's HBM trip)
    float nx[4 * V4], nb[4 * V4];
    const bool norm_wave = xn != nullptr && wave == 3;
    if (norm_wave) {
        const float* xr = xn + row * d;
#pragma unroll
        for (int u = 0; u < 4 * V4; ++u) {
            const int i = lane + 64 * u;
            nx[u] = i < d_true ? xr[i] : 0.f;
            nb[u] = i < d_true ? batch_mean[i] : 0.f;
        }
    }
    // gather the token's candidates: ntn (<= 256: d_sae <= 65536) per-tile lists of <= slots entries
    uint32_t myc = 0;
    if (tid < ntn) myc = tile_cnt[row * ntn + tid];
    tcnt[tid] = tid < ntn ? myc : 0u;
    if (tid == 0) { sh_t = 0u; sh_nr = 0u; sh_bad = 0u; }
    __syncthreads();
    if (tid < ntn && myc == 0xffffffffu) sh_bad = 1u;
    __syncthreads();
    bool bad = sh_bad != 0u;
    uint32_t n = 0, off = 0;
    if (!bad) {
        for (int t = 0; t < ntn; ++t) {
            const uint32_t c = tcnt[t];
            off += t < tid ? c : 0u;
            n += c;
        }
        bad = n > (uint32_t)SEL_CAP || n < (uint32_t)k;
    }
    if (!bad && tid < ntn && myc > 0u) {
        const int2* src = cand + (row * ntn + tid) * slots;
        // (the usual list is 1-4 entries: two independent 16-byte loads instead of a dependent chain)
        const int4 e01 = *reinterpret_cast<const int4*>(src), e23 = *reinterpret_cast<const int4*>(src + 2);
        const int2 first[4] = {{e01.x, e01.y}, {e01.z, e01.w}, {e23.x, e23.y}, {e23.z, e23.w}};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((uint32_t)e < myc) {
                ckey[off + e] = f2ord(__int_as_float(first[e].y));
                cidx[off + e] = first[e].x;
            }
        for (uint32_t e = 4; e < myc; ++e) {
            const int2 c = src[e];
            ckey[off + e] = f2ord(__int_as_float(c.y));
            cidx[off + e] = c.x;
        }
    }
    const uint32_t nc = bad ? 0u : n;
    __syncthreads();
    // t = the k-th largest filter value (with multiplicity).  Round 3 ranked every candidate against every other out of LDS
    // (~190 x 190 compares per token: 75 of the kernel's 133 us, measured by running the pass twice); now ONE wave runs a radix
    // select over the ordered keys in registers -- four keys per lane, a ballot + popcount per key and bit, starting below the
    // keys' common prefix and stopping as soon as the survivors are exactly the ones still wanted (~10 bits on ordinary data) --
    // while the other waves wait at the barrier.  More than 256 candidates (heavy ties, small d_sae): the all-pairs ranking.
#ifdef PV_SEL_RANK2                                            // (timing ablation, tools/build_variant.sh: the pass twice)
    for (int rep = 0; rep < 2; ++rep)
#endif
    if (!bad && wave == 0) {
        // (up to 256 candidates: four keys per lane in registers; more -- 1 % of the bench batch's tokens, which an all-pairs ranking
        // out of LDS kept in this kernel for 55 us while the others were done in 10, the kernel's tail -- the same select with the keys
        // re-read from LDS in every bit round: sixteen reads per lane and bit at the list's capacity)
        uint32_t kth;
        if (nc <= 256u) kth = sel_radix_kth<4>(ckey, nc, (uint32_t)k, lane);
        else kth = sel_radix_kth_lds(ckey, nc, (uint32_t)k, lane);
        if (lane == 0) sh_t = kth;
    }
    __syncthreads();
    const float t = ord2f(sh_t);
    bad = bad || !(t >= sq[row]);                              // band below t not guaranteed to be in the list (or NaN)
    const float lo = t - band[row];
    if (!bad) {
        for (uint32_t c = tid; c < nc; c += 256) {
            if (ord2f(ckey[c]) >= lo) {
                const uint32_t pos = atomicAdd(&sh_nr, 1u);
                if (pos < (uint32_t)PV_SAE_RESCORE_MAX) ridx[pos] = cidx[c];
#ifdef PV_SEL_NO_RESCORE                                       // (timing ablation: the filter value stands in for the exact one)
                if (pos < (uint32_t)PV_SAE_RESCORE_MAX) rval[pos] = ord2f(ckey[c]);
#endif
            }
        }
    }
    __syncthreads();
    const uint32_t nr = sh_nr;
    bad = bad || nr > (uint32_t)PV_SAE_RESCORE_MAX;
    SEL_STAMP(2);                                              // (candidate pass + threshold select done)
    if (norm_wave) {
        float cn = 0.f;
#pragma unroll
        for (int u = 0; u < 4 * V4; ++u) {                     // (terms beyond d_true are exact zeros: they do not move the sum)
            const float c = nx[u] - nb[u];
            cn += c * c;
        }
        cn = wave_sum(cn);
        if (lane == 0) norm_out[row] = sqrtf(cn);
    }
    if (bad) {                                                 // (uniform over the workgroup)
        if constexpr (!INLINE_FB) {
            if (tid == 0) fb_list[atomicAdd(fb_count, 1u)] = (int32_t)row;
            return;
        } else {
            if (tid == 0) atomicAdd(fb_count, 1u);             // (the count stays: NativeSAE.fallback_rows)
            const int d_sae = ntn * 256;
            float4 xq[V4];
#pragma unroll
            for (int i = 0; i < V4; ++i)
                xq[i] = 4 * lane + 256 * i < d ? *reinterpret_cast<const float4*>(sae_in + row * d + 4 * lane + 256 * i)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j0 = wave * 4; j0 < d_sae; j0 += 16) {
                float acc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* w = W_encT + (int64_t)(j0 + u) * d;
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < V4; ++i) {
                        if (4 * lane + 256 * i < d) {
                            const float4 wv = *reinterpret_cast<const float4*>(w + 4 * lane + 256 * i);
                            a = fmaf(xq[i].x, wv.x, a); a = fmaf(xq[i].y, wv.y, a); a = fmaf(xq[i].z, wv.z, a); a = fmaf(xq[i].w, wv.w, a);
                        }
                    }
                    acc[u] = a;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = wave_sum(acc[u]);
                if (lane < 4) {
                    const float av = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
                    hidden[row * d_sae + j0 + lane] = av + b_enc[j0 + lane];
                }
            }
            __threadfence_block();
            __syncthreads();                                   // the row is in memory for the whole workgroup
            sae_topk_row_lds(hidden, idx_out, val_out, d_sae, k, row, feat_cnt, wpos, tcnt, ckey, cidx);
            return;
        }
    }
#ifndef PV_SEL_NO_RESCORE
    // exact fp32 re-scoring: a wave per candidate, four candidates (4 x V4 16-byte loads per lane) in flight
    bool ok[V4];
    int col[V4];
    float4 xr[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        col[i] = 4 * lane + 256 * i;
        ok[i] = col[i] < d;
        xr[i] = ok[i] ? *reinterpret_cast<const float4*>(sae_in + row * d + col[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (uint32_t c0 = wave; c0 < nr; c0 += 16) {
        int jj[4];
        float acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t c = c0 + 4 * u;
            jj[u] = ridx[c < nr ? c : c0];
            const float* w = W_encT + (int64_t)jj[u] * d;
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                if (ok[i]) {
                    const float4 wv = *reinterpret_cast<const float4*>(w + col[i]);
                    a = fmaf(xr[i].x, wv.x, a); a = fmaf(xr[i].y, wv.y, a); a = fmaf(xr[i].z, wv.z, a); a = fmaf(xr[i].w, wv.w, a);
                }
            }
            acc[u] = a;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += __shfl_xor(acc[u], o, 64);
        if (lane < 4) {
            const uint32_t c = c0 + 4 * lane;
            const float av = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
            const int jv = lane == 0 ? jj[0] : (lane == 1 ? jj[1] : (lane == 2 ? jj[2] : jj[3]));
            if (c < nr) rval[c] = av + b_enc[jv];
        }
    }
#endif
    __syncthreads();
    for (uint32_t c = tid; c < nr; c += 256) {
        const float vc = rval[c];
        const int32_t ic = ridx[c];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < nr; ++o) {
            const float vo = rval[o];
            rank += (vo > vc) || (vo == vc && ridx[o] < ic);
        }
        if (rank < (uint32_t)k) {
            const float v = fmaxf(vc, 0.f);                    // postact_fn = ReLU (sae.py:806)
            idx_out[row * k + rank] = ic;
            val_out[row * k + rank] = v;
            // position of this pair in its feature's list (CSR of the backward), drawn while the result is written
            if (feat_cnt) wpos[row * k + rank] = v > 0.f ? atomicAdd(&feat_cnt[ic], 1u) : 0xffffffffu;
        }
    }
#ifdef PV_SEL_TRACE
    __syncthreads();
    SEL_STAMP(1);
#endif
}

// exact fp32 hidden_pre rows of the tokens the filter could not decide -> hidden scratch.  Reads W_enc as W_encT (the fp32
// master the Adam kernel keeps; the parameter's own layout may be stale): a wave per feature, the 3 KB row coalesced against
// the token's row in LDS.
__global__ __launch_bounds__(256) void sae_fb_hidden_kernel(const float* __restrict__ sae_in, const float* __restrict__ W_encT,
                                                            const float* __restrict__ b_enc, const int32_t* __restrict__ fb_list,
                                                            const uint32_t* __restrict__ fb_count, float* __restrict__ hidden,
                                                            int d, int d_sae) {
    __shared__ float xs[1280];
    const uint32_t nfb = *fb_count;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int jper = (d_sae + gridDim.y - 1) / gridDim.y;
    const int j_lo = blockIdx.y * jper, j_hi = min(d_sae, j_lo + jper);
    for (uint32_t s = blockIdx.x; s < nfb; s += gridDim.x) {
        const int64_t row = fb_list[s];
        __syncthreads();
        for (int i = threadIdx.x; i < d; i += 256) xs[i] = sae_in[row * d + i];
        __syncthreads();
        for (int j = j_lo + wv; j < j_hi; j += 4) {
            const float* w = W_encT + (int64_t)j * d;
            float a = 0.f;
            for (int i = 4 * lane; i < d; i += 256) {
                const float4 wv4 = *reinterpret_cast<const float4*>(w + i);
                a = fmaf(xs[i], wv4.x, a); a = fmaf(xs[i + 1], wv4.y, a); a = fmaf(xs[i + 2], wv4.z, a); a = fmaf(xs[i + 3], wv4.w, a);
            }
            a = wave_sum(a);
            if (lane == 0) hidden[row * d_sae + j] = a + b_enc[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// ReLU encoder in sparse form (pv_sae_relu_step): f = relu(sae_in W_enc + b_enc) as per-token lists of its POSITIVE entries.
// "ReLU is top-k with threshold 0 and a variable k": the same fp16 filter GEMM runs with the per-token threshold -B_n (the
// error band of the fp16 product, see the file header: exact_j > 0  =>  a_j > -B_n), every survivor is re-scored EXACTLY in fp32
// against W_encT, and the ones whose exact value is positive are the token's activations -- values those of the exact fp32
// encoder.  A token's list holds at most `cap` pairs; a token with more positives than that, a candidate slot that overflowed or
// more than PV_SAE_CAND_CAP survivors raise the step's `mode` word: the step then runs on the dense GEMMs instead (sae_dense.hip),
// so nothing is ever approximated.
// ---------------------------------------------------------------------------------------------------
// mode[0] = the step's mode word, mode[1] = dense steps in a row whose sparse attempt was skipped.  The attempt itself costs ~0.5 ms when
// it fails (one filter GEMM over all features): while the run is far from the sparse regime -- the previous step ran dense and kept more
// than twice the capacity per token on average (scalars[2] = its l0) -- the step goes dense at once (the filter GEMM and the selection
// leave on the raised word) and only every eighth such step probes.  Which form runs never changes a result.
__global__ __launch_bounds__(256) void relu_thr_kernel(const float* __restrict__ xnorm, const float* __restrict__ wmax_sq, int d_in,
                                                       float* __restrict__ thr, int n_tok, uint32_t* __restrict__ mode,
                                                       const float* __restrict__ scalars, int cap) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n == 0) {
        const bool far = mode[0] == 1u && scalars && scalars[2] > 2.0f * (float)cap;
        const uint32_t streak = far ? mode[1] + 1u : 0u;
        mode[1] = streak;
        mode[0] = (far && (streak & 7u) != 0u) ? 1u : 0u;
    }
    if (n >= n_tok) return;
    const float wmx = sqrtf(*wmax_sq);
    const float c2 = 2.98023224e-8f * sqrtf((float)d_in);
    const float B = xnorm[n] * (enc_c1(d_in) * wmx + c2) + c2 * wmx;
    thr[n] = (B == B) ? -B : -INFINITY;                        // (a NaN bound lets everything through: the slots overflow -> dense)
}

// One workgroup per token: candidates -> exact values -> the positive ones, ranked by (value desc, feature asc) -> idx / val /
// wpos [cap] (holes: val 0, wpos ~0), tok_cnt, the token's sum of activations (the L1 term).
// GATED (pv_sae_gated_step_sparse): b_enc = b_gate, the list = the features whose GATE is open (gate_pre = sae_in W_enc + b_gate > 0,
// sae.py:703-706), ranked by gate_pre.  A pair carries two values: val_out = feature_acts = relu(p e^r_mag + b_mag) (0 where the
// magnitude path is shut: such a pair still carries the gate path's gradient), valg_out = relu(gate_pre).  l0part = the token's count of
// feature_acts > 0.
template <int V4, bool GATED = false>
__global__ __launch_bounds__(256) void relu_select_kernel(
    const float* __restrict__ sae_in, const float* __restrict__ W_encT, const float* __restrict__ b_enc,
    const uint32_t* __restrict__ tile_cnt, const int2* __restrict__ cand, const float* __restrict__ thr, int32_t* __restrict__ idx_out,
    float* __restrict__ val_out, uint32_t* __restrict__ tok_cnt, float* __restrict__ l1part, uint32_t* __restrict__ feat_cnt,
    uint32_t* __restrict__ wpos, uint32_t* __restrict__ mode, int d, int cap, int ntn, int slots,
    const float* __restrict__ r_mag = nullptr, const float* __restrict__ b_mag = nullptr, float* __restrict__ l0part = nullptr,
    float* __restrict__ valg_out = nullptr) {
    __shared__ int32_t cidx[PV_SAE_CAND_CAP];
    __shared__ float rval[PV_SAE_CAND_CAP];
    __shared__ float rmag[GATED ? PV_SAE_CAND_CAP : 1];
    __shared__ uint32_t sh_l0;
    __shared__ uint32_t tcnt[256];
    __shared__ float sval[256];                              // the kept values in rank order (cap <= 256)
    __shared__ float red[4];
    __shared__ uint32_t sh_bad, sh_m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    // another token has already sent the step to the dense GEMMs: nothing this workgroup computes would be used (the mode word only
    // ever goes 0 -> 1 within a step; a stale 0 costs time, never correctness)
    if (tid == 0) { sh_bad = __hip_atomic_load(mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1u : 0u; sh_m = 0u; }
    __syncthreads();                                           // (ONE read of the word per workgroup: every thread sees the same answer)
    const bool already_dense = sh_bad != 0u;
    uint32_t myc = 0;
    if (tid < ntn && !already_dense) myc = tile_cnt[row * ntn + tid];
    tcnt[tid] = tid < ntn ? myc : 0u;
    __syncthreads();
    if (tid < ntn && myc == 0xffffffffu) sh_bad = 1u;
    __syncthreads();
    bool bad = sh_bad != 0u;
    uint32_t n = 0, off = 0;
    if (!bad) {
        for (int t = 0; t < ntn; ++t) {
            const uint32_t c = tcnt[t];
            off += t < tid ? c : 0u;
            n += c;
        }
        bad = n > (uint32_t)PV_SAE_CAND_CAP;
    }
    if (!bad && tid < ntn && myc > 0u) {
        // candidates whose filter value lies above +B_n are positive for certain (exact >= a - B_n > 0): more of those than the list
        // can hold settles the matter before a single row of W_enc is gathered (the published L0 of 600 - 2000 ends here)
        const int2* src = cand + (row * ntn + tid) * slots;
        const float Bn = -thr[row];
        uint32_t sure = 0;
        for (uint32_t e = 0; e < myc; ++e) {
            const int2 c = src[e];
            cidx[off + e] = c.x;
            sure += __int_as_float(c.y) > Bn ? 1u : 0u;
        }
        if (sure) atomicAdd(&sh_m, sure);
    }
    __syncthreads();
    if (!bad && sh_m > (uint32_t)cap) bad = true;
    __syncthreads();
    if (tid == 0) sh_m = 0u;
    const uint32_t nr = bad ? 0u : n;
    // exact fp32 re-scoring of every candidate: a wave per candidate, four in flight (as in sae_select_kernel)
    bool ok[V4];
    int col[V4];
    float4 xr[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        col[i] = 4 * lane + 256 * i;
        ok[i] = col[i] < d;
        xr[i] = ok[i] ? *reinterpret_cast<const float4*>(sae_in + row * d + col[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (uint32_t c0 = wave; c0 < nr; c0 += 16) {
        int jj[4];
        float acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t c = c0 + 4 * u;
            jj[u] = cidx[c < nr ? c : c0];
            const float* w = W_encT + (int64_t)jj[u] * d;
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                if (ok[i]) {
                    const float4 wv = *reinterpret_cast<const float4*>(w + col[i]);
                    a = fmaf(xr[i].x, wv.x, a); a = fmaf(xr[i].y, wv.y, a); a = fmaf(xr[i].z, wv.z, a); a = fmaf(xr[i].w, wv.w, a);
                }
            }
            acc[u] = a;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += __shfl_xor(acc[u], o, 64);
        if (lane < 4) {
            const uint32_t c = c0 + 4 * lane;
            const float av = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
            const int jv = lane == 0 ? jj[0] : (lane == 1 ? jj[1] : (lane == 2 ? jj[2] : jj[3]));
            if (c < nr) {
                rval[c] = av + b_enc[jv];
                if constexpr (GATED) rmag[c] = fmaxf(av * expf(r_mag[jv]) + b_mag[jv], 0.f);      // (as DG_EPI_GENC writes it)
            }
        }
    }
    __syncthreads();
    // how many are positive (NaN counts as not positive, like torch's relu keeps NaN -- a NaN pre-activation poisons the dense path
    // the same way: such a token is sent there)
    uint32_t mine = 0;
    bool nan_seen = false;
    for (uint32_t c = tid; c < nr; c += 256) {
        const float v = rval[c];
        mine += v > 0.f ? 1u : 0u;
        nan_seen = nan_seen || (v != v);
    }
    if (mine) atomicAdd(&sh_m, mine);
    if (nan_seen) sh_bad = 1u;
    __syncthreads();
    const uint32_t m = sh_m;
    bad = bad || sh_bad != 0u || m > (uint32_t)cap;
    if (bad) {                                                 // (uniform) this token cannot be held: the whole step goes dense
        if (tid == 0) {
            // (a workgroup that FOUND the word raised leaves it alone: at the published L0 every one of the 4096 does, and their
            // atomics on the one word were 40 of the 62 us this kernel took to do nothing)
            if (!already_dense) atomicOr(mode, 1u);
            tok_cnt[row] = 0u; l1part[row] = 0.f;
            if constexpr (GATED) l0part[row] = 0.f;
        }
        for (int s = tid; s < cap; s += 256) {
            idx_out[row * cap + s] = 0;
            val_out[row * cap + s] = 0.f;
            wpos[row * cap + s] = 0xffffffffu;
            if constexpr (GATED) valg_out[row * cap + s] = 0.f;
        }
        return;
    }
    sval[tid] = 0.f;
    if (tid == 0) sh_l0 = 0u;
    __syncthreads();
    uint32_t myl0 = 0;
    for (uint32_t c = tid; c < nr; c += 256) {
        const float vc = rval[c];
        if (!(vc > 0.f)) continue;
        const int32_t ic = cidx[c];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < nr; ++o) {
            const float vo = rval[o];
            rank += (vo > vc) || (vo == vc && cidx[o] < ic);
        }
        const uint32_t wp = atomicAdd(&feat_cnt[ic], 1u);
        idx_out[row * cap + rank] = ic;
        wpos[row * cap + rank] = wp;
        if constexpr (GATED) {
            const float f = rmag[c];
            val_out[row * cap + rank] = f;
            valg_out[row * cap + rank] = vc;
            myl0 += f > 0.f ? 1u : 0u;
        } else {
            val_out[row * cap + rank] = vc;
        }
        sval[rank] = vc;
    }
    for (int s = (int)m + tid; s < cap; s += 256) {           // the rest of the row: holes
        idx_out[row * cap + s] = 0;
        val_out[row * cap + s] = 0.f;
        wpos[row * cap + s] = 0xffffffffu;
        if constexpr (GATED) valg_out[row * cap + s] = 0.f;
    }
    if (GATED && myl0) atomicAdd(&sh_l0, myl0);
    __syncthreads();
    // the token's L1 term, summed in rank order (the candidates arrive in whatever order the filter's LDS atomics drew)
    const float lsum = wave_sum(sval[tid]);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
        tok_cnt[row] = m;
        l1part[row] = (red[0] + red[1]) + (red[2] + red[3]);
        if constexpr (GATED) l0part[row] = (float)sh_l0;
    }
}

// CUs of the current device (cached per device id)
static int enc_cus() {
    static std::atomic<int> per_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = per_dev[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        if (n <= 0) n = 256;
        per_dev[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

int launch_enc_gemm(int mode, const EncParams& p, hipStream_t stream) {
    const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
    const dim3 grid(ntm * ntn), block(512);
    if (mode == 0 && g_pv_tuning.gemm_loop == -1 && p.K % 64 == 0 && ntm * ntn <= enc_cus() / 2 && !g_pv_tuning.enc_tm256) {
        // the sample pass where 256-row tiles leave most of the chip idle: 128-row tiles (see the kernel's MB_)
        hipLaunchKernelGGL((sae_enc_gemm_kernel<0, 2, false, 2>), dim3(((p.M + 127) / 128) * ntn), block, 0, stream, p);
        PV_LAUNCH_CHECK("sae_enc_gemm_kernel");
        return PV_OK;
    }
    // K loop as in gemm.hip's launcher: 2 = full-line form (whole 128-byte slabs of fp16: K % 64 == 0), 1 = pipelined 64-byte
    // slabs (K % 32 == 0), 0 = barrier-then-fetch
    int lp = 0;
    if (g_pv_tuning.gemm_loop != 0 && p.K % 32 == 0) lp = (p.K % 64 == 0 && g_pv_tuning.gemm_loop != 1) ? 2 : 1;
#define PV_ENC_LAUNCH(MODE)                                                                              \
    do {                                                                                                 \
        if (lp == 2 && MODE == 1 && p.slots * 128 <= 2 * 256 * 128 / 8 && !g_pv_tuning.enc_rounds)       \
            hipLaunchKernelGGL((sae_enc_gemm_kernel<MODE, 2, MODE == 1>), grid, block, 0, stream, p);    \
        else if (lp == 2) hipLaunchKernelGGL((sae_enc_gemm_kernel<MODE, 2>), grid, block, 0, stream, p); \
        else if (lp == 1) hipLaunchKernelGGL((sae_enc_gemm_kernel<MODE, 1>), grid, block, 0, stream, p); \
        else hipLaunchKernelGGL((sae_enc_gemm_kernel<MODE, 0>), grid, block, 0, stream, p);              \
    } while (0)
    if (mode == 0) PV_ENC_LAUNCH(0);
    else PV_ENC_LAUNCH(1);
#undef PV_ENC_LAUNCH
    PV_LAUNCH_CHECK("sae_enc_gemm_kernel");
    return PV_OK;
}

}  // namespace

// debug: workgroups of the filtered encoder's small kernels the runtime says fit a CU (tools/probes/select_timeline.py counts six select
// workgroups alive per CU where the compiler's register model says eight)
extern "C" int pv_debug_select_occupancy(int32_t* out4) {
    int n = 0;
    PV_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&sae_select_kernel<3, false>), 256, 0));
    out4[0] = n;
    PV_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&sae_thr_kernel<32>), 256, 0));
    out4[1] = n;
    PV_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&relu_select_kernel<3, false>), 256, 0));
    out4[2] = n;
    hipFuncAttributes fa;
    PV_HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&sae_select_kernel<3, false>)));
    out4[3] = fa.numRegs;
    return PV_OK;
}

int sae_encode_fast(const pv_sae_desc& d, const pv_sae_state* st, int N, int32_t* topk_idx, float* topk_val,
                    uint32_t* feat_cnt, uint32_t* wpos, unsigned char* wsb, const SaeWs& ws, hipStream_t stream, const SaePre* pre) {
    PV_REQUIRE(st->W_encT && st->W_enc16T && st->enc_colsq, "encoder shadows (W_encT, W_enc16T, enc_colsq) are required");
    const int S = PV_SAE_SAMPLE_STRIDE, ns = d.d_sae / S, q = pv_sae_sample_q(d.k);
    float* wmax = (float*)(wsb + ws.wmax);
    uint32_t* fb_count = (uint32_t*)(wsb + ws.fb_count);
    int32_t* fb_list = (int32_t*)(wsb + ws.fb_list);
    if (!pre)                                                  // (the fused pre-pass: a workgroup of the prep launch did this)
        hipLaunchKernelGGL(sae_wmax_kernel, dim3(1), dim3(1024), 0, stream, (const float*)st->enc_colsq, d.d_sae, wmax, fb_count, feat_cnt);
    EncParams p = {};
    p.A = wsb + ws.x16; p.M = N; p.K = d.d_in; p.lda = d.d_in;
    // pass 0: every S-th feature
    p.B = st->W_enc16T; p.N = ns; p.ldb_bytes = (uint32_t)S * d.d_in * 2u; p.b_span = (uint32_t)ns * p.ldb_bytes;
    p.bias = st->b_enc; p.bias_stride = S; p.out = (float*)(wsb + ws.sample); p.ldo = ns;
    int rc = launch_enc_gemm(0, p, stream);
    if (rc) return rc;
    const int nb_thr = (N + 3) / 4, nb_cf = (pre && !pre->have_mean) ? (d.d_in + 15) / 16 : 0;
#define THR(V)                                                                                                                \
    hipLaunchKernelGGL((sae_thr_kernel<V>), dim3(nb_thr + nb_cf), dim3(256), 0, stream, (const float*)(wsb + ws.sample), ns,      \
                       (const float*)(wsb + ws.xnorm), (const float*)wmax, q, d.d_in, (float*)(wsb + ws.thr), (float*)(wsb + ws.sq), \
                       (float*)(wsb + ws.band), N, nb_thr, (const float*)(wsb + ws.colpart), (float*)(wsb + ws.batch_mean),          \
                       (N + 15) / 16, 1.0f / (float)N)
    if (ns <= 1024) { THR(16); } else if (ns <= 2048) { THR(32); } else { THR(64); }        // d_sae <= 65536: ns <= 4096
#undef THR
    PV_LAUNCH_CHECK("sae_thr_kernel");
    // filter: all features
    p.N = d.d_sae; p.ldb_bytes = (uint32_t)d.d_in * 2u; p.b_span = (uint32_t)d.d_sae * p.ldb_bytes; p.bias_stride = 1;
    p.out = nullptr; p.thr = (const float*)(wsb + ws.thr); p.cnt = (uint32_t*)(wsb + ws.cand_cnt); p.cand = (int2*)(wsb + ws.cand);
#ifdef PV_ENC_TRACE
    p.trace = reinterpret_cast<uint64_t*>(wsb + ws.hidden) + 4 * 8192;      // (behind the select kernel's stamps)
#endif
    const int ntn = d.d_sae / 256;
    p.slots = pv_sae_tile_slots(d);
    rc = launch_enc_gemm(1, p, stream);
    if (rc) return rc;
#define SEL_ARGS                                                                                                            \
    dim3(N), dim3(256), 0, stream, (const float*)(wsb + ws.sae_in),                                                          \
        (const float*)st->W_encT, (const float*)st->b_enc, (const uint32_t*)(wsb + ws.cand_cnt),                            \
        (const int2*)(wsb + ws.cand), (const float*)(wsb + ws.sq), (const float*)(wsb + ws.band), topk_idx,                 \
        topk_val, fb_list, fb_count, feat_cnt, wpos, d.d_in, d.k, ntn, p.slots,                                             \
        pre ? pre->x : (const float*)nullptr, (const float*)(wsb + ws.batch_mean), (float*)(wsb + ws.norm),                  \
        pre ? pre->d_true : 0, (float*)(wsb + ws.hidden)
    const bool inline_fb = pre && g_pv_tuning.sae_inline_fb;
#define CALL(D)                                                                                                             \
    if (inline_fb) hipLaunchKernelGGL((sae_select_kernel<D, true>), SEL_ARGS);                                              \
    else hipLaunchKernelGGL((sae_select_kernel<D>), SEL_ARGS)
    if (d.d_in <= 256) { CALL(1); } else if (d.d_in <= 768) { CALL(3); } else if (d.d_in <= 1024) { CALL(4); } else { CALL(5); }
#undef CALL
#undef SEL_ARGS
    PV_LAUNCH_CHECK("sae_select_kernel");
    if (inline_fb) return PV_OK;                                // (undecided tokens were recomputed inside the select kernel)
    // undecided tokens: exact rows + the streaming / radix top-k (both launches are empty-handed when the list is empty)
    hipLaunchKernelGGL(sae_fb_hidden_kernel, dim3(PV_SAE_FB_SLOTS, (d.d_sae + 1023) / 1024), dim3(256), 0, stream,
                       (const float*)(wsb + ws.sae_in), (const float*)st->W_encT, (const float*)st->b_enc, (const int32_t*)fb_list,
                       (const uint32_t*)fb_count, (float*)(wsb + ws.hidden), d.d_in, d.d_sae);
    PV_LAUNCH_CHECK("sae_fb_hidden_kernel");
    sae_topk_rows((const float*)(wsb + ws.hidden), topk_idx, topk_val, d.d_sae, d.k, N, fb_list, fb_count, PV_SAE_FB_SLOTS, feat_cnt, wpos, stream);
    return PV_OK;
}

int sae_encode_relu(const pv_sae_desc& d, const pv_sae_state* st, int N, int cap, int32_t* idx, float* val, uint32_t* tok_cnt,
                    float* l1part, uint32_t* cand_cnt, void* cand, uint32_t* feat_cnt, uint32_t* wpos, uint32_t* mode,
                    const float* prev_scalars, unsigned char* wsb, const SaeWs& ws, hipStream_t stream, float* l0part, float* valg) {
    // l0part != NULL: the gated form (see relu_select_kernel) -- the bias is b_gate, val = feature_acts, valg = relu(gate_pre)
    const bool gated = l0part != nullptr;
    const float* bias = gated ? (const float*)st->gt.b_gate : (const float*)st->b_enc;
    PV_REQUIRE(st->W_encT && st->W_enc16T && st->enc_colsq, "encoder shadows (W_encT, W_enc16T, enc_colsq) are required");
    PV_REQUIRE(cap >= 4 && cap <= PV_SAE_RELU_CAP_MAX && cap % 4 == 0, "cap");
    float* wmax = (float*)(wsb + ws.wmax);
    uint32_t* fb_count = (uint32_t*)(wsb + ws.fb_count);
    hipLaunchKernelGGL(sae_wmax_kernel, dim3(1), dim3(1024), 0, stream, (const float*)st->enc_colsq, d.d_sae, wmax, fb_count, feat_cnt);
    hipLaunchKernelGGL(relu_thr_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, (const float*)(wsb + ws.xnorm), (const float*)wmax,
                       d.d_in, (float*)(wsb + ws.thr), N, mode, prev_scalars, cap);
    PV_LAUNCH_CHECK("relu_thr_kernel");
    EncParams p = {};
    p.A = wsb + ws.x16; p.M = N; p.K = d.d_in; p.lda = d.d_in;
    p.B = st->W_enc16T; p.N = d.d_sae; p.ldb_bytes = (uint32_t)d.d_in * 2u; p.b_span = (uint32_t)d.d_sae * p.ldb_bytes;
    p.bias = bias; p.bias_stride = 1; p.out = nullptr; p.thr = (const float*)(wsb + ws.thr);
    p.cnt = cand_cnt; p.cand = (int2*)cand; p.slots = PV_SAE_RELU_SLOTS; p.mode = mode;
    int rc = launch_enc_gemm(1, p, stream);
    if (rc) return rc;
    const int ntn = d.d_sae / 256;
#define CALL(D)                                                                                                                 \
    if (gated)                                                                                                                  \
        hipLaunchKernelGGL((relu_select_kernel<D, true>), dim3(N), dim3(256), 0, stream, (const float*)(wsb + ws.sae_in),          \
                           (const float*)st->W_encT, bias, (const uint32_t*)cand_cnt, (const int2*)cand,                           \
                           (const float*)(wsb + ws.thr), idx, val, tok_cnt, l1part, feat_cnt, wpos, mode, d.d_in, cap, ntn,        \
                           PV_SAE_RELU_SLOTS, (const float*)st->gt.r_mag, (const float*)st->gt.b_mag, l0part, valg);               \
    else                                                                                                                        \
        hipLaunchKernelGGL((relu_select_kernel<D>), dim3(N), dim3(256), 0, stream, (const float*)(wsb + ws.sae_in),                \
                           (const float*)st->W_encT, bias, (const uint32_t*)cand_cnt, (const int2*)cand,                           \
                           (const float*)(wsb + ws.thr), idx, val, tok_cnt, l1part, feat_cnt, wpos, mode, d.d_in, cap, ntn, PV_SAE_RELU_SLOTS)
    if (d.d_in <= 256) { CALL(1); } else if (d.d_in <= 768) { CALL(3); } else if (d.d_in <= 1024) { CALL(4); } else { CALL(5); }
#undef CALL
    PV_LAUNCH_CHECK("relu_select_kernel");
    return PV_OK;
}
