// LDS-tiled MFMA GEMM for gfx950 with fused tap epilogues.
//
//   tile 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA
//   tiles of 32 x 32), K consumed in slabs of 128 BYTES per row (64 bf16 or 32 fp32), so the
//   staging / LDS geometry is identical for both dtypes:
//     - global -> registers: 16-byte chunks, 8 consecutive lanes cover one 128-byte row segment
//     - registers -> LDS rows of 144 bytes (128 + one 16-byte access width of padding): the
//       ds_read_b128 fragment reads (32 distinct rows per half-wave at one chunk column) are
//       bank-conflict free (9*r mod 16 is a permutation over each 16-lane service group)
//     - fragments: lane l reads row (l & 31), chunk 2*j + (l >> 5); for bf16 that IS the
//       v_mfma_f32_32x32x16_bf16 operand (8 consecutive k per lane); for fp32 the 4 floats feed 4
//       consecutive v_mfma_f32_32x32x2_f32 (A and B use the same k <-> lane-half assignment, so
//       any consistent assignment gives the exact dot product)
//     - double-buffered LDS, next slab's global loads issued before the MFMAs of the current one
//   epilogue: accumulators -> per-wave LDS staging (reusing the operand buffers) -> each lane owns
//   8 consecutive columns of a row -> bias / GELU / residual in fp32 -> 16-byte coalesced stores of
//   every requested tap straight from the epilogue (the "fused hook-tap": caching an activation
//   costs one HBM store, nothing is re-read).
//   blockIdx -> tile mapping is XCD-aware: consecutive tiles along N (which share the A rows)
//   are placed on the same XCD so the A slab is served from that XCD's L2.
#include <atomic>

#include "gemm.hpp"

#include <hip/hip_ext.h>
#include "prof.hpp"

namespace {

// debug tracing (pv_debug_gemm_trace_*): stamps on the 100 MHz wall clock, written by thread 0 of each workgroup
__device__ __forceinline__ void trace_stamp(uint64_t* trace, int bid, int slot) {
    if (trace && threadIdx.x == 0) {
        trace[(int64_t)bid * 4 + slot] = wall_clock64();
        if (slot == 0) trace[(int64_t)bid * 4 + 3] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                                      (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int SLAB = 128;            // bytes of K per row per stage
constexpr int ROWB = 144;            // padded LDS row (bytes)
constexpr int TILE_BYTES = BM * ROWB;   // 18432
constexpr int CS_LD = 68;            // fp32 staging row (floats): 64 + 4 pad
constexpr int GEMM_LDS = 4 * TILE_BYTES;   // As[2] + Bs[2] = 73728 >= 4 waves * 64*68*4 = 69632

template <typename T, int AMODE, bool VEC>
struct Loader {
    // one 16-byte chunk of operand A for global row gm at K byte offset kb
    __device__ static __forceinline__ uint4 load_a(const GemmParams& p, int gm, int kb) {
        constexpr int EB = DT<T>::kBytes;
        constexpr int PC = DT<T>::kPerChunk;
        const int Kb = p.K * EB;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (gm >= p.M || kb >= Kb) return z;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(p.A);
        if constexpr (AMODE == PV_A_PLAIN) {
            const unsigned char* ptr = base + (int64_t)gm * p.lda * EB + kb;
            if constexpr (VEC) {
                return *reinterpret_cast<const uint4*>(ptr);
            } else {
                alignas(16) T tmp[PC];
                const T* e = reinterpret_cast<const T*>(ptr);
                const int ke = kb / EB;
#pragma unroll
                for (int i = 0; i < PC; ++i) tmp[i] = (ke + i < p.K) ? e[i] : T(0);
                return *reinterpret_cast<uint4*>(tmp);
            }
        } else {
            // im2col-free patch gather: row gm = (image b, patch py, px); k = (c, i, j)
            const int np = p.pG * p.pG;
            const int b = gm / np, pidx = gm - b * np;
            const int py = pidx / p.pG, px = pidx - py * p.pG;
            const int pp = p.pP * p.pP;
            const int ke = kb / EB;
            if constexpr (VEC) {
                const int c = ke / pp, rem = ke - c * pp;
                const int i = rem / p.pP, j = rem - i * p.pP;
                const int64_t off = (((int64_t)b * p.pC + c) * p.pS + (py * p.pP + i)) * p.pS + px * p.pP + j;
                return *reinterpret_cast<const uint4*>(base + off * EB);
            } else {
                alignas(16) T tmp[PC];
#pragma unroll
                for (int t = 0; t < PC; ++t) {
                    const int k = ke + t;
                    if (k < p.K) {
                        const int c = k / pp, rem = k - c * pp;
                        const int i = rem / p.pP, j = rem - i * p.pP;
                        const int64_t off = (((int64_t)b * p.pC + c) * p.pS + (py * p.pP + i)) * p.pS + px * p.pP + j;
                        tmp[t] = reinterpret_cast<const T*>(base)[off];
                    } else {
                        tmp[t] = T(0);
                    }
                }
                return *reinterpret_cast<uint4*>(tmp);
            }
        }
    }
    __device__ static __forceinline__ uint4 load_b(const GemmParams& p, int gn, int kb) {
        constexpr int EB = DT<T>::kBytes;
        constexpr int PC = DT<T>::kPerChunk;
        const int Kb = p.K * EB;
        if (gn >= p.N || kb >= Kb) return make_uint4(0, 0, 0, 0);
        const unsigned char* ptr = reinterpret_cast<const unsigned char*>(p.Bt) + (int64_t)gn * p.ldb * EB + kb;
        if constexpr (VEC) {
            return *reinterpret_cast<const uint4*>(ptr);
        } else {
            alignas(16) T tmp[PC];
            const T* e = reinterpret_cast<const T*>(ptr);
            const int ke = kb / EB;
#pragma unroll
            for (int i = 0; i < PC; ++i) tmp[i] = (ke + i < p.K) ? e[i] : T(0);
            return *reinterpret_cast<uint4*>(tmp);
        }
    }
};

__device__ __forceinline__ pv_f32x2 unpack2(uint32_t w) {
    return pv_f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}
__device__ __forceinline__ uint32_t pack2(pv_f32x2 v) { return pack_bf16x2(v.x, v.y); }

template <int ACT>
__device__ __forceinline__ pv_f32x2 act2(pv_f32x2 x) {
    if constexpr (ACT == PV_ACT_QUICK_GELU) {                // x * sigmoid(1.702 x), models/activation_fns.py:19
        const pv_f32x2 t = x * (-1.702f * 1.4426950408889634f);
        pv_f32x2 d = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        d = d + 1.0f;
        const pv_f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        return x * r;
    } else if constexpr (ACT == PV_ACT_GELU) {               // 0.5 x (1 + erf(x / sqrt 2)); erf: Abramowitz-Stegun 7.1.26
        const pv_f32x2 z = x * 0.70710678118654752440f;
        const pv_f32x2 az = __builtin_elementwise_abs(z);
        const pv_f32x2 den = __builtin_elementwise_fma(az, pv_f32x2{0.3275911f, 0.3275911f}, pv_f32x2{1.0f, 1.0f});
        const pv_f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        pv_f32x2 q = __builtin_elementwise_fma(t, pv_f32x2{1.061405429f, 1.061405429f}, pv_f32x2{-1.453152027f, -1.453152027f});
        q = __builtin_elementwise_fma(t, q, pv_f32x2{1.421413741f, 1.421413741f});
        q = __builtin_elementwise_fma(t, q, pv_f32x2{-0.284496736f, -0.284496736f});
        q = __builtin_elementwise_fma(t, q, pv_f32x2{0.254829592f, 0.254829592f});
        q = q * t;
        const pv_f32x2 a2 = az * az * (-1.4426950408889634f);
        const pv_f32x2 e = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
        pv_f32x2 r = __builtin_elementwise_fma(-q, e, pv_f32x2{1.0f, 1.0f});
        r = __builtin_elementwise_copysign(r, z);
        const pv_f32x2 hx = x * 0.5f;
        return __builtin_elementwise_fma(hx, r, hx);
    } else {
        return __builtin_elementwise_max(x, pv_f32x2{0.0f, 0.0f});
    }
}

// The activation every bf16 epilogue applies (v4's run-time epilogues included): ONE instruction sequence for all
// kernels, so mlp.hook_post of an image has the same bits whatever GEMM kernel its batch size selects.
template <typename T>
__device__ __forceinline__ float act_any(float x, int act) {
    if constexpr (sizeof(T) == 2) {
        const pv_f32x2 v = {x, x};
        if (act == PV_ACT_GELU) return act2<PV_ACT_GELU>(v).x;
        if (act == PV_ACT_QUICK_GELU) return act2<PV_ACT_QUICK_GELU>(v).x;
        return act2<PV_ACT_RELU>(v).x;
    } else {
        return pv_act<false>(x, act);
    }
}

template <typename T>
__device__ __forceinline__ void epilogue8(const GemmParams& p, float (&v)[8], int gm, int gn) {
    T* out0 = reinterpret_cast<T*>(p.out0);
    T* out1 = reinterpret_cast<T*>(p.out1);
    const T* bias = reinterpret_cast<const T*>(p.bias0);
    int col = gn;
    if (p.epi == PV_EPI_QKV) {
        const int which = gn / p.nsplit;
        col = gn - which * p.nsplit;
        if (which == 1) { out0 = reinterpret_cast<T*>(p.out1); bias = reinterpret_cast<const T*>(p.bias1); }
        if (which == 2) { out0 = reinterpret_cast<T*>(p.out2); bias = reinterpret_cast<const T*>(p.bias2); }
    }
    const int nvalid = min(8, p.N - gn);
    if (p.vec_out && nvalid == 8) {
        if (bias) {
            float b[8];
            load8(bias + col, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += b[i];
        }
        if (p.epi == PV_EPI_BIAS || p.epi == PV_EPI_QKV) {
            store8(out0 + (int64_t)gm * p.ldo + col, v);
        } else if (p.epi == PV_EPI_RESID) {
            // attn_out / mlp_out is rounded to the storage dtype first (it is what the reference
            // adds to the residual: transformer_block.py:122-124, :134)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = DT<T>::round(v[i]);
            if (out0) store8(out0 + (int64_t)gm * p.ldo + gn, v);
            float r[8];
            load8(reinterpret_cast<const T*>(p.resid) + (int64_t)gm * p.ldr + gn, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += r[i];
            store8(out1 + (int64_t)gm * p.ldo + gn, v);
        } else {  // PV_EPI_ACT: mlp.py:67-72, activation applied to the stored pre-activation
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = DT<T>::round(v[i]);
            if (out0) store8(out0 + (int64_t)gm * p.ldo + gn, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = act_any<T>(v[i], p.act);
            store8(out1 + (int64_t)gm * p.ldo + gn, v);
        }
    } else {
        for (int i = 0; i < nvalid; ++i) {
            float x = v[i];
            int c = col + i, g = gn + i;
            T* o0 = out0;
            const T* bs = bias;
            if (p.epi == PV_EPI_QKV) {   // a chunk may straddle outputs when nsplit % 8 != 0
                const int which = g / p.nsplit;
                c = g - which * p.nsplit;
                o0 = reinterpret_cast<T*>(which == 0 ? p.out0 : (which == 1 ? p.out1 : p.out2));
                bs = reinterpret_cast<const T*>(which == 0 ? p.bias0 : (which == 1 ? p.bias1 : p.bias2));
            }
            if (bs) x += DT<T>::load(bs + c);
            if (p.epi == PV_EPI_BIAS || p.epi == PV_EPI_QKV) {
                DT<T>::store(o0 + (int64_t)gm * p.ldo + c, x);
            } else if (p.epi == PV_EPI_RESID) {
                x = DT<T>::round(x);
                if (o0) DT<T>::store(o0 + (int64_t)gm * p.ldo + g, x);
                x += DT<T>::load(reinterpret_cast<const T*>(p.resid) + (int64_t)gm * p.ldr + g);
                DT<T>::store(out1 + (int64_t)gm * p.ldo + g, x);
            } else {
                x = DT<T>::round(x);
                if (o0) DT<T>::store(o0 + (int64_t)gm * p.ldo + g, x);
                DT<T>::store(out1 + (int64_t)gm * p.ldo + g, act_any<T>(x, p.act));
            }
        }
    }
}

// accumulators -> per-wave LDS staging (reusing the operand buffers; caller has passed a barrier after
// the last LDS read) -> each lane owns 8 consecutive columns of a row -> fused epilogue + tap stores
// epilogue8 with the bias values and the residual chunk already in registers (bf16 storage; the full-chunk
// vector case only).  Same arithmetic, same rounding points as epilogue8.
template <typename T>
__device__ __forceinline__ void epilogue8_pre(const GemmParams& p, float (&v)[8], int gm, int gn, int col, T* out0,
                                              const uint4& braw, const uint4& res) {
    T* out1 = reinterpret_cast<T*>(p.out1);
    const uint32_t bw[4] = {braw.x, braw.y, braw.z, braw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] += __uint_as_float(bw[i] << 16);
        v[2 * i + 1] += __uint_as_float(bw[i] & 0xffff0000u);
    }
    if (p.epi == PV_EPI_BIAS || p.epi == PV_EPI_QKV) {
        store8(out0 + (int64_t)gm * p.ldo + col, v);
    } else if (p.epi == PV_EPI_RESID) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = DT<T>::round(v[i]);
        if (out0) store8(out0 + (int64_t)gm * p.ldo + gn, v);
        const uint32_t rw[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] += __uint_as_float(rw[i] << 16);
            v[2 * i + 1] += __uint_as_float(rw[i] & 0xffff0000u);
        }
        store8(out1 + (int64_t)gm * p.ldo + gn, v);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = DT<T>::round(v[i]);
        if (out0) store8(out0 + (int64_t)gm * p.ldo + gn, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_any<T>(v[i], p.act);
        store8(out1 + (int64_t)gm * p.ldo + gn, v);
    }
}

template <typename T>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f32x16 (&acc)[2][2], unsigned char* smem, int m0, int n0,
                                              int wave, int lane, int wm, int wn) {
    float* Cs = reinterpret_cast<float*>(smem) + wave * (64 * CS_LD);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int col = ni * 32 + (lane & 31);
                Cs[row * CS_LD + col] = acc[mi][ni][e];
            }
    __syncthreads();
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int cc = (lane & 7) * 8;
        const int gm = m0 + wm * 64 + row;
        const int gn = n0 + wn * 64 + cc;
        if (gm < p.M && gn < p.N) {
            float v[8];
            const float4 x0 = *reinterpret_cast<const float4*>(Cs + row * CS_LD + cc);
            const float4 x1 = *reinterpret_cast<const float4*>(Cs + row * CS_LD + cc + 4);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w;
            v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            epilogue8<T>(p, v, gm, gn);
        }
    }
}

constexpr int BKN_ROW = 528;         // [K][N]-layout B tile: 128 floats + 16 pad bytes per k row

template <typename T, int AMODE, bool VEC, bool BKN = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    static_assert(!BKN || sizeof(T) == 4, "[K][N] B operand is fp32 only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                      // [2][TILE_BYTES]
    unsigned char* Bs = smem + 2 * TILE_BYTES;     // [2][TILE_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective remap of the linear workgroup id (block b runs on XCD b % 8): each XCD
    // gets a contiguous run of tiles; tiles are ordered N-fastest so a run shares A rows.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int ntn = (p.N + BN - 1) / BN;
    const int tile_m = swz / ntn, tile_n = swz - tile_m * ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    constexpr int EB = DT<T>::kBytes;
    const int Kb = p.K * EB;
    const int nk = (Kb + SLAB - 1) / SLAB;

    // staging assignment: chunk c = tid + 256*i -> row c >> 3, 16-byte column c & 7
    uint4 ra[4], rb[4];
    auto load_slab = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 3, kc = c & 7;
            const int kb = kt * SLAB + kc * 16;
            ra[i] = Loader<T, AMODE, VEC>::load_a(p, m0 + row, kb);
            if constexpr (BKN) {
                // B[k][n]: chunk c -> k row c >> 5 (32 per slab), 4-float column group c & 31
                const int krow = kt * 32 + (c >> 5), nn = n0 + (c & 31) * 4;
                rb[i] = (krow < p.K && nn < p.N)
                            ? *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(p.Bt) + (int64_t)krow * p.ldb + nn)
                            : make_uint4(0, 0, 0, 0);
            } else {
                rb[i] = Loader<T, AMODE, VEC>::load_b(p, n0 + row, kb);
            }
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 3, kc = c & 7;
            *reinterpret_cast<uint4*>(As + buf * TILE_BYTES + row * ROWB + kc * 16) = ra[i];
            if constexpr (BKN)
                *reinterpret_cast<uint4*>(Bs + buf * TILE_BYTES + (c >> 5) * BKN_ROW + (c & 31) * 16) = rb[i];
            else
                *reinterpret_cast<uint4*>(Bs + buf * TILE_BYTES + row * ROWB + kc * 16) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int a_off = (wm * 64 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    const int b_off = (wn * 64 + (lane & 31)) * ROWB + (lane >> 5) * 16;

    load_slab(0);
    store_slab(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_slab(kt + 1);
        const unsigned char* Ab = As + buf * TILE_BYTES;
        const unsigned char* Bb = Bs + buf * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 a[2], b[2];
            a[0] = *reinterpret_cast<const uint4*>(Ab + a_off + j * 32);
            a[1] = *reinterpret_cast<const uint4*>(Ab + a_off + 32 * ROWB + j * 32);
            if constexpr (BKN) {
                // MFMA step (j, e) consumes k = 8 j + 4 (lane >> 5) + e on the A side; fetch the same k
                const float* bk = reinterpret_cast<const float*>(Bb + (8 * j + 4 * (lane >> 5)) * BKN_ROW) + wn * 64 + (lane & 31);
                b[0] = make_uint4(__float_as_uint(bk[0]), __float_as_uint(bk[BKN_ROW / 4]),
                                  __float_as_uint(bk[2 * (BKN_ROW / 4)]), __float_as_uint(bk[3 * (BKN_ROW / 4)]));
                b[1] = make_uint4(__float_as_uint(bk[32]), __float_as_uint(bk[BKN_ROW / 4 + 32]),
                                  __float_as_uint(bk[2 * (BKN_ROW / 4) + 32]), __float_as_uint(bk[3 * (BKN_ROW / 4) + 32]));
            } else {
                b[0] = *reinterpret_cast<const uint4*>(Bb + b_off + j * 32);
                b[1] = *reinterpret_cast<const uint4*>(Bb + b_off + 32 * ROWB + j * 32);
            }
            if constexpr (EB == 2) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]),
                            acc[mi][ni], 0, 0, 0);
            } else {
                const float af[2][4] = {{__uint_as_float(a[0].x), __uint_as_float(a[0].y),
                                         __uint_as_float(a[0].z), __uint_as_float(a[0].w)},
                                        {__uint_as_float(a[1].x), __uint_as_float(a[1].y),
                                         __uint_as_float(a[1].z), __uint_as_float(a[1].w)}};
                const float bf[2][4] = {{__uint_as_float(b[0].x), __uint_as_float(b[0].y),
                                         __uint_as_float(b[0].z), __uint_as_float(b[0].w)},
                                        {__uint_as_float(b[1].x), __uint_as_float(b[1].y),
                                         __uint_as_float(b[1].z), __uint_as_float(b[1].w)}};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                af[mi][e], bf[ni][e], acc[mi][ni], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_slab(buf ^ 1);
        __syncthreads();
    }

    tile_epilogue<T>(p, acc, smem, m0, n0, wave, lane, wm, wn);
}

// ---------------------------------------------------------------------------------------------------
// v4 mainloop: operands DMA'd straight into LDS (buffer_load ... lds), deep enough to cover the latency
// (history of the variants in between -- register staging, 2-stage DMA, 128-byte slabs -- in profiles/r01_notes.md).
//   * K slabs of 64 BYTES per row (32 bf16 / 16 fp32): a stage is 8 KB (A) + 8 KB (B)
//   * THREE-stage LDS ring (48 KB) -> 3 workgroups per CU (12 waves): while one workgroup sits at its
//     barrier or in its store epilogue the other two keep the matrix pipe busy
//   * slab k+2 is issued while slab k is multiplied; the wait before the barrier is a COUNTED
//     s_waitcnt vmcnt(4) (only slab k has to have landed, the 4 DMA instructions of slab k+1 stay in
//     flight across the raw s_barrier) -- __syncthreads() would drain the queue
//   * 64-byte rows: 4 chunks per row, 4 rows per 256-byte bank row; chunk position = k-chunk ^ ((row>>2)&3)
//     applied on the DMA source address and on the fragment reads (conflict-free ds_read_b128)
//   * epilogue staged through LDS in two 32-row halves per wave (34.8 KB <= the ring) 
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;      // destination of an LDS-DMA (buffer_load ... lds)
constexpr int V4_SLAB = 64;                 // bytes of K per row per stage
constexpr int V4_TILE = 128 * V4_SLAB;      // 8 KB per operand
constexpr int V4_STAGE = 2 * V4_TILE;
constexpr int V4_NSTAGE = 3;
static_assert(V4_NSTAGE * V4_STAGE == 49152, "three 16 KB ring slots -> 3 workgroups per CU");

template <typename T>
__global__ __launch_bounds__(256, 3) void gemm_kernel_v4(const GemmParams p) {
    // One static __shared__ object per ring slot: hipcc's waitcnt pass can then prove that the ds_reads of
    // slot i do not alias the DMA writes in flight to slots i+1 / i+2 (alias scopes per LDS variable); with a
    // single array it inserts s_waitcnt vmcnt(0) before the first ds_read of every step and drains the ring.
    __shared__ __attribute__((aligned(16))) unsigned char ring0[V4_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char ring1[V4_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char ring2[V4_STAGE];
    constexpr int EB = DT<T>::kBytes;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // tile order: column blocks of <= 8 N-tiles, M-major inside a block.  The B panel of a block
    // (8 x 128 rows x K) stays resident in the XCD's 4 MB L2 while the M range streams past once per block,
    // instead of the whole weight matrix being re-streamed every few M-tiles (PMC: L2-miss reads were 4x the
    // algorithmic bytes with plain N-fastest order).
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int nblk = (ntn + 7) / 8;
    const int wblk = (ntn + nblk - 1) / nblk;
    const int blk = swz / (ntm * wblk);
    const int rem = swz - blk * (ntm * wblk);
    const int wcur = min(wblk, ntn - blk * wblk);
    const int tile_m = rem / wcur, tile_n = blk * wblk + (rem - tile_m * wcur);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    trace_stamp(p.trace, bid, 0);

    const unsigned Kb = (unsigned)p.K * EB;
    const int nk = (int)((Kb + V4_SLAB - 1) / V4_SLAB);
    const bool ktail = (Kb % V4_SLAB) != 0;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.A), 0, (int)((unsigned)p.M * (unsigned)p.lda * EB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.Bt), 0, (int)((unsigned)p.N * (unsigned)p.ldb * EB), 0x00020000);

    // 8 wave-instructions (1 KiB = 16 rows x 64 B) per operand per slab; wave w issues instructions 2w, 2w+1
    unsigned offA[2], offB[2], kcb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2);
        const int kc = (lane & 3) ^ ((row >> 2) & 3);
        kcb[j] = kc * 16;
        offA[j] = (unsigned)(m0 + row) * (unsigned)p.lda * EB + kc * 16;
        offB[j] = (unsigned)(n0 + row) * (unsigned)p.ldb * EB + kc * 16;
    }
    auto issue = [&](int kt, unsigned char* slot) {
        const unsigned kbase = (unsigned)kt * V4_SLAB;
        unsigned char* Ab = slot + wave * 2048;
        unsigned char* Bb = Ab + V4_TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned oa = offA[j] + kbase, ob = offB[j] + kbase;
            // NOTE: no branch may surround these DMA instructions -- hipcc's waitcnt pass answers a
            // conditionally executed LDS-DMA with s_waitcnt vmcnt(0) before the next ds_read (ring drained)
            // (bitwise, not short-circuit: it has to compile to selects)
            bool oob = (kt >= nk) | (ktail & (kbase + kcb[j] >= Kb));
#ifdef PV_TUNING
            oob |= (((p.dbg & 1) != 0) & (kt >= 2));
#endif
            oa = oob ? 0xffffff00u : oa;
            ob = oob ? 0xffffff00u : ob;
#ifdef PV_TUNING
            oa = (p.dbg & 4) ? lane * 16u : oa;                     // ablation: every DMA hits the same cached 1 KiB
            ob = (p.dbg & 4) ? lane * 16u : ob;
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(Ab + j * 1024), 16, oa, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bb + j * 1024), 16, ob, 0, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int l31 = lane & 31, half = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    const int co0 = ((0 + half) ^ sw) * 16, co1 = ((2 + half) ^ sw) * 16;
    const int a_row = (wm * 64 + l31) * V4_SLAB;
    const int b_row = (wn * 64 + l31) * V4_SLAB;
    auto compute = [&](const unsigned char* slot) {
        const unsigned char* Ab = slot;
        const unsigned char* Bb = slot + V4_TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = j == 0 ? co0 : co1;
            uint4 a[2], b[2];
            a[0] = *reinterpret_cast<const uint4*>(Ab + a_row + co);
            a[1] = *reinterpret_cast<const uint4*>(Ab + a_row + 32 * V4_SLAB + co);
            b[0] = *reinterpret_cast<const uint4*>(Bb + b_row + co);
            b[1] = *reinterpret_cast<const uint4*>(Bb + b_row + 32 * V4_SLAB + co);
            if constexpr (EB == 2) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni], 0, 0, 0);
            } else {
                const uint32_t au[2][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w}};
                const uint32_t bu[2][4] = {{b[0].x, b[0].y, b[0].z, b[0].w}, {b[1].x, b[1].y, b[1].z, b[1].w}};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                __uint_as_float(au[mi][e]), __uint_as_float(bu[ni][e]), acc[mi][ni], 0, 0, 0);
            }
        }
    };
    // one pipeline step: slab kt lives in `cur`, slab kt+2 is issued into `nxt2` (the slot multiplied in step kt-1)
    //   s_waitcnt vmcnt(4): this wave's 4 DMA instructions of slab kt have landed, the 4 of slab kt+1 may
    //   still be in flight (simm16 0x0F74 = vmcnt 4, expcnt 7, lgkmcnt 15: only vmcnt is waited on)
#define PV_V4_STEP(KT, CUR, NXT2)                      \
    __builtin_amdgcn_s_waitcnt(0x0F74);                \
    __builtin_amdgcn_s_barrier();                      \
    issue((KT) + 2, NXT2);                             \
    compute(CUR);

    // Epilogue operands fetched BEFORE the K loop (bf16 only: 40 VGPRs): this lane's 8 bias values and, for the
    // residual epilogues, its 8 x 16 B of residual-stream rows.  They are older than every DMA in the vmcnt
    // queue, so the counted waits of the loop retire them for free and the store epilogue never waits on HBM.
    const int e_gn = n0 + wn * 64 + (lane & 7) * 8;
    const bool e_fast = EB == 2 && p.vec_out && e_gn + 8 <= p.N;
    uint4 e_bias = make_uint4(0, 0, 0, 0);        // 8 raw bf16 (converted in the epilogue: no wait up here)
    uint4 e_res[2][4];
    int e_col = e_gn;
    T* e_out0 = reinterpret_cast<T*>(p.out0);
    if constexpr (EB == 2) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int it = 0; it < 4; ++it) e_res[mi][it] = make_uint4(0, 0, 0, 0);
        if (e_fast) {
            const T* bias = reinterpret_cast<const T*>(p.bias0);
            if (p.epi == PV_EPI_QKV) {
                const int which = e_gn / p.nsplit;
                e_col = e_gn - which * p.nsplit;
                if (which == 1) { e_out0 = reinterpret_cast<T*>(p.out1); bias = reinterpret_cast<const T*>(p.bias1); }
                if (which == 2) { e_out0 = reinterpret_cast<T*>(p.out2); bias = reinterpret_cast<const T*>(p.bias2); }
            }
            if (bias) e_bias = *reinterpret_cast<const uint4*>(bias + e_col);
            if (p.epi == PV_EPI_RESID) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int gm = m0 + wm * 64 + mi * 32 + it * 8 + (lane >> 3);
                        if (gm < p.M)
                            e_res[mi][it] = *reinterpret_cast<const uint4*>(
                                reinterpret_cast<const T*>(p.resid) + (int64_t)gm * p.ldr + e_gn);
                    }
            }
        }
    }

    issue(0, ring0);
    issue(1, ring1);
    int kt = 0;
    for (; kt + 3 <= nk; kt += 3) {
        PV_V4_STEP(kt, ring0, ring2)
        PV_V4_STEP(kt + 1, ring1, ring0)
        PV_V4_STEP(kt + 2, ring2, ring1)
    }
    if (kt < nk) { PV_V4_STEP(kt, ring0, ring2) }
    if (kt + 1 < nk) { PV_V4_STEP(kt + 1, ring1, ring0) }
#undef PV_V4_STEP
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): drain the off-the-end prefetches
    __syncthreads();
    trace_stamp(p.trace, bid, 1);
#ifdef PV_TUNING
    if (p.dbg & 2) {
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out0)[0] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];
        return;
    }
#endif

    // ---- epilogue in two 32-row halves per wave; staging 32 x 64 floats per wave (waves 0,1 in ring0,
    //      waves 2,3 in ring1)
    constexpr int CLD = 64;
    float* Cs = reinterpret_cast<float*>((wave < 2 ? ring0 : ring1) + (wave & 1) * (32 * CLD * 4));
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                Cs[row * CLD + ni * 32 + l31] = acc[mi][ni][e];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int cc = (lane & 7) * 8;
            const int gm = m0 + wm * 64 + mi * 32 + row;
            const int gn = n0 + wn * 64 + cc;
            if (gm < p.M && gn < p.N) {
                float v[8];
                const float4 x0 = *reinterpret_cast<const float4*>(Cs + row * CLD + cc);
                const float4 x1 = *reinterpret_cast<const float4*>(Cs + row * CLD + cc + 4);
                v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w;
                v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                if constexpr (EB == 2) {
                    if (e_fast) epilogue8_pre<T>(p, v, gm, gn, e_col, e_out0, e_bias, e_res[mi][it]);
                    else epilogue8<T>(p, v, gm, gn);
                } else {
                    epilogue8<T>(p, v, gm, gn);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    trace_stamp(p.trace, bid, 2);
}

template <typename T>
int launch_v4(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    {
        constexpr double EBd = DT<T>::kBytes;
        const double mn = (double)p.M * p.N;
        double outs = 1.0;
        if (p.epi == PV_EPI_RESID) outs = 2.0 + (p.out0 ? 1.0 : 0.0);
        if (p.epi == PV_EPI_ACT) outs = 1.0 + (p.out0 ? 1.0 : 0.0);
        ProfScope prof(PV_PROF_GEMM, stream, 2.0 * mn * p.K, ((double)p.M * p.K + (double)p.N * p.K + outs * mn) * EBd);
        hipLaunchKernelGGL((gemm_kernel_v4<T>), dim3(ntm * ntn), dim3(256), 0, stream, p);
    }
    PV_LAUNCH_CHECK("gemm_kernel_v4");
    return PV_OK;
}

template <typename T, int AMODE, bool VEC, bool BKN = false>
int launch(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    static bool attr_done = false;
    if (!attr_done) {
        PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, AMODE, VEC, BKN>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS));
        attr_done = true;
    }
    {
        constexpr double EBd = DT<T>::kBytes;
        const double mn = (double)p.M * p.N;
        double outs = 1.0;                                   // BIAS / QKV: one M x N store
        if (p.epi == PV_EPI_RESID) outs = 2.0 + (p.out0 ? 1.0 : 0.0);   // resid read + out1 (+ tap)
        if (p.epi == PV_EPI_ACT) outs = 1.0 + (p.out0 ? 1.0 : 0.0);
        ProfScope prof(PV_PROF_GEMM, stream, 2.0 * mn * p.K, ((double)p.M * p.K + (double)p.N * p.K + outs * mn) * EBd);
        hipLaunchKernelGGL((gemm_kernel<T, AMODE, VEC, BKN>), dim3(ntm * ntn), dim3(256), GEMM_LDS, stream, p);
    }
    PV_LAUNCH_CHECK("gemm_kernel");
    return PV_OK;
}

// ---------------------------------------------------------------------------------------------------
// bf16 store epilogue with everything known at compile time (v7).  The phase trace (tools/gemm_trace.py)
// showed the generic epilogue8 to be VALU-bound, not store-bound: the MLP-1 tile spent 35 us in it against
// 28 us in its K loop (IEEE division + three activation arms + per-call pointer math, ~100 VALU per element
// for 8 waves per CU).  Here: packed fp32 math (v_pk_*), v_exp / v_rcp instead of expf / division (the value
// is rounded to bf16 two instructions later), no run-time switches.
// ---------------------------------------------------------------------------------------------------
// one 8-element chunk: acc + bias -> (rounded) outputs.  o0 / o1 point at the chunk; b = unpacked bias.
template <int EPI, int ACT>
__device__ __forceinline__ void epi8_bf16(const float4& x0, const float4& x1, const pv_f32x2 (&b)[4], bf16_t* o0, bf16_t* o1,
                                          const uint4& res) {
    pv_f32x2 v[4] = {{x0.x, x0.y}, {x0.z, x0.w}, {x1.x, x1.y}, {x1.z, x1.w}};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] + b[i];
    uint4 r;
    r.x = pack2(v[0]); r.y = pack2(v[1]); r.z = pack2(v[2]); r.w = pack2(v[3]);
    if constexpr (EPI == PV_EPI_BIAS || EPI == PV_EPI_QKV) {
        *reinterpret_cast<uint4*>(o0) = r;
    } else {
        // the stored (bf16-rounded) value is what the reference carries on: transformer_block.py:122-124, :134;
        // mlp.py:67-72
        if (o0) pv_store16_stream<pv_u32x4_a16>(o0, r.x, r.y, r.z, r.w);     // (o0 = the tap-only output of these two epilogues)
        const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
        uint4 y;
        if constexpr (EPI == PV_EPI_RESID) {
            const uint32_t sw[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = unpack2(rw[i]) + unpack2(sw[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = act2<ACT>(unpack2(rw[i]));
        }
        y.x = pack2(v[0]); y.y = pack2(v[1]); y.z = pack2(v[2]); y.w = pack2(v[3]);
        *reinterpret_cast<uint4*>(o1) = y;
    }
}

// ---------------------------------------------------------------------------------------------------
// v7 mainloop: ONE 512-thread workgroup per CU, (64*MB) x 256 tile (MB = 4: 256 x 256, MB = 5: 320 x 256).
//   The 128 x 128 tiles of v4 pull 1/64 byte of operand per flop out of L2: 68 GB per bs=512 B/32 forward,
//   ~15 TB/s sustained at v4's speed -- the L2 -> LDS path, not the matrix pipe, is what v4 saturates
//   (profiles/r01_notes.md).  Here the byte/flop ratio is 1/128 (MB = 4) or 1/142 (MB = 5):
//     * 8 waves as 2 (M) x 4 (N); a wave owns (32*MB) x 64 outputs (14 ds_read_b128 per 20 MFMAs at MB = 5)
//     * slot = (64*MB + 256) rows x 64 B (36 KB at MB = 5); FOUR slots (144 KB of the 160 KB) -> slab k+3 is
//       issued while slab k is multiplied, three slabs (108 KB per CU) in flight
//     * per slab a wave issues NA = ceil(4*MB / 8) A-instructions + 2 B-instructions; at MB = 5 waves 4..7 have
//       only 2 real A pieces -- their third is an out-of-range (zero-fill) DMA into a pad, so that every
//       wave retires the same number and the counted wait stays a compile-time vmcnt(2 * (NA + 2))
//     * MB = 5 makes M = 25600 (512 images x 50 tokens) exactly 80 row tiles: the N = 768 GEMMs (O-projection,
//       MLP-2) are 240 tiles = ONE round of the 256 CUs, the QKV GEMM 720 = three
// ---------------------------------------------------------------------------------------------------
template <typename T, int MB, int EPI, int ACT, int LP = 0>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v7(const GemmParams p) {
    static_assert(sizeof(T) == 2, "v7 is the bf16 kernel");
    constexpr int TM = 64 * MB;
    constexpr int TN = 256;
    constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, SLOT = A_BYTES + B_BYTES;
    constexpr int NA = (4 * MB + 7) / 8;                  // A wave-instructions per wave per slab
    constexpr bool PAD = (4 * MB) % 8 != 0;
    static_assert(4 * SLOT + (PAD ? 8192 : 16) <= 160 * 1024, "one workgroup per CU");
    // LP >= 2 (full-line form below): two slots of 128-byte rows in ring0 / ring1, the other objects shrink to stubs
    __shared__ __attribute__((aligned(16))) unsigned char ring0[LP >= 2 ? 2 * SLOT : SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char ring1[LP >= 2 ? 2 * SLOT : SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char ring2[LP >= 2 ? 16 : SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char ring3[LP >= 2 ? 16 : SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char pad[(PAD && LP < 2) ? 8192 : 16];
    constexpr int EB = DT<T>::kBytes;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int ntn = (p.N + TN - 1) / TN, ntm = (p.M + TM - 1) / TM;
    const int nblk = (ntn + 7) / 8;
    const int wblk = (ntn + nblk - 1) / nblk;
    const int blk = swz / (ntm * wblk);
    const int rem = swz - blk * (ntm * wblk);
    const int wcur = min(wblk, ntn - blk * wblk);
    const int tile_m = rem / wcur, tile_n = blk * wblk + (rem - tile_m * wcur);
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    trace_stamp(p.trace, bid, 0);

    const unsigned Kb = (unsigned)p.K * EB;
    const int nk = (int)((Kb + 63) / 64);
    const bool ktail = (Kb % 64) != 0;
    // PATCH A operand (patch size 32, bf16): the 64 bytes of K that slab kt covers are ONE contiguous run of the
    // NCHW image -- channel kt / 32, patch row kt % 32 -- so the im2col-free gather is only a different row base and
    // slab offset for the same DMA (patch_embedding.py:26-32)
    const bool patch = p.a_mode == PV_A_PATCH;
    const unsigned a_span = patch ? (unsigned)(p.M / (p.pG * p.pG)) * (unsigned)p.pC * (unsigned)p.pS * (unsigned)p.pS * EB
                                  : (unsigned)p.M * (unsigned)p.lda * EB;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)a_span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.Bt), 0, (int)((unsigned)p.N * (unsigned)p.ldb * EB), 0x00020000);

    // a wave-instruction moves 16 rows x 64 B; A has 4*MB of them per slab (instruction j*8 + wave), B has 16
    unsigned offA[NA], kcA[NA], offB[2], kcB[2];
    bool realA[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int ia = j * 8 + wave;
        realA[j] = ia < 4 * MB;
        const int row = ia * 16 + (lane >> 2);
        const int kc = (lane & 3) ^ ((row >> 2) & 3);
        kcA[j] = kc * 16;
        if (patch) {
            const int gm = m0 + row, np = p.pG * p.pG;
            const int b = gm / np, pidx = gm - b * np;
            const int py = pidx / p.pG, px = pidx - py * p.pG;
            offA[j] = (unsigned)(((b * p.pC) * p.pS + py * p.pP) * p.pS + px * p.pP) * EB + kc * 16;      // image b >= B: past the range -> 0
        } else {
            offA[j] = (unsigned)(m0 + row) * (unsigned)p.lda * EB + kc * 16;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 8 + wave) * 16 + (lane >> 2);
        const int kc = (lane & 3) ^ ((row >> 2) & 3);
        kcB[j] = kc * 16;
        offB[j] = (unsigned)(n0 + row) * (unsigned)p.ldb * EB + kc * 16;
    }
    auto issue = [&](int kt, unsigned char* slot) {
        const unsigned kbase = (unsigned)kt * 64;
        bool dead = (kt >= nk);
#ifdef PV_TUNING
        dead |= (((p.dbg & 1) != 0) & (kt >= 3));
#endif
        // (selects only: a branch around an LDS-DMA makes hipcc drain the queue before the next ds_read)
        const unsigned kbaseA = patch ? (unsigned)((kt >> 5) * p.pS * p.pS + (kt & 31) * p.pS) * EB : kbase;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            unsigned o = offA[j] + kbaseA;
            o = (dead | !realA[j] | (ktail & (kbase + kcA[j] >= Kb))) ? 0xffffff00u : o;
            unsigned char* dst = slot + (j * 8 + wave) * 1024;
            if constexpr (PAD) { if (j == NA - 1) dst = realA[j] ? dst : pad + wave * 1024; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)dst, 16, o, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned o = offB[j] + kbase;
            o = (dead | (ktail & (kbase + kcB[j] >= Kb))) ? 0xffffff00u : o;
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
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni], 0, 0, 0);
        }
    };

    // epilogue operands in flight before the K loop (see v4): bias chunk + residual rows of block 0.
    // The launcher guarantees vec_out and N % 8 == 0: whole 8-element chunks only.
    const int e_gn = n0 + wn * 64 + (lane & 7) * 8;
    const bool e_live = e_gn < p.N;
    const int e_rows_left = e_live ? p.M - (m0 + wm * 32 * MB + (lane >> 3)) : 0;    // row mi*32 + it*8 of this lane is real iff < e_rows_left
    uint4 e_bias = make_uint4(0, 0, 0, 0);
    constexpr int NRES = EPI == PV_EPI_RESID ? MB : 1;
    uint4 e_res[NRES][4];
#pragma unroll
    for (int mi = 0; mi < NRES; ++mi)
#pragma unroll
        for (int it = 0; it < 4; ++it) e_res[mi][it] = make_uint4(0, 0, 0, 0);
    int e_col = e_gn;
    T* e_out0 = reinterpret_cast<T*>(p.out0);
    const T* e_rbase = reinterpret_cast<const T*>(p.resid) + (int64_t)(m0 + wm * 32 * MB + (lane >> 3)) * p.ldr + e_gn;
#define PV_V7_FETCH_RES(MI)                                                                          \
    if constexpr (EPI == PV_EPI_RESID) {                                                             \
        _Pragma("unroll") for (int it = 0; it < 4; ++it)                                             \
            if ((MI) * 32 + it * 8 < e_rows_left)                                                    \
                e_res[MI][it] = *reinterpret_cast<const uint4*>(e_rbase + (int64_t)((MI) * 32 + it * 8) * p.ldr); \
    }
    if (e_live) {
        const T* bias = reinterpret_cast<const T*>(p.bias0);
        if constexpr (EPI == PV_EPI_QKV) {
            const int which = e_gn / p.nsplit;
            e_col = e_gn - which * p.nsplit;
            if (which == 1) { e_out0 = reinterpret_cast<T*>(p.out1); bias = reinterpret_cast<const T*>(p.bias1); }
            if (which == 2) { e_out0 = reinterpret_cast<T*>(p.out2); bias = reinterpret_cast<const T*>(p.bias2); }
        }
        if (bias) e_bias = *reinterpret_cast<const uint4*>(bias + e_col);
    }
    PV_V7_FETCH_RES(0)

    // step kt: slab kt must have landed -- the 2 * (NA + 2) DMA instructions of slabs kt+1, kt+2 may stay in flight
    constexpr int NPIECE = NA + 2;
    static_assert(2 * NPIECE <= 15, "vmcnt immediate below uses the low 4 bits only");
#define PV_V7_SYNC() __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NPIECE)); __builtin_amdgcn_s_barrier();
#define PV_V7_STEP(KT, CUR, NXT3)                               \
    PV_V7_SYNC()                                                \
    issue((KT) + 3, NXT3);                                      \
    compute(CUR);

    if constexpr (LP == 0) {
        issue(0, ring0);
        issue(1, ring1);
        issue(2, ring2);
        int kt = 0;
        for (; kt + 4 <= nk; kt += 4) {
            PV_V7_STEP(kt, ring0, ring3)
            PV_V7_STEP(kt + 1, ring1, ring0)
            PV_V7_STEP(kt + 2, ring2, ring1)
            PV_V7_STEP(kt + 3, ring3, ring2)
        }
        if (kt < nk) { PV_V7_STEP(kt, ring0, ring3) }
        if (kt + 1 < nk) { PV_V7_STEP(kt + 1, ring1, ring0) }
        if (kt + 2 < nk) { PV_V7_STEP(kt + 2, ring2, ring1) }
    } else if constexpr (LP == 1) {
        // Software-pipelined form.  The loop above has every wave arrive at the slab's barrier with empty fragment
        // registers: both waves of a SIMD then issue their DMA pieces and their first ds_reads and sit out the LDS
        // latency with the matrix pipe idle.  Here the fragments of a half-slab are fetched while the previous
        // half-slab is multiplied (each A fragment is refilled right behind the two MFMAs that consumed it), the
        // barrier of slab s+1 sits in the MIDDLE of step s (between its two halves: by then every read of slab s has
        // been issued, and the second half's operands are already in registers), and the DMA pieces of slab s+4 go
        // out one per MFMA pair during the second half, into the slot the barrier has just freed.
        //   ring depth as before: at the barrier of slab s+1 the pieces of slabs s+2 and s+3 stay in flight.
        // (plain A operand, whole 64-byte slabs only -- the launcher keeps the loop above for the patch gather and
        // for a K tail: a piece's source offset is then one per-lane base + a uniform term, two VALU per piece)
        const unsigned pA0 = (unsigned)(m0 + wave * 16 + (lane >> 2)) * (unsigned)p.lda * EB + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
        const unsigned pB0 = (unsigned)(n0 + wave * 16 + (lane >> 2)) * (unsigned)p.ldb * EB + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
        const unsigned strideA = 128u * (unsigned)p.lda * EB, strideB = 128u * (unsigned)p.ldb * EB;
        auto issue_piece = [&](int kt, unsigned char* slot, int j) {
            const unsigned kbase = (unsigned)kt * 64;
            const bool dead = (kt >= nk);
            if (j < NA) {
                const bool off = dead | ((j * 8 + wave) >= 4 * MB);                  // uniform
                const unsigned o = off ? 0xffffff00u : pA0 + ((unsigned)j * strideA + kbase);
                unsigned char* dst = slot + (j * 8 + wave) * 1024;
                if constexpr (PAD) { if (j == NA - 1) dst = (j * 8 + wave) < 4 * MB ? dst : pad + wave * 1024; }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)dst, 16, o, 0, 0, 0);
            } else {
                const int jb = j - NA;
                const unsigned o = dead ? 0xffffff00u : pB0 + ((unsigned)jb * strideB + kbase);
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
        static_assert(3 * NPIECE <= 15 && NPIECE <= MB, "prologue vmcnt immediate; one DMA piece per MFMA pair");
        constexpr int BPOS = MB >= 4 ? 1 : 0;           // which MFMA pair the next half's B fragments are fetched behind
        uint4 fa[MB], fb0[2], fb1[2];
        // the issue order below is the schedule: MFMA pair | DMA piece + fragment refill | MFMA pair | ... (left to itself
        // hipcc sinks the refills to the end of the half-slab and waits for them right behind the barrier)
#define PV_V7_PIN() __builtin_amdgcn_sched_barrier(0)
#define PV_V7_PIN2() __builtin_amdgcn_sched_barrier(0)
#define PV_V7_PAIR(MI, FB)                                                                                   \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                     \
            acc[MI][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                           \
                __builtin_bit_cast(bf16x8, fa[MI]), __builtin_bit_cast(bf16x8, FB[ni]), acc[MI][ni], 0, 0, 0);
        // step KT: slab KT in CUR (visible), slab KT+1 in NXT; fa / fb0 hold the first half of slab KT
#define PV_V7_PSTEP(KT, CUR, NXT)                                                                            \
        _Pragma("unroll") for (int mi = 0; mi < MB; ++mi) {                                                  \
            PV_V7_PAIR(mi, fb0)                                                                              \
            PV_V7_PIN2();                                                                                    \
            fa[mi] = rdA(CUR, 1, mi);                                                                        \
            if (mi == BPOS) { fb1[0] = rdB(CUR, 1, 0); fb1[1] = rdB(CUR, 1, 1); }                            \
            PV_V7_PIN();                                                                                     \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NPIECE));                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        _Pragma("unroll") for (int mi = 0; mi < MB; ++mi) {                                                  \
            PV_V7_PAIR(mi, fb1)                                                                              \
            PV_V7_PIN2();                                                                                    \
            fa[mi] = rdA(NXT, 0, mi);                                                                        \
            if (mi == BPOS) { fb0[0] = rdB(NXT, 0, 0); fb0[1] = rdB(NXT, 0, 1); }                            \
            if (mi < NPIECE) issue_piece((KT) + 4, CUR, mi);                                                 \
            PV_V7_PIN();                                                                                     \
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
            PV_V7_PSTEP(kt, ring0, ring1)
            PV_V7_PSTEP(kt + 1, ring1, ring2)
            PV_V7_PSTEP(kt + 2, ring2, ring3)
            PV_V7_PSTEP(kt + 3, ring3, ring0)
        }
        if (kt < nk) { PV_V7_PSTEP(kt, ring0, ring1) }
        if (kt + 1 < nk) { PV_V7_PSTEP(kt + 1, ring1, ring2) }
        if (kt + 2 < nk) { PV_V7_PSTEP(kt + 2, ring2, ring3) }
#undef PV_V7_PSTEP
#undef PV_V7_PAIR
#undef PV_V7_PIN
#undef PV_V7_PIN2
    } else {
        // Full-line form (LP == 2): K slabs of 128 BYTES per row, i.e. whole cache lines -- a DMA piece (1 KiB) is 8 rows x 128 B
        // = 8 lines instead of 16 half lines (the 64-byte slabs above pull every operand line through the L1 twice, 1 us
        // apart, and the texture addresser is busy 55-79 % of the launch: profiles/r02_notes.md).  Two 72 KB slots; the
        // software pipeline of LP == 1 on four 16-element k-steps per slab: the barrier of slab s+1 sits before the LAST k-step
        // of slab s (every read of slab s has been issued by then), the pieces of slab s+2 go out behind the MFMA pairs of
        // that last k-step (into the slot the barrier freed) and of the next slab's first k-step.  One slab of prefetch
        // distance (the slot it lands in is read until the barrier), so the wait before the barrier is vmcnt(0).
        //   LDS row = 128 B = 8 chunks; chunk c of row r at position c ^ ((r >> 1) & 7): the 16 rows of a ds_read_b128 lane group
        //   (8 even, 8 odd) then cover all 64 banks.
        constexpr int A2 = TM * 128;                                   // bytes of the A part of a slot
        constexpr int NP2 = MB + 4;                                    // pieces per wave per slab: MB of A, 4 of B
        static_assert(MB >= 4, "four B pieces ride on the first k-step's MFMA pairs");
        const int prow = lane >> 3;                                    // row of the piece this lane fetches
        const int psw = ((lane >> 4) + 4 * (wave & 1)) & 7;            // (row >> 1) & 7 of that row (row = (j*8 + wave)*8 + prow)
        const unsigned pcol = (unsigned)(((lane & 7) ^ psw) * 16);
        const unsigned pA0 = (unsigned)(m0 + wave * 8 + prow) * (unsigned)p.lda * EB + pcol;
        const unsigned pB0 = (unsigned)(n0 + wave * 8 + prow) * (unsigned)p.ldb * EB + pcol;
        const unsigned strideA = 64u * (unsigned)p.lda * EB, strideB = 64u * (unsigned)p.ldb * EB;
        const int nk2 = (int)(Kb / 128);
        auto issue_piece2 = [&](int kt, unsigned char* slot, int j) {
            const unsigned kbase = (unsigned)kt * 128;
            const bool dead = (kt >= nk2);
            if (j < MB) {
                const unsigned o = dead ? 0xffffff00u : pA0 + ((unsigned)j * strideA + kbase);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(slot + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
            } else {
                const int jb = j - MB;
                const unsigned o = dead ? 0xffffff00u : pB0 + ((unsigned)jb * strideB + kbase);
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
#define PV_V7_PAIR2(MI, H)                                                                                    \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                      \
            acc[MI][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                            \
                __builtin_bit_cast(bf16x8, fa[MI]), __builtin_bit_cast(bf16x8, fb[(H) & 1][ni]), acc[MI][ni], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);
        // slab KT in CUR (visible), slab KT+1 arriving in NXT; fa / fb[0] hold k-step 0 of slab KT
#define PV_V7_FSTEP(KT, CUR, NXT)                                                                             \
        _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                                       \
            if (h == 3) {                                                                                     \
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                           \
                __builtin_amdgcn_s_barrier();                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                            \
            }                                                                                                 \
            _Pragma("unroll") for (int mi = 0; mi < MB; ++mi) {                                               \
                PV_V7_PAIR2(mi, h)                                                                            \
                fa[mi] = h < 3 ? rdA2(CUR, h < 3 ? h + 1 : 0, mi) : rdA2(NXT, 0, mi);                         \
                if (mi == 1) {                                                                                \
                    fb[(h + 1) & 1][0] = h < 3 ? rdB2(CUR, h < 3 ? h + 1 : 0, 0) : rdB2(NXT, 0, 0);           \
                    fb[(h + 1) & 1][1] = h < 3 ? rdB2(CUR, h < 3 ? h + 1 : 0, 1) : rdB2(NXT, 0, 1);           \
                }                                                                                             \
                if (h == 3) issue_piece2((KT) + 2, CUR, mi);                                                  \
                if (h == 0 && mi < 4) issue_piece2((KT) + 1, NXT, MB + mi);                                   \
                __builtin_amdgcn_sched_barrier(0);                                                            \
            }                                                                                                 \
        }
#pragma unroll
        for (int j = 0; j < NP2; ++j) issue_piece2(0, ring0, j);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < MB; ++j) issue_piece2(1, ring1, j);      // (the B pieces of slab 1 ride on step 0's first k-step)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) fa[mi] = rdA2(ring0, 0, mi);
        fb[0][0] = rdB2(ring0, 0, 0); fb[0][1] = rdB2(ring0, 0, 1);
        int kt = 0;
        for (; kt + 2 <= nk2; kt += 2) {
            PV_V7_FSTEP(kt, ring0, ring1)
            PV_V7_FSTEP(kt + 1, ring1, ring0)
        }
        if (kt < nk2) { PV_V7_FSTEP(kt, ring0, ring1) }
#undef PV_V7_FSTEP
#undef PV_V7_PAIR2
    }
#undef PV_V7_STEP
#undef PV_V7_SYNC
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): drain the off-the-end prefetches
    __syncthreads();
    trace_stamp(p.trace, bid, 1);
#ifdef PV_TUNING
    if (p.dbg & 2) {
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out0)[0] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];
        return;
    }
#endif

    constexpr int CLD = 64;
    float* Cs = reinterpret_cast<float*>((wave < 4 ? ring0 : ring1) + (wave & 3) * (32 * CLD * 4));
    const pv_f32x2 e_b[4] = {unpack2(e_bias.x), unpack2(e_bias.y), unpack2(e_bias.z), unpack2(e_bias.w)};
    const int64_t e_row0 = (int64_t)(m0 + wm * 32 * MB + (lane >> 3)) * p.ldo;
    T* const o0_base = e_out0 ? e_out0 + e_row0 + e_col : nullptr;
    T* const o1_base = reinterpret_cast<T*>(p.out1) + e_row0 + e_gn;
    const float* Cr = Cs + (lane >> 3) * CLD + (lane & 7) * 8;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                Cs[row * CLD + ni * 32 + l31] = acc[mi][ni][e];
            }
        __builtin_amdgcn_wave_barrier();
        // residual rows two blocks ahead: staging block mi freed 32 accumulator registers, which now carry the
        // rows of blocks 2mi+1 and 2mi+2 -- from block 2 on a load has two blocks of stores to land behind
        // (indices kept in range: an OOB index in a dead arm defeats SROA)
        if (2 * mi + 1 < MB) { PV_V7_FETCH_RES((2 * mi + 1 < MB ? 2 * mi + 1 : 0)) }
        if (2 * mi + 2 < MB) { PV_V7_FETCH_RES((2 * mi + 2 < MB ? 2 * mi + 2 : 0)) }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (mi * 32 + it * 8 < e_rows_left) {
                const float4 x0 = *reinterpret_cast<const float4*>(Cr + it * 8 * CLD);
                const float4 x1 = *reinterpret_cast<const float4*>(Cr + it * 8 * CLD + 4);
                const int64_t ro = (int64_t)(mi * 32 + it * 8) * p.ldo;
                epi8_bf16<EPI, ACT>(x0, x1, e_b, o0_base ? o0_base + ro : nullptr, o1_base + ro,
                                    e_res[EPI == PV_EPI_RESID ? mi : 0][it]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef PV_V7_FETCH_RES
    trace_stamp(p.trace, bid, 2);
}

// ---------------------------------------------------------------------------------------------------
// v8: the v7 full-line kernel made PERSISTENT.  One 512-thread workgroup per CU walks its tiles (virtual block ids bid,
// bid + grid, ...: the same XCD-aware tile order as v7) and the K slabs of consecutive tiles form ONE stream through the
// two-slot ring: the prefetches v7 issues off the end of its K loop (dead DMAs there) are the NEXT tile's first slab and the
// A pieces of its second, so a tile's store epilogue runs with the next tile's operands already in LDS / in flight, the
// next K loop starts without a launch, a DMA prologue or a drained store queue in front of it, and the workgroups of a
// launch drift apart instead of computing together and then storing together (profiles/r02_gemm_phase_trace.txt).
//   * what changes against v7 LP == 2: the epilogue's transposition no longer borrows the ring (it holds the next tile's
//     slab): 2 KB per wave of dedicated LDS (the 16 KB the two 72 KB slots leave), 8 rows per pass instead of 32;
//     no workgroup barrier and no vmcnt(0) between K loop and epilogue; slab indices past the tile's end address the next
//     tile (selects on a uniform condition: no branch around an LDS-DMA)
//   * the MFMA / fragment-read / DMA-issue order inside the K loop is v7's, and so is the k order of every output's fp32 sum:
//     results are bit-identical to v7's
//   * legal where v7's full-line loop is: plain A operand, K a whole EVEN number of 128-byte slabs (the two slots alternate
//     per slab: an odd count would swap their roles from tile to tile)
// ---------------------------------------------------------------------------------------------------
// the lane id, computed where it is asked for (asm volatile: neither hoisted nor merged with an earlier copy) -- what the epilogue of
// the persistent kernel derives its addresses from, so that none of them lives through the K loop
__device__ __forceinline__ int lane_id_here() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <typename T, int MB, int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v8(const GemmParams p, const int ntiles, const int stagger) {
    static_assert(sizeof(T) == 2, "v8 is a bf16 kernel");
    constexpr int TM = 64 * MB;
    constexpr int TN = 256;
    constexpr int A2 = TM * 128, B2 = TN * 128, SLOT2 = A2 + B2;      // one slot: (TM + TN) rows x 128 bytes of K
    constexpr int STAGE = 8 * 2048;                                   // epilogue transposition: 8 rows x 64 fp32 per wave
    static_assert(MB >= 4, "four B pieces ride on the first k-step's MFMA pairs");
    static_assert(2 * SLOT2 + STAGE <= 160 * 1024, "one workgroup per CU");
    // four LDS objects, the A / B part of either slot (hipcc's alias scopes are per object -- gemm_kernel_v4: the ds_reads of one slot
    // must not be taken to alias the DMA writes in flight to the other)
    __shared__ __attribute__((aligned(16))) unsigned char ringA0[A2];
    __shared__ __attribute__((aligned(16))) unsigned char ringA1[A2];
    __shared__ __attribute__((aligned(16))) unsigned char ringB0[B2];
    __shared__ __attribute__((aligned(16))) unsigned char ringB1[B2];
    __shared__ __attribute__((aligned(16))) unsigned char stage[STAGE];
    constexpr int EB = 2;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, half = lane >> 5;
    const int ntn = (p.N + TN - 1) / TN, ntm = (p.M + TM - 1) / TM;
    const int nblk = (ntn + 7) / 8;
    const int wblk = (ntn + nblk - 1) / nblk;
    // virtual block id -> tile origin (v7's mapping with the virtual grid size: bid and bid + k * gridDim.x sit on the same XCD)
    auto tile_of = [&](int vb, int& m0, int& n0) {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
        const int blk = swz / (ntm * wblk);
        const int rem = swz - blk * (ntm * wblk);
        const int wcur = min(wblk, ntn - blk * wblk);
        const int tile_m = rem / wcur;
        m0 = tile_m * TM;
        n0 = (blk * wblk + (rem - tile_m * wcur)) * TN;
    };
    const unsigned Kb = (unsigned)p.K * EB;
    const int nk2 = (int)(Kb / 128);
    const unsigned ldaB = (unsigned)p.lda * EB, ldbB = (unsigned)p.ldb * EB;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((unsigned)p.M * ldaB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Bt), 0, (int)((unsigned)p.N * ldbB), 0x00020000);

    // DMA pieces (1 KiB = 8 rows x 128 B): per-lane part of the source offset; the tile's origin and the slab are uniform terms
    const int prow = lane >> 3;
    const int psw = ((lane >> 4) + 4 * (wave & 1)) & 7;            // (row >> 1) & 7 of the piece row this lane fetches
    const unsigned pcol = (unsigned)(((lane & 7) ^ psw) * 16);
    const unsigned lpA = (unsigned)(wave * 8 + prow) * ldaB + pcol;
    const unsigned lpB = (unsigned)(wave * 8 + prow) * ldbB + pcol;
    const unsigned strideA = 64u * ldaB, strideB = 64u * ldbB;
    unsigned uA, uB, uAn = 0, uBn = 0;                              // this tile's / the next tile's uniform origin terms
    bool dead_n = true;
    // piece j of slab s (s >= nk2: slab s - nk2 of the NEXT tile) into `slot`
    auto issue_piece = [&](int s, unsigned char* slotA, unsigned char* slotB, int j) {
        const bool nx = s >= nk2;
        const unsigned sb = (unsigned)(nx ? s - nk2 : s) * 128u;
        const bool dead = nx & dead_n;
        if (j < MB) {
            unsigned o = lpA + ((nx ? uAn : uA) + (unsigned)j * strideA + sb);
            o = dead ? 0xffffff00u : o;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(slotA + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
        } else {
            const int jb = j - MB;
            unsigned o = lpB + ((nx ? uBn : uB) + (unsigned)jb * strideB + sb);
            o = dead ? 0xffffff00u : o;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(slotB + (jb * 8 + wave) * 1024), 16, o, 0, 0, 0);
        }
    };
    const int fsw = (l31 >> 1) & 7;
    const int a_row2 = (wm * 32 * MB + l31) * 128, b_row2 = (wn * 64 + l31) * 128;
    int fco[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) fco[h] = ((2 * h + half) ^ fsw) * 16;
    auto rdA2 = [&](const unsigned char* slotA, int h, int mi) {
        return *reinterpret_cast<const uint4*>(slotA + a_row2 + mi * 4096 + fco[h]);
    };
    auto rdB2 = [&](const unsigned char* slotB, int h, int ni) {
        return *reinterpret_cast<const uint4*>(slotB + b_row2 + ni * 4096 + fco[h]);
    };

    f32x16 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
#define PV_V8_PAIR(MI, H)                                                                                     \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                          \
        acc[MI][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                \
            __builtin_bit_cast(bf16x8, fa[MI]), __builtin_bit_cast(bf16x8, fb[(H) & 1][ni]), acc[MI][ni], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
    // slab KT in CUR (visible), slab KT+1 arriving in NXT; fa / fb[0] hold k-step 0 of slab KT (v7's PV_V7_FSTEP)
#define PV_V8_FSTEP(KT, CA, CB, NA, NB)                                                                       \
    _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                                           \
        if (h == 3) {                                                                                         \
            __builtin_amdgcn_s_waitcnt(0x0F70);                                                               \
            __builtin_amdgcn_s_barrier();                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                \
        }                                                                                                     \
        _Pragma("unroll") for (int mi = 0; mi < MB; ++mi) {                                                   \
            PV_V8_PAIR(mi, h)                                                                                 \
            fa[mi] = h < 3 ? rdA2(CA, h < 3 ? h + 1 : 0, mi) : rdA2(NA, 0, mi);                               \
            if (mi == 1) {                                                                                    \
                fb[(h + 1) & 1][0] = h < 3 ? rdB2(CB, h < 3 ? h + 1 : 0, 0) : rdB2(NB, 0, 0);                 \
                fb[(h + 1) & 1][1] = h < 3 ? rdB2(CB, h < 3 ? h + 1 : 0, 1) : rdB2(NB, 0, 1);                 \
            }                                                                                                 \
            if (h == 3) issue_piece((KT) + 2, CA, CB, mi);                                                    \
            if (h == 0 && mi < 4) issue_piece((KT) + 1, NA, NB, MB + mi);                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                \
        }                                                                                                     \
    }
#define PV_V8_EVEN(KT) PV_V8_FSTEP(KT, ringA0, ringB0, ringA1, ringB1)
#define PV_V8_ODD(KT) PV_V8_FSTEP(KT, ringA1, ringB1, ringA0, ringB0)

    int vb = blockIdx.x;
    int m0, n0;
    tile_of(vb, m0, n0);
    uA = (unsigned)m0 * ldaB; uB = (unsigned)n0 * ldbB;
    if (stagger > 0) {
        // (A/B knob: workgroups of a launch start a fraction of a tile apart -- see profiles/r05_notes.md)
        const int k = ((blockIdx.x >> 3) & 3) * stagger;
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
    }
    trace_stamp(p.trace, vb, 0);
    // the stream's head: slab 0 of the first tile entirely, the A pieces of its slab 1, k-step 0's fragments
#pragma unroll
    for (int j = 0; j < (MB + 4); ++j) issue_piece(0, ringA0, ringB0, j);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < MB; ++j) issue_piece(1, ringA1, ringB1, j);

    for (;;) {
        const int vbn = vb + (int)gridDim.x;
        const bool has_next = vbn < ntiles;
        int m0n = 0, n0n = 0;
        if (has_next) tile_of(vbn, m0n, n0n);
        uAn = (unsigned)m0n * ldaB; uBn = (unsigned)n0n * ldbB;
        dead_n = !has_next;

        // epilogue operands of THIS tile in flight under its K loop (v7): bias chunk + residual rows of block 0.  Only the loaded
        // registers live through the loop; every address is recomputed behind it (the loop body runs at the register limit).
        uint4 e_bias = make_uint4(0, 0, 0, 0);
        constexpr int NRES = EPI == PV_EPI_RESID ? MB : 1;
        uint4 e_res[NRES][4];
#pragma unroll
        for (int mi = 0; mi < NRES; ++mi)
#pragma unroll
            for (int it = 0; it < 4; ++it) e_res[mi][it] = make_uint4(0, 0, 0, 0);
#define PV_V8_EPI_GEOMETRY()                                                                                          \
        const int e_lane = lane_id_here();                                                                            \
        const int e_gn = n0 + wn * 64 + (e_lane & 7) * 8;                                                             \
        const bool e_live = e_gn < p.N;                                                                               \
        const int e_rows_left = e_live ? p.M - (m0 + wm * 32 * MB + (e_lane >> 3)) : 0;                               \
        int e_col = e_gn;                                                                                             \
        T* e_out0 = reinterpret_cast<T*>(p.out0);                                                                     \
        const T* e_biasp = reinterpret_cast<const T*>(p.bias0);                                                       \
        if constexpr (EPI == PV_EPI_QKV) {                                                                            \
            const int which = e_gn / p.nsplit;                                                                        \
            e_col = e_gn - which * p.nsplit;                                                                          \
            if (which == 1) { e_out0 = reinterpret_cast<T*>(p.out1); e_biasp = reinterpret_cast<const T*>(p.bias1); } \
            if (which == 2) { e_out0 = reinterpret_cast<T*>(p.out2); e_biasp = reinterpret_cast<const T*>(p.bias2); } \
        }                                                                                                             \
        const T* e_rbase = reinterpret_cast<const T*>(p.resid) + (int64_t)(m0 + wm * 32 * MB + (e_lane >> 3)) * p.ldr + e_gn;
#define PV_V8_FETCH_RES(MI)                                                                          \
        if constexpr (EPI == PV_EPI_RESID) {                                                         \
            _Pragma("unroll") for (int it = 0; it < 4; ++it)                                         \
                if ((MI) * 32 + it * 8 < e_rows_left)                                                \
                    e_res[MI][it] = *reinterpret_cast<const uint4*>(e_rbase + (int64_t)((MI) * 32 + it * 8) * p.ldr); \
        }
        // k-step 0's fragments of the tile's first slab (visible since the last barrier of the previous tile's loop; that loop's
        // last refills fetched the same values, but carrying 36 registers across the epilogue costs the loop body spills)
        uint4 fa[MB], fb[2][2];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) fa[mi] = rdA2(ringA0, 0, mi);
        fb[0][0] = rdB2(ringB0, 0, 0); fb[0][1] = rdB2(ringB0, 0, 1);
        int kt = 0;
        do {                                                       // (nk2 >= 2 and even: the launcher's condition)
            PV_V8_EVEN(kt)
            if (kt == 0) {
                // The epilogue's operands are fetched HERE -- inside the loop, behind the first slab's wait (which also covered the
                // previous tile's stores) and in front of the second's, which retires them: hipcc then knows them landed when the
                // epilogue reads them.  Fetched in front of the loop, their first use behind it (the next tile's pieces in flight)
                // is answered with a vmcnt(0); named as used inside the loop, the queue is flushed in the loop's preheader.
                PV_V8_EPI_GEOMETRY()
                (void)e_out0;
                if (e_live && e_biasp) e_bias = *reinterpret_cast<const uint4*>(e_biasp + e_col);
                if constexpr (MB < 5) { PV_V8_FETCH_RES(0) }        // (MB = 5: sixteen more registers through the loop body spill)
            }
            PV_V8_ODD(kt + 1)
            kt += 2;
        } while (kt + 2 <= nk2);
        trace_stamp(p.trace, vb, 1);

        // ---- store epilogue: 8 rows x 64 columns per pass through the wave's own 2 KB (the ring holds the next tile)
        PV_V8_EPI_GEOMETRY()
        (void)e_biasp;
        const pv_f32x2 e_b[4] = {unpack2(e_bias.x), unpack2(e_bias.y), unpack2(e_bias.z), unpack2(e_bias.w)};
        const int64_t e_row0 = (int64_t)(m0 + wm * 32 * MB + (e_lane >> 3)) * p.ldo;
        T* const o0_base = e_out0 ? e_out0 + e_row0 + e_col : nullptr;
        T* const o1_base = reinterpret_cast<T*>(p.out1) + e_row0 + e_gn;
        float* const Cs = reinterpret_cast<float*>(stage + wave * 2048);
        const float* const Cr = Cs + (e_lane >> 3) * 64 + (e_lane & 7) * 8;
        float* const Cw = Cs + (e_lane >> 5) * 256 + (e_lane & 31);
        if constexpr (MB >= 5) { PV_V8_FETCH_RES(0) }
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        Cw[j * 64 + ni * 32] = acc[mi][ni][4 * it + j];
                __builtin_amdgcn_wave_barrier();
                if (it == 0) {
                    // residual rows two blocks ahead, as v7
                    if (2 * mi + 1 < MB) { PV_V8_FETCH_RES((2 * mi + 1 < MB ? 2 * mi + 1 : 0)) }
                    if (2 * mi + 2 < MB) { PV_V8_FETCH_RES((2 * mi + 2 < MB ? 2 * mi + 2 : 0)) }
                }
                if (mi * 32 + it * 8 < e_rows_left) {
                    const float4 x0 = *reinterpret_cast<const float4*>(Cr);
                    const float4 x1 = *reinterpret_cast<const float4*>(Cr + 4);
                    const int64_t ro = (int64_t)(mi * 32 + it * 8) * p.ldo;
                    epi8_bf16<EPI, ACT>(x0, x1, e_b, o0_base ? o0_base + ro : nullptr, o1_base + ro,
                                        e_res[EPI == PV_EPI_RESID ? mi : 0][it]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#undef PV_V8_EPI_GEOMETRY
#undef PV_V8_FETCH_RES
        trace_stamp(p.trace, vb, 2);
        if (!has_next) break;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        vb = vbn; m0 = m0n; n0 = n0n; uA = uAn; uB = uBn;
        trace_stamp(p.trace, vb, 0);
    }
#undef PV_V8_EVEN
#undef PV_V8_ODD
#undef PV_V8_FSTEP
#undef PV_V8_PAIR
    __builtin_amdgcn_s_waitcnt(0x0F70);     // the dead prefetches behind the last tile must land before the LDS is handed on
}

// CUs a launch may count on: the launch's own budget (a plan's pipeline on CU-masked streams), else the tuning override, else the
// current device's count (cached per device id; rounded down to a multiple of the XCD count: bid and bid + grid on one XCD)
inline int pv_gemm_cus(const GemmParams& p) {
    int n = p.cus > 0 ? p.cus : g_pv_tuning.gemm_cus;
    if (n <= 0) {
        static std::atomic<int> per_dev[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        n = per_dev[dev].load(std::memory_order_relaxed);
        if (n == 0) {
            hipDeviceProp_t prop;
            n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
            if (n <= 0) n = 256;
            per_dev[dev].store(n, std::memory_order_relaxed);
        }
    }
    n -= n % 8;
    return n <= 0 ? 8 : n;
}

template <typename T, int MB>
int launch_v7(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + 64 * MB - 1) / (64 * MB), ntn = (p.N + 255) / 256;
    {
        constexpr double EBd = DT<T>::kBytes;
        const double mn = (double)p.M * p.N;
        double outs = 1.0;
        if (p.epi == PV_EPI_RESID) outs = 2.0 + (p.out0 ? 1.0 : 0.0);
        if (p.epi == PV_EPI_ACT) outs = 1.0 + (p.out0 ? 1.0 : 0.0);
        const dim3 grid(ntm * ntn), block(512);
        // the software-pipelined K loop covers plain A operands and whole 64-byte slabs
        // K loop: 2 = full-line form (whole 128-byte slabs: every B/32, L/14 shape), 1 = pipelined 64-byte slabs, 0 = the
        // barrier-then-fetch loop (patch gather, K tails).  gemm_loop: -1 auto, 0 / 1 force the simpler forms where legal.
        const int64_t kbytes = (int64_t)p.K * DT<T>::kBytes;
        int loop_sel = 0;
        if (p.a_mode == PV_A_PLAIN && kbytes % 64 == 0 && g_pv_tuning.gemm_loop != 0)
            loop_sel = (kbytes % 128 == 0 && g_pv_tuning.gemm_loop != 1) ? 2 : 1;
        // persistent form (v8): the full-line loop's shapes with an even slab count, when the launch has more tiles than CUs
        // (gemm_persist: -1 auto, 0 never, 1 wherever legal)
        const int n_cu = pv_gemm_cus(p);
        const int ntiles = ntm * ntn;
        const bool persist = loop_sel == 2 && (kbytes / 128) % 2 == 0 && g_pv_tuning.gemm_persist != 0 &&
                             (g_pv_tuning.gemm_persist > 0 || ntiles > n_cu);
        const dim3 pgrid(ntiles < n_cu ? ntiles : n_cu);
        const int stagger = g_pv_tuning.gemm_stagger;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        const bool timed = !g_pv_tuning.prof_markers &&
                           pv_prof_events(PV_PROF_GEMM, 2.0 * mn * p.K, ((double)p.M * p.K + (double)p.N * p.K + outs * mn) * EBd, &ev0, &ev1);
        ProfScope prof(timed ? PV_PROF__COUNT : PV_PROF_GEMM, stream, 2.0 * mn * p.K, ((double)p.M * p.K + (double)p.N * p.K + outs * mn) * EBd);
#define PV_V8_LAUNCH(EPI, ACT)                                                                                                   \
    do {                                                                                                                         \
        if (timed) hipExtLaunchKernelGGL((gemm_kernel_v8<T, MB, EPI, ACT>), pgrid, block, 0, stream, ev0, ev1, 0, p, ntiles, stagger); \
        else hipLaunchKernelGGL((gemm_kernel_v8<T, MB, EPI, ACT>), pgrid, block, 0, stream, p, ntiles, stagger);                  \
    } while (0)
#define PV_V7_LAUNCH_LP(EPI, ACT, LP)                                                                                   \
    do {                                                                                                                \
        if (timed) hipExtLaunchKernelGGL((gemm_kernel_v7<T, MB, EPI, ACT, LP>), grid, block, 0, stream, ev0, ev1, 0, p); \
        else hipLaunchKernelGGL((gemm_kernel_v7<T, MB, EPI, ACT, LP>), grid, block, 0, stream, p);                      \
    } while (0)
#define PV_V7_LAUNCH(EPI, ACT)                                                   \
    do {                                                                         \
        if (persist) PV_V8_LAUNCH(EPI, ACT);                                     \
        else if (loop_sel == 2) PV_V7_LAUNCH_LP(EPI, ACT, 2);                    \
        else if (loop_sel) PV_V7_LAUNCH_LP(EPI, ACT, 1);                         \
        else PV_V7_LAUNCH_LP(EPI, ACT, 0);                                       \
    } while (0)
        if (p.epi == PV_EPI_BIAS) PV_V7_LAUNCH(PV_EPI_BIAS, 0);
        else if (p.epi == PV_EPI_QKV) PV_V7_LAUNCH(PV_EPI_QKV, 0);
        else if (p.epi == PV_EPI_RESID) PV_V7_LAUNCH(PV_EPI_RESID, 0);
        else if (p.act == PV_ACT_GELU) PV_V7_LAUNCH(PV_EPI_ACT, PV_ACT_GELU);
        else if (p.act == PV_ACT_QUICK_GELU) PV_V7_LAUNCH(PV_EPI_ACT, PV_ACT_QUICK_GELU);
        else PV_V7_LAUNCH(PV_EPI_ACT, PV_ACT_RELU);
#undef PV_V7_LAUNCH
#undef PV_V7_LAUNCH_LP
#undef PV_V8_LAUNCH
    }
    PV_LAUNCH_CHECK("gemm_kernel_v7");
    return PV_OK;
}

// v4 or v7 for this shape?  Cost model = rounds over the chip x time of one round, the round times being the
// measured per-tile K-loop + epilogue of the two kernels at K = 768 on the B/32 shapes (tools/gemm_trace.py):
// v4 25 us for 3 x (128 x 128) per CU, v7 26.5 us (MB = 4) / 32 us (MB = 5) for one (64*MB) x 256 per CU.
inline int pick_v7(const GemmParams& p) {
    if (g_pv_tuning.gemm_tile >= 0) return g_pv_tuning.gemm_tile;           // 0 = v4, 4 / 5 = v7<MB>
    auto rounds = [](int64_t tiles, int64_t slots) { return (double)((tiles + slots - 1) / slots); };
    const int64_t M = p.M, N = p.N, cus = pv_gemm_cus(p);
    // (the round times predate the full-line loop, which makes the v7 tiles about 10 % faster; scaling them moved only the
    // 512 x 512 head GEMM from 16 small tiles to 4 large ones -- slower -- so the measured constants stay)
    const double c4 = rounds(((M + 127) / 128) * ((N + 127) / 128), 3 * cus) * 25.0;
    const double c74 = rounds(((M + 255) / 256) * ((N + 255) / 256), cus) * 26.5;
    const double c75 = rounds(((M + 319) / 320) * ((N + 255) / 256), cus) * 32.0;
    if (c4 <= c74 && c4 <= c75) return 0;
    return c75 <= c74 ? 5 : 4;
}

template <typename T>
int dispatch(GemmParams& p, hipStream_t stream) {
    constexpr int EB = DT<T>::kBytes;
    const bool kvec = ((int64_t)p.K * EB) % 16 == 0;
    const bool b_vec = kvec && (p.ldb * EB) % 16 == 0 && pv_aligned16(p.Bt);
    bool a_vec;
    if (p.a_mode == PV_A_PLAIN) {
        a_vec = kvec && (p.lda * EB) % 16 == 0 && pv_aligned16(p.A);
    } else {
        a_vec = (p.pP * EB) % 16 == 0 && (p.pS * EB) % 16 == 0 && pv_aligned16(p.A);
    }
    const bool vec = a_vec && b_vec;
    // vector epilogue legality
    bool vo = (p.ldo * EB) % 16 == 0 && pv_aligned16(p.out0) && pv_aligned16(p.out1) && pv_aligned16(p.out2) &&
              pv_aligned16(p.bias0) && pv_aligned16(p.bias1) && pv_aligned16(p.bias2);
    if (p.epi == PV_EPI_RESID) vo = vo && (p.ldr * EB) % 16 == 0 && pv_aligned16(p.resid);
    if (p.epi == PV_EPI_QKV) vo = vo && (p.nsplit % 8) == 0;
    p.vec_out = vo ? 1 : 0;
    if (p.a_mode == PV_A_PLAIN && vec) {
        const uint64_t spanA = ((uint64_t)p.M + BM) * (uint64_t)p.lda * EB, spanB = ((uint64_t)p.N + BN) * (uint64_t)p.ldb * EB;
        if (spanA < 0xffffff00ull && spanB < 0xffffff00ull && !g_pv_tuning.gemm_v1) {
            // 128 x 128 tiles, 3 workgroups / CU (v4) or one 8-wave workgroup with a (64*MB) x 256 tile (v7);
            // history and measurements of the variants in between: profiles/r01_notes.md
            if constexpr (EB == 2) {
                if (p.vec_out && p.N % 8 == 0) {
                    const int pick = pick_v7(p);
                    if (pick == 5) return launch_v7<T, 5>(p, stream);
                    if (pick == 4) return launch_v7<T, 4>(p, stream);
                }
            }
            return launch_v4<T>(p, stream);
        }
    }
    if constexpr (EB == 2) {
        // patch embedding at patch size 32: the tiled kernel gathers the patches itself (see gemm_kernel_v7)
        const uint64_t img_bytes = (uint64_t)(p.pG ? p.M / (p.pG * p.pG) : 0) * p.pC * p.pS * p.pS * EB;
        if (p.a_mode == PV_A_PATCH && vec && p.vec_out && p.N % 8 == 0 && p.pP == 32 && p.K == p.pC * 1024 && p.pG > 0 &&
            p.M % (p.pG * p.pG) == 0 && img_bytes < 0xffffff00ull && (uint64_t)(p.N + 256) * p.ldb * EB < 0xffffff00ull &&
            !g_pv_tuning.gemm_v1 && !g_pv_tuning.gemm_v1patch) {
            const int pick = pick_v7(p);
            if (pick == 5) return launch_v7<T, 5>(p, stream);
            if (pick == 4) return launch_v7<T, 4>(p, stream);
        }
    }
    if (p.a_mode == PV_A_PLAIN) {
        return vec ? launch<T, PV_A_PLAIN, true>(p, stream) : launch<T, PV_A_PLAIN, false>(p, stream);
    }
    return vec ? launch<T, PV_A_PATCH, true>(p, stream) : launch<T, PV_A_PATCH, false>(p, stream);
}

}  // namespace

namespace {
uint64_t* g_trace_dev = nullptr;
int g_trace_countdown = -1;
int32_t g_trace_info[6] = {0, 0, 0, 0, 0, 0};
constexpr int TRACE_MAX_WG = 8192;
}  // namespace

extern "C" int pv_debug_gemm_trace_arm(int32_t launch_idx) {
    if (!g_trace_dev) {
        PV_HIP_CHECK(hipMalloc(&g_trace_dev, (size_t)TRACE_MAX_WG * 4 * sizeof(uint64_t)));
    }
    PV_HIP_CHECK(hipMemset(g_trace_dev, 0, (size_t)TRACE_MAX_WG * 4 * sizeof(uint64_t)));
    g_trace_countdown = launch_idx;
    return PV_OK;
}

extern "C" int pv_debug_gemm_trace_read(uint64_t* host_out, int32_t max_wg, int32_t* info6) {
    PV_REQUIRE(g_trace_dev && host_out && info6, "trace not armed");
    PV_HIP_CHECK(hipDeviceSynchronize());
    const int n = max_wg < g_trace_info[4] ? max_wg : g_trace_info[4];
    PV_HIP_CHECK(hipMemcpy(host_out, g_trace_dev, (size_t)n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) info6[i] = g_trace_info[i];
    return PV_OK;
}

namespace {
template <typename T>
__global__ __launch_bounds__(256) void act_rows_kernel(const T* __restrict__ pre, T* __restrict__ post, int64_t n8, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    load8(pre + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_any<T>(v[e], act);
    store8(post + i * 8, v);
}
}  // namespace

int pv_launch_act(int dtype, int act, const void* pre, void* post, int64_t n, hipStream_t stream) {
    PV_REQUIRE(pre && post && n > 0 && n % 8 == 0 && pv_aligned16(pre) && pv_aligned16(post), "activation launcher arguments");
    const int64_t n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256)), block(256);
    if (dtype == PV_DTYPE_BF16)
        hipLaunchKernelGGL(act_rows_kernel<bf16_t>, grid, block, 0, stream, reinterpret_cast<const bf16_t*>(pre), reinterpret_cast<bf16_t*>(post), n8, act);
    else
        hipLaunchKernelGGL(act_rows_kernel<float>, grid, block, 0, stream, reinterpret_cast<const float*>(pre), reinterpret_cast<float*>(post), n8, act);
    PV_LAUNCH_CHECK("act_rows_kernel");
    return PV_OK;
}

int pv_launch_gemm(int dtype, GemmParams p, hipStream_t stream) {
    // instance tag for the per-instance roofline (bench.py): 1 QKV, 2 O-projection, 3 MLP-1, 4 MLP-2, 0 everything else
    struct TagScope { TagScope(int t) { pv_prof_set_tag(t); } ~TagScope() { pv_prof_set_tag(0); } }
        tag_scope(p.epi == PV_EPI_QKV ? 1 : p.epi == PV_EPI_ACT ? 3 : p.epi == PV_EPI_RESID ? (p.K > p.N ? 4 : 2) : 0);
    p.dbg = g_pv_tuning.gemm_dbg;
    p.trace = nullptr;
    if (g_trace_countdown >= 0 && p.a_mode == PV_A_PLAIN && !p.b_kn) {
        if (g_trace_countdown-- == 0) {
            p.trace = g_trace_dev;
            const bool bf16_big = dtype == PV_DTYPE_BF16 && p.M > 0 && (p.ldo * 2) % 16 == 0 && p.N % 8 == 0;
            const int v7 = bf16_big ? pick_v7(p) : 0;           // (mirrors dispatch(): which kernel will trace this launch)
            const int tm = v7 == 5 ? 320 : (v7 ? 256 : 128), tn = v7 ? 256 : 128;
            g_trace_info[0] = p.M; g_trace_info[1] = p.N; g_trace_info[2] = p.K; g_trace_info[3] = p.epi;
            g_trace_info[4] = ((p.M + tm - 1) / tm) * ((p.N + tn - 1) / tn); g_trace_info[5] = v7 ? 70 + v7 : 4;
            if (g_trace_info[4] > TRACE_MAX_WG) p.trace = nullptr;
        }
    }
    PV_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm dims must be positive");
    PV_REQUIRE(p.A && p.Bt, "gemm operands must be non-null");
    if (p.epi == PV_EPI_BIAS) PV_REQUIRE(p.out0, "EPI_BIAS needs out0");
    if (p.epi == PV_EPI_QKV) PV_REQUIRE(p.out0 && p.out1 && p.out2 && p.nsplit > 0 && p.N == 3 * p.nsplit, "EPI_QKV outputs");
    if (p.epi == PV_EPI_RESID) PV_REQUIRE(p.out1 && p.resid, "EPI_RESID needs out1 and resid");
    if (p.epi == PV_EPI_ACT) PV_REQUIRE(p.out1, "EPI_ACT needs out1");
    if (p.b_kn) {
        PV_REQUIRE(dtype == PV_DTYPE_F32 && p.a_mode == PV_A_PLAIN, "[K][N] B operand: fp32, plain A only");
        PV_REQUIRE(p.N % 4 == 0 && p.ldb % 4 == 0 && pv_aligned16(p.Bt), "[K][N] B operand alignment");
        PV_REQUIRE(((int64_t)p.K * 4) % 16 == 0 && (p.lda * 4) % 16 == 0 && pv_aligned16(p.A), "[K][N] path needs 16-byte aligned A rows");
        p.vec_out = ((p.ldo * 4) % 16 == 0 && pv_aligned16(p.out0) && pv_aligned16(p.out1) && pv_aligned16(p.bias0)) ? 1 : 0;
        return launch<float, PV_A_PLAIN, true, true>(p, stream);
    }
    if (dtype == PV_DTYPE_BF16) return dispatch<bf16_t>(p, stream);
    if (dtype == PV_DTYPE_F32) return dispatch<float>(p, stream);
    pv_set_error("gemm: unsupported dtype");
    return PV_ERR_INVALID;
}
