// SAE training step for gfx950 (C ABI: pv_sae_* in include/pv_native.h).
//
// Reference semantics: StandardSparseAutoencoder.forward (/root/reference/src/vit_prisma/sae/sae.py:
// 557-645), TopK (:795-810), _compute_mse_loss (:144-149), run-time LayerNorm of the input (:78-93),
// and VisionSAETrainer.train_step (sae/train_sae.py:278-411).  fp32 throughout (the reference's SAE
// config has no bf16, sae/config.py:14-45).
//
// Data flow of one step on N tokens (k-sparse everywhere after the encoder):
//   prep        x -> mu, std (unbiased), x_hat = (x - mu)/(std + eps), sae_in = x_hat - b_dec,
//               norm_n = ||x_n - mean_batch(x)||_2                                    (one wave / token)
//   encode      hidden_pre = sae_in @ W_enc + b_enc      fp32 MFMA GEMM, W_enc read in its own [K][N] layout
//   topk        exact per-row radix select of the k largest -> (idx, relu(val))       (one workgroup / token)
//   decode      sae_out = (sum_s val_s W_dec[idx_s] + b_dec) * std + mu ; err ; loss partials ;
//               dY = 2 err std / (norm N_glob d_in) ; dh_s = dY . W_dec[idx_s]        (one wave / token)
//   csr         active (token, slot) pairs grouped by feature: count -> scan -> fill
//   backward    per feature j: gW_dec[j,:] = sum a dY[n,:] ; gW_encT[j,:] = sum g sae_in[n,:] ; gb_enc[j] = sum g
//               (rows of both gradients are written coalesced; W_enc's gradient is kept TRANSPOSED,
//               [d_sae][d_in], and transposed back tile-wise inside the Adam kernel)
//   bias grads  gb_dec = colsum(dY) - W_enc @ gb_enc   (the encoder-input path of b_dec, folded into a GEMV)
//   apply       clip coefficient -> remove component parallel to decoder rows -> Adam, fused per tensor
#include <math.h>
#include <string.h>

#include "sae.hpp"

namespace {

constexpr int MAXK = PV_SAE_MAXK;

// ------------------------------------------------------------------------------------------------
// generic helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// column sums of a [rows][d] fp32 matrix (deterministic two-stage reduction).
// stage 1: partial[blk][c] over CS_ROWS-row blocks -- rows / 16 workgroups (256 at N = 4096: every CU busy), 16
// independent loads in flight per thread
constexpr int CS_ROWS = 16;
__device__ __forceinline__ void colsum_partial_body(int bid, const float* __restrict__ x, float* __restrict__ partial, int rows, int d) {
    const int r0 = bid * CS_ROWS;
    for (int c = threadIdx.x; c < d; c += 256) {
        float v[CS_ROWS];
#pragma unroll
        for (int i = 0; i < CS_ROWS; ++i) v[i] = (r0 + i < rows) ? x[(int64_t)(r0 + i) * d + c] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CS_ROWS; ++i) s += v[i];          // fixed order
        partial[(int64_t)bid * d + c] = s;
    }
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                             int rows, int d) {
    colsum_partial_body(blockIdx.x, x, partial, rows, d);
}
// stage 2: out[c] = scale * sum_blk partial[blk][c]; 64 columns per workgroup, 16 partial streams per column (the
// 256 partial rows of a 4096-token batch are 16 loads per thread instead of a 64-deep serial chain)
// (returns out[c] on the threads that wrote one -- part 0, c < d -- and 0 elsewhere)
__device__ __forceinline__ float colsum_final_body(int bid, const float* __restrict__ partial, float* __restrict__ out,
                                                   int nblk, int d, float scale) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = bid * 64 + lane;
    float s = 0.f;
    if (c < d) {
#pragma unroll 8
        for (int b = part; b < nblk; b += 16) s += partial[(int64_t)b * d + c];      // (unrolled: the loads of 8 trips in flight, the sum in order)
    }
    red[part][lane] = s;
    __syncthreads();
    if (part == 0 && c < d) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][lane];        // fixed order
        out[c] = t * scale;
        return t * scale;
    }
    return 0.f;
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            int nblk, int d, float scale) {
    (void)colsum_final_body(blockIdx.x, partial, out, nblk, d, scale);
}

// ------------------------------------------------------------------------------------------------
// prep: LN-in (sae.py:78-87), sae_in (sae.py:563-565), loss normaliser (sae.py:145-147)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sae_prep_body(int bid, const float* __restrict__ x, const float* __restrict__ b_dec,
                                              const float* __restrict__ batch_mean, float* __restrict__ sae_in,
                                              _Float16* __restrict__ x16, float* __restrict__ xnorm_out,
                                              float* __restrict__ mu_out, float* __restrict__ std_out,
                                              float* __restrict__ norm_out, int n_tok, int d, int use_ln, float eps,
                                              int d_true) {
    // d_true < d (a transcoder whose input is narrower than its output, pv_sae_transcoder.d_in_true: rows are padded to the common
    // width d): the statistics run over the d_true real columns, the padding of sae_in / x16 is written as exact zeros
    const int lane = threadIdx.x & 63;
    const int n = bid * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    const float* xr = x + (int64_t)n * d;
    // Rows of up to 64 * PREP_NE elements live in registers: ONE round trip to memory for the row (all of a lane's loads in flight),
    // the three passes below run out of registers.  The loops further down re-read the row from memory with one dependent load per
    // trip -- 3 x 12 round trips at d = 768, 14 of this kernel's 17 us by its waves' PMC lifetimes; they remain for wider rows.  Same
    // operations in the same order either way.
    constexpr int PREP_NE = 20;
    if (d <= 64 * PREP_NE) {
        float xv[PREP_NE], bm[PREP_NE], bd[PREP_NE];
#pragma unroll
        for (int u = 0; u < PREP_NE; ++u) {
            const int i = lane + 64 * u;
            xv[u] = i < d ? xr[i] : 0.f;
            bm[u] = (batch_mean && i < d_true) ? batch_mean[i] : 0.f;
            bd[u] = i < d_true ? b_dec[i] : 0.f;
        }
        float s = 0.f, cn = 0.f;
#pragma unroll
        for (int u = 0; u < PREP_NE; ++u) {
            if (lane + 64 * u < d_true) {
                s += xv[u];
                if (batch_mean) {
                    const float c = xv[u] - bm[u];
                    cn += c * c;
                }
            }
        }
        const float mu = use_ln == 1 ? wave_sum(s) / (float)d_true : 0.f;
        cn = wave_sum(cn);
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < PREP_NE; ++u) {
            if (lane + 64 * u < d_true) {
                const float c = xv[u] - mu;
                sq += c * c;
            }
        }
        sq = wave_sum(sq);
        const float coeff = use_ln == 2 ? sqrtf((float)d_true) / sqrtf(sq) : 1.f;
        const float sd = use_ln == 1 ? sqrtf(sq / (float)(d_true - 1)) : (use_ln == 2 ? 1.f / coeff : 1.f);
        float s2 = 0.f, amax = 0.f;
#pragma unroll
        for (int u = 0; u < PREP_NE; ++u) {
            const int i = lane + 64 * u;
            if (i < d) {
                const float xh = use_ln == 1 ? (xv[u] - mu) / (sd + eps) : (use_ln == 2 ? xv[u] * coeff : xv[u]);
                const float si = i < d_true ? xh - bd[u] : 0.f;
                sae_in[(int64_t)n * d + i] = si;
                if (x16) x16[(int64_t)n * d + i] = (_Float16)si;
                s2 += si * si;
                amax = fmaxf(amax, fabsf(si));
            }
        }
        s2 = wave_sum(s2);
        amax = wave_max(amax);
        if (lane == 0) {
            mu_out[n] = mu;
            std_out[n] = sd;
            if (batch_mean) norm_out[n] = sqrtf(cn);
            if (xnorm_out) xnorm_out[n] = (amax <= 6.0e4f) ? sqrtf(s2) : INFINITY;
        }
        return;
    }
    // batch_mean == NULL: the loss normaliser is left to a later kernel (the select kernel of the fused pre-pass, SaePre)
    float s = 0.f, cn = 0.f;
    if (batch_mean) {
        for (int i = lane; i < d_true; i += 64) {
            const float v = xr[i];
            s += v;
            const float c = v - batch_mean[i];
            cn += c * c;
        }
    } else {
        for (int i = lane; i < d_true; i += 64) s += xr[i];
    }
    // use_ln: 0 none, 1 "layer_norm" (sae.py:74-90), 2 "constant_norm_rescale" (sae.py:60-72: x * c on the way in with
    // c = sqrt(d_in) / ||x||, / c on the way out -- the step's other kernels only know "out = pre * sd + mu": mu = 0, sd = 1 / c)
    const float mu = use_ln == 1 ? wave_sum(s) / (float)d_true : 0.f;
    cn = wave_sum(cn);
    float sq = 0.f;
    for (int i = lane; i < d_true; i += 64) {
        const float c = xr[i] - mu;
        sq += c * c;
    }
    sq = wave_sum(sq);
    // torch.std: unbiased (divide by d - 1)
    const float coeff = use_ln == 2 ? sqrtf((float)d_true) / sqrtf(sq) : 1.f;
    const float sd = use_ln == 1 ? sqrtf(sq / (float)(d_true - 1)) : (use_ln == 2 ? 1.f / coeff : 1.f);
    float s2 = 0.f, amax = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float xh = use_ln == 1 ? (xr[i] - mu) / (sd + eps) : (use_ln == 2 ? xr[i] * coeff : xr[i]);
        const float si = i < d_true ? xh - b_dec[i] : 0.f;
        sae_in[(int64_t)n * d + i] = si;
        if (x16) x16[(int64_t)n * d + i] = (_Float16)si;       // operand of the filter GEMM (sae_enc.hip)
        s2 += si * si;
        amax = fmaxf(amax, fabsf(si));
    }
    s2 = wave_sum(s2);
    amax = wave_max(amax);
    if (lane == 0) {
        mu_out[n] = mu;
        std_out[n] = sd;
        if (batch_mean) norm_out[n] = sqrtf(cn);
        // ||sae_in||_2 for the filter's error bound; a row outside the fp16 range (or NaN) is sent to the exact path
        if (xnorm_out) xnorm_out[n] = (amax <= 6.0e4f) ? sqrtf(s2) : INFINITY;
    }
}
__global__ __launch_bounds__(256) void sae_prep_kernel(const float* __restrict__ x, const float* __restrict__ b_dec,
                                                       const float* __restrict__ batch_mean, float* __restrict__ sae_in,
                                                       _Float16* __restrict__ x16, float* __restrict__ xnorm_out,
                                                       float* __restrict__ mu_out, float* __restrict__ std_out,
                                                       float* __restrict__ norm_out, int n_tok, int d, int use_ln, float eps,
                                                       int d_true) {
    sae_prep_body(blockIdx.x, x, b_dec, batch_mean, sae_in, x16, xnorm_out, mu_out, std_out, norm_out, n_tok, d, use_ln, eps, d_true);
}
// The first launch of the fused pre-pass (SaePre): workgroup 0 = the weight bound (below), [1, 1 + nblk) = the 16-row partial column
// sums of x (the batch mean's first stage; nblk = 0 when the caller supplied the mean), the other nb_prep = prep without the loss
// normaliser.  The
// weight bound max_j ||W_enc[:, j]||^2 of the filter's error band + the zeroing of the per-feature pair counters and the fallback
// count (sae_wmax_kernel of sae_enc.hip on 256 threads; a maximum does not care about the order)
__global__ __launch_bounds__(256) void sae_prep_roles_kernel(const float* __restrict__ x, const float* __restrict__ b_dec,
                                                             float* __restrict__ sae_in, _Float16* __restrict__ x16,
                                                             float* __restrict__ xnorm_out, float* __restrict__ mu_out,
                                                             float* __restrict__ std_out, int n_tok, int d, int use_ln, float eps,
                                                             int d_true, int nb_prep, float* __restrict__ colpart, int nblk,
                                                             const float* __restrict__ colsq, int d_sae, float* __restrict__ wmax_out,
                                                             uint32_t* __restrict__ fb_count, uint32_t* __restrict__ feat_cnt) {
    // (the one workgroup of the weight bound is the launch's longest job: it goes FIRST, the column sums next, the tokens last)
    const int b = blockIdx.x;
    if (b > nblk) {
        sae_prep_body(b - 1 - nblk, x, b_dec, nullptr, sae_in, x16, xnorm_out, mu_out, std_out, nullptr, n_tok, d, use_ln, eps, d_true);
    } else if (b > 0) {
        colsum_partial_body(b - 1, x, colpart, n_tok, d);
    } else {
        __shared__ float red[4];
        // (one workgroup for d_sae values and the launch's long pole: 16-byte accesses, all of a thread's loads in flight at once;
        // d_sae % 4 == 0 is a plan requirement)
        const int n4 = d_sae >> 2;
        if (feat_cnt)
            for (int j = threadIdx.x; j < n4; j += 256) reinterpret_cast<uint4*>(feat_cnt)[j] = make_uint4(0u, 0u, 0u, 0u);
        float m = 0.f;
        constexpr int WM_FL = 12;                                  // (the bench shape: two round trips to memory; 24 in flight cost the launch its occupancy)
        for (int j0 = threadIdx.x; j0 < n4; j0 += WM_FL * 256) {
            float4 c[WM_FL];
#pragma unroll
            for (int u = 0; u < WM_FL; ++u) c[u] = j0 + u * 256 < n4 ? reinterpret_cast<const float4*>(colsq)[j0 + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < WM_FL; ++u) {
                const float e[4] = {c[u].x, c[u].y, c[u].z, c[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) m = (e[i] == e[i]) ? fmaxf(m, e[i]) : INFINITY;      // a NaN column norm poisons the bound (-> exact fallback)
            }
        }
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            *wmax_out = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            *fb_count = 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// exact top-k per row by radix select on the order-preserving uint image of the floats
// (TopK.forward, sae.py:795-810: torch.topk -> relu -> scatter; only the selected SET matters)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int TOPK_CAP = PV_TOPK_CAP;   // candidate list capacity of the fast path

// Per-row exact top-k, streaming (no per-thread value cache -> ~40 VGPRs, full occupancy):
//   pass 1  each thread streams its share of the row keeping only its maximum
//   T0      the k-th largest of the 256 per-thread maxima is a lower bound of the global k-th largest
//           value (every maximum is an element), found by an all-pairs rank over 256 LDS values
//   pass 2  the row is streamed again (it is L2-resident: 96 KB just read) and the few elements >= T0
//           are compacted into LDS
//   rank    exact rank of every candidate by (value desc, column asc): rank < k -> output slot = rank, so
//           the result is sorted by value like torch.topk and run-to-run deterministic
// Fallback (more than TOPK_CAP candidates: massive ties at the top, e.g. constant rows): MSB-first radix
// select that re-streams the row once per digit.
// `row_list` != nullptr: the workgroups walk rows row_list[blockIdx.x], row_list[blockIdx.x + gridDim.x], ... (the tokens
// the filtered encoder of sae_enc.hip could not decide); otherwise row = blockIdx.x.
// `feat_cnt` / `wpos` (both or neither): every kept (token, slot) with a positive value draws its position inside its
// feature's pair list from feat_cnt[feature] (zeroed by the caller) -- the CSR-by-feature of the backward then needs only
// a scan and an atomic-free scatter.
__device__ void sae_topk_row(const float* __restrict__ hidden, int32_t* __restrict__ idx_out, float* __restrict__ val_out,
                             int d_sae, int k, int64_t row, uint32_t* __restrict__ feat_cnt, uint32_t* __restrict__ wpos);

__global__ __launch_bounds__(256) void sae_topk_kernel(const float* __restrict__ hidden, int32_t* __restrict__ idx_out,
                                                       float* __restrict__ val_out, int d_sae, int k,
                                                       const int32_t* __restrict__ row_list, const uint32_t* __restrict__ n_list,
                                                       uint32_t* __restrict__ feat_cnt, uint32_t* __restrict__ wpos) {
    if (!row_list) {
        sae_topk_row(hidden, idx_out, val_out, d_sae, k, blockIdx.x, feat_cnt, wpos);
        return;
    }
    const uint32_t n = *n_list;
    for (uint32_t s = blockIdx.x; s < n; s += gridDim.x) {
        __syncthreads();
        sae_topk_row(hidden, idx_out, val_out, d_sae, k, row_list[s], feat_cnt, wpos);
    }
}

__device__ void sae_topk_row(const float* __restrict__ hidden, int32_t* __restrict__ idx_out, float* __restrict__ val_out,
                             int d_sae, int k, int64_t row, uint32_t* __restrict__ feat_cnt, uint32_t* __restrict__ wpos) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cand_key[TOPK_CAP];
    __shared__ int32_t cand_idx[TOPK_CAP];
    sae_topk_row_lds(hidden, idx_out, val_out, d_sae, k, row, feat_cnt, wpos, hist, cand_key, cand_idx);      // (sae.hpp)
}

// ------------------------------------------------------------------------------------------------
// decode + LN-out + loss partial + dY + dh    (one wave per token)
// ------------------------------------------------------------------------------------------------
// A wave owns a token; lane l owns the 16-byte column groups 4 l + 256 i (i < V4: d_in <= 256 V4), so every gathered
// W_dec row is fetched as V4 16-byte loads per lane (1 KiB per wave-instruction), four rows in flight.
__device__ __forceinline__ float4 ld4(const float* p, bool ok) {
    return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// The optimizer state (both Adam moments, 302 MB) and the gradients are touched ONCE per step: streamed past the caches
// (nontemporal), so that what the next step gathers by row -- W_dec, W_enc^T and its fp16 copy, 188 MB, written by the same
// kernels -- is what stays in the 256 MB MALL.  -DPV_NO_NT (A/B builds): plain accesses.
typedef float pv_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p, bool ok) {
#ifdef PV_NO_NT
    return ld4(p, ok);
#else
    if (!ok) return make_float4(0.f, 0.f, 0.f, 0.f);
    const pv_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pv_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ float ld1_stream(const float* p) {
#ifdef PV_NO_NT
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}
__device__ __forceinline__ void st1_stream(float* p, float v) {
#ifdef PV_NO_NT
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ void st4_stream(float* p, const float4& v) {
#ifdef PV_NO_NT
    *reinterpret_cast<float4*>(p) = v;
#else
    __builtin_nontemporal_store(pv_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<pv_f32x4*>(p));
#endif
}

// The exclusive scan of csr_scan_kernel as ONE 256-thread workgroup without its LDS staging (a role of the decode launch, see
// sae_decode_kernel: the scan reads the selection's counts only, so it does not have to wait for the decode kernel -- it runs inside it):
// every thread owns a contiguous run of counts, sums it with 16-byte loads, the runs are scanned by shuffles, the offsets are written
// in a second walk over the (cached) counts.  Integer arithmetic: the offsets are csr_scan_kernel's.  Also its other duties: the
// total (offs[d_sae], scalars[2] = l0) and the zeroing of the long-list counters.  The loss reduction that rides in csr_scan_kernel's
// workgroup cannot come along (it needs the decode kernel's output): loss_reduce_body, a role of csr_post_fill_kernel.
struct ScanRole {
    const uint32_t* cnt; uint32_t* offs; uint32_t* n_long; int d_sae; float* scalars; float inv_tokens;
};
__device__ __forceinline__ void scan_body_256(const ScanRole& r) {
    __shared__ uint32_t sc_wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (((r.d_sae + 255) / 256) + 3) & ~3;
    const int lo = min(tid * per, r.d_sae), hi = min(lo + per, r.d_sae);
    const bool quads = (r.d_sae & 3) == 0;                   // (then lo and hi are multiples of four)
    uint32_t s = 0;
    {
        int i = lo;
        if (quads) {
#pragma unroll 8
            for (; i + 4 <= hi; i += 4) {
                const uint4 c = *reinterpret_cast<const uint4*>(r.cnt + i);
                s += c.x + c.y + c.z + c.w;
            }
        }
        for (; i < hi; ++i) s += r.cnt[i];
    }
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(inc, o, 64);
        if (lane >= o) inc += a;
    }
    if (lane == 63) sc_wsum[wv] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        base += w < wv ? sc_wsum[w] : 0u;
        total += sc_wsum[w];
    }
    uint32_t run = base + inc - s;                           // exclusive prefix of this thread's run
    {
        int i = lo;
        if (quads)
#pragma unroll 8
            for (; i + 4 <= hi; i += 4) {
                const uint4 c = *reinterpret_cast<const uint4*>(r.cnt + i);
                uint4 o;
                o.x = run; o.y = run + c.x; o.z = o.y + c.y; o.w = o.z + c.z;
                run = o.w + c.w;
                *reinterpret_cast<uint4*>(r.offs + i) = o;
            }
        for (; i < hi; ++i) {
            const uint32_t c = r.cnt[i];
            r.offs[i] = run;
            run += c;
        }
    }
    if (tid == 255) {
        r.offs[r.d_sae] = total;
        if (r.scalars) r.scalars[2] = (float)total * r.inv_tokens;      // l0 = mean_n #(val > 0), train_sae.py:364
    }
    if (tid == 0) { r.n_long[0] = 0u; r.n_long[1] = 0u; r.n_long[2] = 0u; }      // counters of csr_post_kernel; the ticket of colsum_final_sq_kernel
}

// MODE 0: the whole thing.  The feature-parallel step cuts it at the reconstruction: MODE 1 = this rank's PARTIAL
// reconstruction sum_s val_s W_dec[idx_s] only (written to sae_out, no b_dec, no LN-out); MODE 2 = everything behind it, the
// reconstruction summed over the ranks coming in through pre_sum.
#ifndef PV_DEC_ROWS
#define PV_DEC_ROWS 4
#endif
template <int V4, int MODE = 0>
__global__ __launch_bounds__(256) void sae_decode_kernel(
    const float* __restrict__ x, const float* __restrict__ W_dec, const float* __restrict__ b_dec,
    const int32_t* __restrict__ idx, const float* __restrict__ val, const float* __restrict__ mu,
    const float* __restrict__ sd, const float* __restrict__ norm, float* __restrict__ sae_out,
    float* __restrict__ dY, float* __restrict__ dh, float* __restrict__ loss_partial, int n_tok, int d, int k,
    float grad_scale /* 2 / (N_global * d_in) */, int want_grad, const float* __restrict__ inv_norm,
    const float* __restrict__ pre_sum = nullptr, const float* __restrict__ addend = nullptr, float dh_add = 0.f,
    const uint32_t* __restrict__ tok_cnt = nullptr, const uint32_t* __restrict__ gate = nullptr, const ScanRole scan = ScanRole{}) {
    // scan.cnt != NULL: the launch carries one more workgroup than the tokens need, and that one runs the CSR scan (scan_body_256)
    if (MODE != 1 && scan.cnt && blockIdx.x == gridDim.x - 1) {
        scan_body_256(scan);
        return;
    }
    // the sparse form of the ReLU + L1 step (pv_sae_relu_step): k = the per-token capacity, tok_cnt[n] = the pairs token n holds
    // (front-packed; the rest of the row are holes), dh_add = l1_coefficient / N_global (the L1 term's gradient on every kept
    // activation, sae.py:617-626), gate: the step's mode word -- nonzero = the step runs on the dense GEMMs, leave at once
    // transcoder (pv_sae_state.tc): x = the TARGET, b_dec = b_dec_out, addend = the skip term x_in @ W_skip^T or nullptr
    // inv_norm != nullptr: set_decoder_norm_to_unit_norm is pending -- W_dec still holds the un-normalised rows and row j
    // stands for W_dec[j] * inv_norm[j] (the Adam kernel writes the normalised + updated row; see pv_sae_step)
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    if (gate && *gate != 0u) return;
    bool ok[V4];
    int col[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        col[i] = 4 * lane + 256 * i;
        ok[i] = col[i] < d;
    }
    float4 acc[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int32_t* ir = idx + (int64_t)n * k;
    const float* vr = val + (int64_t)n * k;
    const int k_walk = tok_cnt ? min(k, (int)((tok_cnt[n] + 3u) & ~3u)) : k;       // (slots beyond it are holes)
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < V4; ++i) acc[i] = ld4(pre_sum + (int64_t)n * d + col[i], ok[i]);
    }
    if (MODE == 0 && addend) {
#pragma unroll
        for (int i = 0; i < V4; ++i) acc[i] = ld4(addend + (int64_t)n * d + col[i], ok[i]);
    }
    // DEC_R rows of W_dec in flight per trip (k_walk is a multiple of 4; a trip that reaches past it gathers nothing for those slots: a = 0).
    // 4: 106 VGPRs = four waves per SIMD = all 4096 tokens of the bench batch resident at once.  8 halves the dependent trips and
    // costs a wave per SIMD: 66 -> 90 us (tools/probes/dec_rows_ab.sh, -DPV_DEC_ROWS=8; same bits)
    constexpr int DEC_R = PV_DEC_ROWS;
    for (int s = 0; MODE != 2 && s < k_walk; s += DEC_R) {
        float a[DEC_R];
        float4 w[DEC_R][V4];
#pragma unroll
        for (int u = 0; u < DEC_R; ++u) {
            const int su = min(s + u, k - 1);
            const int ju = ir[su];
            a[u] = s + u < k_walk ? vr[su] : 0.f;
            const bool live = a[u] != 0.f;                        // (wave-uniform) a hole -- a candidate that lost the global top-k of the
                                                                  // feature-parallel step, a clamped negative -- gathers nothing
            if (inv_norm) a[u] *= inv_norm[ju];
            const float* wr = W_dec + (int64_t)ju * d;
#pragma unroll
            for (int i = 0; i < V4; ++i) w[u][i] = ld4(wr + col[i], ok[i] && live);
        }
#pragma unroll
        for (int u = 0; u < DEC_R; ++u)                   // (slot order, as torch's dense matmul would not care; fixed here)
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                acc[i].x += a[u] * w[u][i].x; acc[i].y += a[u] * w[u][i].y;
                acc[i].z += a[u] * w[u][i].z; acc[i].w += a[u] * w[u][i].w;
            }
    }
    if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < V4; ++i)
            if (ok[i]) *reinterpret_cast<float4*>(sae_out + (int64_t)n * d + col[i]) = acc[i];
        return;
    }
    const float m = mu[n], sdv = sd[n], nf = norm[n];
    float lsum = 0.f;
    float4 g[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok[i]) {
            const float4 bd = *reinterpret_cast<const float4*>(b_dec + col[i]);
            const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)n * d + col[i]);
            float4 o, e;
            o.x = (acc[i].x + bd.x) * sdv + m; o.y = (acc[i].y + bd.y) * sdv + m;      // decode (sae.py:583-595) + LN-out (:89-90)
            o.z = (acc[i].z + bd.z) * sdv + m; o.w = (acc[i].w + bd.w) * sdv + m;
            e.x = o.x - xv.x; e.y = o.y - xv.y; e.z = o.z - xv.z; e.w = o.w - xv.w;
            if (sae_out) *reinterpret_cast<float4*>(sae_out + (int64_t)n * d + col[i]) = o;
            lsum += (e.x * e.x) / nf + (e.y * e.y) / nf + (e.z * e.z) / nf + (e.w * e.w) / nf;      // sae.py:146-148
            g[i].x = grad_scale * e.x / nf * sdv; g[i].y = grad_scale * e.y / nf * sdv;            // dL/d(pre-LN-out reconstruction)
            g[i].z = grad_scale * e.z / nf * sdv; g[i].w = grad_scale * e.w / nf * sdv;
            if (want_grad) *reinterpret_cast<float4*>(dY + (int64_t)n * d + col[i]) = g[i];
        }
    }
    lsum = wave_sum(lsum);
    if (lane == 0) loss_partial[n] = lsum;
    if (!want_grad) return;
    for (int s = 0; s < k_walk; s += DEC_R) {
        float dot[DEC_R];
#pragma unroll
        for (int u = 0; u < DEC_R; ++u) {
            const int su = min(s + u, k - 1);
            const int ju = ir[su];
            const bool live = s + u < k_walk && vr[su] > 0.f;     // (wave-uniform) dh of a hole is gated to 0 below: no gather
            const float* wr = W_dec + (int64_t)ju * d;
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < V4; ++i) t += dot4(g[i], ld4(wr + col[i], ok[i] && live));
            dot[u] = inv_norm ? t * inv_norm[ju] : t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < DEC_R; ++u) dot[u] += __shfl_xor(dot[u], o, 64);
        // TopK backward: gradient reaches only selected entries; ReLU gate on the kept value
        if (lane < DEC_R && s + lane < k && s + lane < ((k_walk + 3) & ~3)) {
            float dsel = dot[0];
#pragma unroll
            for (int u = 1; u < DEC_R; ++u) dsel = lane == u ? dot[u] : dsel;
            dh[(int64_t)n * k + s + lane] = vr[s + lane] > 0.f ? dsel + dh_add : 0.f;
        }
    }
}

// deterministic scalar reduction: out[slot] = scale * sum(v[0..n))
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ v, float* __restrict__ out, int n,
                                                         float scale, int slot, int slot2, const uint32_t* __restrict__ gate = nullptr,
                                                         uint32_t want = 0u) {
    __shared__ float red[4];
    if (gate && *gate != want) return;                          // (pv_sae_relu_step: only in the mode this sum belongs to)
    // (a fixed order -- the result is run-to-run identical --, four accumulators over 16-byte loads: the one workgroup reduces up to
    // 24 576 partials, and 96 dependent 4-byte loads per thread made each of the dense steps' four reductions 12 us)
    float s;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int i = threadIdx.x * 4; i < n; i += 1024) {
            const float4 t = *reinterpret_cast<const float4*>(v + i);
            s0 += t.x; s1 += t.y; s2 += t.z; s3 += t.w;
        }
        s = (s0 + s1) + (s2 + s3);
    } else {
        s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        out[slot] = s * scale;
        if (slot2 >= 0) out[slot2] = s * scale;
    }
}

// The step's loss from the decode kernel's per-token terms: scalars[0] = scalars[1] = loss_scale * sum, in the order of the 1024-thread
// reduction that rides in csr_scan_kernel's workgroup (thread t of 1024 takes terms t, t + 1024, ...; 16 wave sums; added in wave
// order) -- here by 256 threads that each play four of those threads, so that the two homes of the reduction agree to the bit.
__device__ __forceinline__ void loss_reduce_body(const float* __restrict__ loss_part, int n_loss, float loss_scale, float* __restrict__ scalars) {
    __shared__ float lsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float t = 0.f;
        for (int i = tid + 256 * q; i < n_loss; i += 1024) t += loss_part[i];
        t = wave_sum(t);
        if (lane == 0) lsum[wv + 4 * q] = t;
    }
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += lsum[w];               // fixed order
        scalars[0] = t * loss_scale;
        scalars[1] = t * loss_scale;
    }
}
// ------------------------------------------------------------------------------------------------
// CSR by feature of the active (token, slot) pairs
// ------------------------------------------------------------------------------------------------
// The pair counts per feature come out of the top-k selection (feat_cnt); csr_scan_kernel turns them into offsets,
// csr_fill_kernel scatters the pairs (no atomics), csr_post_kernel (a) cuts the CSR-ordered pair sequence into the chunks
// the short-list backward's waves own -- nominally BWD_CH pairs each, but a cut that would fall inside a list moves
// forward to that list's end, so a chunk is a run of WHOLE lists of which only the last can be long -- and (b) registers
// the features with more than BWD_LMAX pairs (dense features: on the bench batch 1.5 % of the features hold 36 % of the
// pairs) with their BWD_SEG-pair segments.  No gradient row is ever shared between waves, none needs atomics, every row is
// written exactly once.
constexpr int BWD_CH = 16;
constexpr int BWD_LMAX = 64;
constexpr int BWD_SEG = 32;
constexpr int BWD_RANGES = 8;               // token ranges of the long-list backward = XCDs (one L2 each)
// segments of the long lists: ceil(c / SEG) per list with c > LMAX pairs (count-cut form), or BWD_RANGES per list (token-range
// form; at most n_pairs / (LMAX + 1) long lists)
static inline size_t sae_max_segs(size_t n_pairs) {
    const size_t a = n_pairs / BWD_SEG + n_pairs / BWD_LMAX + 1, b = (size_t)BWD_RANGES * (n_pairs / (BWD_LMAX + 1) + 1);
    return a > b ? a : b;
}
// the token-range form sorts a list through a token-indexed LDS array: N tokens x 4 bytes (+ scratch) must fit the 160 KB
static inline bool sae_long_ranged(int n_tokens) { return (size_t)n_tokens * 4 + 4096 <= 150 * 1024; }

// single-workgroup exclusive scan over the d_sae counts, staged through LDS in blocks of 32768 features (one block for
// the 24 576-feature bench shape, two for the x64 SAEs of docs/sae_table.md: 49 152): coalesced load, per-thread contiguous
// runs scanned out of LDS, shuffles across threads, coalesced store, the running total carried into the next block
// loss_part (optional): this workgroup also reduces the decode kernel's per-token loss terms, scalars[0] = scalars[1] = loss_scale *
// their sum in a fixed order (the step's loss; it used to be a launch of its own in front of this one)
__global__ __launch_bounds__(1024) void csr_scan_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ offs,
                                                        uint32_t* __restrict__ n_long, int d_sae,
                                                        float* __restrict__ scalars, float inv_tokens,
                                                        const float* __restrict__ loss_part = nullptr, int n_loss = 0,
                                                        float loss_scale = 0.f) {
    constexpr int BLK = 32768;
    __shared__ uint32_t buf[BLK];
    __shared__ uint32_t wsum[16];
    __shared__ float lsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (loss_part) {
        float s = 0.f;
        for (int i = tid; i < n_loss; i += 1024) s += loss_part[i];
        s = wave_sum(s);
        if (lane == 0) lsum[wv] = s;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < 16; ++w) t += lsum[w];               // fixed order
            scalars[0] = t * loss_scale;
            scalars[1] = t * loss_scale;
        }
    }
    uint32_t carry = 0;
    for (int b0 = 0; b0 < d_sae; b0 += BLK) {
        const int nb = min(BLK, d_sae - b0);
        __syncthreads();                                 // (the previous block's stores out of buf, its reads of wsum)
        for (int i = tid; i < nb; i += 1024) buf[i] = cnt[b0 + i];
        __syncthreads();
        const int per = (nb + 1023) / 1024;
        const int lo = min(tid * per, nb), hi = min(lo + per, nb);
        uint32_t s = 0;
        for (int i = lo; i < hi; ++i) s += buf[i];
        uint32_t inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= o) inc += a;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t base = carry;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        uint32_t total = 0;
        for (int w = 0; w < 16; ++w) total += wsum[w];
        uint32_t run = base + inc - s;                  // exclusive prefix of this thread's run
        for (int i = lo; i < hi; ++i) {
            const uint32_t c = buf[i];
            buf[i] = run;
            run += c;
        }
        __syncthreads();
        for (int i = tid; i < nb; i += 1024) offs[b0 + i] = buf[i];
        carry += total;
    }
    if (tid == 1023) {
        offs[d_sae] = carry;
        if (scalars) scalars[2] = (float)carry * inv_tokens;            // l0 = mean_n #(val > 0), train_sae.py:364
    }
    if (tid == 0) { n_long[0] = 0u; n_long[1] = 0u; n_long[2] = 0u; }   // counters of csr_post_kernel; the ticket of colsum_final_sq_kernel
}

// Per feature (one thread each), after the scan:
//   * firing statistics (train_sae.py:356-361);
//   * the chunk cuts of the short-list backward: the grid point g = w * BWD_CH that falls inside this feature's list
//     [beg, end) becomes chunk_start[w] = end (beg itself when g == beg) -- every grid point below the total lies in
//     exactly one list, so the scatter is disjoint; the cuts at and beyond the total are filled by the whole grid;
//   * long lists (> BWD_LMAX pairs): registered in long_list {feature, first segment, #segments} with their
//     BWD_SEG-pair segments in seg_range (two device counters, zeroed by the scan kernel).
struct CsrPostArgs {
    const uint32_t* offs; uint32_t* chunk_start; int max_chunks; int32_t* long_list; uint32_t* n_long; uint32_t* seg_range; int max_segs;
    float* act_freq; float* n_since_fired; float* fire_count; int d_sae; int update_stats; float* gb_enc_sparse; float* rowsq_sparse;
    int ranged; const uint32_t* gate;
};
// bid / nblocks: this workgroup's index among the nblocks that run the post pass (a launch of its own, or a block range of
// csr_post_fill_kernel)
__device__ __forceinline__ void csr_post_body(int bid, int nblocks, const uint32_t* __restrict__ offs, uint32_t* __restrict__ chunk_start,
                                                       int max_chunks, int32_t* __restrict__ long_list, uint32_t* __restrict__ n_long,
                                                       uint32_t* __restrict__ seg_range, int max_segs, float* __restrict__ act_freq,
                                                       float* __restrict__ n_since_fired, float* __restrict__ fire_count,
                                                       int d_sae, int update_stats, float* __restrict__ gb_enc_sparse,
                                                       float* __restrict__ rowsq_sparse, int ranged,
                                                       const uint32_t* __restrict__ gate) {
    if (gate && *gate != 0u) update_stats = 0;                    // (pv_sae_relu_step in dense mode: the dense path owns the statistics)
    const int j = bid * 256 + threadIdx.x;
    {   // the cuts at and beyond the total (all threads of the pass: a feature shard of the feature-parallel step keeps a
        // fraction of the N k pairs the cut array is sized for -- 7 of 8 cuts lie beyond the total at world 8)
        const uint32_t total = offs[d_sae];
        for (uint32_t w = (total + BWD_CH - 1) / BWD_CH + (uint32_t)j; w <= (uint32_t)max_chunks; w += (uint32_t)nblocks * 256u) chunk_start[w] = total;
    }
    if (j >= d_sae) return;
    const uint32_t beg = offs[j], end = offs[j + 1], c = end - beg;
    const float cnt = (float)c;
    if (fire_count) fire_count[j] = cnt;
    if (gb_enc_sparse && c == 0) {                    // PV_SAE_SPARSE_GRADS: the rows stay as they are, the scalars are zeroed here
        gb_enc_sparse[j] = 0.f;
        rowsq_sparse[j] = 0.f;
    }
    if (update_stats) {
        act_freq[j] += cnt;
        n_since_fired[j] = cnt > 0.f ? 0.f : n_since_fired[j] + 1.f;
    }
    for (uint32_t w = (beg + BWD_CH - 1) / BWD_CH; w * BWD_CH < end; ++w) chunk_start[w] = w * BWD_CH == beg ? beg : end;
    if (c > (uint32_t)BWD_LMAX) {
        const uint32_t e = atomicAdd(&n_long[0], 1u);
        long_list[3 * e] = j;
        if (ranged) return;                                       // (sae_long_sort_kernel cuts the list at the token ranges)
        const uint32_t nseg = (c + BWD_SEG - 1) / BWD_SEG;
        const uint32_t sb = atomicAdd(&n_long[1], nseg);
        long_list[3 * e + 1] = (int32_t)sb;
        long_list[3 * e + 2] = (int32_t)nseg;
        for (uint32_t sg = 0; sg < nseg && sb + sg < (uint32_t)max_segs; ++sg) {
            seg_range[2 * (sb + sg)] = beg + sg * BWD_SEG;
            seg_range[2 * (sb + sg) + 1] = min(beg + (sg + 1) * BWD_SEG, end);
        }
    }
}

__global__ __launch_bounds__(256) void csr_post_kernel(const uint32_t* __restrict__ offs, uint32_t* __restrict__ chunk_start,
                                                       int max_chunks, int32_t* __restrict__ long_list, uint32_t* __restrict__ n_long,
                                                       uint32_t* __restrict__ seg_range, int max_segs, float* __restrict__ act_freq,
                                                       float* __restrict__ n_since_fired, float* __restrict__ fire_count,
                                                       int d_sae, int update_stats, float* __restrict__ gb_enc_sparse,
                                                       float* __restrict__ rowsq_sparse, int ranged,
                                                       const uint32_t* __restrict__ gate = nullptr) {
    csr_post_body(blockIdx.x, gridDim.x, offs, chunk_start, max_chunks, long_list, n_long, seg_range, max_segs, act_freq, n_since_fired,
                  fire_count, d_sae, update_stats, gb_enc_sparse, rowsq_sparse, ranged, gate);
}

__device__ __forceinline__ void csr_fill_body(int bid, const int32_t* __restrict__ idx, const uint32_t* __restrict__ wpos,
                                              const uint32_t* __restrict__ offs, int32_t* __restrict__ pairs, int n_pairs) {
    const int p = bid * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t w = wpos[p];
    if (w != 0xffffffffu) pairs[offs[idx[p]] + w] = p;
}
__global__ __launch_bounds__(256) void csr_fill_kernel(const int32_t* __restrict__ idx, const uint32_t* __restrict__ wpos,
                                                       const uint32_t* __restrict__ offs, int32_t* __restrict__ pairs, int n_pairs) {
    csr_fill_body(blockIdx.x, idx, wpos, offs, pairs, n_pairs);
}

// The three passes behind the scan that depend on nothing but it and the decode kernel, as ONE launch (they were three): blocks
// [0, nb_post) = csr_post, [nb_post, nb_post + nb_fill) = csr_fill, the rest (cs_x != NULL) = the 16-row partial column sums of dY
// that the bias gradients start from.
// nb_cs (the column-sum workgroups; 0: none) + one more workgroup when loss_part != NULL: the step's loss (loss_reduce_body: the scan ran as a
// role of the decode launch and could not take it)
__global__ __launch_bounds__(256) void csr_post_fill_kernel(const CsrPostArgs a, int nb_post, const int32_t* __restrict__ idx,
                                                            const uint32_t* __restrict__ wpos, int32_t* __restrict__ pairs, int n_pairs,
                                                            int nb_fill, const float* __restrict__ cs_x, float* __restrict__ cs_partial,
                                                            int cs_rows, int cs_d, int nb_cs, const float* __restrict__ loss_part = nullptr,
                                                            int n_loss = 0, float loss_scale = 0.f, float* __restrict__ scalars = nullptr) {
    const int b = blockIdx.x;
    if (b < nb_post)
        csr_post_body(b, nb_post, a.offs, a.chunk_start, a.max_chunks, a.long_list, a.n_long, a.seg_range, a.max_segs, a.act_freq,
                      a.n_since_fired, a.fire_count, a.d_sae, a.update_stats, a.gb_enc_sparse, a.rowsq_sparse, a.ranged, a.gate);
    else if (b < nb_post + nb_fill)
        csr_fill_body(b - nb_post, idx, wpos, a.offs, pairs, n_pairs);
    else if (b < nb_post + nb_fill + nb_cs)
        colsum_partial_body(b - nb_post - nb_fill, cs_x, cs_partial, cs_rows, cs_d);
    else
        loss_reduce_body(loss_part, n_loss, loss_scale, scalars);
}

// The position of a pair inside its feature's list was drawn by an integer atomic in the selection kernel: the SET of a list is
// exact, its ORDER is whatever the atomics produced -- and the backward sums in list order, so gradients would agree from run to run
// only up to fp32 summation order.  Short lists (<= BWD_LMAX = 64 pairs: one wave holds a whole list) are put into ascending pair
// order here (a pair id is token * k + slot and a feature holds a token at most once: token order) with a 64-lane bitonic network;
// the long lists get the same from sae_long_sort_kernel.  With both, every gradient is bit-reproducible.
__device__ __forceinline__ void csr_sort_short_body(int bid, const uint32_t* __restrict__ offs, int32_t* __restrict__ pairs, int d_sae) {
    const int lane = threadIdx.x & 63;
    const int f = bid * 4 + (threadIdx.x >> 6);
    if (f >= d_sae) return;
    const uint32_t o = offs[f];
    const int c = (int)(offs[f + 1] - o);
    if (c < 2 || c > BWD_LMAX) return;                        // (wave-uniform)
    int v = lane < c ? pairs[o + lane] : 0x7fffffff;
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const int other = __shfl_xor(v, j, 64);
            const bool up = (lane & k2) == 0, lower = (lane & j) == 0;
            v = (lower == up) ? min(v, other) : max(v, other);
        }
    if (lane < c) pairs[o + lane] = v;
}
__global__ __launch_bounds__(256) void csr_sort_short_kernel(const uint32_t* __restrict__ offs, int32_t* __restrict__ pairs, int d_sae) {
    csr_sort_short_body(blockIdx.x, offs, pairs, d_sae);
}

// ------------------------------------------------------------------------------------------------
// sparse backward:  gW_dec[j, :] = sum_p a_p dY[n_p, :],  gW_enc^T[j, :] = sum_p g_p sae_in[n_p, :],  gb_enc[j] = sum_p g_p
// over the pairs p = (token n_p, feature j) of feature j's list (a = kept activation, g = dh).  One-wave-per-feature
// collapses on skewed data (8.6 ms / step measured), so the work is split by PAIRS:
//   short lists (<= BWD_LMAX pairs)  sae_backward_kernel: wave w owns the chunk [chunk_start[w], chunk_start[w+1]) of whole
//                                    lists (~BWD_CH pairs), accumulates per feature and stores each finished row once
//   long lists                       cut into BWD_SEG-pair segments: sae_backward_seg_kernel (a wave per segment -> partial
//                                    rows in scratch), then sae_backward_long_kernel (a wave per dense feature sums its
//                                    segments in order, one store).  A feature that fires on all 4096 tokens is 128
//                                    independent waves, not one long chain
// Rows of features that did not fire are zeroed by the caller (this IS their zero_grad).  No atomics.
// Shared inner loop: the pair metadata of a run is fetched up front, one pair per lane, and broadcast with readlane, so the
// only memory operations inside the loop are the row gathers (16 bytes per lane), issued two pairs ahead of their use
// (vmcnt retires loads in order: nothing younger may sit between a gather and its use).
// ------------------------------------------------------------------------------------------------
template <int V4>
struct BwdAcc {
    float4 gd[V4], ge[V4];
    float gb;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < V4; ++i) { gd[i] = make_float4(0.f, 0.f, 0.f, 0.f); ge[i] = gd[i]; }
        gb = 0.f;
    }
    // the finished rows of feature j -> gW_dec[j], gW_encT[j], gb_enc[j]; rowsq[j] = their sum of squares (this feature's
    // term of the clip norm, so that the norm does not have to re-read the 151 MB it was just written to)
    __device__ __forceinline__ void store(int j, float* __restrict__ gW_dec, float* __restrict__ gW_encT, float* __restrict__ gb_enc,
                                          float* __restrict__ rowsq, int d, int lane, const int (&col)[V4], const bool (&ok)[V4]) const {
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < V4; ++i)
            if (ok[i]) {
                *reinterpret_cast<float4*>(gW_dec + (int64_t)j * d + col[i]) = gd[i];
                *reinterpret_cast<float4*>(gW_encT + (int64_t)j * d + col[i]) = ge[i];
                sq += gd[i].x * gd[i].x + gd[i].y * gd[i].y + gd[i].z * gd[i].z + gd[i].w * gd[i].w;
                sq += ge[i].x * ge[i].x + ge[i].y * ge[i].y + ge[i].z * ge[i].z + ge[i].w * ge[i].w;
            }
        sq = wave_sum(sq);
        if (lane == 0) {
            gb_enc[j] = gb;
            rowsq[j] = sq + gb * gb;
        }
    }
};

// accumulate the pairs [q0, q1) (all of ONE feature when STOP_AT_LONG is false).  With STOP_AT_LONG the run is a chunk of
// whole lists: a finished feature's rows are stored, and the walk ends at the first long list (the last list of a chunk).
// DUAL (the gated step, sae_gated_sparse): a pair carries a SECOND decoder term -- gW_dec[j] = sum_p a_p dY[n_p] + b_p dYb[n_p] (b =
// val_b: relu(gate_pre), dYb: the gradient of the reconstruction through the gate) -- gathered and accumulated alongside the first.
template <int V4, bool STOP_AT_LONG, bool DUAL = false>
__device__ __forceinline__ void bwd_walk(BwdAcc<V4>& acc, uint32_t q0, uint32_t q1, const uint32_t* __restrict__ offs,
                                         const int32_t* __restrict__ pairs, const int32_t* __restrict__ idx,
                                         const float* __restrict__ val, const float* __restrict__ dh, const float* __restrict__ dY,
                                         const float* __restrict__ sae_in, float* __restrict__ gW_dec, float* __restrict__ gW_encT,
                                         float* __restrict__ gb_enc, float* __restrict__ rowsq, int d, int k, int lane,
                                         const int (&col)[V4], const bool (&ok)[V4], const float* __restrict__ val_b = nullptr,
                                         const float* __restrict__ dYb = nullptr) {
    int cur = -1;
    auto store_rows = [&](int j) {
        acc.store(j, gW_dec, gW_encT, gb_enc, rowsq, d, lane, col, ok);
        acc.clear();
    };
    bool stop = false;
    for (uint32_t base = q0; base < q1 && !stop; base += 64) {
        const int cnt = (int)min(64u, q1 - base);
        int my_n = 0, my_j = 0;
        float my_a = 0.f, my_g = 0.f, my_b = 0.f;
        if (lane < cnt) {
            const int32_t p = pairs[base + lane];
            my_n = p / k;
            my_j = idx[p];
            my_a = val[p];
            my_g = dh[p];
            if constexpr (DUAL) my_b = val_b[p];
        }
        constexpr int VB = DUAL ? V4 : 1;
        float4 dy0[V4], si0[V4], dy1[V4], si1[V4], db0[VB], db1[VB];
        auto gather = [&](float4 (&dy)[V4], float4 (&si)[V4], float4 (&db)[VB], int t) {
            const int n = __shfl(my_n, t, 64);
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                dy[i] = ld4(dY + (int64_t)n * d + col[i], ok[i]);
                si[i] = ld4(sae_in + (int64_t)n * d + col[i], ok[i]);
                if constexpr (DUAL) db[i] = ld4(dYb + (int64_t)n * d + col[i], ok[i]);
            }
        };
        auto accumulate = [&](const float4 (&dy)[V4], const float4 (&si)[V4], const float4 (&db)[VB], int t) {
            const float a = __shfl(my_a, t, 64), g = __shfl(my_g, t, 64);
            const float b = DUAL ? __shfl(my_b, t, 64) : 0.f;
            if constexpr (STOP_AT_LONG) {
                const int j = __shfl(my_j, t, 64);
                if (j != cur) {
                    if (cur >= 0) store_rows(cur);
                    cur = j;
                    stop = offs[j + 1] - offs[j] > (uint32_t)BWD_LMAX;       // (uniform) the long-list kernel owns it
                }
                if (stop) return;
            }
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                acc.gd[i].x += a * dy[i].x; acc.gd[i].y += a * dy[i].y; acc.gd[i].z += a * dy[i].z; acc.gd[i].w += a * dy[i].w;
                acc.ge[i].x += g * si[i].x; acc.ge[i].y += g * si[i].y; acc.ge[i].z += g * si[i].z; acc.ge[i].w += g * si[i].w;
                if constexpr (DUAL) {
                    acc.gd[i].x += b * db[i].x; acc.gd[i].y += b * db[i].y; acc.gd[i].z += b * db[i].z; acc.gd[i].w += b * db[i].w;
                }
            }
            acc.gb += g;
        };
        gather(dy0, si0, db0, 0);
        if (cnt > 1) gather(dy1, si1, db1, 1);
        for (int t = 0; t < cnt && !stop; t += 2) {
            accumulate(dy0, si0, db0, t);
            if (t + 2 < cnt) gather(dy0, si0, db0, t + 2);
            if (t + 1 < cnt && !stop) {
                accumulate(dy1, si1, db1, t + 1);
                if (t + 3 < cnt) gather(dy1, si1, db1, t + 3);
            }
        }
    }
    if constexpr (STOP_AT_LONG) {
        if (!stop && cur >= 0) store_rows(cur);
    }
}

template <int V4, bool DUAL = false>
__global__ __launch_bounds__(256) void sae_backward_kernel(
    const uint32_t* __restrict__ offs, const uint32_t* __restrict__ chunk_start, const int32_t* __restrict__ pairs,
    const int32_t* __restrict__ idx, const float* __restrict__ val, const float* __restrict__ dh, const float* __restrict__ dY,
    const float* __restrict__ sae_in, float* __restrict__ gW_dec, float* __restrict__ gW_encT, float* __restrict__ gb_enc,
    float* __restrict__ rowsq, int d, int k, int n_chunks, const uint32_t* __restrict__ gate = nullptr,
    const float* __restrict__ val_b = nullptr, const float* __restrict__ dYb = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= n_chunks) return;
    if (gate && *gate != 0u) return;
    const uint32_t q0 = chunk_start[wv], q1 = chunk_start[wv + 1];
    if (q0 >= q1) return;
    bool ok[V4];
    int col[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        col[i] = 4 * lane + 256 * i;
        ok[i] = col[i] < d;
    }
    BwdAcc<V4> acc;
    acc.clear();
    bwd_walk<V4, true, DUAL>(acc, q0, q1, offs, pairs, idx, val, dh, dY, sae_in, gW_dec, gW_encT, gb_enc, rowsq, d, k, lane, col, ok, val_b,
                             dYb);
}

// long lists, stage 0 (token-range form): sort every long list by token and cut it at the BWD_RANGES token-range boundaries.
// The 86 us / 608 MB of the count-cut segments were row gathers out of a 25 MB working set (dY + sae_in of 4096 tokens) that no
// XCD's 4 MB L2 holds: with the lists in token order, segment (feature e, range r) only touches the N / 8 tokens of range r, and
// the segment kernel gives range r to the workgroups of XCD r (3 MB of rows per L2).  A feature's pairs are DISTINCT tokens,
// so the sort is a scatter into a token-indexed LDS array + a compaction -- which also makes the summation order of these
// lists independent of the order the select kernel's atomics drew their positions in.  One workgroup per long feature.
// bid / nblocks: this workgroup's index among the nblocks that sort long lists (a launch of its own, or the leading block range of csr_sort_kernel)
__device__ __forceinline__ void sae_long_sort_body(int bid, int nblocks, int32_t* __restrict__ slot, int32_t* __restrict__ long_list,
                                                   uint32_t* __restrict__ n_long, const uint32_t* __restrict__ offs,
                                                   int32_t* __restrict__ pairs, uint32_t* __restrict__ seg_range, int k, int n_tok,
                                                   int max_segs) {
    __shared__ uint32_t wsum[4], bnd[BWD_RANGES + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t nl = n_long[0];
    if (bid == 0 && tid == 0) n_long[1] = min(nl * BWD_RANGES, (uint32_t)max_segs);
    const int chunk = (n_tok + 255) / 256;
    const int rs = (n_tok + BWD_RANGES - 1) / BWD_RANGES;
    for (uint32_t e = bid; e < nl; e += nblocks) {
        const int j = long_list[3 * e];
        const uint32_t beg = offs[j], c = offs[j + 1] - beg;
        __syncthreads();
        for (int i = tid; i < n_tok; i += 256) slot[i] = -1;
        if (tid <= BWD_RANGES) bnd[tid] = tid == 0 ? 0u : c;
        __syncthreads();
        for (uint32_t i = tid; i < c; i += 256) {
            const int32_t p = pairs[beg + i];
            slot[p / k] = p;
        }
        __syncthreads();
        const int t0 = tid * chunk, t1 = min(t0 + chunk, n_tok);
        uint32_t cnt = 0;
        for (int t = t0; t < t1; ++t) cnt += slot[t] >= 0;
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= o) inc += a;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t base = inc - cnt;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        uint32_t pos = base;
        for (int t = t0; t < t1; ++t) {
            if (t > 0 && t % rs == 0) bnd[t / rs] = pos;       // (first token of range t / rs: everything before it)
            const int32_t p = slot[t];
            if (p >= 0) pairs[beg + pos++] = p;
        }
        __syncthreads();
        if (tid < BWD_RANGES) {
            const uint32_t sg = e * BWD_RANGES + tid;
            if (sg < (uint32_t)max_segs) {
                seg_range[2 * sg] = beg + bnd[tid];
                seg_range[2 * sg + 1] = beg + bnd[tid + 1];
            }
        }
        if (tid == 0) {
            long_list[3 * e + 1] = (int32_t)(e * BWD_RANGES);
            long_list[3 * e + 2] = BWD_RANGES;
        }
    }
}
__global__ __launch_bounds__(256) void sae_long_sort_kernel(int32_t* __restrict__ long_list, uint32_t* __restrict__ n_long,
                                                            const uint32_t* __restrict__ offs, int32_t* __restrict__ pairs,
                                                            uint32_t* __restrict__ seg_range, int k, int n_tok, int max_segs) {
    extern __shared__ int32_t slot[];                      // [n_tok] token -> pair (or -1)
    sae_long_sort_body(blockIdx.x, gridDim.x, slot, long_list, n_long, offs, pairs, seg_range, k, n_tok, max_segs);
}
// Both list sorts as ONE launch (they touch disjoint lists): workgroups [0, nb_long) = the long lists (first: they are the longer jobs),
// the rest = the short lists, four features per workgroup.  The launch carries the long sort's LDS (4 bytes per token).
__global__ __launch_bounds__(256) void csr_sort_kernel(int nb_long, int32_t* __restrict__ long_list, uint32_t* __restrict__ n_long,
                                                       const uint32_t* __restrict__ offs, int32_t* __restrict__ pairs,
                                                       uint32_t* __restrict__ seg_range, int k, int n_tok, int max_segs, int d_sae) {
    extern __shared__ int32_t slot[];
    if ((int)blockIdx.x < nb_long) sae_long_sort_body(blockIdx.x, nb_long, slot, long_list, n_long, offs, pairs, seg_range, k, n_tok, max_segs);
    else csr_sort_short_body(blockIdx.x - nb_long, offs, pairs, d_sae);
}

// long lists, stage 1: one wave per segment -> partial rows in scratch  [segment][gd | ge][d] (+ gb).  Segments are BWD_SEG-pair
// cuts (ranged == 0) or the BWD_RANGES token ranges of a sorted list (ranged == 1: segment 8 e + r goes to a workgroup with
// blockIdx % 8 == r, i.e. to XCD r)
template <int V4, bool DUAL = false>
__global__ __launch_bounds__(512) void sae_backward_seg_kernel(
    const uint32_t* __restrict__ offs, const uint32_t* __restrict__ seg_range, const uint32_t* __restrict__ n_long,
    const int32_t* __restrict__ pairs, const int32_t* __restrict__ idx, const float* __restrict__ val,
    const float* __restrict__ dh, const float* __restrict__ dY, const float* __restrict__ sae_in, float* __restrict__ seg_rows,
    float* __restrict__ seg_b, int d, int k, int max_segs, int ranged, const uint32_t* __restrict__ gate = nullptr,
    const float* __restrict__ val_b = nullptr, const float* __restrict__ dYb = nullptr) {
    if (gate && *gate != 0u) return;                               // (uniform over the grid: no barrier is skipped by a part of a workgroup)
    constexpr int NW = 8;                                          // waves per workgroup (token-range form: 512 threads)
    __shared__ __attribute__((aligned(16))) float part[NW * 2 * 256 * V4];      // [wave][gd | ge][256 V4] (token-range form)
    __shared__ float part_b[NW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nseg = min(n_long[1], (uint32_t)max_segs);
    bool ok[V4];
    int col[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        col[i] = 4 * lane + 256 * i;
        ok[i] = col[i] < d;
    }
    if (!ranged) {
        const uint32_t nw = blockDim.x >> 6;
        for (uint32_t sg = blockIdx.x * nw + wv; sg < nseg; sg += gridDim.x * nw) {
            BwdAcc<V4> acc;
            acc.clear();
            bwd_walk<V4, false, DUAL>(acc, seg_range[2 * sg], seg_range[2 * sg + 1], offs, pairs, idx, val, dh, dY, sae_in, nullptr, nullptr,
                                      nullptr, nullptr, d, k, lane, col, ok, val_b, dYb);
            float* o = seg_rows + (int64_t)sg * 2 * d;
#pragma unroll
            for (int i = 0; i < V4; ++i)
                if (ok[i]) {
                    *reinterpret_cast<float4*>(o + col[i]) = acc.gd[i];
                    *reinterpret_cast<float4*>(o + d + col[i]) = acc.ge[i];
                }
            if (lane == 0) seg_b[sg] = acc.gb;
        }
        return;
    }
    // token-range form: the WORKGROUP owns segment (long feature e, range r = blockIdx % 8 = its XCD); its eight waves take an
    // eighth of the segment's (token-sorted) pairs each -- a dense feature has N / 8 pairs per range, one wave alone would be
    // the kernel's tail -- and their partial rows are summed in wave order through LDS
    constexpr int DP = 256 * V4;                                  // padded row length (floats)
    float* pw = part;
    for (uint32_t e = blockIdx.x >> 3; e * BWD_RANGES < nseg; e += gridDim.x >> 3) {
        const uint32_t sg = e * BWD_RANGES + (blockIdx.x & 7);
        const uint32_t q0 = seg_range[2 * sg], q1 = seg_range[2 * sg + 1];
        const uint32_t per = (q1 - q0 + NW - 1) / NW;
        const uint32_t w0 = min(q0 + wv * per, q1), w1 = min(w0 + per, q1);
        BwdAcc<V4> acc;
        acc.clear();
        bwd_walk<V4, false, DUAL>(acc, w0, w1, offs, pairs, idx, val, dh, dY, sae_in, nullptr, nullptr, nullptr, nullptr, d, k, lane, col, ok,
                                  val_b, dYb);
        __syncthreads();                                          // (the previous segment's reads of part)
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            *reinterpret_cast<float4*>(pw + (wv * 2 + 0) * DP + col[i]) = acc.gd[i];
            *reinterpret_cast<float4*>(pw + (wv * 2 + 1) * DP + col[i]) = acc.ge[i];
        }
        if (lane == 0) part_b[wv] = acc.gb;
        __syncthreads();
        // 2 x d floats out: thread t sums column group t (gd for t < 64 V4 ... ) in wave order
        float* o = seg_rows + (int64_t)sg * 2 * d;
        for (int c4 = threadIdx.x; c4 < 2 * 64 * V4; c4 += NW * 64) {
            const int which = c4 / (64 * V4), cc = (c4 - which * 64 * V4) * 4;
            if (cc < d) {
                float4 t = *reinterpret_cast<const float4*>(pw + (0 * 2 + which) * DP + cc);
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const float4 u = *reinterpret_cast<const float4*>(pw + (w * 2 + which) * DP + cc);
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                *reinterpret_cast<float4*>(o + which * d + cc) = t;
            }
        }
        if (threadIdx.x == 0) {
            float t = part_b[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) t += part_b[w];
            seg_b[sg] = t;
        }
    }
}

// long lists, stage 2: one wave per dense feature sums its segments' partial rows in segment order and stores
template <int V4>
__global__ __launch_bounds__(256) void sae_backward_long_kernel(
    const int32_t* __restrict__ long_list, const uint32_t* __restrict__ n_long, const float* __restrict__ seg_rows,
    const float* __restrict__ seg_b, float* __restrict__ gW_dec, float* __restrict__ gW_encT, float* __restrict__ gb_enc,
    float* __restrict__ rowsq, int d, int max_segs, const uint32_t* __restrict__ gate = nullptr) {
    if (gate && *gate != 0u) return;
    const int lane = threadIdx.x & 63;
    const uint32_t nl = n_long[0];
    bool ok[V4];
    int col[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        col[i] = 4 * lane + 256 * i;
        ok[i] = col[i] < d;
    }
    for (uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6); f < nl; f += gridDim.x * 4) {
        const int j = long_list[3 * f];
        const uint32_t sb = (uint32_t)long_list[3 * f + 1], ns = (uint32_t)long_list[3 * f + 2];
        BwdAcc<V4> acc;
        acc.clear();
        const uint32_t s_end = min(sb + ns, (uint32_t)max_segs);
        for (uint32_t sg = sb; sg < s_end; sg += 4) {                 // four partial rows in flight, summed in segment order
            float4 a[4][V4], b[4][V4];
            float gb4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool live = sg + u < s_end;
                const float* o = seg_rows + (int64_t)(live ? sg + u : sg) * 2 * d;
#pragma unroll
                for (int i = 0; i < V4; ++i) {
                    a[u][i] = ld4(o + col[i], ok[i] && live);
                    b[u][i] = ld4(o + d + col[i], ok[i] && live);
                }
                gb4[u] = live ? seg_b[sg + u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < V4; ++i) {
                    acc.gd[i].x += a[u][i].x; acc.gd[i].y += a[u][i].y; acc.gd[i].z += a[u][i].z; acc.gd[i].w += a[u][i].w;
                    acc.ge[i].x += b[u][i].x; acc.ge[i].y += b[u][i].y; acc.ge[i].z += b[u][i].z; acc.ge[i].w += b[u][i].w;
                }
                acc.gb += gb4[u];
            }
        }
        acc.store(j, gW_dec, gW_encT, gb_enc, rowsq, d, lane, col, ok);
    }
}

// The features no token kept: their gradient rows are zero (this IS their zero_grad).  One wave per feature; the others
// leave at once (their rows are stored, exactly once, by the backward kernels).
template <int V4>
__global__ __launch_bounds__(256) void sae_zero_empty_kernel(const uint32_t* __restrict__ offs, float* __restrict__ gW_dec,
                                                             float* __restrict__ gW_encT, float* __restrict__ gb_enc,
                                                             float* __restrict__ rowsq, int d_sae, int d,
                                                             const uint32_t* __restrict__ gate = nullptr) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gate && *gate != 0u) return;
    if (j >= d_sae || offs[j + 1] != offs[j]) return;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < d) {
            *reinterpret_cast<float4*>(gW_dec + (int64_t)j * d + c) = z;
            *reinterpret_cast<float4*>(gW_encT + (int64_t)j * d + c) = z;
        }
    }
    if (lane == 0) {
        gb_enc[j] = 0.f;
        rowsq[j] = 0.f;
    }
}

// clip norm from the per-feature terms of the backward (+ gb_dec): one workgroup, fixed summation order
__global__ __launch_bounds__(1024) void sqnorm_rowsq_kernel(const float* __restrict__ rowsq, int d_sae, const float* __restrict__ gb_dec,
                                                            int d_in, float* __restrict__ scalars, const float* __restrict__ extra0,
                                                            int n0, const float* __restrict__ extra1, int n1) {
    __shared__ float red[16];
    float s = 0.f;
#pragma unroll 8
    for (int j = threadIdx.x; j < d_sae; j += 1024) s += rowsq[j];             // (8 loads in flight; the sum in order)
    for (int i = threadIdx.x; i < d_in; i += 1024) s += gb_dec[i] * gb_dec[i];
    for (int i = threadIdx.x; i < n0; i += 1024) s += extra0[i] * extra0[i];          // transcoder: gb_dec_out, gW_skip
    for (int i = threadIdx.x; i < n1; i += 1024) s += extra1[i] * extra1[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        scalars[3] = t;
    }
}

// The step's last launch with the clip norm in it (PV_SAE_FUSED_SQNORM).  Workgroup b of the nb that finish gb_dec (64 columns each)
// also takes block b of the clip norm: the squares of its own 64 columns + the b-th of nb slices of the backward's per-feature terms
// (rowsq) -> sqpart[b]; the workgroup that finishes LAST (a ticket from an agent-scope acq_rel counter; the partials travel as
// agent-scope atomics, so no cache of another XCD can hold a stale one) adds the nb partials in block order -> scalars[3] and re-arms
// the counter.  The sum is a fixed function of the inputs: which workgroup comes last does not enter it.  sq_block_terms is that
// block's arithmetic, shared with the one-workgroup form (sqnorm_blocks_kernel: the same bits from a launch of its own).
__device__ __forceinline__ float sq_block_terms(int b, int nb, float g, const float* __restrict__ rowsq, int d_sae, float* red16) {
    // g: this thread's column of block b (threads of wave 0; 0 elsewhere)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slice = (d_sae + nb - 1) / nb, lo = b * slice, hi = min(lo + slice, d_sae);
    float sr = 0.f;
    for (int j = lo + (int)threadIdx.x; j < hi; j += 1024) sr += rowsq[j];
    sr = wave_sum(sr);
    const float gg = wave_sum(g * g);                        // (only wave 0's matters)
    __syncthreads();
    if (lane == 0) red16[wv] = sr;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
        for (int w = 0; w < 16; ++w) t += red16[w];          // fixed order
        t += gg;
    }
    return t;                                                // (thread 0)
}
__global__ __launch_bounds__(1024) void colsum_final_sq_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int d,
                                                               const float* __restrict__ rowsq, int d_sae, float* __restrict__ sqpart,
                                                               uint32_t* __restrict__ ticket, float* __restrict__ scalars) {
    __shared__ float red16[16];
    const int b = blockIdx.x, nb = gridDim.x;
    const float g = colsum_final_body(b, partial, out, nblk, d, 1.0f);
    const float t = sq_block_terms(b, nb, g, rowsq, d_sae, red16);
    __shared__ uint32_t last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&sqpart[b], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t mine = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = mine == (uint32_t)nb - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
        // lane i fetches partial i (one memory round trip for all of them: read one after the other by one thread they were most of
        // this launch's time), lane 0 adds them in block order (nb <= 64: d_in <= 4096)
        const int lane = threadIdx.x;
        const float mine = lane < nb ? __hip_atomic_load(&sqpart[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        float total = 0.f;
        for (int i = 0; i < nb; ++i) total += __shfl(mine, i, 64);
        if (lane == 0) {
            scalars[3] = total;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// the same sum from a launch of its own (gb_dec finished by the launch before): one workgroup walks the blocks
__global__ __launch_bounds__(1024) void sqnorm_blocks_kernel(const float* __restrict__ gb_dec, int d, const float* __restrict__ rowsq,
                                                             int d_sae, float* __restrict__ scalars) {
    __shared__ float red16[16];
    const int nb = (d + 63) / 64;
    float total = 0.f;
    for (int b = 0; b < nb; ++b) {
        const int c = b * 64 + (int)threadIdx.x;
        const float g = (threadIdx.x < 64 && c < d) ? gb_dec[c] : 0.f;
        total += sq_block_terms(b, nb, g, rowsq, d_sae, red16);
    }
    if (threadIdx.x == 0) scalars[3] = total;
}

// gb_dec = colsum(dY) - W_enc gb_enc (the encoder-input path of b_dec: sae_in = x - b_dec).  The second term as partial rows
// for colsum_final_kernel, stacked under the dY partials with the sign folded in: workgroup c sums -gb_enc[j] * W_encT[j][:]
// over its GBD_ROWS features, skipping the features no token kept (gb_enc = 0: half of them on the bench batch), so the
// pass reads the fired rows of W_encT once, coalesced, instead of all of W_enc.
constexpr int GBD_ROWS = 96;
__device__ __forceinline__ void gbdec_partial_body(int bid, const float* __restrict__ W_encT, const float* __restrict__ gb_enc,
                                                   float* __restrict__ partial, int d_sae, int d) {
    __shared__ float red[3][1280];                       // waves 1..3 -> wave 0, up to 1280 columns (d_in <= 256 * 5)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j0 = bid * GBD_ROWS, j1 = min(j0 + GBD_ROWS, d_sae);
    float4 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // (four of the wave's rows in flight: one row per trip was a chain of 24 dependent row fetches, the launch's whole 13 us; the rows
    // are still added in row order, a row no token kept is still neither read nor added)
    for (int jb = j0 + wv; jb < j1; jb += 16) {
        float g[4];
        float4 w[4][5];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = jb + 4 * u;
            g[u] = j < j1 ? gb_enc[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = jb + 4 * u;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int c = 4 * lane + 256 * i;
                w[u][i] = (g[u] != 0.f && c < d) ? *reinterpret_cast<const float4*>(W_encT + (int64_t)j * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (g[u] == 0.f) continue;                   // (wave-uniform)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (4 * lane + 256 * i < d) {
                    acc[i].x -= g[u] * w[u][i].x; acc[i].y -= g[u] * w[u][i].y; acc[i].z -= g[u] * w[u][i].z; acc[i].w -= g[u] * w[u][i].w;
                }
            }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) *reinterpret_cast<float4*>(&red[wv - 1][4 * lane + 256 * i]) = acc[i];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = 4 * lane + 256 * i;
            if (c < d) {
                float4 t = acc[i];
                for (int w = 0; w < 3; ++w) {            // fixed order
                    const float4 o = *reinterpret_cast<const float4*>(&red[w][c]);
                    t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
                }
                *reinterpret_cast<float4*>(partial + (int64_t)bid * d + c) = t;
            }
        }
    }
}
__global__ __launch_bounds__(256) void sae_gbdec_partial_kernel(const float* __restrict__ W_encT, const float* __restrict__ gb_enc,
                                                                float* __restrict__ partial, int d_sae, int d) {
    gbdec_partial_body(blockIdx.x, W_encT, gb_enc, partial, d_sae, d);
}

// ------------------------------------------------------------------------------------------------
// gradient square-norm (two stage, deterministic)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            for (int64_t t = i; t < n; ++t) s += g[t] * g[t];
        }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------
// apply: clip -> project -> Adam  (train_sae.py:394-401; sae.py:279-297; torch.optim.Adam defaults)
// ------------------------------------------------------------------------------------------------
struct AdamC {
    float lr, b1, b2, eps, bc1, bc2_sqrt, max_norm;
};
__device__ __forceinline__ float clip_coef(const float* scalars, float max_norm) {
    if (max_norm <= 0.f) return 1.f;
    const float total = sqrtf(scalars[3]);
    return fminf(max_norm / (total + 1e-6f), 1.f);          // clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
}
__device__ __forceinline__ float adam_update(float w, float g, float& m, float& v, const AdamC& c) {
    m = m + (g - m) * (1.f - c.b1);                         // exp_avg.lerp_(grad, 1 - beta1)
    v = v * c.b2 + (1.f - c.b2) * g * g;                    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    return w - (c.lr / c.bc1) * (m / denom);
}

// two vectors in one launch (b_enc's feature range and b_dec): blocks [0, nb0) take the first
struct AdamVec2 {
    float* W0; const float* G0; float* M0; float* V0; int lo0, hi0, nb0;
    float* W1; const float* G1; float* M1; float* V1; int hi1;
};
__device__ __forceinline__ void adam_vec2_body(int bid, const AdamVec2& a, const float* __restrict__ scalars, const AdamC& c) {
    const bool first = bid < a.nb0;
    const int i = first ? a.lo0 + bid * 256 + threadIdx.x : (bid - a.nb0) * 256 + threadIdx.x;
    if (i >= (first ? a.hi0 : a.hi1)) return;
    float* W = first ? a.W0 : a.W1;
    const float* G = first ? a.G0 : a.G1;
    float* M = first ? a.M0 : a.M1;
    float* V = first ? a.V0 : a.V1;
    const float coef = clip_coef(scalars, c.max_norm);
    float m = M[i], v = V[i];
    W[i] = adam_update(W[i], G[i] * coef, m, v, c);
    M[i] = m;
    V[i] = v;
}
__global__ __launch_bounds__(256) void adam_vec2_kernel(const AdamVec2 a, const float* __restrict__ scalars, AdamC c) {
    adam_vec2_body(blockIdx.x, a, scalars, c);
}

template <int V4>   // W_dec rows [j_lo, j_hi): one wave per row (16 bytes per lane and load), with the parallel-gradient projection
__global__ __launch_bounds__(256) void adam_wdec_kernel(float* __restrict__ W, const float* __restrict__ G,
                                                        float* __restrict__ M, float* __restrict__ V,
                                                        const float* __restrict__ scalars, AdamC c, int j_lo, int j_hi, int d,
                                                        const float* inv_norm, float* inv_next /* may be the same array */,
                                                        const uint32_t* __restrict__ live_offs) {
    const int lane = threadIdx.x & 63;
    const int j = j_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= j_hi) return;
    const bool live = !live_offs || live_offs[j + 1] != live_offs[j];        // (wave-uniform) PV_SAE_SPARSE_GRADS: no pair, g = 0
    const float coef = clip_coef(scalars, c.max_norm);
    const float rn = inv_norm ? inv_norm[j] : 1.f;            // pending set_decoder_norm_to_unit_norm (see pv_sae_step)
    float4 w[V4], g[V4], m[V4], v[V4];
    bool ok[V4];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const int col = 4 * lane + 256 * i;
        ok[i] = col < d;
        const int64_t o = (int64_t)j * d + col;
        w[i] = ld4(W + o, ok[i]);
        g[i] = ld4_stream(G + o, ok[i] && live);
        m[i] = ld4_stream(M + o, ok[i]);
        v[i] = ld4_stream(V + o, ok[i]);
    }
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        w[i].x *= rn; w[i].y *= rn; w[i].z *= rn; w[i].w *= rn;
        g[i].x *= coef; g[i].y *= coef; g[i].z *= coef; g[i].w *= coef;
        dot += g[i].x * w[i].x + g[i].y * w[i].y + g[i].z * w[i].z + g[i].w * w[i].w;
    }
    dot = wave_sum(dot);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        if (!ok[i]) continue;
        const int64_t o = (int64_t)j * d + 4 * lane + 256 * i;
        float4 wn;                                          // g - dot * w: remove_gradient_parallel_to_decoder_directions
        wn.x = adam_update(w[i].x, g[i].x - dot * w[i].x, m[i].x, v[i].x, c);
        wn.y = adam_update(w[i].y, g[i].y - dot * w[i].y, m[i].y, v[i].y, c);
        wn.z = adam_update(w[i].z, g[i].z - dot * w[i].z, m[i].z, v[i].z, c);
        wn.w = adam_update(w[i].w, g[i].w - dot * w[i].w, m[i].w, v[i].w, c);
        *reinterpret_cast<float4*>(W + o) = wn;
        st4_stream(M + o, m[i]);
        st4_stream(V + o, v[i]);
        sq += wn.x * wn.x + wn.y * wn.y + wn.z * wn.z + wn.w * wn.w;
    }
    // the next step's set_decoder_norm_to_unit_norm needs 1 / ||row|| of what was just written: leave it behind
    // (PV_SAE_INV_NORM_VALID) instead of re-reading W_dec for it
    if (inv_next) {
        sq = wave_sum(sq);
        if (lane == 0) inv_next[j] = 1.0f / sqrtf(sq);
    }
}

// W_enc's Adam step when the parameter's own [d_in][d_sae] layout is NOT rewritten (lazy materialisation, pv_sae_state.W_enc ==
// NULL): everything lives in the transposed domain -- the fp32 master WT, gradient, moments, the fp16 shadow and the column
// norms are all [d_sae][d_in] rows -- so this is the row-streaming form of adam_wdec_kernel (a wave per feature row, 16 bytes
// per lane and load, no LDS transposes) instead of the 32 x 64 tile walk of wenc_rows_kernel.  Same arithmetic per element.
template <int V4>
__global__ __launch_bounds__(256) void adam_wenct_kernel(float* __restrict__ WT, _Float16* __restrict__ W16T, float* __restrict__ colsq,
                                                         const float* __restrict__ GT, float* __restrict__ MT, float* __restrict__ VT,
                                                         const float* __restrict__ scalars, AdamC c, int j_lo, int j_hi, int d,
                                                         const uint32_t* __restrict__ live_offs, int nb_rows, const AdamVec2 v2) {
    // workgroups beyond nb_rows (v2.nb0 + the blocks of b_dec of them): the two bias vectors' Adam (adam_vec2_body) -- it was a launch
    if ((int)blockIdx.x >= nb_rows) {
        adam_vec2_body(blockIdx.x - nb_rows, v2, scalars, c);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int j = j_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= j_hi) return;
    const bool live = !live_offs || live_offs[j + 1] != live_offs[j];        // (wave-uniform) PV_SAE_SPARSE_GRADS: no pair, g = 0
    const float coef = clip_coef(scalars, c.max_norm);
    float4 w[V4], g[V4], m[V4], v[V4];
    bool ok[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const int col = 4 * lane + 256 * i;
        ok[i] = col < d;
        const int64_t o = (int64_t)j * d + col;
        w[i] = ld4(WT + o, ok[i]);
        g[i] = ld4_stream(GT + o, ok[i] && live);
        m[i] = ld4_stream(MT + o, ok[i]);
        v[i] = ld4_stream(VT + o, ok[i]);
    }
    float sq = 0.f;
    bool big = false;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        if (!ok[i]) continue;
        const int64_t o = (int64_t)j * d + 4 * lane + 256 * i;
        float4 wn;
        wn.x = adam_update(w[i].x, g[i].x * coef, m[i].x, v[i].x, c);
        wn.y = adam_update(w[i].y, g[i].y * coef, m[i].y, v[i].y, c);
        wn.z = adam_update(w[i].z, g[i].z * coef, m[i].z, v[i].z, c);
        wn.w = adam_update(w[i].w, g[i].w * coef, m[i].w, v[i].w, c);
        *reinterpret_cast<float4*>(WT + o) = wn;
        st4_stream(MT + o, m[i]);
        st4_stream(VT + o, v[i]);
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 hv = {(_Float16)wn.x, (_Float16)wn.y, (_Float16)wn.z, (_Float16)wn.w};
        *reinterpret_cast<h4*>(W16T + o) = hv;
        sq += wn.x * wn.x + wn.y * wn.y + wn.z * wn.z + wn.w * wn.w;
        big = big || !(fabsf(wn.x) <= 6.0e4f) || !(fabsf(wn.y) <= 6.0e4f) || !(fabsf(wn.z) <= 6.0e4f) || !(fabsf(wn.w) <= 6.0e4f);
    }
    sq = wave_sum(sq);
    const bool any_big = __any(big);                        // outside the fp16 range (or NaN): the filter must not be trusted
    if (lane == 0) colsq[j] = any_big ? INFINITY : sq;
}

// ------------------------------------------------------------------------------------------------
// W_enc lives three times: the module's parameter W [d_in][d_sae]; WT = its transpose [d_sae][d_in] in fp32 -- the
// layout the sparse backward writes gradients in and the exact re-scoring of sae_enc.hip gathers rows from, hence
// the layout Adam runs in (gradient, both moments and WT are read and written coalesced, feature rows [j_lo, j_hi) are
// contiguous: the data-parallel trainer shards the optimizer by feature); and W16T = WT rounded to fp16, the B operand of
// the filter GEMM.  One workgroup owns 32 features x all d_in: tile-wise it updates WT / moments / W16T in place,
// transposes the new values through LDS into W, and leaves ||W_enc[:, j]||^2 (the filter's error bound) in colsq.
// MODE 0: Adam step.  MODE 1: rebuild W, W16T, colsq from WT (after an all-gather of WT).  MODE 2: rebuild WT, W16T,
// colsq from W (parameters edited from outside).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void wenc_rows_kernel(float* __restrict__ W, float* __restrict__ WT, _Float16* __restrict__ W16T,
                                                        float* __restrict__ colsq, const float* __restrict__ GT,
                                                        float* __restrict__ MT, float* __restrict__ VT,
                                                        const float* __restrict__ scalars, AdamC c, int d_in, int d_sae,
                                                        int j_lo, int j_hi, const uint32_t* __restrict__ live_offs = nullptr) {
    constexpr int NC = 2;                                     // 32-column chunks per iteration (16 independent loads per array)
    __shared__ float tile[NC][32][33];
    const int j0 = j_lo + blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float coef = 1.f;
    if constexpr (MODE == 0) coef = clip_coef(scalars, c.max_norm);
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
    bool big = false;
    bool live[4] = {true, true, true, true};                  // PV_SAE_SPARSE_GRADS: rows of features without a pair are not read
    if constexpr (MODE == 0) {
        if (live_offs) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + ty + 8 * r;
                live[r] = j < j_hi && live_offs[j + 1] != live_offs[j];
            }
        }
    }
    for (int i0 = 0; i0 < d_in; i0 += 32 * NC) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int u = 0; u < NC; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {                   // read W[i][j0 + tx] coalesced along j
                    const int i = i0 + 32 * u + ty + 8 * r, j = j0 + tx;
                    tile[u][tx][ty + 8 * r] = (i < d_in && j < j_hi) ? W[(int64_t)i * d_sae + j] : 0.f;
                }
            __syncthreads();
        }
        float wv[NC][4], gv[NC][4], mv[NC][4], vv[NC][4];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int u = 0; u < NC; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + ty + 8 * r, i = i0 + 32 * u + tx;
                    const bool in = j < j_hi && i < d_in;
                    const int64_t o = (int64_t)j * d_in + i;
                    wv[u][r] = in ? WT[o] : 0.f;
                    gv[u][r] = (in && live[r]) ? ld1_stream(GT + o) : 0.f;      // (optimizer state and gradients: streamed, see ld4_stream)
                    mv[u][r] = in ? ld1_stream(MT + o) : 0.f;
                    vv[u][r] = in ? ld1_stream(VT + o) : 0.f;
                }
        }
#pragma unroll
        for (int u = 0; u < NC; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + ty + 8 * r, i = i0 + 32 * u + tx;
                if (j < j_hi && i < d_in) {
                    const int64_t o = (int64_t)j * d_in + i;
                    float wn;
                    if constexpr (MODE == 0) {
                        float m = mv[u][r], v = vv[u][r];
                        wn = adam_update(wv[u][r], gv[u][r] * coef, m, v, c);
                        st1_stream(MT + o, m);
                        st1_stream(VT + o, v);
                        WT[o] = wn;
                    } else if constexpr (MODE == 1) {
                        wn = WT[o];
                    } else {
                        wn = tile[u][ty + 8 * r][tx];
                        WT[o] = wn;
                    }
                    W16T[o] = (_Float16)wn;
                    sq[r] += wn * wn;
                    big = big || !(fabsf(wn) <= 6.0e4f);        // outside the fp16 range (or NaN): the filter must not be trusted
                    if constexpr (MODE != 2) { if (W) tile[u][ty + 8 * r][tx] = wn; }
                }
            }
        if constexpr (MODE != 2) {
            if (W) {                                                   // (uniform) W == nullptr: the parameter layout is materialised lazily
                __syncthreads();
#pragma unroll
                for (int u = 0; u < NC; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = i0 + 32 * u + ty + 8 * r, j = j0 + tx;
                        if (i < d_in && j < j_hi) W[(int64_t)i * d_sae + j] = tile[u][tx][ty + 8 * r];
                    }
                __syncthreads();
            }
        } else {
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = big ? INFINITY : sq[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);           // the 32 lanes that share ty
        const int j = j0 + ty + 8 * r;
        if (tx == 0 && j < j_hi) colsq[j] = s;
    }
}

__global__ __launch_bounds__(256) void adam_vec_kernel(float* __restrict__ W, const float* __restrict__ G,
                                                       float* __restrict__ M, float* __restrict__ V,
                                                       const float* __restrict__ scalars, AdamC c, int lo, int hi) {
    const int i = lo + blockIdx.x * 256 + threadIdx.x;
    if (i >= hi) return;
    const float coef = clip_coef(scalars, c.max_norm);
    float m = M[i], v = V[i];
    W[i] = adam_update(W[i], G[i] * coef, m, v, c);
    M[i] = m;
    V[i] = v;
}

// 1 / ||W_dec[j]|| (the read-only half of set_decoder_norm_to_unit_norm): a wave takes four rows, all loads in flight
__global__ __launch_bounds__(256) void dec_inv_norm_kernel(const float* __restrict__ W, float* __restrict__ inv, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (j0 >= rows) return;
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 4 * lane; c < d; c += 256) {
        float4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = ld4(W + (int64_t)min(j0 + u, rows - 1) * d + c, true);
#pragma unroll
        for (int u = 0; u < 4; ++u) sq[u] += w[u].x * w[u].x + w[u].y * w[u].y + w[u].z * w[u].z + w[u].w * w[u].w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < 4; ++u) sq[u] += __shfl_xor(sq[u], o, 64);
    if (lane < 4 && j0 + lane < rows) {
        const float s = lane == 0 ? sq[0] : (lane == 1 ? sq[1] : (lane == 2 ? sq[2] : sq[3]));
        inv[j0 + lane] = 1.0f / sqrtf(s);
    }
}

// set_decoder_norm_to_unit_norm (sae.py:275-277)
template <int DPL>
__global__ __launch_bounds__(256) void renorm_rows_kernel(float* __restrict__ W, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= rows) return;
    float w[DPL];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        const int c = lane + 64 * i;
        w[i] = c < d ? W[(int64_t)j * d + c] : 0.f;
        sq += w[i] * w[i];
    }
    const float nrm = sqrtf(wave_sum(sq));
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        const int c = lane + 64 * i;
        if (c < d) W[(int64_t)j * d + c] = w[i] / nrm;
    }
}

int launch_long_sort(int32_t* long_list, uint32_t* n_long, const uint32_t* offs, int32_t* pairs, uint32_t* seg_range, int k, int N,
                     int max_segs, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sae_long_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         150 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(sae_long_sort_kernel, dim3(512), dim3(256), (size_t)N * 4, stream, long_list, n_long, offs, pairs, seg_range, k, N,
                       max_segs);
    PV_LAUNCH_CHECK("sae_long_sort_kernel");
    return PV_OK;
}
int launch_csr_sort(int32_t* long_list, uint32_t* n_long, const uint32_t* offs, int32_t* pairs, uint32_t* seg_range, int k, int N,
                    int max_segs, int d_sae, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&csr_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         150 * 1024));
        attr_done = true;
    }
    const int nb_long = 512;
    hipLaunchKernelGGL(csr_sort_kernel, dim3(nb_long + (d_sae + 3) / 4), dim3(256), (size_t)N * 4, stream, nb_long, long_list, n_long, offs,
                       pairs, seg_range, k, N, max_segs, d_sae);
    PV_LAUNCH_CHECK("csr_sort_kernel");
    return PV_OK;
}

}  // namespace

void sae_topk_rows(const float* hidden, int32_t* idx_out, float* val_out, int d_sae, int k, int n_rows, const int32_t* row_list,
                   const uint32_t* n_list, int slots, uint32_t* feat_cnt, uint32_t* wpos, hipStream_t stream) {
    hipLaunchKernelGGL(sae_topk_kernel, dim3(row_list ? slots : n_rows), dim3(256), 0, stream, hidden, idx_out, val_out, d_sae, k,
                       row_list, n_list, feat_cnt, wpos);
}

// ------------------------------------------------------------------------------------------------
// plan + C ABI
// ------------------------------------------------------------------------------------------------
SaeWs sae_carve(const pv_sae_desc& d) {
    SaeWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (size_t)pv_align_up((int64_t)bytes, 256); return o; };
    const size_t N = d.max_tokens;
    w.sq_blocks = 1024;
    w.hidden = take(N * (size_t)d.d_sae * 4);        // exact path: hidden_pre; filtered path: rows of undecided tokens only
    w.sae_in = take(N * (size_t)d.d_in * 4);
    w.dY = take(N * (size_t)d.d_in * 4);
    w.mu = take(N * 4);
    w.sd = take(N * 4);
    w.norm = take(N * 4);
    w.dh = take(N * (size_t)d.k * 4);
    w.loss_part = take(N * 4);
    w.cnt = take((size_t)d.d_sae * 4);
    w.offs = take(((size_t)d.d_sae + 1) * 4);
    w.cursor = take((N * (size_t)d.k / BWD_CH + 8) * 4);      // chunk starts of the backward's waves
    w.wpos = take(N * (size_t)d.k * 4);
    w.long_list = take((size_t)d.d_sae * 12);
    w.n_long = take(256);
    {
        const size_t max_segs = sae_max_segs(N * (size_t)d.k);
        w.seg_range = take(max_segs * 8);
        w.seg_rows = take(max_segs * 2 * (size_t)d.d_in * 4);
        w.seg_b = take(max_segs * 4);
    }
    w.pairs = take(N * (size_t)d.k * 4);
    w.colpart = take((size_t)((d.max_tokens + CS_ROWS - 1) / CS_ROWS + (d.d_sae + GBD_ROWS - 1) / GBD_ROWS) * d.d_in * 4);
    w.colsum = take((size_t)d.d_in * 4);
    w.batch_mean = take((size_t)d.d_in * 4);
    w.sqpart = take((size_t)w.sq_blocks * 4);
    w.rowsq = take((size_t)d.d_sae * 4);              // per-feature terms of the clip norm (backward kernels)
    w.x16 = take(N * (size_t)d.d_in * 2);
    w.xnorm = take(N * 4);
    w.sample = take(N * (size_t)(d.d_sae / PV_SAE_SAMPLE_STRIDE + 1) * 4);
    w.thr = take(N * 4);
    w.sq = take(N * 4);
    w.band = take(N * 4);
    const size_t ntn = (size_t)(d.d_sae + 255) / 256;
    w.cand_cnt = take(N * ntn * 4);
    w.cand = take(N * ntn * (size_t)pv_sae_tile_slots(d) * 8);
    w.fb_list = take(N * 4);
    w.fb_count = take(256);
    w.wmax = take(256);
    {   // dense (ReLU + L1) step, sae_dense.hip
        const size_t rblk = (N + 63) / 64, cblk = ((size_t)d.d_sae + 63) / 64;
        w.dense_colpart = take(rblk * (size_t)d.d_sae * 4);
        w.dense_rowpart = take(rblk * cblk * 4);
        w.dense_kpart = take((size_t)PV_SAE_DENSE_SPLITK * N * (size_t)d.d_in * 4);
        w.dense_amax = take((size_t)PV_SAE_AMAX_TENSORS * 256 * 4);
        const bool lp = d.lp_norm != 0.f && d.lp_norm != 1.f;
        w.dense_lp_part = take(lp ? N * cblk * 4 : 0);
        w.dense_lp_tok = take(lp ? N * 4 : 0);
        w.dense_lp_loss = take(lp ? N * 4 : 0);
    }
    w.total = off + 256;
    return w;
}

// The buffers of the sparse ReLU + L1 step that depend on its per-token capacity (pv_sae_relu_step; caller-owned workspace)
ReluWs relu_carve(const pv_sae_desc& d, int n_tokens, int cap) {
    ReluWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (size_t)pv_align_up((int64_t)bytes, 256); return o; };
    const size_t N = (size_t)n_tokens, P = N * (size_t)cap, ntn = (size_t)(d.d_sae + 255) / 256;
    w.mode = take(256);
    w.idx = take(P * 4);
    w.val = take(P * 4);
    w.tok_cnt = take(N * 4);
    w.l1part = take(N * 4);
    w.cand_cnt = take(N * ntn * 4);
    w.cand = take(N * ntn * (size_t)PV_SAE_RELU_SLOTS * 8);
    w.dh = take(P * 4);
    w.cursor = take((P / BWD_CH + 8) * 4);
    w.wpos = take(P * 4);
    w.max_segs = (int)sae_max_segs(P);
    w.seg_range = take((size_t)w.max_segs * 8);
    w.seg_rows = take((size_t)w.max_segs * 2 * (size_t)d.d_in * 4);
    w.seg_b = take((size_t)w.max_segs * 4);
    w.pairs = take(P * 4);
    w.total = off + 256;
    return w;
}

extern "C" int pv_sae_plan_create(const pv_sae_desc* desc, pv_sae_plan** out_plan) {
    PV_REQUIRE(desc && out_plan, "null argument");
    PV_REQUIRE(desc->d_in > 1 && desc->d_in <= 64 * 20 && desc->d_in % 4 == 0, "d_in must be a multiple of 4, <= 1280 (ViT-H/14)");
    PV_REQUIRE(desc->d_sae >= desc->k && desc->d_sae % 4 == 0 && desc->d_sae <= 256 * 256, "d_sae must be a multiple of 4, <= 65536");
    // (k <= 64: the filtered encoder; beyond it, up to 256, the exact fp32 encoder + the streaming top-k serve the plan -- pv_sae_fast_ok)
    PV_REQUIRE(desc->k >= 1 && desc->k <= 256, "k must be in [1, 256]");
    PV_REQUIRE(desc->max_tokens >= 1, "max_tokens");
    PV_REQUIRE(desc->normalize_layer_norm >= 0 && desc->normalize_layer_norm <= 2, "normalize_layer_norm: 0 none, 1 layer_norm, 2 constant_norm_rescale");
    PV_REQUIRE(desc->activation == PV_SAE_ACT_RELU || desc->activation == PV_SAE_ACT_TANH_RELU, "activation: PV_SAE_ACT_RELU or PV_SAE_ACT_TANH_RELU");
    PV_REQUIRE(desc->lp_norm == 0.f || desc->lp_norm >= 1.f, "lp_norm: 0 / 1 (the 1-norm) or p > 1 (p < 1 has no finite gradient at zero activations)");
    pv_sae_plan* p = new pv_sae_plan();
    p->d = *desc;
    *out_plan = p;
    return PV_OK;
}
extern "C" void pv_sae_plan_destroy(pv_sae_plan* plan) { delete plan; }
extern "C" size_t pv_sae_workspace_bytes(const pv_sae_plan* plan) { return plan ? sae_carve(plan->d).total : 0; }
extern "C" int pv_sae_encoder_is_filtered(const pv_sae_plan* plan) { return plan && pv_sae_fast_ok(plan->d) ? 1 : 0; }
// debug / tests: byte offset of a named region of the workspace ("fb_count": uint32 number of tokens of the last encode that
// took the exact fallback; "cand_cnt": uint32 [N] candidates per token), or (size_t)-1
extern "C" size_t pv_debug_sae_ws_offset(const pv_sae_plan* plan, const char* name) {
    if (!plan || !name) return (size_t)-1;
    const SaeWs w = sae_carve(plan->d);
    if (!strcmp(name, "fb_count")) return w.fb_count;
    if (!strcmp(name, "fb_list")) return w.fb_list;
    if (!strcmp(name, "cand_cnt")) return w.cand_cnt;
    if (!strcmp(name, "thr")) return w.thr;
    if (!strcmp(name, "hidden")) return w.hidden;
    return (size_t)-1;
}

#define DPL_DISPATCH(d_in, CALL)            \
    do {                                    \
        if ((d_in) <= 64 * 4) { CALL(4); }  \
        else if ((d_in) <= 64 * 12) { CALL(12); } \
        else if ((d_in) <= 64 * 16) { CALL(16); } \
        else { CALL(20); }                  \
    } while (0)
// kernels whose lanes own 16-byte column groups: d_in <= 256 * V4
#define V4_DISPATCH(d_in, CALL)             \
    do {                                    \
        if ((d_in) <= 256) { CALL(1); }     \
        else if ((d_in) <= 768) { CALL(3); } \
        else if ((d_in) <= 1024) { CALL(4); } \
        else { CALL(5); }                   \
    } while (0)

extern "C" int pv_sae_renorm_decoder(pv_sae_plan* plan, pv_sae_state* st, void* stream_) {
    PV_REQUIRE(plan && st && st->W_dec, "null argument");
    hipStream_t stream = (hipStream_t)stream_;
    const pv_sae_desc& d = plan->d;
    const dim3 grid((d.d_sae + 3) / 4), block(256);
#define CALL(D) hipLaunchKernelGGL((renorm_rows_kernel<D>), grid, block, 0, stream, st->W_dec, d.d_sae, d.d_in)
    DPL_DISPATCH(d.d_in, CALL);
#undef CALL
    PV_LAUNCH_CHECK("renorm_rows_kernel");
    plan->renorm_pending = false;
    return PV_OK;
}

// Rebuild the encoder shadows for features [j_lo, j_hi).  from_transposed = 0: W_enc is the truth (parameters were edited
// outside: load_state_dict, a fresh engine) -> W_encT, W_enc16T, enc_colsq; 1: W_encT is the truth (an all-gather of the
// sharded optimizer's slices just landed) -> W_enc, W_enc16T, enc_colsq.
extern "C" int pv_sae_sync_shadows(pv_sae_plan* plan, pv_sae_state* st, int32_t from_transposed, int32_t j_lo, int32_t j_hi,
                                   void* stream_) {
    PV_REQUIRE(plan && st && st->W_enc && st->W_encT && st->W_enc16T && st->enc_colsq, "null argument");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(j_lo >= 0 && j_lo <= j_hi && j_hi <= d.d_sae, "feature range");
    if (j_lo == j_hi) return PV_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((j_hi - j_lo + 31) / 32), block(256);
    AdamC c = {};
    if (from_transposed)
        hipLaunchKernelGGL((wenc_rows_kernel<1>), grid, block, 0, stream, st->W_enc, st->W_encT, (_Float16*)st->W_enc16T, st->enc_colsq,
                           (const float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, c, d.d_in, d.d_sae, j_lo, j_hi);
    else
        hipLaunchKernelGGL((wenc_rows_kernel<2>), grid, block, 0, stream, st->W_enc, st->W_encT, (_Float16*)st->W_enc16T, st->enc_colsq,
                           (const float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, c, d.d_in, d.d_sae, j_lo, j_hi);
    PV_LAUNCH_CHECK("wenc_rows_kernel");
    return PV_OK;
}

// ---- transcoder (pv_sae_state.tc; sae/transcoder.py:6-116) ----------------------------------------------------------------
// loss normaliser of the target: ||y_n - mean_n(y)||_2 (sae.py:145-147 as called by transcoder.py:78).  One wave per token.
// d_true < d (a transcoder whose output is narrower than its input, pv_sae_transcoder.d_out_true: rows padded to the common width d):
// the norm runs over the real columns, and the padding of the target row is set to the token's LN mean -- exactly what the decoder
// (zero padded columns of W_dec / b_dec_out) reconstructs there after LN-out, so that the padding's error, loss and gradient are exact
// zeros
__global__ __launch_bounds__(256) void sae_target_norm_kernel(float* __restrict__ y, const float* __restrict__ batch_mean,
                                                              float* __restrict__ norm_out, int n_tok, int d, int d_true,
                                                              const float* __restrict__ mu) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    float cn = 0.f;
    for (int i = lane; i < d_true; i += 64) {
        const float c = y[(int64_t)n * d + i] - batch_mean[i];
        cn += c * c;
    }
    cn = wave_sum(cn);
    if (lane == 0) norm_out[n] = sqrtf(cn);
    for (int i = d_true + lane; i < d; i += 64) y[(int64_t)n * d + i] = mu[n];
}

int sae_tc_require(const pv_sae_desc& d, const pv_sae_state* st, int N) {
    const pv_sae_transcoder& t = st->tc;
    PV_REQUIRE(t.b_dec_out && t.gb_dec_out && t.target, "transcoder state: b_dec_out, gb_dec_out and target are required");
    const bool any_skip = t.W_skip || t.gW_skip;
    PV_REQUIRE(!any_skip || (t.W_skip && t.gW_skip), "transcoder state: W_skip and gW_skip come together");
    PV_REQUIRE(t.d_in_true >= 0 && t.d_in_true <= d.d_in && t.d_out_true >= 0 && t.d_out_true <= d.d_in &&
                   (t.d_in_true == 0 || t.d_in_true == d.d_in || t.d_out_true == 0 || t.d_out_true == d.d_in),
               "transcoder widths: d_in_true / d_out_true are 0 (= the plan's d_in) or the real widths of rows padded to max(d_in, d_out)");
    PV_REQUIRE(!t.W_skip || ((t.d_in_true == 0 || t.d_in_true == d.d_in) && (t.d_out_true == 0 || t.d_out_true == d.d_in)),
               "the skip connection needs d_out == d_in (transcoder.py:10, 73-76)");
    if (t.W_skip) {
        PV_REQUIRE(d.d_in % 8 == 0, "the skip connection's GEMMs need d_in to be a multiple of 8");
        PV_REQUIRE(t.scratch && ((uintptr_t)t.scratch & 255) == 0 &&
                       t.scratch_bytes >= ((size_t)N * d.d_in + (size_t)PV_SAE_SKIP_SPLITK * d.d_in * d.d_in) * 4,
                   "transcoder scratch too small / misaligned (pv_sae_transcoder_scratch_bytes)");
    }
    return PV_OK;
}

extern "C" size_t pv_sae_transcoder_scratch_bytes(const pv_sae_plan* plan, int32_t n_tokens) {
    if (!plan || n_tokens < 1) return 0;
    const pv_sae_desc& d = plan->d;
    return (size_t)pv_align_up((int64_t)(((size_t)n_tokens * d.d_in + (size_t)PV_SAE_SKIP_SPLITK * d.d_in * d.d_in) * 4), 256);
}

int sae_tc_target_norm(const pv_sae_desc& d, const pv_sae_state* st, const float* batch_mean, int N, unsigned char* wsb, const SaeWs& ws,
                       hipStream_t stream) {
    float* bmean = (float*)(wsb + ws.batch_mean);
    const float* y = st->tc.target;
    if (batch_mean) {
        PV_HIP_CHECK(hipMemcpyAsync(bmean, batch_mean, (size_t)d.d_in * 4, hipMemcpyDeviceToDevice, stream));
    } else {
        const int nblk = (N + CS_ROWS - 1) / CS_ROWS;
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, stream, y, (float*)(wsb + ws.colpart), N, d.d_in);
        hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream,
                           (const float*)(wsb + ws.colpart), bmean, nblk, d.d_in, 1.0f / (float)N);
    }
    hipLaunchKernelGGL(sae_target_norm_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, const_cast<float*>(y), (const float*)bmean,
                       (float*)(wsb + ws.norm), N, d.d_in, sae_loss_width(d, st), (const float*)(wsb + ws.mu));
    PV_LAUNCH_CHECK("sae_target_norm_kernel");
    return PV_OK;
}

// ---- pieces of the step shared with the dense (ReLU + L1) step of sae_dense.hip --------------------------------------------
// batch mean (given, or computed from x) -> ws.batch_mean; LN-in, sae_in, loss normaliser (+ the fp16 copy / row norms the
// filtered encoder wants) -> ws.sae_in, ws.mu, ws.sd, ws.norm (ws.x16, ws.xnorm)
int sae_prep(const pv_sae_desc& d, const float* x, const float* b_dec, const float* batch_mean, int N, bool want_filter_inputs,
             unsigned char* wsb, const SaeWs& ws, hipStream_t stream, int d_true, SaePre* pre, const pv_sae_state* st, uint32_t* feat_cnt) {
    float* bmean = (float*)(wsb + ws.batch_mean);
    const int nblk = (N + CS_ROWS - 1) / CS_ROWS;
    if (batch_mean) PV_HIP_CHECK(hipMemcpyAsync(bmean, batch_mean, (size_t)d.d_in * 4, hipMemcpyDeviceToDevice, stream));
    if (pre) {                                                   // the fused pre-pass (SaePre): one launch, the mean and the normaliser come later
        pre->x = x; pre->d_true = d_true > 0 ? d_true : d.d_in; pre->have_mean = batch_mean != nullptr;
        const int nb_prep = (N + 3) / 4, nb_cs = batch_mean ? 0 : nblk;
        hipLaunchKernelGGL(sae_prep_roles_kernel, dim3(nb_prep + nb_cs + 1), dim3(256), 0, stream, x, b_dec, (float*)(wsb + ws.sae_in),
                           (_Float16*)(wsb + ws.x16), (float*)(wsb + ws.xnorm), (float*)(wsb + ws.mu), (float*)(wsb + ws.sd), N, d.d_in,
                           d.normalize_layer_norm, d.ln_eps, pre->d_true, nb_prep, (float*)(wsb + ws.colpart), nb_cs,
                           (const float*)st->enc_colsq, d.d_sae, (float*)(wsb + ws.wmax), (uint32_t*)(wsb + ws.fb_count), feat_cnt);
        PV_LAUNCH_CHECK("sae_prep_roles_kernel");
        return PV_OK;
    }
    if (!batch_mean) {
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, stream, x, (float*)(wsb + ws.colpart), N, d.d_in);
        hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream,
                           (const float*)(wsb + ws.colpart), bmean, nblk, d.d_in, 1.0f / (float)N);
    }
    hipLaunchKernelGGL(sae_prep_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, x, b_dec, (const float*)bmean,
                       (float*)(wsb + ws.sae_in), want_filter_inputs ? (_Float16*)(wsb + ws.x16) : (_Float16*)nullptr,
                       want_filter_inputs ? (float*)(wsb + ws.xnorm) : (float*)nullptr, (float*)(wsb + ws.mu), (float*)(wsb + ws.sd),
                       (float*)(wsb + ws.norm), N, d.d_in, d.normalize_layer_norm, d.ln_eps, d_true > 0 ? d_true : d.d_in);
    PV_LAUNCH_CHECK("sae_prep_kernel");
    return PV_OK;
}

// gb_dec = colsum(dY) - W_enc @ gb_enc (the encoder-input path of b_dec): both terms as partial rows of one column sum
// have_colsum: the partial column sums of dY are in ws.colpart already (csr_post_fill_kernel)
int sae_gbdec(const pv_sae_desc& d, const pv_sae_state* st, const float* dY, int N, unsigned char* wsb, const SaeWs& ws,
              hipStream_t stream, bool have_colsum, float* sq_scalars) {
    const int nblk = (N + CS_ROWS - 1) / CS_ROWS, ngb = (d.d_sae + GBD_ROWS - 1) / GBD_ROWS;
    float* colpart = (float*)(wsb + ws.colpart);
    if (!have_colsum) hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, stream, dY, colpart, N, d.d_in);
    hipLaunchKernelGGL(sae_gbdec_partial_kernel, dim3(ngb), dim3(256), 0, stream, (const float*)st->W_encT, (const float*)st->gb_enc,
                       colpart + (size_t)nblk * d.d_in, d.d_sae, d.d_in);
    if (sq_scalars)                 // PV_SAE_FUSED_SQNORM: the clip norm out of the same launch (the ticket word: n_long[2], zeroed by the scan)
        hipLaunchKernelGGL(colsum_final_sq_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream, (const float*)colpart, st->gb_dec,
                           nblk + ngb, d.d_in, (const float*)(wsb + ws.rowsq), d.d_sae, (float*)(wsb + ws.sqpart),
                           (uint32_t*)(wsb + ws.n_long) + 2, sq_scalars);
    else
        hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream, (const float*)colpart, st->gb_dec,
                           nblk + ngb, d.d_in, 1.0f);
    PV_LAUNCH_CHECK("sae bias-grad kernels");
    return PV_OK;
}

int sae_tc_bias_grads(const pv_sae_desc& d, const pv_sae_state* st, const float* dY, int N, unsigned char* wsb, const SaeWs& ws,
                      hipStream_t stream) {
    const int nblk = (N + CS_ROWS - 1) / CS_ROWS, ngb = (d.d_sae + GBD_ROWS - 1) / GBD_ROWS;
    float* colpart = (float*)(wsb + ws.colpart);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, stream, dY, colpart, N, d.d_in);
    hipLaunchKernelGGL(sae_gbdec_partial_kernel, dim3(ngb), dim3(256), 0, stream, (const float*)st->W_encT, (const float*)st->gb_enc,
                       colpart + (size_t)nblk * d.d_in, d.d_sae, d.d_in);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream, (const float*)colpart, st->tc.gb_dec_out,
                       nblk, d.d_in, 1.0f);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream,
                       (const float*)(colpart + (size_t)nblk * d.d_in), st->gb_dec, ngb, d.d_in, 1.0f);
    PV_LAUNCH_CHECK("transcoder bias-grad kernels");
    return PV_OK;
}

// out[c] = scale * sum_rows x[r][c] for a [rows][d] matrix; `partial` = (rows / 16 + 1) * d floats of scratch
int sae_colsum(const float* x, int rows, int d, float* out, float scale, float* partial, hipStream_t stream) {
    const int nblk = (rows + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, stream, x, partial, rows, d);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((d + 63) / 64), dim3(1024), 0, stream, (const float*)partial, out, nblk, d, scale);
    PV_LAUNCH_CHECK("colsum kernels");
    return PV_OK;
}

// out[slot] (and out[slot2] when >= 0) = scale * sum(v[0..n)), one workgroup, fixed order
void sae_reduce_sum(const float* v, float* out, int n, float scale, int slot, int slot2, hipStream_t stream, const uint32_t* gate,
                    uint32_t want) {
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, stream, v, out, n, scale, slot, slot2, gate, want);
}

// want_csr: also count the kept pairs per feature (ws.cnt) and record their positions (ws.wpos) for the backward
// cnt_over / wpos_over: other destinations for those two (the top-k gated step encodes twice); skip_prep: sae_prep has already run on x
static int sae_encode_topk(pv_sae_plan* plan, const pv_sae_state* st, const float* x, int N, const float* batch_mean,
                           int32_t* topk_idx, float* topk_val, bool want_csr, unsigned char* wsb, const SaeWs& ws, hipStream_t stream,
                           uint32_t* cnt_over = nullptr, uint32_t* wpos_over = nullptr, bool skip_prep = false) {
    const pv_sae_desc& d = plan->d;
    uint32_t* feat_cnt = want_csr ? (cnt_over ? cnt_over : (uint32_t*)(wsb + ws.cnt)) : nullptr;
    uint32_t* wpos = want_csr ? (wpos_over ? wpos_over : (uint32_t*)(wsb + ws.wpos)) : nullptr;
    const bool fast = pv_sae_fast_ok(d) && st->W_encT && st->W_enc16T && st->enc_colsq;
    if (want_csr && !fast) PV_HIP_CHECK(hipMemsetAsync(feat_cnt, 0, (size_t)d.d_sae * 4, stream));      // (fast path: its first kernel zeroes them)
    // the pre-pass in fused launches (SaePre)
    SaePre pre = {x, 0, false};
    const bool fused = fast && !skip_prep && !cnt_over && g_pv_tuning.sae_fold;      // (training steps, the feature-parallel encode, inference)
    if (!skip_prep) {
        int rcp = sae_prep(d, x, (const float*)st->b_dec, batch_mean, N, fast, wsb, ws, stream, sae_in_width(d, st), fused ? &pre : nullptr,
                           st, feat_cnt);
        if (rcp) return rcp;
    }
    // algorithmic work of the encoder: 2 N d_in d_sae FLOP; bytes = operands once (x, W_enc as fp16) + the k results
    ProfScope prof(PV_PROF_SAE_ENC, stream, 2.0 * N * (double)d.d_in * d.d_sae,
                   ((double)N * d.d_in + (double)d.d_in * d.d_sae) * (fast ? 2.0 : 4.0) + (double)N * d.k * 8.0 +
                       (fast ? 0.0 : (double)N * d.d_sae * 8.0));
    if (fast) return sae_encode_fast(d, st, N, topk_idx, topk_val, feat_cnt, wpos, wsb, ws, stream, fused ? &pre : nullptr);
    {
        // exact path: hidden_pre = sae_in @ W_enc + b_enc (sae.py:567-574) on the fp32 MFMA, W_enc in its own [d_in][d_sae] layout
        GemmParams g = {};
        g.A = wsb + ws.sae_in; g.lda = d.d_in; g.a_mode = PV_A_PLAIN;
        if (st->W_encT) { g.Bt = st->W_encT; g.ldb = d.d_in; g.b_kn = 0; }          // the transposed master (always current)
        else { g.Bt = st->W_enc; g.ldb = d.d_sae; g.b_kn = 1; }                      // no shadows: the parameter's own [K][N] layout
        g.M = N; g.N = d.d_sae; g.K = d.d_in; g.epi = PV_EPI_BIAS; g.bias0 = st->b_enc; g.out0 = wsb + ws.hidden; g.ldo = d.d_sae;
        int rc = pv_launch_gemm(PV_DTYPE_F32, g, stream);
        if (rc) return rc;
        sae_topk_rows((const float*)(wsb + ws.hidden), topk_idx, topk_val, d.d_sae, d.k, N, nullptr, nullptr, 0, feat_cnt, wpos, stream);
    }
    PV_LAUNCH_CHECK("sae_topk_kernel");
    return PV_OK;
}

extern "C" int pv_sae_encode_topk(pv_sae_plan* plan, const pv_sae_state* st, const float* x, int32_t N,
                                  int32_t* topk_idx, float* topk_val, float* ln_mu, float* ln_std, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    PV_REQUIRE(plan && st && x && topk_idx && topk_val && workspace, "null argument");
    PV_REQUIRE(N >= 1 && N <= plan->d.max_tokens, "n_tokens exceeds plan max_tokens");
    const SaeWs ws = sae_carve(plan->d);
    PV_REQUIRE(workspace_bytes >= ws.total, "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* wsb = (unsigned char*)workspace;
    int rc = sae_encode_topk(plan, st, x, N, nullptr, topk_idx, topk_val, false, wsb, ws, stream);
    if (rc) return rc;
    if (ln_mu) PV_HIP_CHECK(hipMemcpyAsync(ln_mu, wsb + ws.mu, (size_t)N * 4, hipMemcpyDeviceToDevice, stream));
    if (ln_std) PV_HIP_CHECK(hipMemcpyAsync(ln_std, wsb + ws.sd, (size_t)N * 4, hipMemcpyDeviceToDevice, stream));
    return PV_OK;
}

// Inference forward of the module API (StandardSparseAutoencoder.forward / encode + decode, sae.py:557-645): top-k
// encode + sparse decode + LN-out -> sae_out [N, d_in]; no gradients, no statistics.  scalars[1] (optional) receives the
// mse loss over these N tokens taken as ONE batch (sae.py:144-149).
extern "C" int pv_sae_forward(pv_sae_plan* plan, const pv_sae_state* st, const float* x, int32_t N, float* sae_out,
                              int32_t* topk_idx, float* topk_val, float* ln_mu, float* ln_std, float* scalars, void* workspace,
                              size_t workspace_bytes, void* stream_) {
    PV_REQUIRE(!st || !sae_is_gated(st), "this entry point does not serve a gated state (pv_sae_state.gt)");
    PV_REQUIRE(!st || !sae_is_tc(st), "pv_sae_forward: not available for a transcoder state (pv_sae_state.tc)");
    PV_REQUIRE(plan && st && x && sae_out && topk_idx && topk_val && workspace, "null argument");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(N >= 1 && N <= d.max_tokens, "n_tokens exceeds plan max_tokens");
    const SaeWs ws = sae_carve(d);
    PV_REQUIRE(workspace_bytes >= ws.total, "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* wsb = (unsigned char*)workspace;
    int rc = sae_encode_topk(plan, st, x, N, nullptr, topk_idx, topk_val, false, wsb, ws, stream);
    if (rc) return rc;
    const dim3 grid((N + 3) / 4), block(256);
#define CALL(D)                                                                                                      \
    hipLaunchKernelGGL((sae_decode_kernel<D>), grid, block, 0, stream, x, (const float*)st->W_dec, (const float*)st->b_dec, \
                       (const int32_t*)topk_idx, (const float*)topk_val, (const float*)(wsb + ws.mu),                \
                       (const float*)(wsb + ws.sd), (const float*)(wsb + ws.norm), sae_out, (float*)nullptr, (float*)nullptr, \
                       (float*)(wsb + ws.loss_part), N, d.d_in, d.k, 0.0f, 0, (const float*)nullptr)
    V4_DISPATCH(d.d_in, CALL);
#undef CALL
    PV_LAUNCH_CHECK("sae_decode_kernel");
    if (scalars)
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, stream, (const float*)(wsb + ws.loss_part), scalars, N,
                           1.0f / ((float)N * (float)d.d_in), 1, -1);
    if (ln_mu) PV_HIP_CHECK(hipMemcpyAsync(ln_mu, wsb + ws.mu, (size_t)N * 4, hipMemcpyDeviceToDevice, stream));
    if (ln_std) PV_HIP_CHECK(hipMemcpyAsync(ln_std, wsb + ws.sd, (size_t)N * 4, hipMemcpyDeviceToDevice, stream));
    return PV_OK;
}

// The backward of a k-sparse step behind its decode kernel: the CSR by feature (counts and within-list positions came out of the
// selection), then the three backward kernels -- every row of gW_dec / gW_enc^T / gb_enc written exactly once.  N tokens of k slots;
// dY / sae_in hold N rows, dh N k entries.  loss_part (optional): the N per-token loss terms, scalars[0] = scalars[1] = loss_scale * sum.
// cs_here: also the 16-row partial column sums of dY.  val_b / dYb: the second decoder term of a pair (bwd_walk<DUAL>: the gated step).
// The two halves of sae_csr_backward.  sae_csr_build: the CSR by feature -- scan, chunk cuts / long lists / statistics / pair scatter, the
// two list sorts; it reads the selection's output only (counts, positions, indices), so with loss_part == NULL and cs_here false it
// does not wait for the decode kernel.  sae_csr_grads: the backward kernels.
static int sae_csr_build(pv_sae_plan* plan, pv_sae_state* st, int N, int k, const int32_t* topk_idx, const float* dY, float* scalars,
                         float* fire_count, int update_stats, bool sparse, const SaeTail& tb, unsigned char* wsb, const SaeWs& ws,
                         const float* loss_part, float loss_scale, bool cs_here, const uint32_t* gate, hipStream_t stream,
                         const uint32_t* cnt_in, bool folded = false) {
    // folded (pv_sae_step): the scan has run as a role of the decode launch (ScanRole); the loss rides in the post + fill launch, the two
    // list sorts are one launch -- two launches here instead of four
    const pv_sae_desc& d = plan->d;
    const int n_pairs = N * k;
    int rc = PV_OK;
    const dim3 block(256);
    {
        // CSR by feature: counts and within-list positions came out of the top-k selection; scan + atomic-free scatter
        const uint32_t* cnt = cnt_in ? cnt_in : (const uint32_t*)(wsb + ws.cnt);       // (cnt_in: counts that are not the selection's own)
        uint32_t* offs = (uint32_t*)(wsb + ws.offs);
        uint32_t* chunk_start = tb.chunk_start;
        int32_t* pairs = tb.pairs;
        int32_t* long_list = (int32_t*)(wsb + ws.long_list);
        uint32_t* n_long = (uint32_t*)(wsb + ws.n_long);
        const int max_chunks = (n_pairs + BWD_CH - 1) / BWD_CH;
        float* rowsq = (float*)(wsb + ws.rowsq);
        const int max_segs = tb.max_segs;
        uint32_t* seg_range = tb.seg_range;
        // (the scan's workgroup also reduces the loss: loss = mse_loss = sum / (N_global * d_in), sae.py:148; topk: loss == mse_loss,
        // :620-626 -- scalars[0] = scalars[1])
        if (!folded)
            hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, stream, cnt, offs, n_long, d.d_sae,
                               scalars, 1.0f / (float)N, loss_part, loss_part ? N : 0, loss_scale);
        // chunk cuts / long lists / statistics, the scatter of the pairs and (autoencoder) the partial column sums of dY: one launch
        {
            CsrPostArgs pa;
            pa.offs = offs; pa.chunk_start = chunk_start; pa.max_chunks = max_chunks; pa.long_list = long_list; pa.n_long = n_long;
            pa.seg_range = seg_range; pa.max_segs = max_segs; pa.act_freq = st->act_freq_scores; pa.n_since_fired = st->n_fwd_since_fired;
            pa.fire_count = fire_count; pa.d_sae = d.d_sae; pa.update_stats = update_stats; pa.gb_enc_sparse = sparse ? st->gb_enc : nullptr;
            pa.rowsq_sparse = sparse ? rowsq : nullptr; pa.ranged = sae_long_ranged(N) ? 1 : 0; pa.gate = gate;
            const int nb_post = (d.d_sae + 255) / 256, nb_fill = (n_pairs + 255) / 256, nb_cs = cs_here ? (N + CS_ROWS - 1) / CS_ROWS : 0;
            const int nb_loss = (folded && loss_part) ? 1 : 0;
            hipLaunchKernelGGL(csr_post_fill_kernel, dim3(nb_post + nb_fill + nb_cs + nb_loss), block, 0, stream, pa, nb_post, topk_idx,
                               (const uint32_t*)tb.wpos, pairs, n_pairs, nb_fill, (const float*)dY, (float*)(wsb + ws.colpart), N, d.d_in,
                               nb_cs, nb_loss ? loss_part : (const float*)nullptr, N, loss_scale, scalars);
        }
        const int ranged = sae_long_ranged(N) ? 1 : 0;
        if (folded && ranged) {
            rc = launch_csr_sort(long_list, n_long, (const uint32_t*)offs, pairs, seg_range, k, N, max_segs, d.d_sae, stream);
            if (rc) return rc;
            return PV_OK;
        }
        hipLaunchKernelGGL(csr_sort_short_kernel, dim3((d.d_sae + 3) / 4), dim3(256), 0, stream, (const uint32_t*)offs, pairs, d.d_sae);
        if (ranged) {
            rc = launch_long_sort(long_list, n_long, (const uint32_t*)offs, pairs, seg_range, k, N, max_segs, stream);
            if (rc) return rc;
        }
        PV_LAUNCH_CHECK("csr kernels");
    }
    return PV_OK;
}

static int sae_csr_grads(pv_sae_plan* plan, pv_sae_state* st, int N, int k, const int32_t* topk_idx, const float* topk_val, const float* dh,
                         const float* dY, const float* sae_in, bool sparse, const SaeTail& tb, unsigned char* wsb, const SaeWs& ws,
                         const uint32_t* gate, hipStream_t stream, const float* val_b, const float* dYb) {
    const pv_sae_desc& d = plan->d;
    const int n_pairs = N * k;
    const dim3 block(256);
    {
        uint32_t* offs = (uint32_t*)(wsb + ws.offs);
        uint32_t* chunk_start = tb.chunk_start;
        int32_t* pairs = tb.pairs;
        int32_t* long_list = (int32_t*)(wsb + ws.long_list);
        uint32_t* n_long = (uint32_t*)(wsb + ws.n_long);
        const int max_chunks = (n_pairs + BWD_CH - 1) / BWD_CH;
        float* rowsq = (float*)(wsb + ws.rowsq);
        const int max_segs = tb.max_segs;
        uint32_t* seg_range = tb.seg_range;
        float* seg_rows = tb.seg_rows;
        float* seg_b = tb.seg_b;
        const int ranged = sae_long_ranged(N) ? 1 : 0;
        const dim3 gridf((max_chunks + 3) / 4);
        // every gradient row is stored exactly once: by the zero kernel (features no token kept), the short-list kernel or
        // the long-list combine.  PV_SAE_SPARSE_GRADS: the rows of features no token kept are not touched at all -- pv_sae_apply
        // takes their gradient as zero from the feature offsets this step leaves in the workspace (about half of the features
        // on a trained-like batch: 2 x 39 MB not written here and not read there)
        plan->live_offs = sparse ? offs : nullptr;
#define CALL(D)                                                                                                        \
    if (!sparse)                                                                                                       \
        hipLaunchKernelGGL((sae_zero_empty_kernel<D>), dim3((d.d_sae + 3) / 4), block, 0, stream, (const uint32_t*)offs, st->gW_dec, \
                           st->gW_enc, st->gb_enc, rowsq, d.d_sae, d.d_in, gate);                                         \
    if (val_b) {                                                                                                       \
        hipLaunchKernelGGL((sae_backward_kernel<D, true>), gridf, block, 0, stream, (const uint32_t*)offs, (const uint32_t*)chunk_start, \
                           (const int32_t*)pairs, topk_idx, topk_val, (const float*)dh, (const float*)dY, (const float*)sae_in, \
                           st->gW_dec, st->gW_enc, st->gb_enc, rowsq, d.d_in, k, max_chunks, gate, val_b, dYb);           \
        hipLaunchKernelGGL((sae_backward_seg_kernel<D, true>), dim3(ranged ? 2048 : 1024), dim3(ranged ? 512 : 256), 0, stream, \
                           (const uint32_t*)offs, (const uint32_t*)seg_range, (const uint32_t*)n_long, (const int32_t*)pairs, topk_idx, \
                           topk_val, (const float*)dh, (const float*)dY, (const float*)sae_in, seg_rows, seg_b,          \
                           d.d_in, k, max_segs, ranged, gate, val_b, dYb);                                                \
    } else {                                                                                                           \
        hipLaunchKernelGGL((sae_backward_kernel<D>), gridf, block, 0, stream, (const uint32_t*)offs, (const uint32_t*)chunk_start, \
                           (const int32_t*)pairs, topk_idx, topk_val, (const float*)dh,                                  \
                           (const float*)dY, (const float*)sae_in, st->gW_dec, st->gW_enc, st->gb_enc, rowsq, d.d_in, k, max_chunks, gate); \
        hipLaunchKernelGGL((sae_backward_seg_kernel<D>), dim3(ranged ? 2048 : 1024), dim3(ranged ? 512 : 256), 0, stream, (const uint32_t*)offs, \
                           (const uint32_t*)seg_range, (const uint32_t*)n_long, (const int32_t*)pairs, topk_idx,         \
                           topk_val, (const float*)dh, (const float*)dY, (const float*)sae_in, seg_rows, seg_b,          \
                           d.d_in, k, max_segs, ranged, gate);                                                           \
    }                                                                                                                  \
    hipLaunchKernelGGL((sae_backward_long_kernel<D>), dim3(256), block, 0, stream, (const int32_t*)long_list,           \
                       (const uint32_t*)n_long, (const float*)seg_rows, (const float*)seg_b, st->gW_dec, st->gW_enc,   \
                       st->gb_enc, rowsq, d.d_in, max_segs, gate)
        V4_DISPATCH(d.d_in, CALL);
#undef CALL
        PV_LAUNCH_CHECK("sae_backward_kernel");
    }
    return PV_OK;
}

int sae_csr_backward(pv_sae_plan* plan, pv_sae_state* st, int N, int k, const int32_t* topk_idx, const float* topk_val, const float* dh,
                     const float* dY, const float* sae_in, float* scalars, float* fire_count, int update_stats, bool sparse,
                     const SaeTail& tb, unsigned char* wsb, const SaeWs& ws, const float* loss_part, float loss_scale, bool cs_here,
                     const uint32_t* gate, hipStream_t stream, const float* val_b, const float* dYb, const uint32_t* cnt_in) {
    int rc = sae_csr_build(plan, st, N, k, topk_idx, dY, scalars, fire_count, update_stats, sparse, tb, wsb, ws, loss_part, loss_scale,
                           cs_here, gate, stream, cnt_in);
    if (rc) return rc;
    return sae_csr_grads(plan, st, N, k, topk_idx, topk_val, dh, dY, sae_in, sparse, tb, wsb, ws, gate, stream, val_b, dYb);
}

// Everything of the k-sparse step behind the selection: decode + LN-out + loss + dY + dh, the CSR by feature, the sparse backward,
// the bias gradients.  Shared by pv_sae_step (k = the plan's k) and by the sparse form of the ReLU + L1 step (pv_sae_relu_step:
// k = the per-token capacity, tok_cnt / dh_add / gate as described at sae_decode_kernel; the k-dependent buffers come in through tb).
int sae_sparse_tail(pv_sae_plan* plan, pv_sae_state* st, const float* x, int N, int n_global, int k, const int32_t* topk_idx,
                    const float* topk_val, float* sae_out, float* scalars, float* fire_count, int update_stats, bool sparse,
                    const float* inv_norm, const SaeTail& tb, unsigned char* wsb, const SaeWs& ws, const float* y, const float* bdo,
                    const float* skip, bool tc, float dh_add, const uint32_t* tok_cnt, const uint32_t* gate, hipStream_t stream,
                    bool bias_grads, float* sq_scalars) {
    const pv_sae_desc& d = plan->d;
    const int n_pairs = N * k;
    int rc = PV_OK;
    float* dY = (float*)(wsb + ws.dY);
    float* dh = tb.dh;
    float* sae_in = (float*)(wsb + ws.sae_in);
    {
        ProfScope prof(PV_PROF_SAE_BWD, stream, 4.0 * n_pairs * (double)d.d_in * 2.0, 0.0);
        const float grad_scale = 2.0f / ((float)n_global * (float)sae_loss_width(d, st));      // (a transcoder: the mean is over N x d_out)
        const float loss_scale = 1.0f / ((float)n_global * (float)sae_loss_width(d, st));
        const dim3 grid((N + 3) / 4), block(256);
        // the CSR scan reads the selection's counts only -- it runs as one more workgroup of the decode launch (ScanRole).  (The sparse
        // form of the ReLU step on a batch that turns out dense: the scan walks counts nobody will use and writes scalars the dense
        // form overwrites -- as the scan's own launch did)
        const bool folded = g_pv_tuning.sae_fold != 0;
        ScanRole scan = {};
        if (folded) {
            scan.cnt = (const uint32_t*)(wsb + ws.cnt); scan.offs = (uint32_t*)(wsb + ws.offs); scan.n_long = (uint32_t*)(wsb + ws.n_long);
            scan.d_sae = d.d_sae; scan.scalars = scalars; scan.inv_tokens = 1.0f / (float)N;
        }
        const dim3 grid_dec((N + 3) / 4 + (folded ? 1 : 0));
#define CALL(D)                                                                                                      \
    hipLaunchKernelGGL((sae_decode_kernel<D>), grid_dec, block, 0, stream, y, (const float*)st->W_dec, bdo,          \
                       topk_idx, topk_val, (const float*)(wsb + ws.mu),      \
                       (const float*)(wsb + ws.sd), (const float*)(wsb + ws.norm), sae_out, dY, dh,             \
                       (float*)(wsb + ws.loss_part), N, d.d_in, k, grad_scale, 1, inv_norm, (const float*)nullptr, skip, dh_add, tok_cnt, gate, scan)
        V4_DISPATCH(d.d_in, CALL);
#undef CALL
        PV_LAUNCH_CHECK("sae_decode_kernel");
        const bool cs_here = bias_grads && !tc;
        rc = sae_csr_build(plan, st, N, k, topk_idx, dY, scalars, fire_count, update_stats, sparse, tb, wsb, ws,
                           (const float*)(wsb + ws.loss_part), loss_scale, cs_here, gate, stream, nullptr, folded);
        if (rc) return rc;
        rc = sae_csr_grads(plan, st, N, k, topk_idx, topk_val, dh, dY, sae_in, sparse, tb, wsb, ws, gate, stream, nullptr, nullptr);
        if (rc) return rc;
        // gb_dec = colsum(dY) - W_enc @ gb_enc: both terms as partial rows of one column sum
        // (bias_grads false: pv_sae_relu_step runs them once, behind whichever of its two forms produced dY and gb_enc)
        if (bias_grads) {
            rc = tc ? sae_tc_bias_grads(d, st, dY, N, wsb, ws, stream) : sae_gbdec(d, st, dY, N, wsb, ws, stream, cs_here, sq_scalars);
            if (rc) return rc;
            if (tc) {
                rc = sae_tc_skip_backward(d, st, x, dY, N, stream);
                if (rc) return rc;
            }
        }
    }
    return PV_OK;
}

// ------------------------------------------------------------------------------------------------
// The gated step in sparse form (pv_sae_gated_step_sparse; the dense form and the mathematics: pv_sae_gated_step, sae_dense.hip).
// A gated SAE's forward is sparse in the OPEN GATES: feature_acts, relu(gate_pre) and every gradient behind them vanish where
// gate_pre <= 0 (sae.py:703-716, :773-792).  The selection (relu_select_kernel<GATED>) leaves, per token, the list of its open gates
// with both values per pair {f = feature_acts, g = relu(gate_pre)}; from there the step is the k-sparse machinery with TWO decoder
// terms per pair:
//     decode (ONE gather of W_dec[j] per pair):  reconstruction = sum f W_dec[j] (+ LN-out, mse, dY, dM = (dY . W_dec[j]) [f > 0]),
//                                                via the gate   = sum g W_dec[j] (aux loss against sae_in, dVia, dG = dVia . W_dec[j] + l1 / N)
//     backward (bwd_walk<DUAL>):  gW_dec[j] = sum f dY[n] + g dVia[n],  gW_enc^T[j] = sum dP sae_in[n],  gb_enc[j] = sum dP = colsum(dP)
//     with dP = dM e^r + dG (parked in gb_enc for gb_dec, as the dense form does)
// ------------------------------------------------------------------------------------------------
namespace {
template <int V4>
__global__ __launch_bounds__(256) void gated_decode_kernel(
    const float* __restrict__ x, const float* __restrict__ sae_in, const float* __restrict__ W_dec, const float* __restrict__ b_dec,
    const int32_t* __restrict__ idx, const float* __restrict__ valf, const float* __restrict__ valg, const float* __restrict__ mu,
    const float* __restrict__ sd, const float* __restrict__ norm, float* __restrict__ sae_out, float* __restrict__ dY,
    float* __restrict__ dVia, float* __restrict__ dM, float* __restrict__ dG, float* __restrict__ mse_part, float* __restrict__ aux_part,
    int n_tok, int d, int k, float grad_scale /* 2 / (N d_in) */, float aux_scale /* 2 / N */, float dh_add /* l1 / N */,
    const uint32_t* __restrict__ tok_cnt, const uint32_t* __restrict__ gate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    if (*gate != 0u) return;
    bool ok[V4];
    int col[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        col[i] = 4 * lane + 256 * i;
        ok[i] = col[i] < d;
    }
    float4 af[V4], ag[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) { af[i] = make_float4(0.f, 0.f, 0.f, 0.f); ag[i] = af[i]; }
    const int32_t* ir = idx + (int64_t)n * k;
    const float* fr = valf + (int64_t)n * k;
    const float* gr = valg + (int64_t)n * k;
    const int k_walk = min(k, (int)((tok_cnt[n] + 3u) & ~3u));       // (slots beyond it are holes: g = 0)
    for (int s = 0; s < k_walk; s += 4) {
        float a[4], b[4];
        float4 w[4][V4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ju = ir[s + u];
            a[u] = fr[s + u];
            b[u] = gr[s + u];
            const float* wr = W_dec + (int64_t)ju * d;
#pragma unroll
            for (int i = 0; i < V4; ++i) w[u][i] = ld4(wr + col[i], ok[i] && b[u] != 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)                       // (slot order: fixed)
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                af[i].x += a[u] * w[u][i].x; af[i].y += a[u] * w[u][i].y; af[i].z += a[u] * w[u][i].z; af[i].w += a[u] * w[u][i].w;
                ag[i].x += b[u] * w[u][i].x; ag[i].y += b[u] * w[u][i].y; ag[i].z += b[u] * w[u][i].z; ag[i].w += b[u] * w[u][i].w;
            }
    }
    const float m = mu[n], sdv = sd[n], nf = norm[n];
    float lf = 0.f, lg = 0.f;
    float4 gf[V4], gg[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        gf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gg[i] = gf[i];
        if (ok[i]) {
            const float4 bd = *reinterpret_cast<const float4*>(b_dec + col[i]);
            const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)n * d + col[i]);
            const float4 sv = *reinterpret_cast<const float4*>(sae_in + (int64_t)n * d + col[i]);
            float4 o, e;
            o.x = (af[i].x + bd.x) * sdv + m; o.y = (af[i].y + bd.y) * sdv + m;          // as sae_decode_kernel / gated_finish_kernel
            o.z = (af[i].z + bd.z) * sdv + m; o.w = (af[i].w + bd.w) * sdv + m;
            e.x = o.x - xv.x; e.y = o.y - xv.y; e.z = o.z - xv.z; e.w = o.w - xv.w;
            if (sae_out) *reinterpret_cast<float4*>(sae_out + (int64_t)n * d + col[i]) = o;
            lf += (e.x * e.x) / nf + (e.y * e.y) / nf + (e.z * e.z) / nf + (e.w * e.w) / nf;
            gf[i].x = grad_scale * e.x / nf * sdv; gf[i].y = grad_scale * e.y / nf * sdv;
            gf[i].z = grad_scale * e.z / nf * sdv; gf[i].w = grad_scale * e.w / nf * sdv;
            *reinterpret_cast<float4*>(dY + (int64_t)n * d + col[i]) = gf[i];
            e.x = ag[i].x + bd.x - sv.x; e.y = ag[i].y + bd.y - sv.y; e.z = ag[i].z + bd.z - sv.z; e.w = ag[i].w + bd.w - sv.w;
            lg += e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w;                          // sae.py:786-792
            gg[i].x = aux_scale * e.x; gg[i].y = aux_scale * e.y; gg[i].z = aux_scale * e.z; gg[i].w = aux_scale * e.w;
            *reinterpret_cast<float4*>(dVia + (int64_t)n * d + col[i]) = gg[i];
        }
    }
    lf = wave_sum(lf);
    lg = wave_sum(lg);
    if (lane == 0) { mse_part[n] = lf; aux_part[n] = lg; }
    for (int s = 0; s < k_walk; s += 4) {
        float df[4], dg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ju = ir[s + u];
            const bool live = gr[s + u] > 0.f;                    // (wave-uniform)
            const float* wr = W_dec + (int64_t)ju * d;
            float tf = 0.f, tg = 0.f;
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                const float4 wv = ld4(wr + col[i], ok[i] && live);
                tf += dot4(gf[i], wv);
                tg += dot4(gg[i], wv);
            }
            df[u] = tf;
            dg[u] = tg;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                df[u] += __shfl_xor(df[u], o, 64);
                dg[u] += __shfl_xor(dg[u], o, 64);
            }
        if (lane < 4) {
            const float sf = lane == 0 ? df[0] : (lane == 1 ? df[1] : (lane == 2 ? df[2] : df[3]));
            const float sg = lane == 0 ? dg[0] : (lane == 1 ? dg[1] : (lane == 2 ? dg[2] : dg[3]));
            dM[(int64_t)n * k + s + lane] = fr[s + lane] > 0.f ? sf : 0.f;                    // (dY W_dec^T) [f > 0]
            dG[(int64_t)n * k + s + lane] = gr[s + lane] > 0.f ? sg + dh_add : 0.f;           // (dVia W_dec^T + l1 / N) [gate_pre > 0]
        }
    }
}

// dP = dM e^r + dG per pair (sae.py:708-712 backwards); holes (never written by the decode kernel) -> 0
__global__ __launch_bounds__(256) void gated_pairs_kernel(const int32_t* __restrict__ idx, const uint32_t* __restrict__ wpos,
                                                          const float* __restrict__ dM, const float* __restrict__ dG,
                                                          const float* __restrict__ r_mag, float* __restrict__ dP, int n_pairs,
                                                          const uint32_t* __restrict__ gate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs || *gate != 0u) return;
    dP[i] = wpos[i] != 0xffffffffu ? dM[i] * expf(r_mag[idx[i]]) + dG[i] : 0.f;
}

// per feature, over its list (ascending token order): gb_gate = sum dG, gb_mag = sum dM, gr_mag = sum dM (mag_pre - b_mag) = sum dM f -
// b_mag gb_mag, pgsum = sum relu(gate_pre) (the decoder-norm factor of the L1 term), the firing count of feature_acts + the
// statistics (train_sae.py:356-361).  One wave per feature.
__global__ __launch_bounds__(256) void gated_feat_kernel(const uint32_t* __restrict__ offs, const int32_t* __restrict__ pairs,
                                                         const float* __restrict__ valf, const float* __restrict__ valg,
                                                         const float* __restrict__ dM, const float* __restrict__ dG,
                                                         const float* __restrict__ b_mag, int F, float* __restrict__ gb_gate,
                                                         float* __restrict__ gb_mag, float* __restrict__ gr_mag, float* __restrict__ pgsum,
                                                         float* __restrict__ fire_count, float* __restrict__ act_freq,
                                                         float* __restrict__ n_since_fired, int update_stats,
                                                         const uint32_t* __restrict__ gate) {
    if (*gate != 0u) return;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= F) return;
    const uint32_t beg = offs[j], end = offs[j + 1];
    float sg = 0.f, sm = 0.f, smf = 0.f, sp = 0.f, fired = 0.f;
#pragma unroll 4
    for (uint32_t q = beg + lane; q < end; q += 64) {      // (unrolled: the gathers of four trips in flight; the sums in order)
        const int p = pairs[q];
        const float f = valf[p], m = dM[p];
        sg += dG[p];
        sm += m;
        smf += m * f;
        sp += valg[p];
        fired += f > 0.f ? 1.f : 0.f;
    }
    sg = wave_sum(sg); sm = wave_sum(sm); smf = wave_sum(smf); sp = wave_sum(sp); fired = wave_sum(fired);
    if (lane != 0) return;
    gb_gate[j] = sg;
    gb_mag[j] = sm;
    gr_mag[j] = smf - b_mag[j] * sm;
    pgsum[j] = sp;
    if (fire_count) fire_count[j] = fired;
    if (update_stats) {
        act_freq[j] += fired;
        n_since_fired[j] = fired > 0.f ? 0.f : n_since_fired[j] + 1.f;
    }
}

// scalars: 1 mse, 6 aux, 4 l1, 2 l0, 0 their sum (sae.py:748); fixed summation order
__global__ __launch_bounds__(256) void gated_sparse_scalars_kernel(const float* __restrict__ mse_part, const float* __restrict__ aux_part,
                                                                   const float* __restrict__ l1part, const float* __restrict__ l0part,
                                                                   int n_tok, float s_mse, float s_aux, float s_l1, float s_l0,
                                                                   float* __restrict__ scalars, const uint32_t* __restrict__ gate) {
    if (*gate != 0u) return;
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float a = 0.f, b = 0.f, c = 0.f, e = 0.f;
#pragma unroll 4
    for (int i = threadIdx.x; i < n_tok; i += 256) { a += mse_part[i]; b += aux_part[i]; c += l1part[i]; e += l0part[i]; }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); e = wave_sum(e);
    if (lane == 0) { red[0][wv] = a; red[1][wv] = b; red[2][wv] = c; red[3][wv] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mse = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * s_mse;
        const float aux = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * s_aux;
        const float l1 = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) * s_l1;
        scalars[1] = mse; scalars[6] = aux; scalars[4] = l1;
        scalars[2] = ((red[3][0] + red[3][1]) + (red[3][2] + red[3][3])) * s_l0;
        scalars[0] = mse + l1 + aux;
    }
}
}  // namespace

GatedSparseWs gated_sparse_carve(const pv_sae_desc& d, int n_tokens, int cap) {
    GatedSparseWs w;
    w.rw = relu_carve(d, n_tokens, cap);
    size_t off = w.rw.total;
    auto take = [&](size_t bytes) { size_t o = off; off += (size_t)pv_align_up((int64_t)bytes, 256); return o; };
    const size_t N = (size_t)n_tokens, P = N * (size_t)cap;
    w.valg = take(P * 4);
    w.dM = take(P * 4);
    w.dG = take(P * 4);
    w.l0part = take(N * 4);
    w.total = off + 256;
    return w;
}

// dYs: [2N, d_in] = [dY; dVia] (the dense form's layout), auxpart: [N], pgsum: [d_sae] -- regions of the gated scratch the two
// forms share, so that everything behind them (the decoder-norm term, gb_dec) runs once.  Leaves colsum(dP) in st->gb_enc.
int sae_gated_sparse(pv_sae_plan* plan, pv_sae_state* st, const float* x, int N, int n_global, int cap, float l1_coefficient,
                     int update_stats, unsigned char* rwb, const GatedSparseWs& gs, pv_sae_out* out, unsigned char* wsb, const SaeWs& ws,
                     float* dYs, float* auxpart, float* pgsum, hipStream_t stream) {
    const pv_sae_desc& d = plan->d;
    const pv_sae_gated& t = st->gt;
    const ReluWs& rw = gs.rw;
    const int D = d.d_in, F = d.d_sae, n_pairs = N * cap;
    const float ng = (float)n_global;
    uint32_t* mode = (uint32_t*)(rwb + rw.mode);
    int32_t* idx = (int32_t*)(rwb + rw.idx);
    float* valf = (float*)(rwb + rw.val);
    float* valg = (float*)(rwb + gs.valg);
    uint32_t* wpos = (uint32_t*)(rwb + rw.wpos);
    uint32_t* tok_cnt = (uint32_t*)(rwb + rw.tok_cnt);
    float* l1part = (float*)(rwb + rw.l1part);
    float* l0part = (float*)(rwb + gs.l0part);
    float* dM = (float*)(rwb + gs.dM);
    float* dG = (float*)(rwb + gs.dG);
    float* dP = (float*)(rwb + rw.dh);
    const float* sae_in = (const float*)(wsb + ws.sae_in);
    float* dVia = dYs + (size_t)N * D;
    int rc;
    {
        ProfScope prof(PV_PROF_SAE_ENC, stream, 2.0 * N * (double)D * F, ((double)N * D + (double)D * F) * 2.0);
        rc = sae_encode_relu(d, st, N, cap, idx, valf, tok_cnt, l1part, (uint32_t*)(rwb + rw.cand_cnt), rwb + rw.cand,
                             (uint32_t*)(wsb + ws.cnt), wpos, mode, (const float*)out->scalars, wsb, ws, stream, l0part, valg);
        if (rc) return rc;
    }
    ProfScope prof(PV_PROF_SAE_BWD, stream, 6.0 * n_pairs * (double)D * 2.0, 0.0);
    const dim3 grid((N + 3) / 4), block(256);
#define CALL(V)                                                                                                                   \
    hipLaunchKernelGGL((gated_decode_kernel<V>), grid, block, 0, stream, x, sae_in, (const float*)st->W_dec, (const float*)st->b_dec, \
                       (const int32_t*)idx, (const float*)valf, (const float*)valg, (const float*)(wsb + ws.mu),                    \
                       (const float*)(wsb + ws.sd), (const float*)(wsb + ws.norm), out->sae_out, dYs, dVia, dM, dG,                 \
                       (float*)(wsb + ws.loss_part), auxpart, N, D, cap, 2.0f / (ng * (float)D), 2.0f / ng, l1_coefficient / ng,   \
                       (const uint32_t*)tok_cnt, (const uint32_t*)mode)
    V4_DISPATCH(D, CALL);
#undef CALL
    PV_LAUNCH_CHECK("gated_decode_kernel");
    hipLaunchKernelGGL(gated_pairs_kernel, dim3((n_pairs + 255) / 256), block, 0, stream, (const int32_t*)idx, (const uint32_t*)wpos,
                       (const float*)dM, (const float*)dG, (const float*)t.r_mag, dP, n_pairs, (const uint32_t*)mode);
    PV_LAUNCH_CHECK("gated_pairs_kernel");
    SaeTail tb;
    tb.dh = dP; tb.chunk_start = (uint32_t*)(rwb + rw.cursor); tb.wpos = wpos; tb.seg_range = (uint32_t*)(rwb + rw.seg_range);
    tb.seg_rows = (float*)(rwb + rw.seg_rows); tb.seg_b = (float*)(rwb + rw.seg_b); tb.pairs = (int32_t*)(rwb + rw.pairs);
    tb.max_segs = rw.max_segs;
    rc = sae_csr_backward(plan, st, N, cap, idx, valf, dP, dYs, sae_in, out->scalars, nullptr, 0, false, tb, wsb, ws, nullptr, 0.f,
                          false, mode, stream, valg, dVia);
    if (rc) return rc;
    hipLaunchKernelGGL(gated_feat_kernel, dim3((F + 3) / 4), block, 0, stream, (const uint32_t*)(wsb + ws.offs), (const int32_t*)tb.pairs,
                       (const float*)valf, (const float*)valg, (const float*)dM, (const float*)dG, (const float*)t.b_mag, F, t.gb_gate,
                       t.gb_mag, t.gr_mag, pgsum, out->fire_count, st->act_freq_scores, st->n_fwd_since_fired, update_stats,
                       (const uint32_t*)mode);
    hipLaunchKernelGGL(gated_sparse_scalars_kernel, dim3(1), block, 0, stream, (const float*)(wsb + ws.loss_part), (const float*)auxpart,
                       (const float*)l1part, (const float*)l0part, N, 1.0f / (ng * (float)D), 1.0f / ng, l1_coefficient / ng,
                       1.0f / (float)N, out->scalars, (const uint32_t*)mode);
    PV_LAUNCH_CHECK("gated sparse kernels");
    return PV_OK;
}

// ------------------------------------------------------------------------------------------------
// The TOP-K form of the gated SAE (activation_fn_str = "topk" on a GatedSparseAutoencoder, sae.py:699-716, 741-745, 773-778):
//     feature_acts = [gate_pre > 0] TopK(mag_pre),   mag_pre  = sae_in (W_enc e^r_mag) + b_mag
//     pi_gate_act  = TopK(gate_pre),                 gate_pre = sae_in W_enc + b_gate
//     loss = mse(feature_acts W_dec + b_dec) + aux(pi_gate_act W_dec + b_dec against sae_in);  no L1 term
// Two k-sparse lists per token, so the step is the k-sparse machinery twice over:
//   * the magnitude path's top-k runs on the SAME filtered encoder against a scaled copy of the encoder shadows (W_magT[j] = W_encT[j]
//     e^r_j in fp32 and fp16, its column norms: one pass per step, gated_scale_rows_kernel), the gate path's on the shadows themselves
//     with b_gate as the bias;
//   * the gate of every kept magnitude is evaluated EXACTLY (gated_topk_mask_kernel: the fp32 dot product of the exact re-scoring);
//   * decode once per list (sae_decode_kernel: the second against sae_in with constant LN-out terms), then ONE CSR + sparse backward
//     over the two lists stacked as 2N "tokens" of k slots -- rows [0, N): {f, dY, dh = dM e^r}, rows [N, 2N): {g, dVia, dh = dG},
//     sae_in repeated -- so that gW_dec, gW_enc^T and colsum(dP) (parked in gb_enc) come out of the unmodified backward kernels;
//   * per-feature sums over the CSR lists: gb_mag, gr_mag, gb_gate, firing statistics.
// The plan must be created for 2 x the tokens of a step (its k-dependent buffers hold both lists).
// ------------------------------------------------------------------------------------------------
namespace {
struct GatedTopkWs {
    size_t total, wmagT, wmag16, colsq, cnt_m, cnt2, dM, mu0, one, l0part, auxpart, tmpd, cspart;
};
GatedTopkWs gated_topk_carve(const pv_sae_desc& d, int n_tokens) {
    GatedTopkWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (size_t)pv_align_up((int64_t)bytes, 256); return o; };
    const size_t F = d.d_sae, D = d.d_in, N = n_tokens;
    w.wmagT = take(F * D * 4);
    w.wmag16 = take(F * D * 2);
    w.colsq = take(F * 4);
    w.cnt_m = take(F * 4);
    w.cnt2 = take(F * 4);
    w.dM = take(N * (size_t)d.k * 4);
    w.mu0 = take(N * 4);
    w.one = take(N * 4);
    w.l0part = take(N * 4);
    w.auxpart = take(N * 4);
    w.tmpd = take(D * 4);
    w.cspart = take((N / 16 + 2) * D * 4);
    w.total = off + 256;
    return w;
}

// W_magT[j] = W_encT[j] e^r_j (fp32 + fp16) and its squared column norm (INFINITY outside the fp16 range, as adam_wenct_kernel);
// one wave per feature
__global__ __launch_bounds__(256) void gated_scale_rows_kernel(const float* __restrict__ WT, const float* __restrict__ r_mag,
                                                               float* __restrict__ MT, _Float16* __restrict__ M16T,
                                                               float* __restrict__ colsq, int F, int d) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= F) return;
    const float e = expf(r_mag[j]);
    float sq = 0.f;
    bool big = false;
    for (int c = 4 * lane; c < d; c += 256) {
        float4 w = *reinterpret_cast<const float4*>(WT + (int64_t)j * d + c);
        w.x *= e; w.y *= e; w.z *= e; w.w *= e;
        *reinterpret_cast<float4*>(MT + (int64_t)j * d + c) = w;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 hv = {(_Float16)w.x, (_Float16)w.y, (_Float16)w.z, (_Float16)w.w};
        *reinterpret_cast<h4*>(M16T + (int64_t)j * d + c) = hv;
        sq += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
        big = big || !(fabsf(w.x) <= 6.0e4f) || !(fabsf(w.y) <= 6.0e4f) || !(fabsf(w.z) <= 6.0e4f) || !(fabsf(w.w) <= 6.0e4f);
    }
    sq = wave_sum(sq);
    const bool any_big = __any(big);
    if (lane == 0) colsq[j] = any_big ? INFINITY : sq;
}

// a wave per token: the gate of each kept magnitude, exactly (gate_pre = sae_in . W_encT[j] + b_gate[j] in the summation order of the
// exact re-scoring), feature_acts = [gate_pre > 0] relu(top-k magnitude) written over the magnitude; the token's count of f > 0 (l0);
// the constant LN-out terms of the pass through the gate
template <int V4>
__global__ __launch_bounds__(256) void gated_topk_mask_kernel(const float* __restrict__ sae_in, const float* __restrict__ W_encT,
                                                              const float* __restrict__ b_gate, const int32_t* __restrict__ idx,
                                                              float* __restrict__ val, float* __restrict__ l0part,
                                                              float* __restrict__ mu0, float* __restrict__ one, int n_tok, int d, int k) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    bool ok[V4];
    int col[V4];
    float4 xr[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        col[i] = 4 * lane + 256 * i;
        ok[i] = col[i] < d;
        xr[i] = ld4(sae_in + (int64_t)n * d + col[i], ok[i]);
    }
    float cnt = 0.f;
    for (int s = 0; s < k; s += 4) {
        float acc[4];
        int jj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int su = min(s + u, k - 1);
            jj[u] = idx[(int64_t)n * k + su];
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
        if (lane < 4 && s + lane < k) {
            const float av = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
            const int jv = lane == 0 ? jj[0] : (lane == 1 ? jj[1] : (lane == 2 ? jj[2] : jj[3]));
            const float m = val[(int64_t)n * k + s + lane];
            const float f = (av + b_gate[jv] > 0.f && m > 0.f) ? m : 0.f;
            val[(int64_t)n * k + s + lane] = f;
            cnt += f > 0.f ? 1.f : 0.f;
        }
    }
    cnt = wave_sum(cnt);
    if (lane == 0) { l0part[n] = cnt; mu0[n] = 0.f; one[n] = 1.f; }
}

// threads [0, n_half): dh of the two stacked lists (dM e^r | dG is already in place), the second list's positions behind the first's;
// threads [n_half, n_half + F): the summed counts
__global__ __launch_bounds__(256) void gated_topk_pairs_kernel(const int32_t* __restrict__ idx, uint32_t* __restrict__ wpos,
                                                               const uint32_t* __restrict__ cnt_m, const uint32_t* __restrict__ cnt_g,
                                                               uint32_t* __restrict__ cnt2, const float* __restrict__ dM,
                                                               const float* __restrict__ r_mag, float* __restrict__ dh2, int n_half, int F) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_half) {
        const int j = i - n_half;
        if (j < F) cnt2[j] = cnt_m[j] + cnt_g[j];
        return;
    }
    dh2[i] = dM[i] * expf(r_mag[idx[i]]);                       // d mag_pre / d p = e^r (sae.py:708-712 backwards)
    const uint32_t w = wpos[n_half + i];
    if (w != 0xffffffffu) wpos[n_half + i] = w + cnt_m[idx[n_half + i]];
}

// per feature over its (token-ordered) list: pairs of the first list -> gb_mag = sum dM, gr_mag = sum dM (mag_pre - b_mag) (dM != 0
// only where f = mag_pre > 0), the firing count of feature_acts; pairs of the second -> gb_gate = sum dG.  One wave per feature.
__global__ __launch_bounds__(256) void gated_topk_feat_kernel(const uint32_t* __restrict__ offs, const int32_t* __restrict__ pairs,
                                                              const float* __restrict__ val2, const float* __restrict__ dM,
                                                              const float* __restrict__ dh2, const float* __restrict__ b_mag, int n_half,
                                                              int F, float* __restrict__ gb_gate, float* __restrict__ gb_mag,
                                                              float* __restrict__ gr_mag, float* __restrict__ fire_count,
                                                              float* __restrict__ act_freq, float* __restrict__ n_since_fired,
                                                              int update_stats) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= F) return;
    const uint32_t beg = offs[j], end = offs[j + 1];
    float sg = 0.f, sm = 0.f, smf = 0.f, fired = 0.f;
    for (uint32_t q = beg + lane; q < end; q += 64) {
        const int p = pairs[q];
        if (p < n_half) {
            const float f = val2[p], m = dM[p];
            sm += m;
            smf += m * f;
            fired += f > 0.f ? 1.f : 0.f;
        } else {
            sg += dh2[p];
        }
    }
    sg = wave_sum(sg); sm = wave_sum(sm); smf = wave_sum(smf); fired = wave_sum(fired);
    if (lane != 0) return;
    gb_gate[j] = sg;
    gb_mag[j] = sm;
    gr_mag[j] = smf - b_mag[j] * sm;
    if (fire_count) fire_count[j] = fired;
    if (update_stats) {
        act_freq[j] += fired;
        n_since_fired[j] = fired > 0.f ? 0.f : n_since_fired[j] + 1.f;
    }
}

__global__ __launch_bounds__(256) void gated_topk_scalars_kernel(const float* __restrict__ mse_part, const float* __restrict__ aux_part,
                                                                 const float* __restrict__ l0part, int n_tok, float s_mse, float s_aux,
                                                                 float s_l0, float* __restrict__ scalars) {
    __shared__ float red[3][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float a = 0.f, b = 0.f, e = 0.f;
#pragma unroll 4
    for (int i = threadIdx.x; i < n_tok; i += 256) { a += mse_part[i]; b += aux_part[i]; e += l0part[i]; }
    a = wave_sum(a); b = wave_sum(b); e = wave_sum(e);
    if (lane == 0) { red[0][wv] = a; red[1][wv] = b; red[2][wv] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mse = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * s_mse;
        const float aux = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * s_aux;
        scalars[1] = mse; scalars[6] = aux; scalars[4] = 0.f;               // (no L1 term in the top-k form, sae.py:741-745)
        scalars[2] = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) * s_l0;
        scalars[0] = mse + aux;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += x[i];
}
}  // namespace

extern "C" size_t pv_sae_gated_topk_scratch_bytes(const pv_sae_plan* plan, int32_t n_tokens) {
    if (!plan || n_tokens < 1) return 0;
    return gated_topk_carve(plan->d, n_tokens).total;
}

extern "C" int pv_sae_gated_topk_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, const float* batch_mean,
                                      int32_t n_global, int32_t flags, pv_sae_out* out, void* workspace, size_t workspace_bytes,
                                      void* scratch, size_t scratch_bytes, void* stream_) {
    const int update_stats = (flags & PV_SAE_UPDATE_STATS) ? 1 : 0;
    PV_REQUIRE(plan && st && x && out && workspace && scratch && out->scalars && out->topk_idx && out->topk_val, "null argument");
    PV_REQUIRE(sae_is_gated(st) && !sae_is_tc(st), "pv_sae_gated_topk_step needs a gated state (pv_sae_state.gt) and no transcoder");
    const pv_sae_gated& t = st->gt;
    PV_REQUIRE(t.r_mag && t.b_mag && t.gb_gate && t.gr_mag && t.gb_mag, "gated state");
    PV_REQUIRE(st->W_dec && st->b_dec && st->gW_enc && st->gW_dec && st->gb_enc && st->gb_dec && st->W_encT, "state");
    PV_REQUIRE(!update_stats || (st->act_freq_scores && st->n_fwd_since_fired), "stats buffers");
    PV_REQUIRE(flags & PV_SAE_RENORM_DECODER, "pv_sae_gated_topk_step: PV_SAE_RENORM_DECODER is required (train_sae.py:307 is part of the step)");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(N >= 1 && 2 * (int64_t)N <= d.max_tokens, "the plan must be created for 2 x the step's tokens (both k-sparse lists)");
    PV_REQUIRE(n_global >= N, "n_global must be >= n_tokens");
    const SaeWs ws = sae_carve(d);
    PV_REQUIRE(workspace_bytes >= ws.total && ((uintptr_t)workspace & 255) == 0, "workspace too small / misaligned");
    const GatedTopkWs gw = gated_topk_carve(d, N);
    PV_REQUIRE(scratch_bytes >= gw.total && ((uintptr_t)scratch & 255) == 0, "scratch too small / misaligned (pv_sae_gated_topk_scratch_bytes)");
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* wsb = (unsigned char*)workspace;
    unsigned char* gb = (unsigned char*)scratch;
    plan->live_offs = nullptr;
    plan->renorm_pending = false;
    const int F = d.d_sae, D = d.d_in, k = d.k, n_half = N * k;
    const float ng = (float)n_global;
    int rc = pv_sae_renorm_decoder(plan, st, stream_);
    if (rc) return rc;
    int32_t* idx2 = out->topk_idx;
    float* val2 = out->topk_val;
    uint32_t* wpos2 = (uint32_t*)(wsb + ws.wpos);
    uint32_t* cnt_g = (uint32_t*)(wsb + ws.cnt);
    uint32_t* cnt_m = (uint32_t*)(gb + gw.cnt_m);
    uint32_t* cnt2 = (uint32_t*)(gb + gw.cnt2);
    float* dM = (float*)(gb + gw.dM);
    float* dh2 = (float*)(wsb + ws.dh);
    float* dY2 = (float*)(wsb + ws.dY);
    float* sae_in = (float*)(wsb + ws.sae_in);
    float* auxpart = (float*)(gb + gw.auxpart);
    float* l0part = (float*)(gb + gw.l0part);
    const dim3 block(256);
    // the magnitude path's operands + its top-k (list 1)
    const bool shadows = st->W_enc16T && st->enc_colsq;
    hipLaunchKernelGGL(gated_scale_rows_kernel, dim3((F + 3) / 4), block, 0, stream, (const float*)st->W_encT, (const float*)t.r_mag,
                       (float*)(gb + gw.wmagT), (_Float16*)(gb + gw.wmag16), (float*)(gb + gw.colsq), F, D);
    PV_LAUNCH_CHECK("gated_scale_rows_kernel");
    pv_sae_state sm = *st;
    sm.W_encT = (float*)(gb + gw.wmagT);
    sm.W_enc16T = shadows ? (uint16_t*)(gb + gw.wmag16) : nullptr;
    sm.enc_colsq = shadows ? (float*)(gb + gw.colsq) : nullptr;
    sm.b_enc = t.b_mag;
    rc = sae_encode_topk(plan, &sm, x, N, batch_mean, idx2, val2, true, wsb, ws, stream, cnt_m, wpos2);
    if (rc) return rc;
#define CALL(V)                                                                                                                   \
    hipLaunchKernelGGL((gated_topk_mask_kernel<V>), dim3((N + 3) / 4), block, 0, stream, (const float*)sae_in, (const float*)st->W_encT, \
                       (const float*)t.b_gate, (const int32_t*)idx2, val2, l0part, (float*)(gb + gw.mu0), (float*)(gb + gw.one), N, D, k)
    V4_DISPATCH(D, CALL);
#undef CALL
    PV_LAUNCH_CHECK("gated_topk_mask_kernel");
    // the gate path's top-k (list 2)
    pv_sae_state sg = *st;
    sg.b_enc = t.b_gate;
    rc = sae_encode_topk(plan, &sg, x, N, batch_mean, idx2 + n_half, val2 + n_half, true, wsb, ws, stream, cnt_g, wpos2 + n_half, true);
    if (rc) return rc;
    PV_HIP_CHECK(hipMemcpyAsync(sae_in + (size_t)N * D, sae_in, (size_t)N * D * 4, hipMemcpyDeviceToDevice, stream));
    {
        ProfScope prof(PV_PROF_SAE_BWD, stream, 8.0 * n_half * (double)D * 2.0, 0.0);
        const dim3 grid((N + 3) / 4);
#define CALL(V)                                                                                                                   \
    hipLaunchKernelGGL((sae_decode_kernel<V>), grid, block, 0, stream, x, (const float*)st->W_dec, (const float*)st->b_dec,           \
                       (const int32_t*)idx2, (const float*)val2, (const float*)(wsb + ws.mu), (const float*)(wsb + ws.sd),           \
                       (const float*)(wsb + ws.norm), out->sae_out, dY2, dM, (float*)(wsb + ws.loss_part), N, D, k,                  \
                       2.0f / (ng * (float)D), 1, (const float*)nullptr);                                                             \
    hipLaunchKernelGGL((sae_decode_kernel<V>), grid, block, 0, stream, (const float*)sae_in, (const float*)st->W_dec,                  \
                       (const float*)st->b_dec, (const int32_t*)(idx2 + n_half), (const float*)(val2 + n_half),                       \
                       (const float*)(gb + gw.mu0), (const float*)(gb + gw.one), (const float*)(gb + gw.one), (float*)nullptr,        \
                       dY2 + (size_t)N * D, dh2 + n_half, auxpart, N, D, k, 2.0f / ng, 1, (const float*)nullptr)
        V4_DISPATCH(D, CALL);
#undef CALL
        PV_LAUNCH_CHECK("sae_decode_kernel (top-k gated)");
        hipLaunchKernelGGL(gated_topk_pairs_kernel, dim3((n_half + F + 255) / 256), block, 0, stream, (const int32_t*)idx2, wpos2,
                           (const uint32_t*)cnt_m, (const uint32_t*)cnt_g, cnt2, (const float*)dM, (const float*)t.r_mag, dh2, n_half, F);
        PV_LAUNCH_CHECK("gated_topk_pairs_kernel");
        SaeTail tb;
        tb.dh = dh2; tb.chunk_start = (uint32_t*)(wsb + ws.cursor); tb.wpos = wpos2; tb.seg_range = (uint32_t*)(wsb + ws.seg_range);
        tb.seg_rows = (float*)(wsb + ws.seg_rows); tb.seg_b = (float*)(wsb + ws.seg_b); tb.pairs = (int32_t*)(wsb + ws.pairs);
        tb.max_segs = (int)sae_max_segs((size_t)2 * n_half);
        rc = sae_csr_backward(plan, st, 2 * N, k, idx2, val2, dh2, dY2, sae_in, out->scalars, nullptr, 0, false, tb, wsb, ws, nullptr, 0.f,
                              false, nullptr, stream, nullptr, nullptr, cnt2);
        if (rc) return rc;
        hipLaunchKernelGGL(gated_topk_feat_kernel, dim3((F + 3) / 4), block, 0, stream, (const uint32_t*)(wsb + ws.offs),
                           (const int32_t*)tb.pairs, (const float*)val2, (const float*)dM, (const float*)dh2, (const float*)t.b_mag, n_half,
                           F, t.gb_gate, t.gb_mag, t.gr_mag, out->fire_count, st->act_freq_scores, st->n_fwd_since_fired, update_stats);
        hipLaunchKernelGGL(gated_topk_scalars_kernel, dim3(1), block, 0, stream, (const float*)(wsb + ws.loss_part), (const float*)auxpart,
                           (const float*)l0part, N, 1.0f / (ng * (float)D), 1.0f / ng, 1.0f / (float)N, out->scalars);
        PV_LAUNCH_CHECK("top-k gated kernels");
        // gb_dec = colsum(dY) + 2 colsum(dVia) - W_enc colsum(dP) (as pv_sae_gated_step); b_enc takes no part
        rc = sae_gbdec(d, st, dY2, N, wsb, ws, stream);
        if (rc) return rc;
        rc = sae_colsum(dY2 + (size_t)N * D, N, D, (float*)(gb + gw.tmpd), 2.0f, (float*)(gb + gw.cspart), stream);
        if (rc) return rc;
        hipLaunchKernelGGL(axpy_kernel, dim3((D + 255) / 256), block, 0, stream, st->gb_dec, (const float*)(gb + gw.tmpd), D);
        PV_LAUNCH_CHECK("axpy_kernel");
        PV_HIP_CHECK(hipMemsetAsync(st->gb_enc, 0, (size_t)F * 4, stream));
    }
    return PV_OK;
}

int sae_dec_inv_norm(const pv_sae_desc& d, const pv_sae_state* st, hipStream_t stream) {
    hipLaunchKernelGGL(dec_inv_norm_kernel, dim3((d.d_sae + 15) / 16), dim3(256), 0, stream, (const float*)st->W_dec, st->dec_inv_norm,
                       d.d_sae, d.d_in);
    PV_LAUNCH_CHECK("dec_inv_norm_kernel");
    return PV_OK;
}

extern "C" int pv_sae_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t N, const float* batch_mean,
                           int32_t n_global, int32_t flags, pv_sae_out* out, void* workspace,
                           size_t workspace_bytes, void* stream_) {
    const int update_stats = (flags & PV_SAE_UPDATE_STATS) ? 1 : 0;
    const bool renorm = (flags & PV_SAE_RENORM_DECODER) != 0;
    const bool sparse = (flags & PV_SAE_SPARSE_GRADS) != 0;
    const bool fused_sq = (flags & PV_SAE_FUSED_SQNORM) != 0;
    PV_REQUIRE(plan && st && x && out && workspace, "null argument");
    PV_REQUIRE(out->topk_idx && out->topk_val && out->scalars, "pv_sae_out buffers");
    PV_REQUIRE(!fused_sq || (st->tc.b_dec_out == nullptr && plan->d.d_in <= 4096),
               "PV_SAE_FUSED_SQNORM: autoencoder states of d_in <= 4096 (a transcoder's clip norm has more terms: pv_sae_grad_sqnorm_step)");
    PV_REQUIRE(st->W_dec && st->b_enc && st->b_dec && st->gW_enc && st->gW_dec && st->gb_enc && st->gb_dec, "state");
    PV_REQUIRE(st->W_encT, "pv_sae_step needs the transposed encoder copy (pv_sae_state.W_encT, see pv_sae_sync_shadows)");
    PV_REQUIRE(!update_stats || (st->act_freq_scores && st->n_fwd_since_fired), "stats buffers");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(N >= 1 && N <= d.max_tokens, "n_tokens exceeds plan max_tokens");
    PV_REQUIRE(n_global >= N, "n_global must be >= n_tokens");
    const SaeWs ws = sae_carve(d);
    PV_REQUIRE(workspace_bytes >= ws.total, "workspace too small");
    PV_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace alignment");
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* wsb = (unsigned char*)workspace;
    const int k = d.k, n_pairs = N * k;

    // set_decoder_norm_to_unit_norm (train_sae.py:307) as part of the step: instead of a read-modify-write pass over
    // W_dec (151 MB), only the inverse row norms are computed (75 MB read); decode / dh use W_dec[j] * inv_norm[j] on the
    // fly and pv_sae_apply writes the normalised + updated rows -- W_dec holds exactly what the reference's holds at every
    // step boundary (un-normalised after the optimizer step, as there).
    PV_REQUIRE(!renorm || st->dec_inv_norm, "PV_SAE_RENORM_DECODER needs pv_sae_state.dec_inv_norm");
    const float* inv_norm = nullptr;
    plan->renorm_pending = renorm;
    if (renorm && !(flags & PV_SAE_INV_NORM_VALID)) {
        hipLaunchKernelGGL(dec_inv_norm_kernel, dim3((d.d_sae + 15) / 16), dim3(256), 0, stream, (const float*)st->W_dec,
                           st->dec_inv_norm, d.d_sae, d.d_in);
    }
    if (renorm) inv_norm = st->dec_inv_norm;
    PV_REQUIRE(!sae_is_gated(st), "pv_sae_step does not serve a gated state: pv_sae_gated_step");
    // transcoder (pv_sae_state.tc): the loss is taken against tc.target, the decoder adds b_dec_out and the skip term
    const bool tc = sae_is_tc(st);
    if (tc) {
        // (token-sharded form: batch_mean = the TARGET's global mean -- the x-side normaliser sae_prep derives from it is overwritten by
        // sae_tc_target_norm below, as in the dense step)
        const int rq = sae_tc_require(d, st, N);
        if (rq) return rq;
    }
    int rc = sae_encode_topk(plan, st, x, N, batch_mean, out->topk_idx, out->topk_val, true, wsb, ws, stream);
    if (rc) return rc;
    const float* skip = nullptr;
    if (tc) {
        rc = sae_tc_target_norm(d, st, batch_mean, N, wsb, ws, stream);
        if (rc) return rc;
        rc = sae_tc_skip_forward(d, st, x, N, &skip, stream);
        if (rc) return rc;
    }
    const float* y = tc ? st->tc.target : x;
    const float* bdo = tc ? (const float*)st->tc.b_dec_out : (const float*)st->b_dec;

    SaeTail tb;
    tb.dh = (float*)(wsb + ws.dh); tb.chunk_start = (uint32_t*)(wsb + ws.cursor); tb.wpos = (uint32_t*)(wsb + ws.wpos);
    tb.seg_range = (uint32_t*)(wsb + ws.seg_range); tb.seg_rows = (float*)(wsb + ws.seg_rows); tb.seg_b = (float*)(wsb + ws.seg_b);
    tb.pairs = (int32_t*)(wsb + ws.pairs); tb.max_segs = (int)sae_max_segs((size_t)n_pairs);
    rc = sae_sparse_tail(plan, st, x, N, n_global, k, out->topk_idx, out->topk_val, out->sae_out, out->scalars, out->fire_count,
                         update_stats, sparse, inv_norm, tb, wsb, ws, y, bdo, skip, tc, 0.0f, nullptr, nullptr, stream, true,
                         (fused_sq && g_pv_tuning.sae_fold) ? out->scalars : (float*)nullptr);
    if (rc) return rc;
    if (fused_sq && !g_pv_tuning.sae_fold) {                          // (the A/B of the folds: the same sum as a launch of its own)
        hipLaunchKernelGGL(sqnorm_blocks_kernel, dim3(1), dim3(1024), 0, stream, (const float*)st->gb_dec, d.d_in,
                           (const float*)(wsb + ws.rowsq), d.d_sae, out->scalars);
        PV_LAUNCH_CHECK("sqnorm_blocks_kernel");
    }
    return PV_OK;
}

// ------------------------------------------------------------------------------------------------
// Feature-parallel step: this rank's engine covers a SHARD of the features; the top-k is taken over all ranks'
// candidates by the caller, the local candidates that did not make it arrive here with value 0 -- and a pair with
// value <= 0 is a hole everywhere (no decode contribution, no CSR entry, dh gated off), like the reference's ReLU behind
// its top-k.  pv_sae_step cut at the reconstruction:
//   pv_sae_encode_topk (this rank's k candidates per token)  ->  [all-gather of candidate values, global top-k: caller]
//   pv_sae_tp_partial  (partial reconstruction of the kept pairs)  ->  [all-reduce of the partials: caller]
//   pv_sae_tp_finish   (LN-out, loss, dY, dh, CSR, sparse backward, statistics for the shard's features)
//   -> [gb_dec / clip norm / l0 all-reduces: caller] -> pv_sae_apply
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sae_recount_kernel(const int32_t* __restrict__ idx, const float* __restrict__ val,
                                                          uint32_t* __restrict__ cnt, uint32_t* __restrict__ wpos, int n_pairs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    wpos[i] = val[i] > 0.f ? atomicAdd(&cnt[idx[i]], 1u) : 0xffffffffu;
}

extern "C" int pv_sae_tp_partial(pv_sae_plan* plan, pv_sae_state* st, const int32_t* topk_idx, const float* topk_val, int32_t N,
                                 int32_t flags, float* partial, void* stream_) {
    PV_REQUIRE(!st || !sae_is_gated(st), "this entry point does not serve a gated state (pv_sae_state.gt)");
    PV_REQUIRE(!st || !sae_is_tc(st), "pv_sae_tp_partial: not available for a transcoder state (pv_sae_state.tc)");
    PV_REQUIRE(plan && st && topk_idx && topk_val && partial && st->W_dec, "null argument");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(N >= 1 && N <= d.max_tokens, "n_tokens exceeds plan max_tokens");
    hipStream_t stream = (hipStream_t)stream_;
    const bool renorm = (flags & PV_SAE_RENORM_DECODER) != 0;
    PV_REQUIRE(!renorm || st->dec_inv_norm, "PV_SAE_RENORM_DECODER needs pv_sae_state.dec_inv_norm");
    plan->renorm_pending = renorm;
    plan->live_offs = nullptr;
    if (renorm && !(flags & PV_SAE_INV_NORM_VALID))
        hipLaunchKernelGGL(dec_inv_norm_kernel, dim3((d.d_sae + 15) / 16), dim3(256), 0, stream, (const float*)st->W_dec,
                           st->dec_inv_norm, d.d_sae, d.d_in);
    const float* inv_norm = renorm ? (const float*)st->dec_inv_norm : nullptr;
    const dim3 grid((N + 3) / 4), block(256);
#define CALL(D)                                                                                                          \
    hipLaunchKernelGGL((sae_decode_kernel<D, 1>), grid, block, 0, stream, (const float*)nullptr, (const float*)st->W_dec, \
                       (const float*)nullptr, topk_idx, topk_val, (const float*)nullptr, (const float*)nullptr,          \
                       (const float*)nullptr, partial, (float*)nullptr, (float*)nullptr, (float*)nullptr, N, d.d_in, d.k, \
                       0.f, 0, inv_norm, (const float*)nullptr)
    V4_DISPATCH(d.d_in, CALL);
#undef CALL
    PV_LAUNCH_CHECK("sae_decode_kernel (partial)");
    return PV_OK;
}

// `workspace` must be the one the preceding pv_sae_encode_topk ran in, on the same x (it holds the LN statistics, the
// loss normaliser and sae_in); pre_sum [N, d_in] = the reconstruction summed over the ranks (without b_dec).  Gradients of
// the shard's rows are WRITTEN into st->g*; st->gb_dec receives colsum(dY) - W_enc[:, shard] gb_enc[shard]: the caller
// combines the ranks' encoder terms.  scalars[0..1] = loss (over the full batch, replicated), scalars[2] = this rank's
// kept pairs per token (the ranks' values add up to l0).
extern "C" int pv_sae_tp_finish(pv_sae_plan* plan, pv_sae_state* st, const float* x, const float* pre_sum, const int32_t* topk_idx,
                                const float* topk_val, int32_t N, int32_t n_global, int32_t flags, pv_sae_out* out,
                                void* workspace, size_t workspace_bytes, void* stream_) {
    PV_REQUIRE(!st || !sae_is_gated(st), "this entry point does not serve a gated state (pv_sae_state.gt)");
    PV_REQUIRE(!st || !sae_is_tc(st), "pv_sae_tp_finish: not available for a transcoder state (pv_sae_state.tc)");
    const int update_stats = (flags & PV_SAE_UPDATE_STATS) ? 1 : 0;
    PV_REQUIRE(plan && st && x && pre_sum && topk_idx && topk_val && out && workspace, "null argument");
    PV_REQUIRE(out->scalars, "pv_sae_out.scalars");
    PV_REQUIRE(st->W_dec && st->b_enc && st->b_dec && st->gW_enc && st->gW_dec && st->gb_enc && st->gb_dec, "state");
    PV_REQUIRE(st->W_encT, "the transposed encoder copy (pv_sae_state.W_encT) is required");
    PV_REQUIRE(!update_stats || (st->act_freq_scores && st->n_fwd_since_fired), "stats buffers");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(N >= 1 && N <= d.max_tokens, "n_tokens exceeds plan max_tokens");
    PV_REQUIRE(n_global >= N, "n_global must be >= n_tokens");
    const SaeWs ws = sae_carve(d);
    PV_REQUIRE(workspace_bytes >= ws.total, "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* wsb = (unsigned char*)workspace;
    const int k = d.k, n_pairs = N * k;
    const float* inv_norm = plan->renorm_pending ? (const float*)st->dec_inv_norm : nullptr;       // as pv_sae_tp_partial left it
    plan->live_offs = nullptr;
    float* dY = (float*)(wsb + ws.dY);
    float* dh = (float*)(wsb + ws.dh);
    float* sae_in = (float*)(wsb + ws.sae_in);
    uint32_t* cnt = (uint32_t*)(wsb + ws.cnt);
    uint32_t* wposp = (uint32_t*)(wsb + ws.wpos);
    const dim3 block(256);
    // the pairs that survived the global top-k: counts and within-list positions afresh
    PV_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)d.d_sae * 4, stream));
    hipLaunchKernelGGL(sae_recount_kernel, dim3((n_pairs + 255) / 256), block, 0, stream, topk_idx, topk_val, cnt, wposp, n_pairs);
    if (g_pv_tuning.sae_fold) {
        // the folded form (as sae_sparse_tail): the CSR scan as one more workgroup of the decode launch, the loss and dY's
        // partial column sums as roles of the post + fill launch, one launch for both list sorts -- 11 launches instead of 16
        const float grad_scale = 2.0f / ((float)n_global * (float)d.d_in);
        ScanRole scan = {};
        scan.cnt = (const uint32_t*)cnt; scan.offs = (uint32_t*)(wsb + ws.offs); scan.n_long = (uint32_t*)(wsb + ws.n_long);
        scan.d_sae = d.d_sae; scan.scalars = out->scalars; scan.inv_tokens = 1.0f / (float)N;
        const dim3 grid((N + 3) / 4 + 1);
#define CALL(D)                                                                                                        \
    hipLaunchKernelGGL((sae_decode_kernel<D, 2>), grid, block, 0, stream, x, (const float*)st->W_dec, (const float*)st->b_dec, \
                       topk_idx, topk_val, (const float*)(wsb + ws.mu), (const float*)(wsb + ws.sd),                    \
                       (const float*)(wsb + ws.norm), out->sae_out, dY, dh, (float*)(wsb + ws.loss_part), N, d.d_in, k, \
                       grad_scale, 1, inv_norm, pre_sum, (const float*)nullptr, 0.f, (const uint32_t*)nullptr,          \
                       (const uint32_t*)nullptr, scan)
        V4_DISPATCH(d.d_in, CALL);
#undef CALL
        PV_LAUNCH_CHECK("sae_decode_kernel (finish)");
        SaeTail tb;
        tb.dh = dh; tb.chunk_start = (uint32_t*)(wsb + ws.cursor); tb.wpos = wposp;
        tb.seg_range = (uint32_t*)(wsb + ws.seg_range); tb.seg_rows = (float*)(wsb + ws.seg_rows); tb.seg_b = (float*)(wsb + ws.seg_b);
        tb.pairs = (int32_t*)(wsb + ws.pairs); tb.max_segs = (int)sae_max_segs((size_t)n_pairs);
        int rc = sae_csr_build(plan, st, N, k, topk_idx, dY, out->scalars, out->fire_count, update_stats, false, tb, wsb, ws,
                               (const float*)(wsb + ws.loss_part), 1.0f / ((float)n_global * (float)d.d_in), true, nullptr, stream, nullptr,
                               true);
        if (rc) return rc;
        rc = sae_csr_grads(plan, st, N, k, topk_idx, topk_val, dh, dY, sae_in, false, tb, wsb, ws, nullptr, stream, nullptr, nullptr);
        if (rc) return rc;
        const int nblk = (N + CS_ROWS - 1) / CS_ROWS, ngb = (d.d_sae + GBD_ROWS - 1) / GBD_ROWS;
        float* colpart = (float*)(wsb + ws.colpart);
        hipLaunchKernelGGL(sae_gbdec_partial_kernel, dim3(ngb), dim3(256), 0, stream, (const float*)st->W_encT, (const float*)st->gb_enc,
                           colpart + (size_t)nblk * d.d_in, d.d_sae, d.d_in);
        // PV_SAE_TP_ENC_TERM_ONLY: every rank holds the same dY; only one of them contributes its column sum to the all-reduce
        if (flags & PV_SAE_TP_ENC_TERM_ONLY)
            hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream,
                               (const float*)(colpart + (size_t)nblk * d.d_in), st->gb_dec, ngb, d.d_in, 1.0f);
        else
            hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream, (const float*)colpart, st->gb_dec,
                               nblk + ngb, d.d_in, 1.0f);
        PV_LAUNCH_CHECK("sae bias-grad kernels");
        return PV_OK;
    }
    {
        const float grad_scale = 2.0f / ((float)n_global * (float)d.d_in);
        const dim3 grid((N + 3) / 4);
#define CALL(D)                                                                                                        \
    hipLaunchKernelGGL((sae_decode_kernel<D, 2>), grid, block, 0, stream, x, (const float*)st->W_dec, (const float*)st->b_dec, \
                       topk_idx, topk_val, (const float*)(wsb + ws.mu), (const float*)(wsb + ws.sd),                    \
                       (const float*)(wsb + ws.norm), out->sae_out, dY, dh, (float*)(wsb + ws.loss_part), N, d.d_in, k, \
                       grad_scale, 1, inv_norm, pre_sum)
        V4_DISPATCH(d.d_in, CALL);
#undef CALL
        PV_LAUNCH_CHECK("sae_decode_kernel (finish)");
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, stream, (const float*)(wsb + ws.loss_part), out->scalars, N,
                           1.0f / ((float)n_global * (float)d.d_in), 1, 0);
        uint32_t* offs = (uint32_t*)(wsb + ws.offs);
        uint32_t* chunk_start = (uint32_t*)(wsb + ws.cursor);
        int32_t* pairs = (int32_t*)(wsb + ws.pairs);
        int32_t* long_list = (int32_t*)(wsb + ws.long_list);
        uint32_t* n_long = (uint32_t*)(wsb + ws.n_long);
        const int max_chunks = (n_pairs + BWD_CH - 1) / BWD_CH;
        float* rowsq = (float*)(wsb + ws.rowsq);
        const int max_segs = (int)sae_max_segs((size_t)n_pairs);
        uint32_t* seg_range = (uint32_t*)(wsb + ws.seg_range);
        float* seg_rows = (float*)(wsb + ws.seg_rows);
        float* seg_b = (float*)(wsb + ws.seg_b);
        hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t*)cnt, offs, n_long, d.d_sae,
                           out->scalars, 1.0f / (float)N);
        hipLaunchKernelGGL(csr_post_kernel, dim3((d.d_sae + 255) / 256), block, 0, stream, (const uint32_t*)offs, chunk_start,
                           max_chunks, long_list, n_long, seg_range, max_segs, st->act_freq_scores, st->n_fwd_since_fired,
                           out->fire_count, d.d_sae, update_stats, (float*)nullptr, (float*)nullptr, sae_long_ranged(N) ? 1 : 0);
        hipLaunchKernelGGL(csr_fill_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, stream, topk_idx,
                           (const uint32_t*)wposp, (const uint32_t*)offs, pairs, n_pairs);
        hipLaunchKernelGGL(csr_sort_short_kernel, dim3((d.d_sae + 3) / 4), dim3(256), 0, stream, (const uint32_t*)offs, pairs, d.d_sae);
        const int ranged = sae_long_ranged(N) ? 1 : 0;
        if (ranged) {
            const int rcs = launch_long_sort(long_list, n_long, (const uint32_t*)offs, pairs, seg_range, k, N, max_segs, stream);
            if (rcs) return rcs;
        }
        PV_LAUNCH_CHECK("csr kernels");
        const dim3 gridf((max_chunks + 3) / 4);
#define CALL(D)                                                                                                        \
    hipLaunchKernelGGL((sae_zero_empty_kernel<D>), dim3((d.d_sae + 3) / 4), block, 0, stream, (const uint32_t*)offs, st->gW_dec, \
                       st->gW_enc, st->gb_enc, rowsq, d.d_sae, d.d_in);                                                   \
    hipLaunchKernelGGL((sae_backward_kernel<D>), gridf, block, 0, stream, (const uint32_t*)offs, (const uint32_t*)chunk_start, \
                       (const int32_t*)pairs, topk_idx, topk_val, (const float*)dh,                                       \
                       (const float*)dY, (const float*)sae_in, st->gW_dec, st->gW_enc, st->gb_enc, rowsq, d.d_in, k, max_chunks); \
    hipLaunchKernelGGL((sae_backward_seg_kernel<D>), dim3(ranged ? 2048 : 1024), dim3(ranged ? 512 : 256), 0, stream, (const uint32_t*)offs, \
                       (const uint32_t*)seg_range, (const uint32_t*)n_long, (const int32_t*)pairs, topk_idx,           \
                       topk_val, (const float*)dh, (const float*)dY, (const float*)sae_in, seg_rows, seg_b,            \
                       d.d_in, k, max_segs, ranged);                                                                   \
    hipLaunchKernelGGL((sae_backward_long_kernel<D>), dim3(256), block, 0, stream, (const int32_t*)long_list,           \
                       (const uint32_t*)n_long, (const float*)seg_rows, (const float*)seg_b, st->gW_dec, st->gW_enc,   \
                       st->gb_enc, rowsq, d.d_in, max_segs)
        V4_DISPATCH(d.d_in, CALL);
#undef CALL
        PV_LAUNCH_CHECK("sae_backward_kernel");
        const int nblk = (N + CS_ROWS - 1) / CS_ROWS, ngb = (d.d_sae + GBD_ROWS - 1) / GBD_ROWS;
        float* colpart = (float*)(wsb + ws.colpart);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, stream, (const float*)dY, colpart, N, d.d_in);
        hipLaunchKernelGGL(sae_gbdec_partial_kernel, dim3(ngb), dim3(256), 0, stream, (const float*)st->W_encT, (const float*)st->gb_enc,
                           colpart + (size_t)nblk * d.d_in, d.d_sae, d.d_in);
        // PV_SAE_TP_ENC_TERM_ONLY: every rank holds the same dY; only one of them contributes its column sum to the all-reduce
        if (flags & PV_SAE_TP_ENC_TERM_ONLY)
            hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream,
                               (const float*)(colpart + (size_t)nblk * d.d_in), st->gb_dec, ngb, d.d_in, 1.0f);
        else
            hipLaunchKernelGGL(colsum_final_kernel, dim3((d.d_in + 63) / 64), dim3(1024), 0, stream, (const float*)colpart, st->gb_dec,
                               nblk + ngb, d.d_in, 1.0f);
        PV_LAUNCH_CHECK("sae bias-grad kernels");
    }
    return PV_OK;
}

// ---- feature-parallel glue kernels: the global top-k over the ranks' candidates, and the ONE small all-reduce bucket ----
// gathered [W][2][N][k]: rank r's piece = its k candidate values per token (float bits, sorted or not) followed by their
// LOCAL feature indices (global index = r * shard + local).  A wave per token: lane s owns this rank's candidate s and
// counts the candidates of all ranks that come before it in the order (value desc, global feature index asc) -- torch.topk's
// order on the dense row, as the oracle -- out of an LDS copy of the token's W k candidates; rank < k = kept, otherwise
// the value becomes 0 (a hole).  Replaces two argsorts over [N, W k] and a scatter on the host side.
__global__ __launch_bounds__(256) void sae_tp_merge_kernel(const int32_t* __restrict__ gathered, int W, int rank, int n_tok, int k,
                                                           int shard, float* __restrict__ val_kept) {
    __shared__ float sv[4][8 * MAXK];
    __shared__ int32_t sg[4][8 * MAXK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wv;
    if (n >= n_tok) return;                                    // (whole waves leave; no workgroup barrier below)
    const int64_t piece = 2 * (int64_t)n_tok * k;
    const int total = W * k;
    for (int c = lane; c < total; c += 64) {
        const int r = c / k, t = c - r * k;
        const int32_t* pr = gathered + r * piece + (int64_t)n * k + t;
        sv[wv][c] = __int_as_float(pr[0]);
        sg[wv][c] = r * shard + pr[(int64_t)n_tok * k];
    }
    __builtin_amdgcn_wave_barrier();                            // (a wave's LDS operations execute in order)
    if (lane < k) {
        const float v = sv[wv][rank * k + lane];
        const int32_t g = sg[wv][rank * k + lane];
        int before = 0;
        for (int c = 0; c < total; ++c) {
            const float vo = sv[wv][c];
            before += (vo > v) || (vo == v && sg[wv][c] < g);
        }
        val_kept[(int64_t)n * k + lane] = before < k ? v : 0.f;
    }
}

extern "C" int pv_sae_tp_merge(const int32_t* gathered, int32_t world, int32_t rank, int32_t N, int32_t k, int32_t shard,
                               float* val_kept, void* stream_) {
    PV_REQUIRE(gathered && val_kept, "null argument");
    PV_REQUIRE(world >= 1 && world <= 8 && rank >= 0 && rank < world, "world size must be in [1, 8]");
    PV_REQUIRE(k >= 1 && k <= MAXK && N >= 1 && shard >= k, "shape");
    hipLaunchKernelGGL(sae_tp_merge_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream_, gathered, world, rank, N, k, shard,
                       val_kept);
    PV_LAUNCH_CHECK("sae_tp_merge_kernel");
    return PV_OK;
}

// The bucket every rank all-reduces once per step: [gb_dec (d_in) | rows term of the clip norm | kept pairs per token | 2 pad |
// firing counts of ALL features (d_sae_total)].  pv_sae_tp_finish has already written this rank's gb_dec term and its
// shard's firing counts in place (the caller points st->gb_dec / out->fire_count into the bucket); pack adds the two
// scalars and zeroes the other ranks' firing counts.  After the all-reduce, unpack forms the clip norm of the GLOBAL
// gradient: the ranks' row terms + ||gb_dec||^2 of the summed gb_dec.
__global__ __launch_bounds__(1024) void sae_tp_bucket_pack_kernel(const float* __restrict__ rowsq, int d_shard, const float* __restrict__ scalars,
                                                                  float* __restrict__ bucket, int d_in, int j_lo, int d_total) {
    __shared__ float red[16];
    float s = 0.f;
    for (int j = threadIdx.x; j < d_shard; j += 1024) s += rowsq[j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    float* fire = bucket + d_in + 4;
    for (int j = threadIdx.x; j < d_total; j += 1024)
        if (j < j_lo || j >= j_lo + d_shard) fire[j] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        bucket[d_in] = t;
        bucket[d_in + 1] = scalars[2];
        bucket[d_in + 2] = 0.f;
        bucket[d_in + 3] = 0.f;
    }
}
__global__ __launch_bounds__(1024) void sae_tp_bucket_unpack_kernel(const float* __restrict__ bucket, int d_in, float* __restrict__ scalars) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < d_in; i += 1024) s += bucket[i] * bucket[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        scalars[3] = bucket[d_in] + t;
        scalars[2] = bucket[d_in + 1];
    }
}

extern "C" int pv_sae_tp_bucket_pack(pv_sae_plan* plan, const void* workspace, const float* scalars, float* bucket, int32_t j_lo,
                                     int32_t d_sae_total, void* stream_) {
    PV_REQUIRE(plan && workspace && scalars && bucket, "null argument");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(j_lo >= 0 && j_lo + d.d_sae <= d_sae_total, "shard range");
    const SaeWs ws = sae_carve(d);
    hipLaunchKernelGGL(sae_tp_bucket_pack_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream_,
                       (const float*)((const unsigned char*)workspace + ws.rowsq), d.d_sae, scalars, bucket, d.d_in, j_lo, d_sae_total);
    PV_LAUNCH_CHECK("sae_tp_bucket_pack_kernel");
    return PV_OK;
}
extern "C" int pv_sae_tp_bucket_unpack(pv_sae_plan* plan, const float* bucket, float* scalars, void* stream_) {
    PV_REQUIRE(plan && bucket && scalars, "null argument");
    hipLaunchKernelGGL(sae_tp_bucket_unpack_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream_, bucket, plan->d.d_in, scalars);
    PV_LAUNCH_CHECK("sae_tp_bucket_unpack_kernel");
    return PV_OK;
}

// the same number from the per-feature terms the last pv_sae_step left in its workspace
extern "C" int pv_sae_grad_sqnorm_step(pv_sae_plan* plan, const pv_sae_state* st, const void* workspace, float* scalars, void* stream_) {
    PV_REQUIRE(plan && st && workspace && scalars && st->gb_dec, "null argument");
    const pv_sae_desc& d = plan->d;
    const SaeWs ws = sae_carve(d);
    const bool tc = sae_is_tc(st);
    PV_REQUIRE(!tc || st->tc.gb_dec_out, "transcoder state");
    hipLaunchKernelGGL(sqnorm_rowsq_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream_,
                       (const float*)((const unsigned char*)workspace + ws.rowsq), d.d_sae, (const float*)st->gb_dec, d.d_in, scalars,
                       tc ? (const float*)st->tc.gb_dec_out : (const float*)nullptr, tc ? d.d_in : 0,
                       (tc && st->tc.gW_skip) ? (const float*)st->tc.gW_skip : (const float*)nullptr,
                       (tc && st->tc.gW_skip) ? d.d_in * d.d_in : 0);
    PV_LAUNCH_CHECK("sqnorm_rowsq_kernel");
    return PV_OK;
}

// sum of squares of a flat gradient buffer -> scalars[3]; `partial` = 1024 floats of scratch
extern "C" int pv_sae_grad_sqnorm(const float* flat_grads, int64_t n, float* partial, float* scalars, void* stream_) {
    PV_REQUIRE(flat_grads && partial && scalars && n > 0, "null argument");
    PV_REQUIRE(pv_aligned16(flat_grads), "gradient buffer must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(1024), dim3(256), 0, stream, flat_grads, n, partial);
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, scalars, 1024, 1.0f, 3, -1);
    PV_LAUNCH_CHECK("sqnorm kernels");
    return PV_OK;
}

// The same over the gradient rows of features [j_lo, j_hi) only (gW_enc^T, gW_dec, gb_enc slices; + gb_dec when
// include_b_dec): the local term of the sharded optimizer's clip norm (the ranks' terms are summed by one scalar all-reduce).
extern "C" int pv_sae_grad_sqnorm_rows(pv_sae_plan* plan, const pv_sae_state* st, int32_t j_lo, int32_t j_hi, int32_t include_b_dec,
                                       float* partial, float* scalars, void* stream_) {
    PV_REQUIRE(plan && st && partial && scalars && st->gW_enc && st->gW_dec && st->gb_enc && st->gb_dec, "null argument");
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(j_lo >= 0 && j_lo <= j_hi && j_hi <= d.d_sae, "feature range");
    hipStream_t stream = (hipStream_t)stream_;
    PV_HIP_CHECK(hipMemsetAsync(partial, 0, 1024 * 4, stream));
    const int64_t nrow = (int64_t)(j_hi - j_lo) * d.d_in;
    if (nrow > 0) {
        PV_REQUIRE(((int64_t)j_lo * d.d_in) % 4 == 0, "slice must start on a 16-byte boundary");
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(384), dim3(256), 0, stream, (const float*)st->gW_enc + (int64_t)j_lo * d.d_in, nrow, partial);
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(384), dim3(256), 0, stream, (const float*)st->gW_dec + (int64_t)j_lo * d.d_in, nrow, partial + 384);
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(8), dim3(256), 0, stream, (const float*)st->gb_enc + j_lo, (int64_t)(j_hi - j_lo), partial + 768);
    }
    if (include_b_dec)
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(8), dim3(256), 0, stream, (const float*)st->gb_dec, (int64_t)d.d_in, partial + 776);
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, scalars, 1024, 1.0f, 3, -1);
    PV_LAUNCH_CHECK("sqnorm kernels");
    return PV_OK;
}

extern "C" int pv_sae_apply(pv_sae_plan* plan, pv_sae_state* st, const float* scalars, float max_grad_norm, float lr,
                            int32_t step, int32_t j_lo, int32_t j_hi, void* stream_) {
    PV_REQUIRE(plan && st && scalars && step >= 1, "null argument / step must be >= 1");
    PV_REQUIRE(st->mW_enc && st->mW_dec && st->mb_enc && st->mb_dec && st->vW_enc && st->vW_dec && st->vb_enc && st->vb_dec, "Adam state");
    PV_REQUIRE(st->W_encT && st->W_enc16T && st->enc_colsq, "encoder shadows");
    hipStream_t stream = (hipStream_t)stream_;
    const pv_sae_desc& d = plan->d;
    PV_REQUIRE(j_lo >= 0 && j_lo <= j_hi && j_hi <= d.d_sae, "feature range");
    AdamC c;
    c.lr = lr; c.b1 = 0.9f; c.b2 = 0.999f; c.eps = 1e-8f; c.max_norm = max_grad_norm;
    c.bc1 = (float)(1.0 - pow(0.9, (double)step));
    c.bc2_sqrt = (float)sqrt(1.0 - pow(0.999, (double)step));
    const int nj = j_hi - j_lo;
    // algorithmic bytes: 7 x 4 per parameter (w, g, m, v read; w, m, v written) + the two extra copies of W_enc (fp32 + fp16)
    ProfScope prof(PV_PROF_SAE_APPLY, stream, 0.0, 7.0 * 4.0 * (2.0 * d.d_in * (double)nj + nj + d.d_in) + 6.0 * d.d_in * (double)nj);
    const dim3 block(256);
    AdamVec2 v2;
    v2.W0 = st->b_enc; v2.G0 = (const float*)st->gb_enc; v2.M0 = st->mb_enc; v2.V0 = st->vb_enc; v2.lo0 = j_lo; v2.hi0 = j_hi;
    v2.nb0 = (nj + 255) / 256; v2.W1 = st->b_dec; v2.G1 = (const float*)st->gb_dec; v2.M1 = st->mb_dec; v2.V1 = st->vb_dec; v2.hi1 = d.d_in;
    bool vec2_done = false;
    if (nj > 0) {
        const float* inv_norm = plan->renorm_pending ? (const float*)st->dec_inv_norm : nullptr;
#define CALL(D) hipLaunchKernelGGL((adam_wdec_kernel<D>), dim3((nj + 3) / 4), block, 0, stream, st->W_dec, (const float*)st->gW_dec, st->mW_dec, st->vW_dec, scalars, c, j_lo, j_hi, d.d_in, inv_norm, st->dec_inv_norm, plan->live_offs)
        V4_DISPATCH(d.d_in, CALL);
#undef CALL
        if (st->W_enc) {
            hipLaunchKernelGGL((wenc_rows_kernel<0>), dim3((nj + 31) / 32), block, 0, stream, st->W_enc, st->W_encT, (_Float16*)st->W_enc16T,
                               st->enc_colsq, (const float*)st->gW_enc, st->mW_enc, st->vW_enc, scalars, c, d.d_in, d.d_sae, j_lo, j_hi,
                               plan->live_offs);
        } else {                                                  // lazy parameter layout: everything stays in the transposed domain
            // (the two bias vectors ride at the end of this launch)
            vec2_done = g_pv_tuning.sae_fold != 0;
            const int nb_rows = (nj + 3) / 4, nb_v2 = vec2_done ? v2.nb0 + (d.d_in + 255) / 256 : 0;
#define CALL(D) hipLaunchKernelGGL((adam_wenct_kernel<D>), dim3(nb_rows + nb_v2), block, 0, stream, st->W_encT, (_Float16*)st->W_enc16T, st->enc_colsq, (const float*)st->gW_enc, st->mW_enc, st->vW_enc, scalars, c, j_lo, j_hi, d.d_in, plan->live_offs, nb_rows, v2)
            V4_DISPATCH(d.d_in, CALL);
#undef CALL
        }
    }
    if (!vec2_done)     // b_enc's feature range and b_dec: one launch
        hipLaunchKernelGGL(adam_vec2_kernel, dim3(v2.nb0 + (d.d_in + 255) / 256), block, 0, stream, v2, scalars, c);
    if (sae_is_tc(st)) {                                               // transcoder: plain Adam on the decoder's own bias and the skip matrix
        const pv_sae_transcoder& t = st->tc;
        PV_REQUIRE(t.gb_dec_out && t.mb_dec_out && t.vb_dec_out && (!t.W_skip || (t.gW_skip && t.mW_skip && t.vW_skip)), "transcoder Adam state");
        hipLaunchKernelGGL(adam_vec_kernel, dim3((d.d_in + 255) / 256), block, 0, stream, t.b_dec_out, (const float*)t.gb_dec_out,
                           t.mb_dec_out, t.vb_dec_out, scalars, c, 0, d.d_in);
        if (t.W_skip)
            hipLaunchKernelGGL(adam_vec_kernel, dim3((d.d_in * d.d_in + 255) / 256), block, 0, stream, t.W_skip, (const float*)t.gW_skip,
                               t.mW_skip, t.vW_skip, scalars, c, 0, d.d_in * d.d_in);
    }
    if (sae_is_gated(st)) {                                            // gated SAE: plain Adam on b_gate, r_mag, b_mag
        const pv_sae_gated& t = st->gt;
        PV_REQUIRE(t.r_mag && t.b_mag && t.gb_gate && t.gr_mag && t.gb_mag && t.mb_gate && t.mr_mag && t.mb_mag && t.vb_gate && t.vr_mag && t.vb_mag,
                   "gated Adam state");
        PV_REQUIRE(j_lo == 0 && j_hi == d.d_sae, "gated SAE: the whole feature range");
        const dim3 gv((d.d_sae + 255) / 256);
        hipLaunchKernelGGL(adam_vec_kernel, gv, block, 0, stream, t.b_gate, (const float*)t.gb_gate, t.mb_gate, t.vb_gate, scalars, c, 0, d.d_sae);
        hipLaunchKernelGGL(adam_vec_kernel, gv, block, 0, stream, t.r_mag, (const float*)t.gr_mag, t.mr_mag, t.vr_mag, scalars, c, 0, d.d_sae);
        hipLaunchKernelGGL(adam_vec_kernel, gv, block, 0, stream, t.b_mag, (const float*)t.gb_mag, t.mb_mag, t.vb_mag, scalars, c, 0, d.d_sae);
    }
    PV_LAUNCH_CHECK("adam kernels");
    if (j_lo == 0 && j_hi == d.d_sae) plan->renorm_pending = false;      // (a sharded apply leaves the other ranks' rows to the all-gather)
    return PV_OK;
}
