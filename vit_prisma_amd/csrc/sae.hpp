// Shared definitions of the SAE step (sae.hip: prep / decode / backward / Adam; sae_enc.hip: the encoder + top-k).
#pragma once
#include "gemm.hpp"
#include "prof.hpp"

constexpr int PV_SAE_MAXK = 64;
constexpr int PV_SAE_SAMPLE_STRIDE = 16;     // pass 0 looks at every 16th feature
constexpr int PV_SAE_CAND_CAP = 1024;        // candidates of a token the select kernel can rank (more -> exact fallback row)
constexpr int PV_SAE_TILE_SLOTS_MIN = 16;    // candidate slots per (token, 256-feature tile) of the filter GEMM, at least
constexpr int PV_SAE_RESCORE_MAX = 192;      // candidates re-scored exactly per token (more -> exact fallback row)
constexpr int PV_SAE_FB_SLOTS = 32;          // workgroup columns of the fallback kernels

struct pv_sae_plan {
    pv_sae_desc d;
    bool renorm_pending = false;     // the last pv_sae_step deferred set_decoder_norm_to_unit_norm to pv_sae_apply
    const uint32_t* live_offs = nullptr;   // PV_SAE_SPARSE_GRADS: feature offsets of the last pv_sae_step (in ITS workspace), else null
};

// The pre-pass of the filtered top-k step (pv_sae_step) in three launches instead of six: the prep launch also carries the 16-row partial
// column sums of x (the batch mean) and the weight bound's workgroup (sae_prep_roles_kernel), the column sums are finished by a few
// extra workgroups of the threshold launch, and the loss normaliser ||x_n - mean||_2 (sae.py:145-147) -- which needs that mean -- is
// taken by an idle wave of the select kernel instead of by prep.  have_mean: the caller supplied the batch mean (no column sums).
struct SaePre {
    const float* x; int d_true; bool have_mean;
};

struct SaeWs {
    size_t total;
    size_t dense_colpart, dense_rowpart, dense_kpart;     // sae_dense.hip: column partials [N/64][d_sae], per-wave sums, split-K partials
    size_t dense_amax;                                    // sae_dense.hip: PV_SAE_AMAX_TENSORS x 256 partial maxima (operand scales of the split-fp16 GEMMs)
    size_t dense_lp_part, dense_lp_tok, dense_lp_loss;    // sae_dense.hip, lp_norm > 1: [N][d_sae / 64] sums of f^p, [N] gradient factors, [N] norms
    size_t hidden, sae_in, dY, mu, sd, norm, dh, loss_part, cnt, offs, cursor, wpos, long_list, n_long, seg_range, seg_rows, seg_b, pairs, colpart, colsum, batch_mean, sqpart, rowsq;
    // fast encoder (sae_enc.hip)
    size_t x16, xnorm, sample, thr, sq, band, cand_cnt, cand, fb_list, fb_count, wmax;
    int sq_blocks;
};
SaeWs sae_carve(const pv_sae_desc& d);

// out[c] = scale * sum_blk partial[blk][c] for the 16 columns of workgroup `bid`, on 256 threads in the summation order of the 1024-thread
// colsum_final_kernel (sae.hip: 16 partial streams per column, then their sum in stream order): thread (part, column) walks stream
// `part` of its column -- 16 independent loads -- as a thread of that kernel does; a workgroup just spans 16 columns instead of 64
// (a first form kept 64 columns per workgroup and let every thread play four parts one after the other: 24 us, PMC-measured wave
// lifetimes, for a 786 KB sum).  (d + 15) / 16 workgroups.
#ifdef __HIPCC__
__device__ __forceinline__ void colsum_final_body_256(int bid, const float* __restrict__ partial, float* __restrict__ out, int nblk, int d,
                                                      float scale) {
    __shared__ float cf_red[16][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = bid * 16 + cl;
    float s = 0.f;
    if (c < d) {
#pragma unroll 8
        for (int b = part; b < nblk; b += 16) s += partial[(int64_t)b * d + c];
    }
    cf_red[part][cl] = s;
    __syncthreads();
    if (part == 0 && c < d) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += cf_red[q][cl];          // fixed order
        out[c] = t * scale;
    }
}
#endif

// The exact top-k of one row of pre-activations (the algorithm: sae.hip, above sae_topk_kernel), shared by sae_topk_kernel (sae.hip) and
// by the select kernel's inline exact path (sae_enc.hip)
constexpr int PV_TOPK_CAP = 1024;            // candidate list capacity of the streaming pass (more: the radix-select passes)
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t pv_f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// LDS of the caller: hist[256], cand_key[PV_TOPK_CAP], cand_idx[PV_TOPK_CAP] (sae_select_kernel lends the arrays of its candidate
// pass: the same sizes); the scalars are this function's own
__device__ __forceinline__ void sae_topk_row_lds(const float* __restrict__ hidden, int32_t* __restrict__ idx_out, float* __restrict__ val_out,
                                                 int d_sae, int k, int64_t row, uint32_t* __restrict__ feat_cnt, uint32_t* __restrict__ wpos,
                                                 uint32_t* hist, uint32_t* cand_key, int32_t* cand_idx) {
    constexpr int TOPK_CAP = PV_TOPK_CAP;
    __shared__ uint32_t sh_T0, sh_ncand, sh_prefix, sh_k, sh_wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* h = hidden + row * d_sae;
    const int nvec = d_sae >> 2;                     // d_sae % 4 == 0 (checked at plan creation)

    uint32_t lmax = 0;
    for (int v = tid; v < nvec; v += 256) {
        const float4 x = *reinterpret_cast<const float4*>(h + 4 * v);
        lmax = max(max(lmax, pv_f2ord(x.x)), max(pv_f2ord(x.y), max(pv_f2ord(x.z), pv_f2ord(x.w))));
    }
    hist[tid] = lmax;
    if (tid == 0) { sh_T0 = 0u; sh_ncand = 0u; }
    __syncthreads();
    if (k <= 256) {
        uint32_t rank = 0;
        for (int u = 0; u < 256; ++u) {
            const uint32_t o = hist[u];               // same address for all lanes: LDS broadcast
            rank += (o > lmax) || (o == lmax && u < tid);
        }
        if (rank == (uint32_t)(k - 1)) sh_T0 = lmax;
    }
    __syncthreads();
    const uint32_t T0 = sh_T0;
    for (int v = tid; v < nvec; v += 256) {
        const float4 x = *reinterpret_cast<const float4*>(h + 4 * v);
        const uint32_t kx[4] = {pv_f2ord(x.x), pv_f2ord(x.y), pv_f2ord(x.z), pv_f2ord(x.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (kx[e] >= T0) {
                const uint32_t pos = atomicAdd(&sh_ncand, 1u);
                if (pos < TOPK_CAP) { cand_key[pos] = kx[e]; cand_idx[pos] = 4 * v + e; }
            }
        }
    }
    __syncthreads();
    const uint32_t ncand = sh_ncand;
    if (ncand <= TOPK_CAP) {
        for (uint32_t c = tid; c < ncand; c += 256) {
            const uint32_t kc = cand_key[c];
            const int32_t ic = cand_idx[c];
            uint32_t rank = 0;
            for (uint32_t o = 0; o < ncand; ++o) {
                const uint32_t ko = cand_key[o];
                rank += (ko > kc) || (ko == kc && cand_idx[o] < ic);
            }
            if (rank < (uint32_t)k) {
                const float v = fmaxf(h[ic], 0.f);                // postact_fn = ReLU (sae.py:806)
                idx_out[row * k + rank] = ic;
                val_out[row * k + rank] = v;
                if (feat_cnt) wpos[row * k + rank] = v > 0.f ? atomicAdd(&feat_cnt[ic], 1u) : 0xffffffffu;
            }
        }
        return;
    }
    // ---------------------------------------------------- fallback: radix select, re-streaming the row
    uint32_t prefix = 0, kk = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        __syncthreads();
        hist[tid] = 0;
        __syncthreads();
        const uint32_t hi_mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        for (int c = tid; c < d_sae; c += 256) {
            const uint32_t key = pv_f2ord(h[c]);
            if ((key & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            int dsel = 0;
            for (int dgt = 255; dgt >= 0; --dgt) {
                if (acc + hist[dgt] >= kk) { dsel = dgt; break; }
                acc += hist[dgt];
            }
            sh_prefix = prefix | ((uint32_t)dsel << shift);
            sh_k = kk - acc;
        }
        __syncthreads();
        prefix = sh_prefix;
        kk = sh_k;
    }
    // prefix = k-th largest key; take everything above it and the first kk (lowest column) ties
    uint32_t ngt = 0, neq = 0;
    for (int c = tid; c < d_sae; c += 256) {
        const uint32_t key = pv_f2ord(h[c]);
        ngt += key > prefix;
        neq += key == prefix;
    }
    auto block_scan = [&](uint32_t cnt, uint32_t& total) -> uint32_t {
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= o) inc += a;
        }
        __syncthreads();
        if (lane == 63) sh_wcnt[wave] = inc;
        __syncthreads();
        uint32_t base = 0;
        total = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) base += sh_wcnt[w];
            total += sh_wcnt[w];
        }
        return base + inc - cnt;
    };
    uint32_t tot_gt, tot_eq;
    uint32_t pos_gt = block_scan(ngt, tot_gt);
    uint32_t pos_eq = block_scan(neq, tot_eq);
    for (int c = tid; c < d_sae; c += 256) {
        const uint32_t key = pv_f2ord(h[c]);
        int slot = -1;
        if (key > prefix) slot = (int)pos_gt++;
        else if (key == prefix) { if (pos_eq < kk) slot = (int)(tot_gt + pos_eq); pos_eq++; }
        if (slot >= 0 && slot < k) {
            const float v = fmaxf(h[c], 0.f);
            idx_out[row * k + slot] = c;
            val_out[row * k + slot] = v;
            if (feat_cnt) wpos[row * k + slot] = v > 0.f ? atomicAdd(&feat_cnt[c], 1u) : 0xffffffffu;
        }
    }
}

#endif

// number of sampled values per token that bound the k-th largest from below (order statistic taken in pass 0)
static inline int pv_sae_sample_q(int k) { return k * 3 / 8 > 8 ? k * 3 / 8 : 8; }

// Slots per (token, tile): the threshold lets ~16 q candidates per token through, spread over d_sae / 256 tiles; 4 x the
// mean + 6, rounded up to a power of two in [16, 256] (beyond that the plan takes the exact path).
static inline int pv_sae_tile_slots(const pv_sae_desc& d) {
    const int ntn = (d.d_sae + 255) / 256;
    const int want = 4 * PV_SAE_SAMPLE_STRIDE * pv_sae_sample_q(d.k) / ntn + 6;
    int s = PV_SAE_TILE_SLOTS_MIN;
    while (s < want) s *= 2;
    return s;
}

// Is the filtered (fp16 MFMA) encoder applicable to this plan?  Otherwise the exact-fp32 GEMM + streaming top-k runs.
static inline bool pv_sae_fast_ok(const pv_sae_desc& d) {
    return d.d_sae % 256 == 0 && d.d_sae >= 2048 && d.d_in % 8 == 0 && d.d_in >= 32 && d.k <= PV_SAE_MAXK &&
           pv_sae_sample_q(d.k) * PV_SAE_SAMPLE_STRIDE * 2 <= PV_SAE_CAND_CAP && pv_sae_tile_slots(d) <= 256 && !g_pv_tuning.sae_exact;
}

// sae_enc.hip: hidden_pre top-k of N tokens through the fp16 filter GEMM + exact re-scoring (see the file header).
// Requires the shadows (W_encT, W_enc16T, enc_colsq) of `st` to be current.  prep (sae.hip) has already filled sae_in,
// x16, xnorm.  Rows the filter cannot decide are recomputed exactly (hidden scratch + sae_topk_rows).
// feat_cnt / wpos (both or neither; feat_cnt is zeroed here): per-feature pair counts and each kept pair's position in
// its feature's list, for the backward's CSR.
int sae_encode_fast(const pv_sae_desc& d, const pv_sae_state* st, int N, int32_t* topk_idx, float* topk_val,
                    uint32_t* feat_cnt, uint32_t* wpos, unsigned char* wsb, const SaeWs& ws, hipStream_t stream, const SaePre* pre = nullptr);

// sae.hip: exact streaming / radix top-k of rows of `hidden`; row_list == nullptr: rows 0..n_rows-1 (one workgroup each),
// else the rows row_list[0 .. *n_list) are walked by `slots` workgroups.
void sae_topk_rows(const float* hidden, int32_t* idx_out, float* val_out, int d_sae, int k, int n_rows, const int32_t* row_list,
                   const uint32_t* n_list, int slots, uint32_t* feat_cnt, uint32_t* wpos, hipStream_t stream);

// ---- the sparse form of the ReLU + L1 step (pv_sae_relu_step, sae_dense.hip) -------------------------------------------------
constexpr int PV_SAE_RELU_SLOTS = 32;        // candidate slots per (token, 256-feature tile) of the ReLU filter
constexpr int PV_SAE_RELU_CAP_MAX = 256;     // kept activations per token the sparse form can hold, at most
// the fp16 filter + exact re-scoring applies (otherwise pv_sae_relu_step always runs the dense GEMMs)
static inline bool pv_sae_relu_sparse_ok(const pv_sae_desc& d) {
    return d.d_sae % 256 == 0 && d.d_sae >= 2048 && d.d_in % 8 == 0 && d.d_in >= 32 && !g_pv_tuning.sae_exact;
}
struct SaeTail {                             // the k-dependent buffers of sae_sparse_tail
    float* dh; uint32_t* chunk_start; uint32_t* wpos; uint32_t* seg_range; float* seg_rows; float* seg_b; int32_t* pairs;
    int max_segs;
};
struct ReluWs {
    size_t total, mode, idx, val, tok_cnt, l1part, cand_cnt, cand, dh, cursor, wpos, seg_range, seg_rows, seg_b, pairs;
    int max_segs;
};
ReluWs relu_carve(const pv_sae_desc& d, int n_tokens, int cap);
// sae_enc.hip: the positive entries of relu(sae_in W_enc + b_enc) per token (see the definition); raises *mode when a token cannot be held
int sae_encode_relu(const pv_sae_desc& d, const pv_sae_state* st, int N, int cap, int32_t* idx, float* val, uint32_t* tok_cnt,
                    float* l1part, uint32_t* cand_cnt, void* cand, uint32_t* feat_cnt, uint32_t* wpos, uint32_t* mode,
                    const float* prev_scalars, unsigned char* wsb, const SaeWs& ws, hipStream_t stream, float* l0part = nullptr,
                    float* valg = nullptr);
// sae.hip: the backward of a k-sparse step behind its decode kernel (see the definition)
int sae_csr_backward(pv_sae_plan* plan, pv_sae_state* st, int N, int k, const int32_t* topk_idx, const float* topk_val, const float* dh,
                     const float* dY, const float* sae_in, float* scalars, float* fire_count, int update_stats, bool sparse,
                     const SaeTail& tb, unsigned char* wsb, const SaeWs& ws, const float* loss_part, float loss_scale, bool cs_here,
                     const uint32_t* gate, hipStream_t stream, const float* val_b = nullptr, const float* dYb = nullptr,
                     const uint32_t* cnt_in = nullptr);
// sae.hip: the gated step in sparse form (pv_sae_gated_step_sparse; see the definition)
struct GatedSparseWs {
    ReluWs rw;
    size_t total, valg, dM, dG, l0part;
};
GatedSparseWs gated_sparse_carve(const pv_sae_desc& d, int n_tokens, int cap);
int sae_gated_sparse(pv_sae_plan* plan, pv_sae_state* st, const float* x, int N, int n_global, int cap, float l1_coefficient,
                     int update_stats, unsigned char* rwb, const GatedSparseWs& gs, pv_sae_out* out, unsigned char* wsb, const SaeWs& ws,
                     float* dYs, float* auxpart, float* pgsum, hipStream_t stream);
// sae.hip: everything of the k-sparse step behind the selection (see the definition)
int sae_sparse_tail(pv_sae_plan* plan, pv_sae_state* st, const float* x, int N, int n_global, int k, const int32_t* topk_idx,
                    const float* topk_val, float* sae_out, float* scalars, float* fire_count, int update_stats, bool sparse,
                    const float* inv_norm, const SaeTail& tb, unsigned char* wsb, const SaeWs& ws, const float* y, const float* bdo,
                    const float* skip, bool tc, float dh_add, const uint32_t* tok_cnt, const uint32_t* gate, hipStream_t stream,
                    bool bias_grads = true, float* sq_scalars = nullptr);
// sae.hip: dec_inv_norm[j] = 1 / ||W_dec[j]|| (the read-only half of set_decoder_norm_to_unit_norm)
int sae_dec_inv_norm(const pv_sae_desc& d, const pv_sae_state* st, hipStream_t stream);

// sae.hip, shared with sae_dense.hip: see the definitions
// pre (with st): the fused pre-pass of the filtered top-k step, see SaePre -- ws.norm and the finished batch mean are left to later launches
int sae_prep(const pv_sae_desc& d, const float* x, const float* b_dec, const float* batch_mean, int N, bool want_filter_inputs,
             unsigned char* wsb, const SaeWs& ws, hipStream_t stream, int d_true = 0, SaePre* pre = nullptr,
             const pv_sae_state* st = nullptr, uint32_t* feat_cnt = nullptr);
// sq_scalars: also leave sq_scalars[3] = the gradient's sum of squares (PV_SAE_FUSED_SQNORM)
int sae_gbdec(const pv_sae_desc& d, const pv_sae_state* st, const float* dY, int N, unsigned char* wsb, const SaeWs& ws,
              hipStream_t stream, bool have_colsum = false, float* sq_scalars = nullptr);
void sae_reduce_sum(const float* v, float* out, int n, float scale, int slot, int slot2, hipStream_t stream,
                    const uint32_t* gate = nullptr, uint32_t want = 0u);
int sae_colsum(const float* x, int rows, int d, float* out, float scale, float* partial, hipStream_t stream);
// transcoder (pv_sae_state.tc, sae/transcoder.py): helpers shared by the top-k step (sae.hip) and the dense step (sae_dense.hip)
static inline bool sae_is_tc(const pv_sae_state* st) { return st->tc.b_dec_out != nullptr; }
// a transcoder between hook points of different width (pv_sae_transcoder.d_in_true / d_out_true): every row is padded to the plan's
// d_in = max(d_in, d_out); what the loss is a mean over, and what LN-in's statistics run over
static inline int sae_loss_width(const pv_sae_desc& d, const pv_sae_state* st) {
    return (st && sae_is_tc(st) && st->tc.d_out_true > 0) ? st->tc.d_out_true : d.d_in;
}
static inline int sae_in_width(const pv_sae_desc& d, const pv_sae_state* st) {
    return (st && sae_is_tc(st) && st->tc.d_in_true > 0) ? st->tc.d_in_true : d.d_in;
}
constexpr int PV_SAE_SKIP_SPLITK = 8;        // token splits of gW_skip = dY^T x (36 output tiles at d_in = 768 otherwise)
int sae_tc_require(const pv_sae_desc& d, const pv_sae_state* st, int N);                                  // sae.hip: field checks
// sae.hip: the loss normaliser of the TARGET (sae.py:145-147 on y) -> ws.norm; batch_mean = mean_n(target) or NULL (computed here)
int sae_tc_target_norm(const pv_sae_desc& d, const pv_sae_state* st, const float* batch_mean, int N, unsigned char* wsb, const SaeWs& ws,
                       hipStream_t stream);
// sae.hip: gb_dec_out = colsum(dY), gb_dec = -W_enc gb_enc (the two roles of the autoencoder's b_dec are two parameters here)
int sae_tc_bias_grads(const pv_sae_desc& d, const pv_sae_state* st, const float* dY, int N, unsigned char* wsb, const SaeWs& ws,
                      hipStream_t stream);
// sae_dense.hip: skip term [N, d_in] = x @ W_skip^T at the head of tc.scratch (returns nullptr through *skip when there is no
// W_skip); gW_skip = dY^T x
int sae_tc_skip_forward(const pv_sae_desc& d, const pv_sae_state* st, const float* x, int N, const float** skip, hipStream_t stream);
int sae_tc_skip_backward(const pv_sae_desc& d, const pv_sae_state* st, const float* x, const float* dY, int N, hipStream_t stream);
// the plain ReLU + L1 form (what the sparse form of pv_sae_relu_step serves): ReLU activation, the 1-norm as the sparsity term
static inline bool sae_plain_relu(const pv_sae_desc& d) {
    return d.activation == PV_SAE_ACT_RELU && (d.lp_norm == 0.f || d.lp_norm == 1.f);
}
static inline bool sae_is_gated(const pv_sae_state* st) { return st->gt.b_gate != nullptr; }
constexpr int PV_SAE_AMAX_TENSORS = 16;      // slot arrays of operand maxima a dense step may track (SaeWs.dense_amax)
constexpr int PV_SAE_DENSE_SPLITK = 4;       // K splits of the dense decoder GEMM (M = tokens, N = d_in: too few tiles otherwise)
