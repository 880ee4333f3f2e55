/*
 * pv_native.h -- C ABI of libpvnative.so, the MI355X (gfx950) implementation of ViT-Prisma's
 * run_with_cache forward path and SAE training step.
 *
 * The reference (Prisma-Multimodal/ViT-Prisma) is 100 % Python and has NO plugin / FFI interface
 * for this path; its boundary is the Python class surface.  This header is
 * therefore the narrow C ABI *underneath* that surface.  Each entry point names the reference
 * Python code it replaces (paths relative to /root/reference/src/vit_prisma/).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  Every pointer is a DEVICE pointer on the
 *     current HIP device unless stated otherwise.
 *   - every call returns an int status (PV_OK == 0); no exception crosses the ABI;
 *     pv_last_error() gives the message of the last failure on the calling thread.
 *   - every compute call is asynchronous on the supplied hipStream_t (passed as void*; 0 = the
 *     legacy default stream).  The caller passes torch.cuda.current_stream().cuda_stream.
 *   - the library never allocates device memory: weights are borrowed from the caller's module
 *     parameters, MFMA-layout weight shadows, workspaces and every tap destination are caller-owned
 *     (sizes from the *_bytes queries).  Plans are not thread-safe; one plan per device.
 *     Two debug / measurement facilities are the exceptions and say so where they are declared: pv_debug_gemm_trace_arm
 *     (hipMalloc of a 256 KB stamp buffer on first use, never freed) and pv_prof_enable (a pool of HIP events).
 */
#ifndef PV_NATIVE_H
#define PV_NATIVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PV_OK 0
#define PV_ERR_INVALID 1      /* bad argument / unsupported configuration          */
#define PV_ERR_HIP 2          /* a HIP runtime call or kernel launch failed        */
#define PV_ERR_WORKSPACE 3    /* caller-provided buffer too small                  */

#define PV_DTYPE_F32 0
#define PV_DTYPE_BF16 1

#define PV_ACT_GELU 0         /* exact erf GELU, models/layers/mlp.py:43-44        */
#define PV_ACT_QUICK_GELU 1   /* models/activation_fns.py:19                       */
#define PV_ACT_RELU 2

/* ABI version; bumped on any struct / signature / flag change (10: PV_SAE_SPARSE_GRADS, pv_sae_tp_partial / pv_sae_tp_finish;
 * 11: pv_sae_tp_merge / pv_sae_tp_bucket_*, pv_build_id, the dense ReLU + L1 step pv_sae_dense_*; 17: pv_gemm_epilogue, the
 * gemm_persist / gemm_stagger tuning keys; 22: pv_sae_desc.activation / lp_norm,
 * normalize_layer_norm = 2 (constant_norm_rescale), the split-fp16 dense GEMMs and their tuning key dense_fp32, gemm_cus). */
#define PV_ABI_VERSION 23
int pv_abi_version(void);
/* Hash of the sources this binary was built from (sha256 over the .hip / .hpp files of vit_prisma_amd/csrc and this header, names and
 * contents, sorted; first 32 hex digits): the prebuilt library travels next to the sources, and the Python binding refuses
 * a library whose id differs from the sources it finds (vit_prisma_amd/build.py:source_id). */
const char* pv_build_id(void);
/* Copies the calling thread's last error message (NUL terminated) into buf. */
void pv_last_error(char* buf, size_t len);

/* ------------------------------------------------------------------------------------------ */
/* ViT forward with taps (HookedViT.forward, models/base_vit.py:152-217, with the caching hooks */
/* of prisma_tools/hooked_root_module.py:289-332 fused in as tap stores)                        */
/* ------------------------------------------------------------------------------------------ */

typedef struct pv_vit_desc {
    int32_t n_layers, d_model, n_heads, d_head, d_mlp;
    int32_t n_channels, patch_size, image_size;
    int32_t n_tokens;          /* patches (+1 with cls token)                                 */
    int32_t n_classes;         /* head width; ignored when has_head == 0                      */
    int32_t use_cls_token;     /* models/base_vit.py:171-175                                  */
    int32_t layer_norm_pre;    /* models/base_vit.py:183-185                                  */
    int32_t has_head;          /* return_type != "pre_logits", models/base_vit.py:210         */
    int32_t normalize_output;  /* F.normalize(dim=-1), models/base_vit.py:214-215             */
    int32_t activation;        /* PV_ACT_*                                                    */
    int32_t dtype;             /* PV_DTYPE_*: storage dtype of params, residual stream, taps  */
    float eps;                 /* LayerNorm eps, models/layers/layer_norm.py:88               */
    float attn_scale;          /* sqrt(d_head) or 1, models/layers/attention.py:96-99         */
} pv_vit_desc;

/* Parameters in the reference's own layouts (the state-dict layout produced by
 * models/weight_conversion.py:276-313, 345-429); element type = desc.dtype. */
typedef struct pv_vit_layer_weights {
    const void *ln1_w, *ln1_b;             /* [d]                                            */
    const void *W_Q, *W_K, *W_V;           /* [H, d, dh]   models/layers/attention.py:37-61  */
    const void *b_Q, *b_K, *b_V;           /* [H, dh]                                        */
    const void *W_O;                       /* [H, dh, d]   attention.py:62-69                */
    const void *b_O;                       /* [d]                                            */
    const void *ln2_w, *ln2_b;             /* [d]                                            */
    const void *W_in, *b_in;               /* [d, d_mlp], [d_mlp]  models/layers/mlp.py:24-29 */
    const void *W_out, *b_out;             /* [d_mlp, d], [d]      mlp.py:30-35              */
} pv_vit_layer_weights;

typedef struct pv_vit_weights {
    const void *cls_token;                 /* [1,1,d]  (NULL if !use_cls_token)              */
    const void *conv_w, *conv_b;           /* [d, C, p, p], [d]  patch_embedding.py:14-20    */
    const void *W_pos;                     /* [T, d]   position_embedding.py:28-30           */
    const void *ln_pre_w, *ln_pre_b;       /* [d]      (NULL if !layer_norm_pre)             */
    const void *ln_final_w, *ln_final_b;   /* [d]                                            */
    const void *W_H, *b_H;                 /* [d, n_classes], [n_classes]  head.py:20-25     */
    const pv_vit_layer_weights* layers;    /* HOST array of n_layers entries                 */
} pv_vit_weights;

/* Unique activation buffers of one forward.  Several HookPoint names may alias one buffer
 * (e.g. blocks.l.hook_resid_post == blocks.l+1.hook_resid_pre); that mapping lives in the Python
 * tap planner.  "T" = desc.dtype.  *_NORM_F32 exist only for dtype == BF16, where the reference
 * fires hook_normalized on the fp32 value before the down-cast (layer_norm.py:89-93). */
enum pv_slot {
    /* global slots (layer field ignored) */
    PV_SLOT_EMBED = 0,        /* hook_embed                         [B, P, d]    T            */
    PV_SLOT_FULL_EMBED = 1,   /* hook_full_embed                    [B, T, d]    T            */
    PV_SLOT_LNPRE_SCALE = 2,  /* ln_pre.hook_scale                  [B, T, 1]    f32          */
    PV_SLOT_LNPRE_NORM_F32 = 3, /* ln_pre.hook_normalized (bf16 mode) [B, T, d]  f32          */
    PV_SLOT_LNPRE_OUT = 4,    /* hook_ln_pre == blocks.0.hook_resid_pre [B,T,d]  T            */
    PV_SLOT_LNF_SCALE = 5,    /* ln_final.hook_scale                [B, T, 1]    f32          */
    PV_SLOT_LNF_NORM_F32 = 6, /* ln_final.hook_normalized (bf16 mode) [B, T, d]  f32          */
    PV_SLOT_LNF_OUT = 7,      /* hook_ln_final                      [B, T, d]    T            */
    PV_SLOT_HEAD_OUT = 8,     /* hook_post_head_pre_normalize       [B, n_cls] or [B, d]  T   */
    /* per-layer slots */
    PV_SLOT_LN1_SCALE = 16,   /* blocks.l.ln1.hook_scale            [B, T, 1]    f32          */
    PV_SLOT_LN1_NORM_F32 = 17,
    PV_SLOT_LN1_OUT = 18,     /* ln1 output in T (fp32 mode: == ln1.hook_normalized)          */
    PV_SLOT_Q = 19,           /* attn.hook_q                        [B, T, H, dh] T           */
    PV_SLOT_K = 20,
    PV_SLOT_V = 21,
    PV_SLOT_SCORES = 22,      /* attn.hook_attn_scores              [B, H, T, T]  T           */
    PV_SLOT_PATTERN = 23,     /* attn.hook_pattern                  [B, H, T, T]  T           */
    PV_SLOT_Z = 24,           /* attn.hook_z                        [B, T, H, dh] T           */
    PV_SLOT_ATTN_OUT = 25,    /* hook_attn_out                      [B, T, d]     T           */
    PV_SLOT_RESID_MID = 26,   /* hook_resid_mid                     [B, T, d]     T           */
    PV_SLOT_LN2_SCALE = 27,
    PV_SLOT_LN2_NORM_F32 = 28,
    PV_SLOT_LN2_OUT = 29,
    PV_SLOT_MLP_PRE = 30,     /* mlp.hook_pre                       [B, T, d_mlp] T           */
    PV_SLOT_MLP_POST = 31,    /* mlp.hook_post                      [B, T, d_mlp] T           */
    PV_SLOT_MLP_OUT = 32,     /* hook_mlp_out                       [B, T, d]     T           */
    PV_SLOT_RESID_POST = 33,  /* hook_resid_post == next hook_resid_pre [B,T,d]   T           */
    PV_SLOT__COUNT = 34
};

/* "store this buffer at dst": the fused replacement for one caching hook
 * (hooked_root_module.py:312-316 `cache[hook.name] = tensor.detach().to(device)`). */
typedef struct pv_tap {
    int32_t slot;   /* enum pv_slot */
    int32_t layer;  /* block index for per-layer slots, 0 otherwise */
    void* dst;      /* caller-owned device memory of the slot's shape/dtype, 256-byte aligned */
} pv_tap;

typedef struct pv_vit_plan pv_vit_plan;

/* Replaces HookedViT.__init__'s shape bookkeeping (models/base_vit.py:68-150). */
int pv_vit_plan_create(const pv_vit_desc* desc, pv_vit_plan** out_plan);
void pv_vit_plan_destroy(pv_vit_plan* plan);

/* Bytes of the MFMA-layout weight shadow ([N][K] K-contiguous copies of W_QKV, W_O, W_in, W_out,
 * W_H) the caller must provide to pv_vit_plan_set_weights. */
size_t pv_vit_shadow_bytes(const pv_vit_plan* plan);
/* Borrows the parameter pointers (never owns/copies them) and (re)packs the shadow on `stream`.
 * Call again whenever parameters change (weight edits, .to(dtype), load_state_dict). */
int pv_vit_plan_set_weights(pv_vit_plan* plan, const pv_vit_weights* w, void* shadow,
                            size_t shadow_bytes, void* stream);

/* Workspace for the non-tapped intermediates of one forward at batch size B. */
size_t pv_vit_workspace_bytes(const pv_vit_plan* plan, int32_t batch);

/* HookedViT.forward (models/base_vit.py:152-217) + run_with_cache's caching hooks.
 *   images         [B, C, S, S] NCHW, element type desc.dtype
 *   n_blocks       number of transformer blocks to run: n_layers for a full forward, k for
 *                  stop_at_layer=k (the caller resolves negative indices, base_vit.py:187)
 *   run_head       0: stop after the blocks (stop_at_layer semantics, base_vit.py:189-190; the
 *                  caller taps the last residual), 1: ln_final + cls + head (+normalise) -> out
 *   taps           HOST array; buffers not listed are kept in the workspace or never materialised
 *   out            [B, n_classes] (or [B, d] without head) T; may be NULL when run_head == 0
 */
int pv_vit_forward(pv_vit_plan* plan, const void* images, int32_t batch, int32_t n_blocks,
                   int32_t run_head, const pv_tap* taps, int32_t n_taps, void* workspace,
                   size_t workspace_bytes, void* out, void* stream);

/* The same forward RESUMED at a block boundary: a Python forward hook on blocks.{L}.hook_resid_post /
 * blocks.{L+1}.hook_resid_pre (SAE substitution, zero-ablation: sae/evals/evals.py:321-392) sees the residual the
 * first call tapped, may return a replacement, and the blocks [first_block, n_blocks) (+ head) continue from it --
 * hooked_root_module.py:176-210 semantics ("split the plan at the hook point -> Python -> resume").
 *   resid_in       [B, n_tokens, d_model] T, the residual stream entering block `first_block`
 *                  (first_block == n_layers: only ln_final + head run)
 * Taps of stages before first_block are ignored. */
int pv_vit_forward_from(pv_vit_plan* plan, const void* resid_in, int32_t batch, int32_t first_block,
                        int32_t n_blocks, int32_t run_head, const pv_tap* taps, int32_t n_taps,
                        void* workspace, size_t workspace_bytes, void* out, void* stream);

/* General form of the two calls above: one SEGMENT of the forward, for hooks that sit in the middle of a block too
 * (blocks.L.hook_attn_out / hook_resid_mid).  Positions are (block, half): half 0 = the block's entry, half 1 = after
 * its attention half (the residual stream is resid_mid there).
 *   images / resid_in   exactly one is non-NULL: start from the pixels (first_block = entry_mid = 0) or resume from a
 *                       residual -- the one entering block first_block, or with entry_mid = 1 its resid_mid
 *   end_block, exit_mid run up to the entry of block end_block, or with exit_mid = 1 also through the attention half of
 *                       block end_block (tap PV_SLOT_RESID_MID there to get the result)
 *   run_head            only with end_block == n_layers and exit_mid == 0 */
int pv_vit_forward_seg(pv_vit_plan* plan, const void* images, const void* resid_in, int32_t batch,
                       int32_t first_block, int32_t entry_mid, int32_t end_block, int32_t exit_mid,
                       int32_t run_head, const pv_tap* taps, int32_t n_taps, void* workspace,
                       size_t workspace_bytes, void* out, void* stream);

/* The finest form: segments that start / stop at any of a block's ten hookable positions, so that mutating hooks INSIDE the
 * attention half and the MLP keep the native path too (head ablation through attn.hook_z, neuron ablation through
 * mlp.hook_post, edits of attn.hook_q / hook_k / hook_v, of attn.hook_attn_scores, of attn.hook_pattern, of mlp.hook_pre and of the
 * LayerNorm points ln1 / ln2 .hook_scale / .hook_normalized ("frozen LayerNorm"): prisma_tools/hook_point.py:44-45 -- a hook's
 * return value replaces the activation -- with models/layers/attention.py:135-152, 186-281, mlp.py:65-80 and layer_norm.py:75-93).
 * Positions inside block L, in the order the block passes them: */
#define PV_STAGE_ENTRY 0      /* the residual stream entering the block (hook_resid_pre)                                  */
#define PV_STAGE_LN1 1        /* ln1 taken (ln1.hook_scale, ln1.hook_normalized)                                           */
#define PV_STAGE_QKV 2        /* q, k, v computed (attn.hook_q / hook_k / hook_v)                                          */
#define PV_STAGE_SCORES 3     /* attention scores computed (attn.hook_attn_scores)                                         */
#define PV_STAGE_PATTERN 4    /* softmax taken (attn.hook_pattern)                                                         */
#define PV_STAGE_Z 5          /* attention core done (attn.hook_z)                                                        */
#define PV_STAGE_MID 6        /* after the O-projection + residual add (hook_attn_out, hook_resid_mid) == entry_mid above  */
#define PV_STAGE_LN2 7        /* ln2 taken (ln2.hook_scale, ln2.hook_normalized)                                           */
#define PV_STAGE_MLP_PRE 8    /* MLP pre-activation computed (mlp.hook_pre)                                                */
#define PV_STAGE_MLP_POST 9   /* MLP activation computed (mlp.hook_post)                                                  */
/*   exit_stage   != 0: the segment also runs block end_block up to that position and stops; the caller taps what the hook
 *                needs (PV_SLOT_LN1_SCALE / LN1_NORM_F32 (bf16 mode; LN1_OUT in fp32 mode), PV_SLOT_Q / K / V, PV_SLOT_SCORES,
 *                PV_SLOT_PATTERN, PV_SLOT_Z, PV_SLOT_RESID_MID, PV_SLOT_LN2_*, PV_SLOT_MLP_PRE, PV_SLOT_MLP_POST; for SCORES / PATTERN also
 *                PV_SLOT_V, which the resumed attention core reads) and -- to resume -- the residual stream the rest of the block
 *                adds to (the block's resid_pre up to Z, its resid_mid behind MID).  The attention core is one kernel and so is
 *                the first MLP GEMM: an exit at SCORES / PATTERN / MLP_PRE runs all of it and the later positions' results are
 *                simply not used.
 *   entry_stage  != 0: resume block first_block behind that position: resid_in = that residual stream, act_in0..2 = the
 *                (possibly edited) activations of the position: LN1 / LN2: the normalized tensor [B, T, d_model] in FP32 (what
 *                hook_normalized carries in either dtype mode; it is rounded to the storage dtype here, layer_norm.py:93) | q, k, v
 *                [B, T, H, dh] | scores [B, H, T, T], v | pattern [B, H, T, T], v | z [B, T, H, dh] | pre [B, T, d_mlp] (the
 *                activation function is applied here) | post [B, T, d_mlp] (NULL for ENTRY / MID).  Behind SCORES / PATTERN the
 *                rest of the attention core (softmax, NaN -> 0, pattern tap; pattern v) runs on a one-wave-per-row kernel.
 *                images: only with first_block = entry_stage = 0. */
int pv_vit_forward_stage(pv_vit_plan* plan, const void* images, const void* resid_in, const void* act_in0, const void* act_in1,
                         const void* act_in2, int32_t batch, int32_t first_block, int32_t entry_stage, int32_t end_block,
                         int32_t exit_stage, int32_t run_head, const pv_tap* taps, int32_t n_taps, void* workspace,
                         size_t workspace_bytes, void* out, void* stream);

/* Kernel-level entry points (used by the unit tests and by bench.py's roofline leg). */
/* C[M,N] = A[M,K] @ Bt[N,K]^T + bias[N]; dtype T for A, Bt, bias, C. */
int pv_gemm_bias(int32_t dtype, const void* A, int64_t lda, const void* Bt, int64_t ldb,
                 const void* bias, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                 void* stream);
/* The same product with one of the forward's fused epilogues (kernel-level tests and A/B timings of the epilogues the ViT plan uses;
 * models/layers/attention.py:186-244, mlp.py:65-80, transformer_block.py:122-134):
 *   epi 0 (bias):      out0 = acc + bias
 *   epi 2 (residual):  t = round_T(acc + bias); out0 (may be NULL) = t; out1 = t + resid
 *   epi 3 (activation): t = round_T(acc + bias); out0 (may be NULL) = t; out1 = act(t)         (act = PV_ACT_*)
 * A [M, K] (lda), Bt [N, K] (ldb), bias [N], resid [M, ldr], outputs [M, ldo]; all of dtype T. */
#define PV_GEMM_EPI_BIAS 0
#define PV_GEMM_EPI_RESID 2
#define PV_GEMM_EPI_ACT 3
int pv_gemm_epilogue(int32_t dtype, int32_t epi, int32_t act, const void* A, int64_t lda, const void* Bt, int64_t ldb,
                     const void* bias, const void* resid, int64_t ldr, void* out0, void* out1, int64_t ldo,
                     int32_t M, int32_t N, int32_t K, void* stream);
/* out[b][c][r] = in[b][r][c], element size 2 or 4 bytes. */
int pv_transpose_batched(int32_t elem_bytes, const void* in, void* out, int32_t batch, int32_t R,
                         int32_t C, void* stream);

/* Opt-in per-kernel timing with HIP events recorded on the launch stream (bench.py roofline leg).
 * kind: 0 GEMM, 1 attention, 2 layernorm/embed, 3 SAE encoder+topk, 4 SAE backward, 5 SAE apply, 6 misc.
 * pv_prof_read synchronises the events and returns launch count, summed kernel milliseconds and the
 * summed ALGORITHMIC flops / bytes of those launches.
 * pv_prof_enable(on): bit 0 = on/off; bits 8.. = mask of kinds to time (bit 8+kind), 0 = all kinds -- every
 * timed launch costs two event records on the stream, so a throughput run times only the kernel it reports. */
int pv_prof_enable(int32_t on);
int pv_prof_reset(void);
int pv_prof_read(int32_t kind, int64_t* launches, double* total_ms, double* flops, double* bytes);
/* The same over the launches of that family that carried one instance tag (GEMMs: 1 = QKV, 2 = O-projection, 3 = MLP-1, 4 = MLP-2,
 * 0 = the others; tagged by pv_vit_forward's GEMM launcher from epilogue and shape): per-instance roofline fractions. */
int pv_prof_read_tag(int32_t kind, int32_t tag, int64_t* launches, double* total_ms, double* flops, double* bytes);

/* Debug only (tools/gemm_trace.py): per-workgroup phase stamps {t_start, t_loop_end, t_end, hw_id} (100 MHz wall
 * clock) of the launch_idx-th plain GEMM launch from now; read blocks until the device is idle.
 * info6 = {M, N, K, epilogue, n_workgroups, kernel id}.  NOT part of the product path: the first arm hipMalloc's a
 * 256 KB device buffer owned by the library (the one place it allocates device memory) and hipMemset's it. */
int pv_debug_gemm_trace_arm(int32_t launch_idx);
int pv_debug_gemm_trace_read(uint64_t* host_out, int32_t max_wg, int32_t* info6);

/* Debug only (tests, A/B measurements): kernel-choice overrides.  The launch path never reads the environment;
 * these process-global switches are the only way to force a kernel.  Keys: "gemm_tile" (-1 auto, 0 = 128 x 128,
 * 4 / 5 = one-workgroup-per-CU 256 / 320 x 256), "gemm_v1", "gemm_v1patch", "attn_wg", "prof_markers", "sae_exact",
 * "gemm_dbg" (ablations, -DPV_TUNING builds only), "gemm_loop", "gemm_persist" (-1 auto / 0 never / 1 wherever legal: the persistent
 * form of the one-workgroup-per-CU GEMM), "gemm_stagger" (its start-skew A/B knob); key "reset" restores every default.
 * pv_debug_get_tuning("any") = 1 when anything is overridden: bench.py records it and refuses to measure then. */
int pv_debug_set_tuning(const char* key, int32_t value);
int pv_debug_get_tuning(const char* key, int32_t* value);

/* ------------------------------------------------------------------------------------------ */
/* SAE training step (sae/sae.py:557-645 forward; sae/train_sae.py:278-411 step)                */
/* ------------------------------------------------------------------------------------------ */

typedef struct pv_sae_desc {
    int32_t d_in, d_sae, k;          /* sae/config.py: d_in (<= 1280, % 4), d_sae = d_in*expansion_factor (<= 65536), topk k (<= 256; the fp16-filtered encoder up to 64, the exact fp32 encoder beyond) */
    int32_t normalize_layer_norm;    /* cfg.normalize_activations: 0 none, 1 "layer_norm" (sae.py:74-90), 2 "constant_norm_rescale" (sae.py:60-72) */
    int32_t max_tokens;              /* largest N a step will be called with                        */
    float ln_eps;                    /* 1e-5, sae.py:80                                             */
    int32_t activation;              /* the dense ReLU + L1 step (pv_sae_dense_step / pv_sae_relu_step): PV_SAE_ACT_RELU, or PV_SAE_ACT_TANH_RELU =
                                      * tanh(relu(.)), cfg.activation_fn_str = "tanh-relu" (sae.py:823-830; always on the dense GEMMs) */
    float lp_norm;                   /* the same step: p of the sparsity term l1_coefficient * mean_n ||f_n||_p (sae.py:617); 0 or 1 = the
                                      * 1-norm (the sparse form of pv_sae_relu_step serves that one only), otherwise p > 1 */
} pv_sae_desc;
#define PV_SAE_ACT_RELU 0
#define PV_SAE_ACT_TANH_RELU 1

/* Transcoder (sae/transcoder.py:6-116; train_sae.py:299-301 hands train_step the pair (input, target)): the coder encodes
 * one activation and reconstructs ANOTHER of the same width.  b_dec_out != NULL switches pv_sae_step / pv_sae_dense_step /
 * pv_sae_grad_sqnorm_step / pv_sae_apply to it:
 *   - pv_sae_state.b_dec only centres the encoder input (transcoder.py:35-37; its gradient is -W_enc gb_enc alone), the
 *     decoder adds b_dec_out (:54-64; gradient colsum(dY));
 *   - W_skip [d_in, d_in] (transcoder_with_skip_connection) or NULL: sae_out += x @ W_skip^T on the RAW input, before LN-out
 *     (:73-76); gW_skip = dY^T x;
 *   - loss, normaliser and `batch_mean` are the TARGET's (`target` [n_tokens, d_in], set before every step; :78);
 *   - the clip norm covers the two extra tensors, pv_sae_apply runs plain Adam on them with the same clip coefficient
 *     (PV_SAE_SPARSE_GRADS works as for the plain step).
 * pv_sae_step and pv_sae_dense_step: also with tokens sharded over ranks (batch_mean = the TARGET's global mean, n_global = the global
 * token count; the caller all-reduces the flat gradient buffer); the feature-parallel entry points refuse a transcoder state.
 * d_out != d_in: see d_in_true / d_out_true below. */
typedef struct pv_sae_transcoder {
    float *b_dec_out, *gb_dec_out, *mb_dec_out, *vb_dec_out;   /* [d_in]                                              */
    float *W_skip, *gW_skip, *mW_skip, *vW_skip;               /* [d_in, d_in] (row o = output coordinate) or all NULL */
    const float* target;                                       /* [n_tokens, d_in] of the coming step                  */
    void* scratch;                                             /* pv_sae_transcoder_scratch_bytes (only with W_skip)   */
    size_t scratch_bytes;
    /* d_out != d_in (skip-less transcoders between hook points of different width): the plan's d_in is max(d_in, d_out) and EVERY
     * row of width d_in above is padded to it with zeros -- W_enc rows / b_dec / x beyond d_in_true, W_dec columns / b_dec_out beyond
     * d_out_true (they stay exact zeros: their gradients are) -- d_in_true / d_out_true = the real widths (0 = the plan's d_in).  LN-in
     * runs over d_in_true columns, the loss is the mean over n_tokens x d_out_true, and the step WRITES the padding columns of `target`
     * (the token's LN mean: what the zero-padded decoder reconstructs there). */
    int32_t d_in_true, d_out_true;
} pv_sae_transcoder;

/* Gated SAE (GatedSparseAutoencoder, sae.py:648-792, activation_fn_str = "relu"): gate path (sae_in @ W_enc + b_gate) > 0,
 * magnitude path with shared weights sae_in @ (W_enc * exp(r_mag)) + b_mag, L1 on relu(gate pre-activation) weighted by the
 * decoder row norms, auxiliary reconstruction of sae_in through the gate.  b_gate != NULL makes a state a gated one:
 * pv_sae_gated_step is its train step, pv_sae_apply adds plain Adam on the three vectors.  pv_sae_state.b_enc exists (the
 * reference keeps the parameter) but takes no part: its gradient is written as zero. */
typedef struct pv_sae_gated {
    float *b_gate, *r_mag, *b_mag;                 /* [d_sae]                                            */
    float *gb_gate, *gr_mag, *gb_mag;
    float *mb_gate, *mr_mag, *mb_mag;
    float *vb_gate, *vr_mag, *vb_mag;
    void* scratch;                                 /* pv_sae_gated_scratch_bytes                         */
    size_t scratch_bytes;
} pv_sae_gated;

/* fp32 master parameters in the reference's layouts (sae.py:537-555), their gradients, Adam
 * moments and training statistics -- all caller-owned (torch tensors), fp32 unless stated. */
typedef struct pv_sae_state {
    float *W_enc, *W_dec, *b_enc, *b_dec;          /* [d_in,d_sae] [d_sae,d_in] [d_sae] [d_in]   */
                                                   /* W_enc may be NULL in pv_sae_step / pv_sae_dense_step / pv_sae_tp_* /
                                                      pv_sae_apply / pv_sae_encode_topk / pv_sae_forward: the kernels read the
                                                      encoder through W_encT (below), and an apply without W_enc leaves the
                                                      parameter's own layout stale (saves its 75 MB transposed write per step at
                                                      768 -> 24576) until pv_sae_sync_shadows(from_transposed = 1) rewrites it --
                                                      the host materialises it when somebody reads the parameter               */
    float *gW_enc, *gW_dec, *gb_enc, *gb_dec;      /* gradients; gW_enc is stored TRANSPOSED,      */
                                                   /* [d_sae, d_in] (coalesced sparse backward; the */
                                                   /* Adam kernel transposes it back tile-wise).    */
                                                   /* ONE contiguous flat buffer is recommended so  */
                                                   /* a single all-reduce + one norm pass cover it  */
    float *mW_enc, *mW_dec, *mb_enc, *mb_dec;      /* Adam exp_avg; mW_enc / vW_enc are kept in the */
    float *vW_enc, *vW_dec, *vb_enc, *vb_dec;      /* gradient's TRANSPOSED [d_sae, d_in] layout    */
    float *act_freq_scores;                        /* [d_sae]  train_sae.py:360                  */
    float *n_fwd_since_fired;                      /* [d_sae]  train_sae.py:357-358              */
    /* Encoder shadows (caller-owned memory, contents owned by the library: pv_sae_apply keeps them in step with W_enc,
     * pv_sae_sync_shadows rebuilds them).  NULL disables the filtered encoder (exact fp32 GEMM + top-k instead);
     * pv_sae_apply requires them. */
    float *W_encT;                                 /* [d_sae, d_in] fp32 transpose of W_enc: the layout Adam runs in and
                                                      the exact re-scoring gathers rows from                          */
    uint16_t *W_enc16T;                            /* [d_sae, d_in] fp16 (B operand of the filter GEMM)               */
    float *enc_colsq;                              /* [d_sae] ||W_enc[:, j]||^2 (error bound of the filter)           */
    float *dec_inv_norm;                           /* [d_sae] scratch of PV_SAE_RENORM_DECODER (1 / ||W_dec[j]||), or NULL */
    pv_sae_transcoder tc;                          /* all NULL: a plain autoencoder                                   */
    pv_sae_gated gt;                               /* all NULL: not a gated SAE                                       */
} pv_sae_state;

/* Per-step outputs, caller-owned. */
typedef struct pv_sae_out {
    float* sae_out;        /* [N, d_in] reconstruction (after LN-out), may be NULL                */
    int32_t* topk_idx;     /* [N, k]   selected feature indices                                  */
    float* topk_val;       /* [N, k]   relu(hidden_pre) at those indices (feature_acts, sparse)   */
    float* scalars;        /* [8]: 0 loss, 1 mse_loss, 2 l0, 3 grad sum-of-squares (pre-clip), 4 l1_loss (dense step) */
    float* fire_count;     /* [d_sae] tokens of this call on which each feature fired, or NULL     */
} pv_sae_out;

typedef struct pv_sae_plan pv_sae_plan;
int pv_sae_plan_create(const pv_sae_desc* desc, pv_sae_plan** out_plan);
size_t pv_sae_transcoder_scratch_bytes(const pv_sae_plan* plan, int32_t n_tokens);   /* skip term + split-K partials of gW_skip */
void pv_sae_plan_destroy(pv_sae_plan* plan);
size_t pv_sae_workspace_bytes(const pv_sae_plan* plan);

/* set_decoder_norm_to_unit_norm (sae.py:275-277): W_dec /= ||W_dec||_row, in place. */
int pv_sae_renorm_decoder(pv_sae_plan* plan, pv_sae_state* st, void* stream);

/* Forward + backward + statistics of one train step on N tokens x [N, d_in] fp32
 * (train_sae.py:328-392 between zero_grad and clip).  Gradients are WRITTEN (not accumulated)
 * into st->g*.  `batch_mean` [d_in] is mean_n(x) over the GLOBAL batch (sae.py:145) -- pass NULL to
 * have it computed from x (single process); `n_global` scales the loss mean (N for 1 process).
 * flags: PV_SAE_UPDATE_STATS    apply did_fire / act_freq updates (train_sae.py:356-361);
 *        PV_SAE_RENORM_DECODER  set_decoder_norm_to_unit_norm (train_sae.py:307) as part of this step: the forward and
 *                               backward use the unit-norm rows, the physical rewrite of W_dec is deferred to (and fused
 *                               into) the following pv_sae_apply (needs st->dec_inv_norm);
 *        PV_SAE_INV_NORM_VALID  (with PV_SAE_RENORM_DECODER) st->dec_inv_norm already holds 1 / ||W_dec[j]|| of the current
 *                               rows: a pv_sae_apply over ALL features leaves them there, and the caller vouches that W_dec
 *                               has not been touched since -- saves re-reading W_dec for the norms;
 *        PV_SAE_SPARSE_GRADS    single-process training only: the gradient rows (gW_enc^T, gW_dec) of features that kept no
 *                               token this step are NOT written (they are zero by definition; on a trained-like batch that is
 *                               half of the 151 MB).  The following pv_sae_apply takes them as zero from the per-feature
 *                               offsets this step leaves in `workspace`, so the workspace must stay untouched until then;
 *                               pv_sae_grad_sqnorm_step stays valid, every reader of the raw gradient buffers (pv_sae_grad_sqnorm,
 *                               pv_sae_grad_sqnorm_rows, a gradient all-reduce) must not be used with this flag. */
#define PV_SAE_UPDATE_STATS 1
#define PV_SAE_RENORM_DECODER 2
#define PV_SAE_INV_NORM_VALID 4
#define PV_SAE_SPARSE_GRADS 8
#define PV_SAE_TP_ENC_TERM_ONLY 16   /* pv_sae_tp_finish: st->gb_dec = -W_enc[:, shard] gb_enc[shard] only (without colsum(dY)) */
#define PV_SAE_FUSED_SQNORM 32       /* pv_sae_step (autoencoder states): the step's last launch also leaves scalars[3] = the gradient's sum of
                                        squares -- pv_sae_grad_sqnorm_step's terms, added block-wise (a fixed order of its own, so the
                                        last bits may differ from that call's); the caller then goes straight to pv_sae_apply */
int pv_sae_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t n_tokens,
                const float* batch_mean, int32_t n_global, int32_t flags, pv_sae_out* out,
                void* workspace, size_t workspace_bytes, void* stream);

/* ---- feature-parallel training step (new functionality) ------------------------------------
 * This rank's plan / state cover a SHARD of the features (all tokens of the batch on every rank).  pv_sae_step cut at the
 * reconstruction, with the global top-k taken by the caller in between:
 *   pv_sae_encode_topk   this rank's k candidates per token (the workspace keeps LN statistics, normaliser, sae_in)
 *   [caller: all-gather of the candidate values, global top-k per token; local candidates that lose get value 0 -- a pair
 *    with value <= 0 is a hole in every kernel, exactly like the reference's ReLU behind its top-k (sae.py:795-810)]
 *   pv_sae_tp_partial    partial [N, d_in] = sum_s val_s W_dec[idx_s] over this rank's kept pairs (no b_dec, no LN-out);
 *                        flags: PV_SAE_RENORM_DECODER / PV_SAE_INV_NORM_VALID as in pv_sae_step
 *   [caller: all-reduce of the partials]
 *   pv_sae_tp_finish     pre_sum [N, d_in] = the summed reconstruction: LN-out, loss, dY, dh, CSR, sparse backward and
 *                        (PV_SAE_UPDATE_STATS) firing statistics for the shard's features; SAME workspace and x as the
 *                        encode; gradients written into st->g*; st->gb_dec = colsum(dY) - W_enc[:, shard] gb_enc[shard]
 *                        (the caller sums the ranks' encoder terms: every rank but one passes PV_SAE_TP_ENC_TERM_ONLY, which
 *                        leaves colsum(dY) out, and all-reduces gb_dec); scalars[0..1] = loss (replicated), scalars[2] = this
 *                        rank's kept pairs per token (the ranks' values add up to l0); out->topk_idx / topk_val are unused
 *   [caller: all-reduces of gb_dec's encoder term, of the clip-norm terms (pv_sae_grad_sqnorm_rows) and of l0]
 *   pv_sae_apply(0, d_sae_shard)                                                                                         */
int pv_sae_tp_partial(pv_sae_plan* plan, pv_sae_state* st, const int32_t* topk_idx, const float* topk_val, int32_t n_tokens,
                      int32_t flags, float* partial, void* stream);
int pv_sae_tp_finish(pv_sae_plan* plan, pv_sae_state* st, const float* x, const float* pre_sum, const int32_t* topk_idx,
                     const float* topk_val, int32_t n_tokens, int32_t n_global, int32_t flags, pv_sae_out* out,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Global top-k over the ranks' candidates (the step between pv_sae_encode_topk and pv_sae_tp_partial above).
 * gathered [world][2][n_tokens][k] int32 = the all-gather of every rank's {candidate values (float bits), LOCAL feature
 * indices}; a candidate's global feature index is rank * shard + local.  val_kept [n_tokens][k] = this rank's candidate
 * values where they are among the k best of ALL ranks' candidates of their token in the order (value descending, global
 * feature index ascending) -- torch.topk's order on the dense row (sae.py:795-810) -- and 0 elsewhere.  world <= 8. */
int pv_sae_tp_merge(const int32_t* gathered, int32_t world, int32_t rank, int32_t n_tokens, int32_t k, int32_t shard,
                    float* val_kept, void* stream);
/* The one small all-reduce of the feature-parallel step.  bucket = [gb_dec (d_in) | clip-norm rows term | kept pairs per
 * token | 2 floats of padding | firing counts of all d_sae_total features].  The caller points pv_sae_state.gb_dec at
 * bucket and pv_sae_out.fire_count at bucket + d_in + 4 + j_lo for pv_sae_tp_finish; pack (after it, same workspace) adds
 * the sum of squares of the shard's gradient rows and scalars[2], and zeroes the other ranks' firing counts; unpack
 * (after the all-reduce) sets scalars[3] = sum of the ranks' row terms + ||gb_dec||^2 (the clip norm of the GLOBAL
 * gradient, train_sae.py:394-397) and scalars[2] = l0. */
int pv_sae_tp_bucket_pack(pv_sae_plan* plan, const void* workspace, const float* scalars, float* bucket, int32_t j_lo,
                          int32_t d_sae_total, void* stream);
int pv_sae_tp_bucket_unpack(pv_sae_plan* plan, const float* bucket, float* scalars, void* stream);

/* ---- dense step: ReLU + L1 SAEs (activation_fn_str = "relu": sae.py:557-645 with the L1 sparsity term :617-626; the kind
 * every published CLIP SAE of the reference is, docs/sae_table.md) ---------------------------------------------------------
 * Same contract as pv_sae_step (forward + backward + statistics of one train step on N tokens, gradients WRITTEN into
 * st->g*, complete buffers), for feature activations that are not k-sparse: five dense GEMMs on the exact fp32 matrix
 * instruction with the elementwise work in their epilogues (sae_dense.hip).  plan->d.k is ignored.  flags:
 * PV_SAE_UPDATE_STATS, PV_SAE_RENORM_DECODER (set_decoder_norm_to_unit_norm is applied to W_dec in place, first).
 * scalars: 0 loss = mse + l1 (+ ghost), 1 mse_loss, 2 l0 (mean_n #(f > 0)), 4 l1_loss = l1_coefficient * mean_n ||f_n||_1,
 * 5 ghost residual loss (0 without pv_sae_ghost).
 * out->topk_idx / topk_val are unused; out->fire_count [d_sae] and out->sae_out [N, d_in] are optional.
 * Follow with pv_sae_grad_sqnorm (the whole flat gradient buffer) and pv_sae_apply.  d_in % 8 == 0, d_sae % 8 == 0. */
/* Ghost gradients (use_ghost_grads: SparseAutoencoder._compute_ghost_residual_loss sae.py:151-179, train_sae.py:337-346): the
 * caller lists the features that count as dead BEFORE this step (n_forward_passes_since_fired > dead_feature_window) --
 * dead_idx [n_dead] ascending, dead_slot [d_sae] = position of feature j in that list or -1 -- and owns the extra workspace
 * (pv_sae_ghost_workspace_bytes).  The step then adds the ghost residual loss (scalars[5]; the reference adds it even when
 * no feature is dead) and its gradient: exp(hidden_pre) of the dead columns leaves the encoder GEMM's epilogue, three small
 * GEMMs over the dead columns do the rest, the result joins dH in the G3 epilogue and gW_dec by a row scatter-add.
 * NULL = no ghost gradients.
 * Tokens sharded over ranks (n_global > n_tokens): the ghost term normalises by the residual's column mean and rescales by the mse
 * loss OF THE WHOLE BATCH (sae.py:156, :172), so the caller provides both -- err_colmean [d_in] = mean over the global batch of
 * (sae_out - target) and mse_global (device scalar) -- from a forward of the same batch (pv_sae_topk_ghost: the pv_sae_step before it;
 * pv_sae_dense_step: a pass without ghost gradients) and an all-reduce; n_global (pv_sae_topk_ghost; the dense step takes its own
 * argument) = the global token count.  All three zero / NULL: single process. */
typedef struct pv_sae_ghost {
    int32_t n_dead;
    const int32_t* dead_idx;
    const int32_t* dead_slot;
    void* workspace;
    size_t workspace_bytes;
    const float* err_colmean;
    const float* mse_global;
    int32_t n_global;
    int32_t reserved;
} pv_sae_ghost;
size_t pv_sae_ghost_workspace_bytes(const pv_sae_plan* plan, int32_t n_tokens, int32_t n_dead);
int pv_sae_dense_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t n_tokens, const float* batch_mean,
                      int32_t n_global, int32_t flags, float l1_coefficient, const pv_sae_ghost* ghost, pv_sae_out* out,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Ghost gradients of a TOP-K SAE (use_ghost_grads with activation_fn_str = "topk"): run AFTER pv_sae_step on the same batch -- complete
 * gradient buffers (no PV_SAE_SPARSE_GRADS), the decoder renormalised in place beforehand (pv_sae_renorm_decoder; not the deferred
 * PV_SAE_RENORM_DECODER), out->sae_out = the reconstruction that step wrote.  Adds the ghost residual loss (scalars[5]; scalars[0] = mse +
 * ghost) and its gradient to the dead features' rows of gW_dec / gW_enc^T / gb_enc and to gb_dec.  exp(hidden_pre) of the dead features comes
 * from one small GEMM (the k-sparse encoder never materialises hidden_pre).  `ghost` as for pv_sae_dense_step (the dead list is taken BEFORE
 * the step's statistics).  Follow with pv_sae_grad_sqnorm over the whole flat buffer and pv_sae_apply.  Single process, no transcoder.
 * Replaces sae/sae.py:151-179 behind TopK (:795-810), train_sae.py:330-346. */
int pv_sae_topk_ghost(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t n_tokens, const pv_sae_ghost* ghost, pv_sae_out* out,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The ReLU + L1 step, sparse where the batch allows it ("ReLU is top-k with threshold 0 and a variable k"): same contract, flags,
 * scalars and follow-up calls as pv_sae_dense_step (no ghost gradients here), same results -- exact fp32 values on both paths.
 * ONE product over all features (the fp16 MFMA filter of pv_sae_step with the per-token threshold -B_n, B_n = the proven error band
 * of the fp16 product, and the exact fp32 re-scoring of every survivor against W_encT) yields each token's POSITIVE activations as
 * a list of at most sp->cap (feature, value) pairs; decode, the CSR by feature, the sparse backward and the bias gradients then run
 * on pv_sae_step's kernels with k = cap (L1 term = the sum of the kept values, its gradient l1_coefficient / N on every kept pair).
 * If some token of the batch cannot be held -- more positives than cap, a candidate slot overflow (the first, dense steps of a
 * training run; the published x64 SAEs with L0 ~ 600-2000, docs/sae_table.md) -- the device-side word *mode is raised, the sparse
 * kernels leave at once and the five dense GEMMs of pv_sae_dense_step run: decided on the GPU, no host round trip.
 *   sp->cap          kept activations per token the sparse form can hold: a multiple of 4 in [4, 256]
 *   sp->workspace    caller-owned, pv_sae_relu_workspace_bytes(plan, n_tokens, cap) bytes, 256-byte aligned.  Its first word is the
 *                    mode of the last step (0 = ran sparse, 1 = ran dense); the kept pairs follow (tests: pv_debug_sae_relu_offset)
 * sp == NULL, or a plan the filter does not cover (d_sae % 256, d_sae < 2048, d_in % 8): the dense step.
 * Two more flags than pv_sae_dense_step takes:
 *   PV_SAE_RENORM_DECODER (+ PV_SAE_INV_NORM_VALID) is DEFERRED as in pv_sae_step where the sparse form applies and
 *       st->dec_inv_norm is given (inverse row norms now, the rows rewritten by pv_sae_apply; a step that turns out dense rewrites
 *       them itself, in place, before its GEMMs and leaves dec_inv_norm = 1); in place, first, otherwise.
 *   PV_SAE_SPARSE_GRADS (single-process training): a step that ran sparse leaves the gradient rows of features no token kept
 *       unwritten, as pv_sae_step does; a step that ran dense marks every feature live and derives the per-feature clip-norm terms
 *       from its complete rows.  Only pv_sae_grad_sqnorm_step and pv_sae_apply may follow.
 * Requires the encoder shadows of pv_sae_state (W_encT, W_enc16T, enc_colsq) to be current.
 * Replaces sae/sae.py:557-645 (L1 :617-626) + sae/train_sae.py:328-392, like pv_sae_dense_step. */
typedef struct pv_sae_relu_sparse {
    int32_t cap;
    int32_t reserved;
    void* workspace;
    size_t workspace_bytes;
} pv_sae_relu_sparse;
size_t pv_sae_relu_workspace_bytes(const pv_sae_plan* plan, int32_t n_tokens, int32_t cap);
/* byte offset of a named region of that workspace ("mode": uint32; "idx": int32 [n_tokens][cap]; "val": float [n_tokens][cap];
 * "tok_cnt": uint32 [n_tokens]), or (size_t)-1 */
size_t pv_debug_sae_relu_offset(const pv_sae_plan* plan, int32_t n_tokens, int32_t cap, const char* name);
int pv_sae_relu_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t n_tokens, const float* batch_mean,
                     int32_t n_global, int32_t flags, float l1_coefficient, const pv_sae_relu_sparse* sp, pv_sae_out* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* One train step of a gated SAE (batch_mean / n_global as in pv_sae_step: tokens may be sharded over ranks, the caller all-reduces
 * the flat gradient buffer): forward + backward + statistics on the dense GEMM kernel -- the shared
 * product sae_in @ W_enc once (both paths in its epilogue), the two decoder products (feature_acts and relu(gate)) as ONE GEMM
 * over stacked rows, likewise their two backward products and the two terms of gW_dec.  Gradients WRITTEN into st->g* and
 * st->gt.g* (gb_enc = 0); flags: PV_SAE_UPDATE_STATS, PV_SAE_RENORM_DECODER (REQUIRED: the L1 term's decoder norms are taken
 * as 1, which is what train_sae.py:307 establishes before every forward).  scalars: 0 loss = mse + l1 + aux, 1 mse_loss, 2 l0,
 * 4 l1_loss, 6 auxiliary reconstruction loss.  Follow with pv_sae_grad_sqnorm over the flat gradient buffer and pv_sae_apply.
 * d_in % 8 == 0, d_sae % 8 == 0. */
size_t pv_sae_gated_scratch_bytes(const pv_sae_plan* plan, int32_t n_tokens);
int pv_sae_gated_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t n_tokens, const float* batch_mean, int32_t n_global,
                      int32_t flags, float l1_coefficient, pv_sae_out* out, void* workspace, size_t workspace_bytes, void* stream);
/* The same step, sparse where the batch allows it (as pv_sae_relu_step is to pv_sae_dense_step): a gated SAE's forward and every
 * gradient behind it vanish where the gate is shut (gate_pre <= 0, sae.py:703-716), so ONE fp16-filtered product over all features
 * + the exact fp32 re-scoring of its survivors gives each token's OPEN gates as a list of at most sp->cap pairs {feature_acts,
 * relu(gate_pre)}; the two decoder products, the CSR by feature and the sparse backward then run on the k-sparse kernels over
 * [feature_acts; relu(gate_pre)] stacked as 2 n_tokens rows -- the dense form's stacking.  A batch some token of which cannot be held
 * raises the device-side mode word (sp->workspace, as pv_sae_relu_step) and the dense GEMMs run instead; exact fp32 values either
 * way.  sp->workspace: pv_sae_gated_sparse_workspace_bytes.  Needs the encoder shadows (W_enc16T, enc_colsq) current, else dense. */
/* The TOP-K form of the gated SAE (activation_fn_str = "topk" on a GatedSparseAutoencoder: TopK on the magnitudes AND on the gate
 * activations, no L1 term -- sae.py:699-716, 741-745, 773-778): two k-sparse lists per token (k = the plan's k).  The magnitude
 * path's top-k runs on the filtered encoder against a per-step scaled copy of the encoder shadows (W_encT e^r_mag), the gate path's
 * on the shadows with b_gate as the bias; each kept magnitude's gate is evaluated exactly; one decode per list, ONE CSR + sparse
 * backward over the two lists stacked as 2 n_tokens rows.  The PLAN must be created for 2 x n_tokens (max_tokens >= 2 n_tokens:
 * its k-dependent buffers hold both lists) and out->topk_idx / topk_val hold 2 n_tokens x k entries: rows [0, n_tokens) = the
 * magnitude list (feature_acts), rows [n_tokens, 2 n_tokens) = the gate list (relu'd gate activations).  flags, scalars (4 = 0),
 * batch_mean / n_global and the follow-up calls as pv_sae_gated_step; scratch: pv_sae_gated_topk_scratch_bytes (gt.scratch is not
 * used). */
size_t pv_sae_gated_topk_scratch_bytes(const pv_sae_plan* plan, int32_t n_tokens);
int pv_sae_gated_topk_step(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t n_tokens, const float* batch_mean,
                           int32_t n_global, int32_t flags, pv_sae_out* out, void* workspace, size_t workspace_bytes, void* scratch,
                           size_t scratch_bytes, void* stream);
size_t pv_sae_gated_sparse_workspace_bytes(const pv_sae_plan* plan, int32_t n_tokens, int32_t cap);
int pv_sae_gated_step_sparse(pv_sae_plan* plan, pv_sae_state* st, const float* x, int32_t n_tokens, const float* batch_mean,
                             int32_t n_global, int32_t flags, float l1_coefficient, const pv_sae_relu_sparse* sp, pv_sae_out* out,
                             void* workspace, size_t workspace_bytes, void* stream);

/* sum of squares of the flat gradient buffer (all four tensors) -> scalars[3] (device), for
 * clip_grad_norm_ (train_sae.py:394-397); called after the (optional) gradient all-reduce.
 * partial_1024: 1024 floats of scratch.  Deterministic two-stage reduction. */
int pv_sae_grad_sqnorm(const float* flat_grads, int64_t n, float* partial_1024, float* scalars, void* stream);
/* The same number (up to summation order) from the per-feature sums of squares the backward kernels of the last pv_sae_step
 * left in `workspace`, without re-reading the gradient: valid only while the gradient buffers are as that step wrote them
 * (single process: no all-reduce in between). */
int pv_sae_grad_sqnorm_step(pv_sae_plan* plan, const pv_sae_state* st, const void* workspace, float* scalars, void* stream);
/* The same over the gradient rows of features [j_lo, j_hi) only (+ gb_dec when include_b_dec): one rank's term of the
 * clip norm when the optimizer is sharded by feature (new functionality; j_lo % 4 == 0). */
int pv_sae_grad_sqnorm_rows(pv_sae_plan* plan, const pv_sae_state* st, int32_t j_lo, int32_t j_hi, int32_t include_b_dec,
                            float* partial_1024, float* scalars, void* stream);

/* clip (coef from scalars[3] on device, max_norm <= 0 disables) -> remove gradient parallel to
 * decoder rows (sae.py:279-297) -> Adam(betas .9/.999, eps 1e-8, wd 0; train_sae.py:229) with
 * learning rate lr at 1-based step `step` -- one fused pass over parameters + moments, for the features
 * [j_lo, j_hi) (0, d_sae = everything; a sub-range = this rank's shard of the optimizer) and always b_dec.
 * Keeps W_enc, W_encT, W_enc16T and enc_colsq of those features in step. */
int pv_sae_apply(pv_sae_plan* plan, pv_sae_state* st, const float* scalars, float max_grad_norm,
                 float lr, int32_t step, int32_t j_lo, int32_t j_hi, void* stream);

/* Rebuild the encoder shadows of features [j_lo, j_hi): from_transposed = 0 takes W_enc as the truth (parameters loaded
 * or edited by the caller), 1 takes W_encT (e.g. after an all-gather of optimizer shards) and rewrites W_enc. */
int pv_sae_sync_shadows(pv_sae_plan* plan, pv_sae_state* st, int32_t from_transposed, int32_t j_lo, int32_t j_hi, void* stream);
/* 1 when this plan's shapes take the filtered encoder (fp16 MFMA filter + exact fp32 re-scoring, sae_enc.hip), 0 when
 * they take the exact fp32 GEMM + streaming top-k. */
int pv_sae_encoder_is_filtered(const pv_sae_plan* plan);
/* Debug / tests: byte offset of a named workspace region ("fb_count": uint32, tokens of the last encode that took the
 * exact fallback; "fb_list", "cand_cnt", "thr"), (size_t)-1 if unknown. */
size_t pv_debug_sae_ws_offset(const pv_sae_plan* plan, const char* name);

/* Inference-side pieces for the module API (StandardSparseAutoencoder.encode/decode). */
int pv_sae_encode_topk(pv_sae_plan* plan, const pv_sae_state* st, const float* x, int32_t n_tokens,
                       int32_t* topk_idx, float* topk_val, float* ln_mu, float* ln_std,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Inference forward for the module API (StandardSparseAutoencoder.forward, sae.py:597-645, in eval / no-grad use such
 * as the SAE-substitution evals, sae/evals/evals.py:321-392): top-k encode + sparse decode + LN-out -> sae_out [N, d_in]
 * plus the sparse feature activations (topk_idx / topk_val [N, k]).  scalars (optional, >= 2 floats): [1] = mse loss of
 * these N tokens as one batch.  ln_mu / ln_std optional. */
int pv_sae_forward(pv_sae_plan* plan, const pv_sae_state* st, const float* x, int32_t n_tokens, float* sae_out,
                   int32_t* topk_idx, float* topk_val, float* ln_mu, float* ln_std, float* scalars, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline: get_clip_val_transforms (transforms/model_transforms.py:9-20) for a batch of decoded
 * uint8 RGB images [B][H][W][3] (device) -> [B][3][S][S] fp32 / bf16 in one kernel:
 *   Resize(S, BICUBIC, antialias) -- Pillow's two-pass fixed-point resampler, which is what torchvision's Resize runs on the
 *   PIL images the reference feeds it: bounds [n][2] = (first input index, tap count) and taps [n][ksize] (int32, 22
 *   fractional bits) per output column / row, built on the host the way Resample.c precompute_coeffs + normalize_coeffs_8bpc
 *   do (vit_prisma_amd/transforms.py:_pil_coeffs); horizontal pass, uint8 rounding, vertical pass, uint8 rounding --
 *   CenterCrop (`left`, `top` inside the new_w x new_h resized image) -- ToTensor (/ 255) -- Normalize ((x - mean) / std).
 * Bit-identical to the reference's CPU pipeline for RGB uint8 input. */
int pv_clip_preprocess(const uint8_t* images, int32_t B, int32_t H, int32_t W, const int32_t* xbounds, const int32_t* xtaps,
                       int32_t ksize_x, const int32_t* ybounds, const int32_t* ytaps, int32_t ksize_y, int32_t new_w,
                       int32_t new_h, int32_t left, int32_t top, int32_t S, const float* mean3_host, const float* std3_host,
                       int32_t out_dtype, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PV_NATIVE_H */
