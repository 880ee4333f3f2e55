// Host-side plan + launch sequence of the tapped ViT forward (C ABI in include/pv_native.h).
// Mirrors HookedViT.forward (/root/reference/src/vit_prisma/models/base_vit.py:152-217) and
// TransformerBlock.forward (models/layers/transformer_block.py:80-138); ln1 is evaluated ONCE per
// block (the reference evaluates it three times with identical results, :106-109).
#include <string.h>

#include <vector>

#include "attention.hpp"
#include "gemm.hpp"
#include "rowops.hpp"

static thread_local std::string g_last_error;
void pv_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" int pv_abi_version(void) { return PV_ABI_VERSION; }
extern "C" void pv_last_error(char* buf, size_t len) {
    if (!buf || len == 0) return;
    strncpy(buf, g_last_error.c_str(), len - 1);
    buf[len - 1] = 0;
}

struct LayerShadow {
    unsigned char* Wqkv;   // [3*H*dh][d]
    unsigned char* WoT;    // [d][H*dh]
    unsigned char* WinT;   // [d_mlp][d]
    unsigned char* WoutT;  // [d][d_mlp]
};

struct pv_vit_plan {
    pv_vit_desc d;
    int P, G, T, HD, EB;
    pv_vit_weights w;
    std::vector<pv_vit_layer_weights> lw;
    std::vector<LayerShadow> sh;
    unsigned char* WhT;
    // bf16, even patch size other than 32: the patches are packed into GEMM rows once per batch (Kp = K rounded up to 8
    // columns) so that the patch embedding runs on the tiled DMA GEMM instead of the register-staged gather kernel
    // (L/14@336: 1.06 ms -> pack + GEMM); conv_wp = the weights padded to Kp columns (the parameter itself when Kp == K)
    bool patch_prepack;
    int Kp;
    unsigned char* conv_wp;
    bool weights_set;
};

static size_t seg(size_t elems, int eb) { return (size_t)pv_align_up((int64_t)elems * eb, 256); }

extern "C" int pv_vit_plan_create(const pv_vit_desc* desc, pv_vit_plan** out_plan) {
    PV_REQUIRE(desc && out_plan, "null argument");
    const pv_vit_desc& d = *desc;
    PV_REQUIRE(d.dtype == PV_DTYPE_F32 || d.dtype == PV_DTYPE_BF16, "dtype must be fp32 or bf16");
    PV_REQUIRE(d.n_layers >= 0 && d.d_model > 0 && d.n_heads > 0 && d.d_head > 0 && d.d_mlp > 0, "dims");
    PV_REQUIRE(d.d_model % 8 == 0 && d.d_model <= 2048, "d_model must be a multiple of 8 and <= 2048");
    PV_REQUIRE(d.d_mlp % 8 == 0, "d_mlp must be a multiple of 8");
    PV_REQUIRE(d.patch_size > 0 && d.image_size >= d.patch_size, "patch/image size");
    const int G = d.image_size / d.patch_size;
    const int P = G * G;
    PV_REQUIRE(d.n_tokens == P + (d.use_cls_token ? 1 : 0), "n_tokens != patches (+cls)");
    PV_REQUIRE(pv_attention_supported(d.n_tokens, d.d_head), "attention shape unsupported (T <= 640, d_head in {32,64})");
    PV_REQUIRE(d.activation >= PV_ACT_GELU && d.activation <= PV_ACT_RELU, "activation");
    if (d.has_head) PV_REQUIRE(d.n_classes > 0, "n_classes");
    pv_vit_plan* p = new pv_vit_plan();
    p->d = d;
    p->G = G;
    p->P = P;
    p->T = d.n_tokens;
    p->HD = d.n_heads * d.d_head;
    p->EB = d.dtype == PV_DTYPE_BF16 ? 2 : 4;
    p->WhT = nullptr;
    p->conv_wp = nullptr;
    p->Kp = (d.n_channels * d.patch_size * d.patch_size + 7) / 8 * 8;
    p->patch_prepack = d.dtype == PV_DTYPE_BF16 && d.patch_size != 32 && d.patch_size % 2 == 0 && d.image_size % 2 == 0;
    p->weights_set = false;
    *out_plan = p;
    return PV_OK;
}

extern "C" void pv_vit_plan_destroy(pv_vit_plan* plan) { delete plan; }

extern "C" size_t pv_vit_shadow_bytes(const pv_vit_plan* p) {
    if (!p) return 0;
    const pv_vit_desc& d = p->d;
    size_t per_layer = seg((size_t)3 * p->HD * d.d_model, p->EB) + seg((size_t)d.d_model * p->HD, p->EB) +
                       2 * seg((size_t)d.d_model * d.d_mlp, p->EB);
    size_t total = per_layer * d.n_layers;
    if (d.has_head) total += seg((size_t)d.n_classes * d.d_model, p->EB);
    if (p->patch_prepack) total += seg((size_t)d.d_model * p->Kp, p->EB);
    return total + 256;
}

extern "C" int pv_vit_plan_set_weights(pv_vit_plan* p, const pv_vit_weights* w, void* shadow,
                                       size_t shadow_bytes, void* stream_) {
    PV_REQUIRE(p && w && shadow, "null argument");
    PV_REQUIRE(shadow_bytes >= pv_vit_shadow_bytes(p), "shadow buffer too small");
    PV_REQUIRE(((uintptr_t)shadow & 255) == 0, "shadow buffer must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const pv_vit_desc& d = p->d;
    PV_REQUIRE(w->conv_w && w->conv_b && w->W_pos && w->ln_final_w && w->ln_final_b, "missing global weights");
    if (d.use_cls_token) PV_REQUIRE(w->cls_token, "missing cls_token");
    if (d.layer_norm_pre) PV_REQUIRE(w->ln_pre_w && w->ln_pre_b, "missing ln_pre weights");
    if (d.has_head) PV_REQUIRE(w->W_H && w->b_H, "missing head weights");
    PV_REQUIRE(d.n_layers == 0 || w->layers, "missing layer weights");
    p->w = *w;
    p->lw.assign(w->layers, w->layers + d.n_layers);
    p->w.layers = p->lw.data();
    p->sh.resize(d.n_layers);
    unsigned char* cur = (unsigned char*)shadow;
    const int EB = p->EB, HD = p->HD, dm = d.d_model, dmlp = d.d_mlp, H = d.n_heads, dh = d.d_head;
    for (int l = 0; l < d.n_layers; ++l) {
        const pv_vit_layer_weights& L = p->lw[l];
        PV_REQUIRE(L.ln1_w && L.ln1_b && L.W_Q && L.W_K && L.W_V && L.b_Q && L.b_K && L.b_V && L.W_O && L.b_O &&
                       L.ln2_w && L.ln2_b && L.W_in && L.b_in && L.W_out && L.b_out,
                   "missing block weights");
        LayerShadow& S = p->sh[l];
        S.Wqkv = cur; cur += seg((size_t)3 * HD * dm, EB);
        S.WoT = cur;  cur += seg((size_t)dm * HD, EB);
        S.WinT = cur; cur += seg((size_t)dm * dmlp, EB);
        S.WoutT = cur; cur += seg((size_t)dm * dmlp, EB);
        // W_Q[h][k][e] -> Wqkv[(h*dh+e)][k]   (attention.py:216-220: x @ W[h] + b[h])
        int rc;
        if ((rc = pv_launch_transpose(EB, L.W_Q, S.Wqkv, H, dm, dh, stream))) return rc;
        if ((rc = pv_launch_transpose(EB, L.W_K, S.Wqkv + (size_t)HD * dm * EB, H, dm, dh, stream))) return rc;
        if ((rc = pv_launch_transpose(EB, L.W_V, S.Wqkv + (size_t)2 * HD * dm * EB, H, dm, dh, stream))) return rc;
        // W_O[h][e][n] == [K = h*dh+e][N = d] -> [N][K]   (attention.py:155-167)
        if ((rc = pv_launch_transpose(EB, L.W_O, S.WoT, 1, HD, dm, stream))) return rc;
        if ((rc = pv_launch_transpose(EB, L.W_in, S.WinT, 1, dm, dmlp, stream))) return rc;
        if ((rc = pv_launch_transpose(EB, L.W_out, S.WoutT, 1, dmlp, dm, stream))) return rc;
    }
    if (d.has_head) {
        p->WhT = cur;
        cur += seg((size_t)d.n_classes * dm, EB);
        int rc;
        if ((rc = pv_launch_transpose(EB, w->W_H, p->WhT, 1, dm, d.n_classes, stream))) return rc;
    }
    if (p->patch_prepack) {
        const int K = d.n_channels * d.patch_size * d.patch_size;
        if (p->Kp == K && pv_aligned16(w->conv_w)) {
            p->conv_wp = (unsigned char*)const_cast<void*>(w->conv_w);
        } else {
            p->conv_wp = cur;
            cur += seg((size_t)dm * p->Kp, EB);
            int rc;
            if ((rc = pv_launch_pad_rows_bf16(w->conv_w, p->conv_wp, dm, K, p->Kp, stream))) return rc;
        }
    }
    p->weights_set = true;
    return PV_OK;
}

namespace {
struct Workspace {
    size_t total;
    size_t embed, resid_a, resid_b, ln_out, q, k, v, z, resid_mid, mlp_post, lnf, head, patches;
};
Workspace carve(const pv_vit_plan* p, int B) {
    const pv_vit_desc& d = p->d;
    const size_t M = (size_t)B * p->T;
    const int EB = p->EB;
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t elems) { size_t o = off; off += seg(elems, EB); return o; };
    w.embed = take((size_t)B * p->P * d.d_model);
    w.resid_a = take(M * d.d_model);
    w.resid_b = take(M * d.d_model);
    w.ln_out = take(M * d.d_model);
    w.q = take(M * p->HD);
    w.k = take(M * p->HD);
    w.v = take(M * p->HD);
    w.z = take(M * p->HD);
    w.resid_mid = take(M * d.d_model);
    w.mlp_post = take(M * d.d_mlp);
    w.lnf = take(M * d.d_model);
    w.head = take((size_t)B * (d.has_head ? d.n_classes : d.d_model));
    w.patches = p->patch_prepack ? take((size_t)B * p->P * p->Kp) : 0;
    w.total = off + 256;
    return w;
}
}  // namespace

extern "C" size_t pv_vit_workspace_bytes(const pv_vit_plan* p, int32_t batch) {
    if (!p || batch <= 0) return 0;
    return carve(p, batch).total;
}

namespace {
// entry_stage / exit_stage: PV_STAGE_* (positions inside a block: entry, ln1, q/k/v ready, scores, pattern, z ready, resid_mid, ln2, mlp pre, mlp post ready)
int vit_forward_impl(pv_vit_plan* p, const void* images, const void* resid_in, const void* const* act_in, int32_t B,
                     int32_t first_block, int32_t entry_stage, int32_t n_blocks, int32_t exit_stage, int32_t run_head,
                     const pv_tap* taps, int32_t n_taps, void* workspace, size_t workspace_bytes, void* out, void* stream_);
}  // namespace

extern "C" int pv_vit_forward(pv_vit_plan* p, const void* images, int32_t B, int32_t n_blocks,
                              int32_t run_head, const pv_tap* taps, int32_t n_taps, void* workspace,
                              size_t workspace_bytes, void* out, void* stream_) {
    PV_REQUIRE(images, "null argument");
    return vit_forward_impl(p, images, nullptr, nullptr, B, 0, 0, n_blocks, 0, run_head, taps, n_taps, workspace, workspace_bytes, out, stream_);
}

extern "C" int pv_vit_forward_seg(pv_vit_plan* p, const void* images, const void* resid_in, int32_t B, int32_t first_block,
                                  int32_t entry_mid, int32_t end_block, int32_t exit_mid, int32_t run_head, const pv_tap* taps,
                                  int32_t n_taps, void* workspace, size_t workspace_bytes, void* out, void* stream_) {
    PV_REQUIRE(p && ((images != nullptr) != (resid_in != nullptr)), "exactly one of images / resid_in");
    PV_REQUIRE(!resid_in || pv_aligned16(resid_in), "resid_in must be 16-byte aligned");
    PV_REQUIRE(images ? (first_block == 0 && !entry_mid) : (first_block >= 0 && first_block <= p->d.n_layers), "segment start");
    PV_REQUIRE(first_block <= end_block && end_block <= p->d.n_layers, "segment end");
    PV_REQUIRE(!entry_mid || first_block < p->d.n_layers, "entry_mid needs a block to resume");
    PV_REQUIRE(!exit_mid || (end_block < p->d.n_layers && !run_head), "exit_mid stops inside block end_block");
    PV_REQUIRE(!(entry_mid && first_block == end_block && !exit_mid), "a mid-block entry must finish its block");
    return vit_forward_impl(p, images, resid_in, nullptr, B, first_block, entry_mid ? PV_STAGE_MID : 0, end_block,
                            exit_mid ? PV_STAGE_MID : 0, run_head, taps, n_taps, workspace, workspace_bytes, out, stream_);
}

extern "C" int pv_vit_forward_stage(pv_vit_plan* p, const void* images, const void* resid_in, const void* act_in0, const void* act_in1,
                                    const void* act_in2, int32_t B, int32_t first_block, int32_t entry_stage, int32_t end_block,
                                    int32_t exit_stage, int32_t run_head, const pv_tap* taps, int32_t n_taps, void* workspace,
                                    size_t workspace_bytes, void* out, void* stream_) {
    PV_REQUIRE(p && ((images != nullptr) != (resid_in != nullptr)), "exactly one of images / resid_in");
    PV_REQUIRE(!resid_in || pv_aligned16(resid_in), "resid_in must be 16-byte aligned");
    PV_REQUIRE(entry_stage >= 0 && entry_stage <= PV_STAGE_MLP_POST && exit_stage >= 0 && exit_stage <= PV_STAGE_MLP_POST, "stage");
    PV_REQUIRE(images ? (first_block == 0 && !entry_stage) : (first_block >= 0 && first_block <= p->d.n_layers), "segment start");
    PV_REQUIRE(first_block <= end_block && end_block <= p->d.n_layers, "segment end");
    PV_REQUIRE(!entry_stage || first_block < p->d.n_layers, "a mid-block entry needs a block to resume");
    PV_REQUIRE(!exit_stage || (end_block < p->d.n_layers && !run_head), "a mid-block exit stops inside block end_block");
    PV_REQUIRE(first_block < end_block || entry_stage < exit_stage || (first_block == end_block && !exit_stage && !entry_stage),
               "empty or backward segment");
    if (entry_stage == PV_STAGE_QKV) PV_REQUIRE(act_in0 && act_in1 && act_in2, "entry at PV_STAGE_QKV needs q, k, v");
    if (entry_stage == PV_STAGE_Z || entry_stage == PV_STAGE_MLP_POST || entry_stage == PV_STAGE_LN1 || entry_stage == PV_STAGE_LN2 ||
        entry_stage == PV_STAGE_MLP_PRE)
        PV_REQUIRE(act_in0, "entry at PV_STAGE_LN1 / Z / LN2 / MLP_PRE / MLP_POST needs the activation");
    if (entry_stage == PV_STAGE_SCORES || entry_stage == PV_STAGE_PATTERN) PV_REQUIRE(act_in0 && act_in1, "entry at PV_STAGE_SCORES / PV_STAGE_PATTERN needs the activation and v");
    const void* act[3] = {act_in0, act_in1, act_in2};
    for (int i = 0; i < 3; ++i) PV_REQUIRE(!act[i] || pv_aligned16(act[i]), "activation inputs must be 16-byte aligned");
    return vit_forward_impl(p, images, resid_in, act, B, first_block, entry_stage, end_block, exit_stage, run_head, taps, n_taps,
                            workspace, workspace_bytes, out, stream_);
}

extern "C" int pv_vit_forward_from(pv_vit_plan* p, const void* resid_in, int32_t B, int32_t first_block, int32_t n_blocks,
                                   int32_t run_head, const pv_tap* taps, int32_t n_taps, void* workspace,
                                   size_t workspace_bytes, void* out, void* stream_) {
    PV_REQUIRE(resid_in && pv_aligned16(resid_in), "resid_in must be a 16-byte aligned device pointer");
    PV_REQUIRE(p && first_block >= 0 && first_block <= p->d.n_layers && first_block <= n_blocks, "first_block out of range");
    return vit_forward_impl(p, nullptr, resid_in, nullptr, B, first_block, 0, n_blocks, 0, run_head, taps, n_taps, workspace, workspace_bytes, out, stream_);
}

namespace {
int vit_forward_impl(pv_vit_plan* p, const void* images, const void* resid_in, const void* const* act_in, int32_t B,
                     int32_t first_block, int32_t entry_stage, int32_t n_blocks, int32_t exit_stage, int32_t run_head,
                     const pv_tap* taps, int32_t n_taps, void* workspace, size_t workspace_bytes, void* out, void* stream_) {
    PV_REQUIRE(p && (images || resid_in) && workspace, "null argument");
    PV_REQUIRE(p->weights_set, "pv_vit_plan_set_weights has not been called");
    PV_REQUIRE(B > 0, "batch must be positive");
    const pv_vit_desc& d = p->d;
    PV_REQUIRE(n_blocks >= 0 && n_blocks <= d.n_layers, "n_blocks out of range");
    PV_REQUIRE(!run_head || n_blocks == d.n_layers, "run_head requires all blocks");
    PV_REQUIRE(!run_head || out, "run_head requires an output buffer");
    PV_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    PV_REQUIRE(!images || pv_aligned16(images), "images must be 16-byte aligned");
    const Workspace ws = carve(p, B);
    PV_REQUIRE(workspace_bytes >= ws.total, "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char* wsb = (unsigned char*)workspace;
    const int dt = d.dtype, T = p->T, dm = d.d_model, HD = p->HD, dmlp = d.d_mlp;
    const int M = B * T;
    const bool bf16 = dt == PV_DTYPE_BF16;

    // tap table
    std::vector<void*> tap((size_t)PV_SLOT__COUNT * (size_t)(d.n_layers + 1), nullptr);
    auto tap_at = [&](int slot, int layer) -> void*& { return tap[(size_t)layer * PV_SLOT__COUNT + slot]; };
    for (int i = 0; i < n_taps; ++i) {
        const pv_tap& t = taps[i];
        PV_REQUIRE(t.slot >= 0 && t.slot < PV_SLOT__COUNT && t.dst, "bad tap slot");
        const int layer = t.slot >= PV_SLOT_LN1_SCALE ? t.layer : 0;
        PV_REQUIRE(layer >= 0 && layer < d.n_layers + 1, "bad tap layer");
        PV_REQUIRE(pv_aligned16(t.dst), "tap destinations must be 16-byte aligned");
        if (!bf16) {
            PV_REQUIRE(t.slot != PV_SLOT_LNPRE_NORM_F32 && t.slot != PV_SLOT_LNF_NORM_F32 &&
                           t.slot != PV_SLOT_LN1_NORM_F32 && t.slot != PV_SLOT_LN2_NORM_F32,
                       "*_NORM_F32 slots exist only in bf16 mode");
        }
        tap_at(t.slot, layer) = t.dst;
    }
    auto pick = [&](int slot, int layer, size_t ws_off) -> void* {
        void* t = tap_at(slot, layer);
        return t ? t : (void*)(wsb + ws_off);
    };
    int rc;

    void* resid;
    if (resid_in) {
        // resumed forward (pv_vit_forward_from): the caller hands in the residual stream entering `first_block`
        // (the value a Python hook at a block boundary returned); the embedding stages are skipped
        resid = const_cast<void*>(resid_in);
    } else {
    // ---- patch embedding: stride-p conv as an im2col-free GEMM (patch_embedding.py:26-32)
    void* embed = pick(PV_SLOT_EMBED, 0, ws.embed);
    {
        GemmParams g = {};
        if (p->patch_prepack && ((uintptr_t)images & 3) == 0) {
            void* rows = wsb + ws.patches;
            if ((rc = pv_launch_patch_pack_bf16(images, rows, B, d.n_channels, d.image_size, d.patch_size, p->G, p->Kp, stream))) return rc;
            g.A = rows; g.lda = p->Kp; g.a_mode = PV_A_PLAIN; g.Bt = p->conv_wp; g.ldb = p->Kp; g.K = p->Kp;
        } else {
            g.A = images; g.a_mode = PV_A_PATCH; g.pC = d.n_channels; g.pP = d.patch_size; g.pS = d.image_size; g.pG = p->G;
            g.Bt = p->w.conv_w; g.ldb = (int64_t)d.n_channels * d.patch_size * d.patch_size;
            g.K = d.n_channels * d.patch_size * d.patch_size;
        }
        g.M = B * p->P; g.N = dm;
        g.epi = PV_EPI_BIAS; g.bias0 = p->w.conv_b; g.out0 = embed; g.ldo = dm;
        if ((rc = pv_launch_gemm(dt, g, stream))) return rc;
    }
    // ---- cls + pos (+ ln_pre) (base_vit.py:171-185)
    {
        LnParams L = {};
        L.x = embed; L.rows = M; L.d = dm; L.eps = d.eps; L.embed = 1; L.T = T; L.use_cls = d.use_cls_token ? 1 : 0;
        L.cls = p->w.cls_token; L.pos = p->w.W_pos;
        if (d.layer_norm_pre) {
            L.do_ln = 1; L.w = p->w.ln_pre_w; L.b = p->w.ln_pre_b;
            L.full_out = tap_at(PV_SLOT_FULL_EMBED, 0);
            L.scale_out = (float*)tap_at(PV_SLOT_LNPRE_SCALE, 0);
            L.norm_f32_out = (float*)tap_at(PV_SLOT_LNPRE_NORM_F32, 0);
            resid = pick(PV_SLOT_LNPRE_OUT, 0, ws.resid_a);
            L.out = resid;
        } else {
            L.do_ln = 0;
            resid = pick(PV_SLOT_FULL_EMBED, 0, ws.resid_a);
            L.full_out = resid;
        }
        if ((rc = pv_launch_ln(dt, L, stream))) return rc;
    }
    }
    bool resid_in_a = true;   // which workspace residual buffer may hold `resid`

    // blocks [first_block, n_blocks) in full; with exit_stage also block n_blocks up to that stage (the caller taps what it
    // needs there); with entry_stage the first block resumes behind that stage: `resid` is then the residual stream the
    // rest of the block adds to (resid_pre for PV_STAGE_QKV / PV_STAGE_Z, resid_mid for PV_STAGE_MID / PV_STAGE_MLP_POST) and
    // act_in holds the (hook-edited) activations of the stage: q, k, v | z | mlp post
    const int last_block = exit_stage ? n_blocks + 1 : n_blocks;
    for (int l = first_block; l < last_block; ++l) {
        const pv_vit_layer_weights& W = p->lw[l];
        const LayerShadow& S = p->sh[l];
        const int es = l == first_block ? entry_stage : 0;
        const int xs = (exit_stage && l == n_blocks) ? exit_stage : 99;
        void* resid_pre = resid;
        void* resid_mid = nullptr;
        void *q = nullptr, *k = nullptr, *v = nullptr, *z = nullptr;
        if (es >= PV_STAGE_MID) {
            resid_mid = resid;
        } else {
        void* ln1 = nullptr;
        if (es < PV_STAGE_LN1) {
        // ln1 (transformer_block.py:106-109 ; layer_norm.py:75-93)
        ln1 = pick(PV_SLOT_LN1_OUT, l, ws.ln_out);
        {
            LnParams L = {};
            L.x = resid_pre; L.ldx = dm; L.rows = M; L.d = dm; L.eps = d.eps; L.do_ln = 1;
            L.w = W.ln1_w; L.b = W.ln1_b;
            L.scale_out = (float*)tap_at(PV_SLOT_LN1_SCALE, l);
            L.norm_f32_out = (float*)tap_at(PV_SLOT_LN1_NORM_F32, l);
            L.out = ln1;
            if ((rc = pv_launch_ln(dt, L, stream))) return rc;
        }
        } else if (es == PV_STAGE_LN1) {
            // behind a hooked ln1.hook_scale / hook_normalized: the (edited) fp32 normalized tensor, rounded to the storage
            // dtype like the module's own last step (layer_norm.py:93)
            if (bf16) {
                ln1 = wsb + ws.ln_out;
                if ((rc = pv_launch_cast_from_f32(dt, (const float*)act_in[0], ln1, (int64_t)M * dm, stream))) return rc;
            } else {
                ln1 = const_cast<void*>(act_in[0]);
            }
        }
        if (xs == PV_STAGE_LN1) break;
        if (es < PV_STAGE_QKV) {
        // q, k, v (attention.py:186-244) as one GEMM against the packed [3*H*dh][d] shadow
        q = pick(PV_SLOT_Q, l, ws.q);
        k = pick(PV_SLOT_K, l, ws.k);
        v = pick(PV_SLOT_V, l, ws.v);
        {
            GemmParams g = {};
            g.A = ln1; g.lda = dm; g.a_mode = PV_A_PLAIN; g.Bt = S.Wqkv; g.ldb = dm;
            g.M = M; g.N = 3 * HD; g.K = dm; g.epi = PV_EPI_QKV; g.nsplit = HD;
            g.bias0 = W.b_Q; g.bias1 = W.b_K; g.bias2 = W.b_V; g.out0 = q; g.out1 = k; g.out2 = v; g.ldo = HD;
            if ((rc = pv_launch_gemm(dt, g, stream))) return rc;
        }
        } else if (es == PV_STAGE_QKV) {
            q = const_cast<void*>(act_in[0]); k = const_cast<void*>(act_in[1]); v = const_cast<void*>(act_in[2]);
        }
        if (xs == PV_STAGE_QKV) break;
        if (es < PV_STAGE_SCORES) {
        // scores / pattern / z (attention.py:135-152, 246-281)
        z = pick(PV_SLOT_Z, l, ws.z);
        {
            AttnParams a = {};
            a.q = q; a.k = k; a.v = v; a.z = z;
            a.scores = tap_at(PV_SLOT_SCORES, l); a.pattern = tap_at(PV_SLOT_PATTERN, l);
            a.B = B; a.T = T; a.H = d.n_heads; a.dh = d.d_head; a.attn_scale = d.attn_scale;
            if ((rc = pv_launch_attention(dt, a, stream))) return rc;
        }
        } else if (es == PV_STAGE_SCORES || es == PV_STAGE_PATTERN) {
            // behind a hooked hook_attn_scores / hook_pattern: the rest of the core from the edited tensor and v
            z = pick(PV_SLOT_Z, l, ws.z);
            AttnParams a = {};
            a.v = act_in[1]; a.z = z;
            if (es == PV_STAGE_SCORES) { a.scores = const_cast<void*>(act_in[0]); a.pattern = tap_at(PV_SLOT_PATTERN, l); }
            else a.pattern = const_cast<void*>(act_in[0]);
            a.B = B; a.T = T; a.H = d.n_heads; a.dh = d.d_head; a.attn_scale = d.attn_scale;
            if ((rc = pv_launch_attention_resume(dt, a, es == PV_STAGE_SCORES ? 1 : 0, stream))) return rc;
        } else {
            z = const_cast<void*>(act_in[0]);
        }
        if (xs == PV_STAGE_SCORES || xs == PV_STAGE_PATTERN || xs == PV_STAGE_Z) break;
        // attn_out = z W_O + b_O ; resid_mid = resid_pre + attn_out (attention.py:155-167 ; block :117-124)
        resid_mid = pick(PV_SLOT_RESID_MID, l, ws.resid_mid);
        {
            GemmParams g = {};
            g.A = z; g.lda = HD; g.a_mode = PV_A_PLAIN; g.Bt = S.WoT; g.ldb = HD;
            g.M = M; g.N = dm; g.K = HD; g.epi = PV_EPI_RESID; g.bias0 = W.b_O;
            g.out0 = tap_at(PV_SLOT_ATTN_OUT, l); g.out1 = resid_mid; g.ldo = dm; g.resid = resid_pre; g.ldr = dm;
            if ((rc = pv_launch_gemm(dt, g, stream))) return rc;
        }
        }
        if (xs == PV_STAGE_MID) {
            resid = resid_mid;
            break;
        }
        void* post;
        if (es < PV_STAGE_MLP_POST) {
        void* ln2 = nullptr;
        if (es < PV_STAGE_LN2) {
        // ln2 (block :130)
        ln2 = pick(PV_SLOT_LN2_OUT, l, ws.ln_out);
        {
            LnParams L = {};
            L.x = resid_mid; L.ldx = dm; L.rows = M; L.d = dm; L.eps = d.eps; L.do_ln = 1;
            L.w = W.ln2_w; L.b = W.ln2_b;
            L.scale_out = (float*)tap_at(PV_SLOT_LN2_SCALE, l);
            L.norm_f32_out = (float*)tap_at(PV_SLOT_LN2_NORM_F32, l);
            L.out = ln2;
            if ((rc = pv_launch_ln(dt, L, stream))) return rc;
        }
        } else if (es == PV_STAGE_LN2) {
            if (bf16) {
                ln2 = wsb + ws.ln_out;
                if ((rc = pv_launch_cast_from_f32(dt, (const float*)act_in[0], ln2, (int64_t)M * dm, stream))) return rc;
            } else {
                ln2 = const_cast<void*>(act_in[0]);
            }
        }
        if (xs == PV_STAGE_LN2) break;
        post = pick(PV_SLOT_MLP_POST, l, ws.mlp_post);
        if (es < PV_STAGE_MLP_PRE) {
        // mlp (mlp.py:65-80): pre -> act -> post
        {
            GemmParams g = {};
            g.A = ln2; g.lda = dm; g.a_mode = PV_A_PLAIN; g.Bt = S.WinT; g.ldb = dm;
            g.M = M; g.N = dmlp; g.K = dm; g.epi = PV_EPI_ACT; g.act = d.activation; g.bias0 = W.b_in;
            g.out0 = tap_at(PV_SLOT_MLP_PRE, l); g.out1 = post; g.ldo = dmlp;
            if ((rc = pv_launch_gemm(dt, g, stream))) return rc;
        }
        } else {
            // behind a hooked mlp.hook_pre: the activation function on the edited pre-activation (mlp.py:67-72)
            if ((rc = pv_launch_act(dt, d.activation, act_in[0], post, (int64_t)M * dmlp, stream))) return rc;
        }
        if (xs == PV_STAGE_MLP_PRE) break;
        } else {
            post = const_cast<void*>(act_in[0]);
        }
        if (xs == PV_STAGE_MLP_POST) break;
        // mlp_out ; resid_post = resid_mid + mlp_out (block :131-134)
        void* resid_post = tap_at(PV_SLOT_RESID_POST, l);
        if (!resid_post) {
            // the workspace buffer NOT holding resid_pre (resid_pre is dead after the O-proj, but keep
            // it simple and safe: ping-pong)
            resid_post = wsb + (resid_in_a ? ws.resid_b : ws.resid_a);
            resid_in_a = !resid_in_a;
        }
        {
            GemmParams g = {};
            g.A = post; g.lda = dmlp; g.a_mode = PV_A_PLAIN; g.Bt = S.WoutT; g.ldb = dmlp;
            g.M = M; g.N = dm; g.K = dmlp; g.epi = PV_EPI_RESID; g.bias0 = W.b_out;
            g.out0 = tap_at(PV_SLOT_MLP_OUT, l); g.out1 = resid_post; g.ldo = dm; g.resid = resid_mid; g.ldr = dm;
            if ((rc = pv_launch_gemm(dt, g, stream))) return rc;
        }
        resid = resid_post;
    }
    if (!run_head) return PV_OK;

    // ---- ln_final on all tokens only if one of its taps is requested, else on the CLS rows only
    const bool lnf_all = tap_at(PV_SLOT_LNF_SCALE, 0) || tap_at(PV_SLOT_LNF_NORM_F32, 0) || tap_at(PV_SLOT_LNF_OUT, 0);
    void* lnf = pick(PV_SLOT_LNF_OUT, 0, ws.lnf);
    int64_t lnf_ld;   // row stride (elements) between CLS rows of consecutive images
    {
        LnParams L = {};
        L.x = resid; L.d = dm; L.eps = d.eps; L.do_ln = 1; L.w = p->w.ln_final_w; L.b = p->w.ln_final_b;
        if (lnf_all) {
            L.ldx = dm; L.rows = M;
            L.scale_out = (float*)tap_at(PV_SLOT_LNF_SCALE, 0);
            L.norm_f32_out = (float*)tap_at(PV_SLOT_LNF_NORM_F32, 0);
            lnf_ld = (int64_t)T * dm;
        } else {
            L.ldx = (int64_t)T * dm; L.rows = B;
            lnf_ld = dm;
        }
        L.out = lnf;
        if ((rc = pv_launch_ln(dt, L, stream))) return rc;
    }
    // ---- cls row -> head (head.py:27-37) -> hook_post_head_pre_normalize -> F.normalize
    const int nout = d.has_head ? d.n_classes : dm;
    void* head_tap = tap_at(PV_SLOT_HEAD_OUT, 0);
    void* hb = head_tap ? head_tap : (d.normalize_output ? (void*)(wsb + ws.head) : out);
    if (d.has_head) {
        GemmParams g = {};
        g.A = lnf; g.lda = lnf_ld; g.a_mode = PV_A_PLAIN; g.Bt = p->WhT; g.ldb = dm;
        g.M = B; g.N = nout; g.K = dm; g.epi = PV_EPI_BIAS; g.bias0 = p->w.b_H; g.out0 = hb; g.ldo = nout;
        if ((rc = pv_launch_gemm(dt, g, stream))) return rc;
    } else {
        PV_HIP_CHECK(hipMemcpy2DAsync(hb, (size_t)dm * p->EB, lnf, (size_t)lnf_ld * p->EB, (size_t)dm * p->EB, B,
                                      hipMemcpyDeviceToDevice, stream));
    }
    if (d.normalize_output) {
        if ((rc = pv_launch_l2norm(dt, hb, out, B, nout, stream))) return rc;
    } else if (hb != out) {
        PV_HIP_CHECK(hipMemcpyAsync(out, hb, (size_t)B * nout * p->EB, hipMemcpyDeviceToDevice, stream));
    }
    return PV_OK;
}
}  // namespace

// ---- kernel-level entry points -------------------------------------------------------------
extern "C" int pv_gemm_bias(int32_t dtype, const void* A, int64_t lda, const void* Bt, int64_t ldb,
                            const void* bias, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                            void* stream) {
    GemmParams g = {};
    g.A = A; g.lda = lda; g.a_mode = PV_A_PLAIN; g.Bt = Bt; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    g.epi = PV_EPI_BIAS; g.bias0 = bias; g.out0 = C; g.ldo = ldc;
    return pv_launch_gemm(dtype, g, (hipStream_t)stream);
}

extern "C" int pv_gemm_epilogue(int32_t dtype, int32_t epi, int32_t act, const void* A, int64_t lda, const void* Bt, int64_t ldb,
                                const void* bias, const void* resid, int64_t ldr, void* out0, void* out1, int64_t ldo,
                                int32_t M, int32_t N, int32_t K, void* stream) {
    PV_REQUIRE(epi == PV_GEMM_EPI_BIAS || epi == PV_GEMM_EPI_RESID || epi == PV_GEMM_EPI_ACT, "pv_gemm_epilogue: epi");
    GemmParams g = {};
    g.A = A; g.lda = lda; g.a_mode = PV_A_PLAIN; g.Bt = Bt; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    g.epi = epi; g.act = act; g.bias0 = bias; g.out0 = out0; g.out1 = out1; g.ldo = ldo; g.resid = resid; g.ldr = ldr;
    return pv_launch_gemm(dtype, g, (hipStream_t)stream);
}

extern "C" int pv_transpose_batched(int32_t elem_bytes, const void* in, void* out, int32_t batch,
                                    int32_t R, int32_t C, void* stream) {
    PV_REQUIRE(in && out, "null argument");
    return pv_launch_transpose(elem_bytes, in, out, batch, R, C, (hipStream_t)stream);
}
