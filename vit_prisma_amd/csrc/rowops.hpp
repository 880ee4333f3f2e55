#pragma once
#include "pv_common.hpp"

struct LnParams {
    const void* x;        // [rows][ldx] T  (EMBED: patch embeddings [B*(T-cls)][d])
    int64_t ldx;
    const void* w;        // [d] T
    const void* b;        // [d] T
    int32_t rows, d;
    float eps;
    float* scale_out;     // [rows] fp32 or NULL                      (hook_scale)
    float* norm_f32_out;  // [rows][d] fp32 or NULL (bf16 mode tap)   (hook_normalized)
    void* out;            // [rows][d] T or NULL                      (LN output in storage dtype)
    // embed-assembly mode
    int32_t embed;        // 1: build rows from cls/patch + pos
    int32_t do_ln;        // 0: only assemble (layer_norm_pre == False)
    int32_t T;            // tokens per image
    int32_t use_cls;
    const void* cls;      // [d] T
    const void* pos;      // [T][d] T
    void* full_out;       // [rows][d] T or NULL                      (hook_full_embed)
};

int pv_launch_ln(int dtype, const LnParams& p, hipStream_t stream);
// out[i] = T(in[i]) (round to the storage dtype; a plain copy in fp32 mode): the LayerNorm output rebuilt from an edited
// hook_normalized tensor (pv_vit_forward_stage, PV_STAGE_LN1 / PV_STAGE_LN2).  n % 4 == 0.
int pv_launch_cast_from_f32(int dtype, const float* in, void* out, int64_t n, hipStream_t stream);
int pv_launch_l2norm(int dtype, const void* x, void* out, int rows, int n, hipStream_t stream);
int pv_launch_transpose(int elem_bytes, const void* in, void* out, int batch, int R, int C, hipStream_t stream);
// bf16 NCHW images -> [B*G*G][Kp] patch rows (im2col once per batch, zero-padded to Kp columns), and the matching row padding
// of the [d][K] patch-embedding weights: the operands of the tiled GEMM for patch sizes its in-kernel gather does not cover
int pv_launch_patch_pack_bf16(const void* images, void* out, int B, int C, int S, int p, int G, int Kp, hipStream_t stream);
int pv_launch_pad_rows_bf16(const void* in, void* out, int R, int K, int Kp, hipStream_t stream);
