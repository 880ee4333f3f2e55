#pragma once
#include "pv_common.hpp"

struct AttnParams {
    const void *q, *k, *v;     // [B, T, H, dh] T
    void* scores;              // [B, H, T, T] T or NULL (tap)
    void* pattern;             // [B, H, T, T] T or NULL (tap)
    void* z;                   // [B, T, H, dh] T
    int32_t B, T, H, dh;
    float attn_scale;
    int32_t Tpad;              // filled by the launcher
};

int pv_attention_supported(int T, int dh);
int pv_launch_attention(int dtype, AttnParams p, hipStream_t stream);
// The attention core resumed behind a hooked activation (pv_vit_forward_stage, PV_STAGE_SCORES / PV_STAGE_PATTERN):
// from_scores: p.scores is the INPUT (the edited hook_attn_scores tensor): pattern = softmax, NaN -> 0 (attention.py:148-150),
// written to p.pattern when non-NULL, z = pattern v; otherwise p.pattern is the INPUT (the edited hook_pattern tensor) and
// z = pattern v.  q / k are unused.  A simple one-wave-per-query-row kernel: interventions, not the all-hooks hot path.
int pv_launch_attention_resume(int dtype, AttnParams p, int from_scores, hipStream_t stream);
