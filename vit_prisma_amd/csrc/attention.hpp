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
