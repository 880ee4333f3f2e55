// MFMA GEMM with fused bias / activation / residual / tap-store epilogues (gfx950).
#pragma once
#include "pv_common.hpp"

enum { PV_EPI_BIAS = 0, PV_EPI_QKV = 1, PV_EPI_RESID = 2, PV_EPI_ACT = 3 };
enum { PV_A_PLAIN = 0, PV_A_PATCH = 1 };

// C[M,N] = A[M,K] * Bt[N,K]^T  (+ epilogue).  All tensors have element type T (fp32 | bf16),
// accumulation is fp32 on the matrix cores.
struct GemmParams {
    // A operand
    const void* A;
    int64_t lda;             // elements (PLAIN)
    int32_t a_mode;          // PV_A_PLAIN | PV_A_PATCH (im2col-free patch gather from NCHW)
    int32_t pC, pP, pS, pG;  // PATCH: channels, patch size, image size, patches per row
    // B operand, [N][K] K-contiguous ("Bt")
    const void* Bt;
    int64_t ldb;             // elements
    int32_t b_kn;            // 1: B is given as [K][N] row-major (N-contiguous), fp32 only (SAE W_enc)
    int32_t M, N, K;
    // epilogue
    int32_t epi;             // PV_EPI_*
    int32_t act;             // PV_ACT_* (EPI_ACT)
    int32_t nsplit;          // EPI_QKV: columns per output
    const void* bias0;       // [N] (or [nsplit] x3 for QKV); may be NULL
    const void* bias1;
    const void* bias2;
    void* out0;              // BIAS: C ; QKV: q ; RESID: (acc+bias) tap or NULL ; ACT: pre tap or NULL
    void* out1;              // QKV: k ; RESID: resid + (acc+bias) ; ACT: act(pre)
    void* out2;              // QKV: v
    int64_t ldo;             // elements, all outputs
    const void* resid;       // RESID: [M][ldr]
    int64_t ldr;
    int32_t vec_out;         // set by the launcher: 16-byte vector epilogue legal
    uint64_t* trace;         // debug: per-workgroup {t_start, t_loop_end, t_end, hw_id} (100 MHz wall clock) or NULL
    int32_t dbg;             // ablation switches for kernel tuning (PV_GEMM_DBG): 1 = no DMA in the loop, 2 = no epilogue
    int32_t cus;             // CUs this launch may count on (a plan's pipeline launches on CU-masked streams): 0 = the device's
};

// dtype: PV_DTYPE_*.  Returns PV_OK / error code (pv_last_error has the message).
int pv_launch_gemm(int dtype, GemmParams p, hipStream_t stream);
// post[i] = act(pre[i]) with the instruction sequence of the GEMM epilogues (PV_EPI_ACT): the MLP resumed behind an edited
// mlp.hook_pre (pv_vit_forward_stage, PV_STAGE_MLP_PRE).  n % 8 == 0, 16-byte aligned pointers.
int pv_launch_act(int dtype, int act, const void* pre, void* post, int64_t n, hipStream_t stream);

// debug: arm per-workgroup phase tracing for the `launch_idx`-th GEMM launch from now (0 = next), read it back
// (blocks until the device is idle).  info = {M, N, K, epi, n_workgroups, kernel version}.
extern "C" int pv_debug_gemm_trace_arm(int32_t launch_idx);
extern "C" int pv_debug_gemm_trace_read(uint64_t* host_out, int32_t max_wg, int32_t* info6);
