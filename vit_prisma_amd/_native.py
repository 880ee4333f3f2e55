"""ctypes binding of libpvnative.so (C ABI declared in include/pv_native.h).

The product path has no CPU or PyTorch-eager substitute for these kernels: if the library is
missing or its ABI version is wrong, ``lib()`` raises -- loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PV_NATIVE_LIB") or os.path.join(HERE, "libpvnative.so")     # (override: kernel A/B builds)
ABI_VERSION = 23

PV_DTYPE_F32, PV_DTYPE_BF16 = 0, 1
PV_ACT = {"gelu": 0, "quick_gelu": 1, "relu": 2}

# enum pv_slot
SLOT = dict(
    EMBED=0, FULL_EMBED=1, LNPRE_SCALE=2, LNPRE_NORM_F32=3, LNPRE_OUT=4,
    LNF_SCALE=5, LNF_NORM_F32=6, LNF_OUT=7, HEAD_OUT=8,
    LN1_SCALE=16, LN1_NORM_F32=17, LN1_OUT=18, Q=19, K=20, V=21, SCORES=22, PATTERN=23, Z=24,
    ATTN_OUT=25, RESID_MID=26, LN2_SCALE=27, LN2_NORM_F32=28, LN2_OUT=29, MLP_PRE=30, MLP_POST=31,
    MLP_OUT=32, RESID_POST=33,
)


class NativeError(RuntimeError):
    pass


class VitDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_layers", "d_model", "n_heads", "d_head", "d_mlp", "n_channels", "patch_size", "image_size",
        "n_tokens", "n_classes", "use_cls_token", "layer_norm_pre", "has_head", "normalize_output",
        "activation", "dtype")] + [("eps", C.c_float), ("attn_scale", C.c_float)]


class VitLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_w", "ln1_b", "W_Q", "W_K", "W_V", "b_Q", "b_K", "b_V", "W_O", "b_O", "ln2_w", "ln2_b",
        "W_in", "b_in", "W_out", "b_out")]


class VitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "cls_token", "conv_w", "conv_b", "W_pos", "ln_pre_w", "ln_pre_b", "ln_final_w", "ln_final_b",
        "W_H", "b_H")] + [("layers", C.POINTER(VitLayerWeights))]


class Tap(C.Structure):
    _fields_ = [("slot", C.c_int32), ("layer", C.c_int32), ("dst", C.c_void_p)]


class SaeDesc(C.Structure):
    _fields_ = [("d_in", C.c_int32), ("d_sae", C.c_int32), ("k", C.c_int32),
                ("normalize_layer_norm", C.c_int32), ("max_tokens", C.c_int32), ("ln_eps", C.c_float),
                ("activation", C.c_int32), ("lp_norm", C.c_float)]


class SaeTranscoder(C.Structure):
    """pv_sae_transcoder: all NULL = a plain autoencoder."""
    _fields_ = [(n, C.c_void_p) for n in ("b_dec_out", "gb_dec_out", "mb_dec_out", "vb_dec_out", "W_skip", "gW_skip", "mW_skip",
                                           "vW_skip", "target", "scratch")] + [("scratch_bytes", C.c_size_t),
                                                                                  ("d_in_true", C.c_int32), ("d_out_true", C.c_int32)]


class SaeGated(C.Structure):
    """pv_sae_gated: all NULL = not a gated SAE."""
    _fields_ = [(n, C.c_void_p) for n in ("b_gate", "r_mag", "b_mag", "gb_gate", "gr_mag", "gb_mag", "mb_gate", "mr_mag", "mb_mag",
                                           "vb_gate", "vr_mag", "vb_mag", "scratch")] + [("scratch_bytes", C.c_size_t)]


class SaeState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "W_enc", "W_dec", "b_enc", "b_dec", "gW_enc", "gW_dec", "gb_enc", "gb_dec",
        "mW_enc", "mW_dec", "mb_enc", "mb_dec", "vW_enc", "vW_dec", "vb_enc", "vb_dec",
        "act_freq_scores", "n_fwd_since_fired", "W_encT", "W_enc16T", "enc_colsq", "dec_inv_norm")] + [("tc", SaeTranscoder), ("gt", SaeGated)]


class SaeGhost(C.Structure):
    _fields_ = [("n_dead", C.c_int32), ("dead_idx", C.c_void_p), ("dead_slot", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("err_colmean", C.c_void_p), ("mse_global", C.c_void_p), ("n_global", C.c_int32),
                ("reserved", C.c_int32)]


class SaeReluSparse(C.Structure):
    _fields_ = [("cap", C.c_int32), ("reserved", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class SaeOut(C.Structure):
    _fields_ = [("sae_out", C.c_void_p), ("topk_idx", C.c_void_p), ("topk_val", C.c_void_p),
                ("scalars", C.c_void_p), ("fire_count", C.c_void_p)]


_lib: Optional[C.CDLL] = None

# every symbol include/pv_native.h declares
EXPORTS = [
    "pv_abi_version", "pv_build_id", "pv_last_error",
    "pv_vit_plan_create", "pv_vit_plan_destroy", "pv_vit_shadow_bytes", "pv_vit_plan_set_weights",
    "pv_vit_workspace_bytes", "pv_vit_forward", "pv_vit_forward_from", "pv_vit_forward_seg", "pv_vit_forward_stage", "pv_gemm_bias", "pv_gemm_epilogue", "pv_transpose_batched",
    "pv_prof_enable", "pv_prof_reset", "pv_prof_read", "pv_prof_read_tag",
    "pv_sae_plan_create", "pv_sae_plan_destroy", "pv_sae_workspace_bytes", "pv_sae_renorm_decoder",
    "pv_sae_step", "pv_sae_grad_sqnorm", "pv_sae_grad_sqnorm_step", "pv_sae_grad_sqnorm_rows", "pv_sae_apply", "pv_sae_encode_topk",
    "pv_sae_sync_shadows", "pv_sae_encoder_is_filtered", "pv_debug_sae_ws_offset", "pv_sae_forward",
    "pv_sae_tp_partial", "pv_sae_tp_finish", "pv_sae_tp_merge", "pv_sae_tp_bucket_pack", "pv_sae_tp_bucket_unpack",
    "pv_sae_dense_step", "pv_sae_topk_ghost", "pv_sae_relu_step", "pv_sae_relu_workspace_bytes", "pv_debug_sae_relu_offset", "pv_sae_ghost_workspace_bytes", "pv_sae_transcoder_scratch_bytes", "pv_sae_gated_scratch_bytes", "pv_sae_gated_step", "pv_sae_gated_sparse_workspace_bytes", "pv_sae_gated_step_sparse", "pv_sae_gated_topk_scratch_bytes", "pv_sae_gated_topk_step",
    "pv_debug_gemm_trace_arm", "pv_debug_gemm_trace_read", "pv_debug_set_tuning", "pv_debug_get_tuning",
    "pv_clip_preprocess",
]


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
            "(run `python -m vit_prisma_amd.build`). There is no fallback for the native path.")
    L = C.CDLL(LIB_PATH)
    L.pv_abi_version.restype = C.c_int
    if L.pv_abi_version() != ABI_VERSION:
        raise NativeError(f"libpvnative ABI {L.pv_abi_version()} != binding {ABI_VERSION}; rebuild")
    L.pv_build_id.restype = C.c_char_p
    # a prebuilt library with the right ABI number but built from OTHER sources than the ones beside it (a stale .so that
    # travelled with an edited tree) must not load silently: include/pv_native.h promises it
    from .build import source_id
    got, want = L.pv_build_id().decode(), source_id()
    if got != want and os.environ.get("PV_ALLOW_STALE_LIB", "0") != "1":
        raise NativeError(f"{LIB_PATH} was built from other sources (build id {got}, tree {want}): run "
                          "`python -m vit_prisma_amd.build` (PV_ALLOW_STALE_LIB=1 loads it anyway)")
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    L.pv_last_error.argtypes = [C.c_char_p, sz]
    L.pv_last_error.restype = None
    L.pv_vit_plan_create.argtypes = [C.POINTER(VitDesc), C.POINTER(vp)]
    L.pv_vit_plan_destroy.argtypes = [vp]
    L.pv_vit_plan_destroy.restype = None
    L.pv_vit_shadow_bytes.argtypes = [vp]
    L.pv_vit_shadow_bytes.restype = sz
    L.pv_vit_plan_set_weights.argtypes = [vp, C.POINTER(VitWeights), vp, sz, vp]
    L.pv_vit_workspace_bytes.argtypes = [vp, i32]
    L.pv_vit_workspace_bytes.restype = sz
    L.pv_vit_forward.argtypes = [vp, vp, i32, i32, i32, C.POINTER(Tap), i32, vp, sz, vp, vp]
    L.pv_vit_forward_from.argtypes = [vp, vp, i32, i32, i32, i32, C.POINTER(Tap), i32, vp, sz, vp, vp]
    L.pv_vit_forward_seg.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(Tap), i32, vp, sz, vp, vp]
    L.pv_vit_forward_stage.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(Tap), i32, vp, sz, vp, vp]
    L.pv_gemm_bias.argtypes = [i32, vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, vp]
    L.pv_gemm_epilogue.argtypes = [i32, i32, i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, i64, i32, i32, i32, vp]
    L.pv_transpose_batched.argtypes = [i32, vp, vp, i32, i32, i32, vp]
    L.pv_prof_enable.argtypes = [i32]
    L.pv_prof_read.argtypes = [i32, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_double)]
    L.pv_debug_set_tuning.argtypes = [C.c_char_p, i32]
    L.pv_debug_get_tuning.argtypes = [C.c_char_p, C.POINTER(i32)]
    if hasattr(L, "pv_sae_plan_create"):
        L.pv_sae_plan_create.argtypes = [C.POINTER(SaeDesc), C.POINTER(vp)]
        L.pv_sae_plan_destroy.argtypes = [vp]
        L.pv_sae_plan_destroy.restype = None
        L.pv_sae_workspace_bytes.argtypes = [vp]
        L.pv_sae_workspace_bytes.restype = sz
        L.pv_sae_renorm_decoder.argtypes = [vp, C.POINTER(SaeState), vp]
        L.pv_sae_step.argtypes = [vp, C.POINTER(SaeState), vp, i32, vp, i32, i32, C.POINTER(SaeOut), vp, sz, vp]
        L.pv_sae_tp_partial.argtypes = [vp, C.POINTER(SaeState), vp, vp, i32, i32, vp, vp]
        L.pv_sae_tp_finish.argtypes = [vp, C.POINTER(SaeState), vp, vp, vp, vp, i32, i32, i32, C.POINTER(SaeOut), vp, sz, vp]
        L.pv_sae_dense_step.argtypes = [vp, C.POINTER(SaeState), vp, i32, vp, i32, i32, C.c_float, C.POINTER(SaeGhost),
                                        C.POINTER(SaeOut), vp, sz, vp]
        L.pv_sae_topk_ghost.argtypes = [vp, C.POINTER(SaeState), vp, i32, C.POINTER(SaeGhost), C.POINTER(SaeOut), vp, sz, vp]
        L.pv_sae_relu_step.argtypes = [vp, C.POINTER(SaeState), vp, i32, vp, i32, i32, C.c_float, C.POINTER(SaeReluSparse),
                                       C.POINTER(SaeOut), vp, sz, vp]
        L.pv_sae_relu_workspace_bytes.argtypes = [vp, i32, i32]
        L.pv_sae_relu_workspace_bytes.restype = sz
        L.pv_debug_sae_relu_offset.argtypes = [vp, i32, i32, C.c_char_p]
        L.pv_debug_sae_relu_offset.restype = sz
        L.pv_sae_ghost_workspace_bytes.argtypes = [vp, i32, i32]
        L.pv_sae_ghost_workspace_bytes.restype = sz
        L.pv_sae_transcoder_scratch_bytes.argtypes = [vp, i32]
        L.pv_sae_transcoder_scratch_bytes.restype = sz
        L.pv_sae_gated_scratch_bytes.argtypes = [vp, i32]
        L.pv_sae_gated_scratch_bytes.restype = sz
        L.pv_sae_gated_step.argtypes = [vp, C.POINTER(SaeState), vp, i32, vp, i32, i32, C.c_float, C.POINTER(SaeOut), vp, sz, vp]
        L.pv_sae_gated_topk_scratch_bytes.argtypes = [vp, i32]
        L.pv_sae_gated_topk_scratch_bytes.restype = sz
        L.pv_sae_gated_topk_step.argtypes = [vp, C.POINTER(SaeState), vp, i32, vp, i32, i32, C.POINTER(SaeOut), vp, sz, vp, sz, vp]
        L.pv_sae_gated_sparse_workspace_bytes.argtypes = [vp, i32, i32]
        L.pv_sae_gated_sparse_workspace_bytes.restype = sz
        L.pv_sae_gated_step_sparse.argtypes = [vp, C.POINTER(SaeState), vp, i32, vp, i32, i32, C.c_float, C.POINTER(SaeReluSparse),
                                               C.POINTER(SaeOut), vp, sz, vp]
        L.pv_sae_tp_merge.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
        L.pv_sae_tp_bucket_pack.argtypes = [vp, vp, vp, vp, i32, i32, vp]
        L.pv_sae_tp_bucket_unpack.argtypes = [vp, vp, vp, vp]
        L.pv_sae_grad_sqnorm.argtypes = [vp, i64, vp, vp, vp]
        L.pv_sae_grad_sqnorm_step.argtypes = [vp, C.POINTER(SaeState), vp, vp, vp]
        L.pv_sae_grad_sqnorm_rows.argtypes = [vp, C.POINTER(SaeState), i32, i32, i32, vp, vp, vp]
        L.pv_sae_apply.argtypes = [vp, C.POINTER(SaeState), vp, C.c_float, C.c_float, i32, i32, i32, vp]
        L.pv_sae_sync_shadows.argtypes = [vp, C.POINTER(SaeState), i32, i32, i32, vp]
        L.pv_sae_encoder_is_filtered.argtypes = [vp]
        L.pv_sae_forward.argtypes = [vp, C.POINTER(SaeState), vp, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.pv_debug_sae_ws_offset.argtypes = [vp, C.c_char_p]
        L.pv_clip_preprocess.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32,
                                         C.POINTER(C.c_float), C.POINTER(C.c_float), i32, vp, vp]
        L.pv_debug_sae_ws_offset.restype = sz
        L.pv_sae_encode_topk.argtypes = [vp, C.POINTER(SaeState), vp, i32, vp, vp, vp, vp, vp, sz, vp]
    _lib = L
    return L


def build_id() -> str:
    """Source hash baked into the loaded library (pv_build_id)."""
    return lib().pv_build_id().decode()


def source_id() -> str:
    """The same hash computed from the sources next to this file (vit_prisma_amd/build.py)."""
    from .build import source_id as _sid
    return _sid()


def built_from_these_sources() -> bool:
    """True when the loaded libpvnative.so was built from exactly the sources in this tree."""
    return build_id() == source_id()


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().pv_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeError(f"{what} failed (status {rc}): {last_error()}")


PROF_KINDS = {"gemm": 0, "attention": 1, "layernorm": 2, "sae_encode_topk": 3, "sae_backward": 4,
              "sae_apply": 5, "misc": 6}


def prof_enable(on: bool = True, kinds=None) -> None:
    """HIP-event timing of the kernel families in ``kinds`` (names of PROF_KINDS; None = all)."""
    mask = 0
    for k in kinds or ():
        mask |= 1 << PROF_KINDS[k]
    lib().pv_prof_enable(int(bool(on)) | (mask << 8))


def prof_reset() -> None:
    lib().pv_prof_reset()


def prof_read(kind: str) -> dict:
    """{'launches', 'ms', 'flops', 'bytes'} of one kernel family since the last reset."""
    n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
    check(lib().pv_prof_read(PROF_KINDS[kind], C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "pv_prof_read")
    return {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}


GEMM_TAGS = {"qkv": 1, "o_proj": 2, "mlp1": 3, "mlp2": 4, "other": 0}


def prof_read_tag(kind: str, tag: int) -> dict:
    """``prof_read`` restricted to the launches that carried one instance tag (GEMM_TAGS)."""
    n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
    check(lib().pv_prof_read_tag(PROF_KINDS[kind], int(tag), C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "pv_prof_read_tag")
    return {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}


def set_tuning(key: str, value: int = 0) -> None:
    """Kernel-choice override for tests / A-B measurements (``key="reset"`` restores every default).  The library
    never reads the environment on the launch path; this is the only switch."""
    check(lib().pv_debug_set_tuning(key.encode(), int(value)), "pv_debug_set_tuning")


def get_tuning(key: str = "any") -> int:
    v = C.c_int32()
    check(lib().pv_debug_get_tuning(key.encode(), C.byref(v)), "pv_debug_get_tuning")
    return v.value
