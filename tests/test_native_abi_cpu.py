"""CPU checks of the C-ABI library: it builds/loads without a GPU and exports every symbol that
include/pv_native.h declares (no compute calls here)."""
import ctypes as C
import os
import re

from vit_prisma_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pv_native.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pv_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _native.lib()
    assert lib.pv_abi_version() == _native.ABI_VERSION
    syms = declared_symbols()
    assert len(syms) >= 18
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_native.EXPORTS) == syms       # the ctypes binding covers exactly the header
    # the library is a prebuilt file next to the sources: it must have been built from exactly these sources
    assert _native.build_id() == _native.source_id(), "libpvnative.so is stale: run python -m vit_prisma_amd.build"


def test_plan_queries_and_error_reporting_without_gpu():
    lib = _native.lib()
    desc = _native.VitDesc(n_layers=12, d_model=768, n_heads=12, d_head=64, d_mlp=3072, n_channels=3, patch_size=32,
                           image_size=224, n_tokens=50, n_classes=512, use_cls_token=1, layer_norm_pre=1, has_head=1,
                           normalize_output=1, activation=0, dtype=1, eps=1e-5, attn_scale=8.0)
    plan = C.c_void_p()
    assert lib.pv_vit_plan_create(C.byref(desc), C.byref(plan)) == 0
    assert lib.pv_vit_shadow_bytes(plan) > 2 * 85_000_000          # bf16 shadow of the 85 M matrix weights
    assert lib.pv_vit_workspace_bytes(plan, 512) > 0
    lib.pv_vit_plan_destroy(plan)
    desc.n_tokens = 49                                             # inconsistent with patches + cls
    assert lib.pv_vit_plan_create(C.byref(desc), C.byref(plan)) == 1
    assert "n_tokens" in _native.last_error()
    sdesc = _native.SaeDesc(d_in=768, d_sae=24576, k=32, normalize_layer_norm=1, max_tokens=4096, ln_eps=1e-5)
    splan = C.c_void_p()
    assert lib.pv_sae_plan_create(C.byref(sdesc), C.byref(splan)) == 0
    assert lib.pv_sae_workspace_bytes(splan) > 4096 * 24576 * 4
    lib.pv_sae_plan_destroy(splan)
    sdesc.k = 100                                  # (beyond the filtered encoder's 64: the exact encoder serves the plan)
    assert lib.pv_sae_plan_create(C.byref(sdesc), C.byref(splan)) == 0 and lib.pv_sae_encoder_is_filtered(splan) == 0
    lib.pv_sae_plan_destroy(splan)
    sdesc.k = 300
    assert lib.pv_sae_plan_create(C.byref(sdesc), C.byref(splan)) == 1
    sdesc.k, sdesc.d_in = 32, 1284
    assert lib.pv_sae_plan_create(C.byref(sdesc), C.byref(splan)) == 1
