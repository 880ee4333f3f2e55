"""vit_prisma_amd -- MI355X-native run_with_cache + SAE training path behind ViT-Prisma's API.

    from vit_prisma_amd import HookedViT, HookedViTConfig, ActivationCache, HookPoint

The arithmetic of the hot path lives in libpvnative.so (hand-written HIP for gfx950, C ABI in
include/pv_native.h); this package is the host-side mirror of the reference's Python interface.
"""
from .activation_cache import ActivationCache
from .configs import HookedViTConfig
from .hook_points import HookPoint, LensHandle
from .hooked_root_module import HookedRootModule
from .compat import install_as
from .vit import (Attention, Head, HookedViT, LayerNorm, LayerNormPre, MLP, PatchEmbedding, PosEmbedding,
                  TransformerBlock)
from .sae_vit import HookedSAEViT

__all__ = [
    "ActivationCache", "HookedViTConfig", "HookPoint", "LensHandle", "HookedRootModule", "HookedViT", "HookedSAEViT",
    "Attention", "Head", "LayerNorm", "LayerNormPre", "MLP", "PatchEmbedding", "PosEmbedding",
    "TransformerBlock", "install_as",
]
