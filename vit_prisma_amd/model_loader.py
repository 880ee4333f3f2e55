"""``load_hooked_model`` for the two target architectures, OFFLINE (drop-in for the notebook entry point
/root/reference/src/vit_prisma/models/model_loader.py:278-368; aliased as ``vit_prisma.models.model_loader`` by ``install_as``).

The reference resolves ``model_name`` to a config and to weights by downloading from HuggingFace (model_loader.py:397-407, 775-784);
there is no network here.  This loader therefore takes the config from the architecture tables of ``synth.ARCHS`` plus the
per-name overrides of the reference registry (``NAME_OVERRIDES``) -- what ``load_config`` would have produced for these names, SURVEY.md 8a -- and the weights from a LOCAL open_clip / HuggingFace CLIP checkpoint
(``local_path=...``: a ``.safetensors`` / ``.pt`` / ``.bin`` file, converted by ``weights.py`` exactly as the reference's converters do) --
or none (``pretrained=False``: the reference's initialisation).  Everything else about the signature is the reference's; the options
that rewrite weights (``fold_ln``, ``center_writing_weights``, ``refactor_factored_attn_matrices``) are not implemented and raise
when set; ``fold_value_biases`` (on by default, as in the reference: b_O absorbs the value biases, model_loader.py:286, 352-358 ->
base_vit.py:498-532) IS applied to loaded weights -- ``attn.hook_v`` / ``hook_z`` of a loaded model are those of the folded weights.
"""
from __future__ import annotations

import logging
from typing import Any, Optional, Type

import torch

from .configs import HookedViTConfig
from .synth import ARCHS
from .vit import HookedViT
from .weights import load_clip_vision_weights

# reference model names (model_loader.py:97-123) -> architecture table
MODEL_ARCH = {
    "open-clip:laion/CLIP-ViT-B-32-DataComp.XL-s13B-b90K": "clip-vit-b32",
    "open-clip:laion/CLIP-ViT-B-32-DataComp.M-s128M-b4K": "clip-vit-b32",
    "open-clip:laion/CLIP-ViT-B-32-DataComp.S-s13M-b4K": "clip-vit-b32",
    "open-clip:laion/CLIP-ViT-B-32-laion2B-s34B-b79K": "clip-vit-b32",
    "openai/clip-vit-base-patch32": "clip-vit-b32",
    "openai/clip-vit-large-patch14-336": "clip-vit-l14-336",
}
# per-name overrides the reference registry applies on top of the architecture (models/model_config_registry.py:83-92, applied by
# its load_config :201-203): the HuggingFace OpenAI B/32 entry runs with eps = 1e-6 and WITHOUT the L2-normalised output, unlike the
# open_clip B/32 entries (BASE_OPEN_CLIP_CONFIGS["ViT-B"], :43-49).  Pinned by tests/golden/model_registry.json (values read out of
# the reference's registry by tests/golden/gen_golden_model_registry.py).
NAME_OVERRIDES = {
    "openai/clip-vit-base-patch32": {"eps": 1e-6, "normalize_output": False},
}
DTYPE_FROM_STRING = {"float32": torch.float32, "fp32": torch.float32, "float16": torch.float16, "fp16": torch.float16,
                     "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}


def list_available_models():
    return sorted(MODEL_ARCH)


def load_config(model_name: str, dtype: torch.dtype = torch.float32, device: str = "cuda", **overrides) -> HookedViTConfig:
    if model_name not in MODEL_ARCH:
        raise ValueError(f"{model_name!r}: offline build knows {list_available_models()} (no network: configs cannot be downloaded)")
    kw = dict(ARCHS[MODEL_ARCH[model_name]])
    kw["model_name"] = model_name
    kw.update(NAME_OVERRIDES.get(model_name, {}))
    kw.update(overrides)
    return HookedViTConfig(**kw, dtype=dtype, device=device)


def load_hooked_model(model_name: str, model_class: Optional[Type] = None, model_type: Any = None, device: str = "cuda",
                      dtype: torch.dtype = torch.float32, pretrained: bool = True, fold_ln: bool = False,
                      center_writing_weights: bool = False, fold_value_biases: bool = True,
                      refactor_factored_attn_matrices: bool = False, move_to_device: bool = True, use_attn_result: bool = False,
                      allow_failing: bool = False, local_path: Optional[str] = None, **kwargs) -> HookedViT:
    assert not (kwargs.get("load_in_8bit", False) or kwargs.get("load_in_4bit", False)), "Quantization not supported"
    if isinstance(dtype, str):
        dtype = DTYPE_FROM_STRING[dtype]
    if "torch_dtype" in kwargs:
        dtype = kwargs.pop("torch_dtype")
    if fold_ln or center_writing_weights or refactor_factored_attn_matrices:
        raise NotImplementedError("fold_ln / center_writing_weights / refactor_factored_attn_matrices are not implemented in this build")
    if model_type is not None and str(getattr(model_type, "name", model_type)).upper() not in ("VISION", "MODELTYPE.VISION"):
        raise NotImplementedError("only the vision tower is built here (SURVEY.md section 8: the text tower is out of scope)")
    # cfg.dtype is set too: the reference's loader forgets it and its bf16 forward raises -- deliberate fix
    cfg = load_config(model_name, dtype=dtype, device=device)
    model = (model_class or HookedViT)(cfg)
    if pretrained:
        if local_path is None:
            raise FileNotFoundError(f"pretrained=True needs local_path=<checkpoint file> for {model_name!r}: this build cannot download "
                                    "weights (no network); pass pretrained=False for the reference's random initialisation")
        load_clip_vision_weights(model, local_path, fold_value_biases=bool(fold_value_biases))
    model = model.to(dtype)
    if move_to_device:
        model = model.to(device)
    model.set_use_attn_result(use_attn_result)
    logging.info(f"Loaded {'pretrained ' if pretrained else ''}model {model_name} into HookedViT")
    return model
