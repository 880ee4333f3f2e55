"""Name shorthand + slicing helpers consumed by ActivationCache.

Behaviour-compatible with /root/reference/src/vit_prisma/utils/prisma_utils.py:99-302
(``Slice``, ``get_act_name``).
"""
from __future__ import annotations

import re
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

SliceInput = Optional[Union[int, Tuple[int, ...], List[int], torch.Tensor, np.ndarray]]


def to_numpy(tensor) -> np.ndarray:
    """utils/prisma_utils.py:304-318 (parameter name the reference's: callers pass ``tensor=``)."""
    if isinstance(tensor, np.ndarray):
        return tensor
    if isinstance(tensor, (list, tuple)):
        return np.array(tensor)
    if isinstance(tensor, (torch.Tensor, torch.nn.Parameter)):
        return tensor.detach().cpu().numpy()
    if isinstance(tensor, (int, float, bool, str)):
        return np.array(tensor)
    raise ValueError(f"Input to to_numpy has invalid type: {type(tensor)}")


class Slice:
    """Index helper with five modes -- int (drops the dim), tuple/slice, array (gather) and None
    (identity) -- applied along one chosen dimension (prisma_utils.py:99-198)."""

    def __init__(self, input_slice: SliceInput = None):
        kind = type(input_slice)
        if kind == tuple:
            self.slice, self.mode = slice(*input_slice), "slice"
        elif kind == int:
            self.slice, self.mode = input_slice, "int"
        elif kind == slice:
            self.slice, self.mode = input_slice, "slice"
        elif kind in (list, torch.Tensor, np.ndarray):
            self.slice, self.mode = to_numpy(input_slice), "array"
        elif input_slice is None:
            self.slice, self.mode = slice(None), "identity"
        else:
            raise ValueError(f"Invalid input_slice {input_slice}")

    def apply(self, tensor: torch.Tensor, dim: int = 0) -> torch.Tensor:
        index = [slice(None)] * tensor.ndim
        index[dim] = self.slice
        return tensor[tuple(index)]

    def indices(self, max_ctx: Optional[int] = None) -> np.ndarray:
        if self.mode == "int":
            return np.array([self.slice], dtype=np.int64)
        if max_ctx is None:
            raise ValueError("max_ctx must be specified if slice is not an integer")
        return np.arange(max_ctx, dtype=np.int64)[self.slice]

    def __repr__(self) -> str:
        return f"Slice: {self.slice} Mode: {self.mode} "


_ATTN_ACTS = {"k", "v", "q", "z", "rot_k", "rot_q", "result", "pattern", "attn_scores"}
_MLP_ACTS = {"pre", "post", "mid", "pre_linear"}
_ACT_ALIASES = {"attn": "pattern", "attn_logits": "attn_scores", "key": "k", "query": "q", "value": "v",
                "mlp_pre": "pre", "mlp_mid": "mid", "mlp_post": "post"}
_LAYER_TYPE_ALIASES = {"a": "attn", "m": "mlp", "b": "", "block": "", "blocks": "", "attention": "attn"}
_SHORTHAND = re.compile(r"([a-z]+)(\d+)([a-z]?.*)")


def get_act_name(name: str, layer: Optional[Union[int, str]] = None, layer_type: Optional[str] = None) -> str:
    """Shorthand -> full HookPoint name, e.g. ('k', 6, 'a') -> 'blocks.6.attn.hook_k',
    'pre5' -> 'blocks.5.mlp.hook_pre', 'scale4ln1' -> 'blocks.4.ln1.hook_scale',
    'normalized' -> 'ln_final.hook_normalized' (prisma_utils.py:202-302)."""
    if ("." in name or name.startswith("hook_")) and layer is None and layer_type is None:
        return name
    m = _SHORTHAND.match(name)
    if m is not None:
        name, layer, layer_type = m.groups(0)
    name = _ACT_ALIASES.get(name, name)
    if name in _ATTN_ACTS:
        layer_type = "attn"
    elif name in _MLP_ACTS:
        layer_type = "mlp"
    elif layer_type in _LAYER_TYPE_ALIASES:
        layer_type = _LAYER_TYPE_ALIASES[layer_type]
    parts = []
    if layer is not None:
        parts.append(f"blocks.{layer}")
    if layer_type:
        parts.append(layer_type)
    parts.append(f"hook_{name}")
    full = ".".join(parts)
    if name in ("scale", "normalized") and layer is None:
        full = f"ln_final.{full}"
    return full


def transpose(tensor: torch.Tensor) -> torch.Tensor:
    return tensor.transpose(-1, -2)
