"""ActivationCache: dict-like wrapper around the tensors produced by ``run_with_cache``.

Keys, shapes, dtypes, insertion order and indexing conventions follow
/root/reference/src/vit_prisma/prisma_tools/activation_cache.py:29-158 exactly (bit-exact hook
names are part of the drop-in contract).  On the native path the values are views into one HBM
tap slab (see native_vit.TapArena); they stay valid for as long as they are referenced.

The residual-stream analysis helpers (accumulated_resid, decompose_resid, stack_head_results,
apply_ln_to_stack, ...) operate purely on cached tensors + model weights.
"""
from __future__ import annotations

import logging
from typing import Dict, Iterator, List, Optional, Tuple, Union

import torch

from .utils import Slice, SliceInput, get_act_name


class ActivationCache:
    def __init__(self, cache_dict: Dict[str, torch.Tensor], model, has_batch_dim: bool = True):
        self.cache_dict = cache_dict
        self.model = model
        self.has_batch_dim = has_batch_dim
        self.has_embed = "hook_embed" in cache_dict
        self.has_pos_embed = "hook_pos_embed" in cache_dict

    # ---- mapping protocol -------------------------------------------------------------------
    def __getitem__(self, key) -> torch.Tensor:
        """Full name, shorthand string ('pattern3', 'resid_pre') or tuple
        (name, layer[, layer_type]) with negative layers counted from the end."""
        try:
            if key in self.cache_dict:
                return self.cache_dict[key]
        except TypeError:
            pass
        if type(key) == str:  # noqa: E721
            return self.cache_dict[get_act_name(key)]
        if len(key) > 1 and key[1] is not None and key[1] < 0:
            key = (key[0], self.model.cfg.n_layers + key[1], *key[2:])
        return self.cache_dict[get_act_name(*key)]

    def __len__(self) -> int:
        return len(self.cache_dict)

    def __iter__(self) -> Iterator[str]:
        return iter(self.cache_dict)

    def __contains__(self, key) -> bool:
        return key in self.cache_dict

    def __repr__(self) -> str:
        return f"ActivationCache with keys {list(self.cache_dict.keys())}"

    def keys(self):
        return self.cache_dict.keys()

    def values(self):
        return self.cache_dict.values()

    def items(self):
        return self.cache_dict.items()

    # ---- housekeeping -----------------------------------------------------------------------
    def remove_batch_dim(self) -> "ActivationCache":
        if not self.has_batch_dim:
            logging.warning("Tried removing batch dimension after already having removed it.")
            return self
        for key, val in self.cache_dict.items():
            assert val.size(0) == 1, (f"Cannot remove batch dimension from cache with batch size > 1, "
                                      f"for key {key} with shape {val.shape}")
            self.cache_dict[key] = val[0]
        self.has_batch_dim = False
        return self

    def to(self, device, move_model: bool = False) -> "ActivationCache":
        self.cache_dict = {k: v.to(device) for k, v in self.cache_dict.items()}
        if move_model:
            self.model.to(device)
        return self

    # ---- residual-stream helpers ------------------------------------------------------------
    def accumulated_resid(self, layer: Optional[int] = None, incl_mid: bool = False, apply_ln: bool = False,
                          pos_slice: Union[Slice, SliceInput] = None, mlp_input: bool = False,
                          return_labels: bool = False):
        """Residual stream at the input of every layer up to ``layer`` (activation_cache.py:160-292)."""
        if not isinstance(pos_slice, Slice):
            pos_slice = Slice(pos_slice)
        n_layers = self.model.cfg.n_layers
        if layer is None or layer == -1:
            layer = n_layers
        assert isinstance(layer, int)
        labels: List[str] = []
        parts: List[torch.Tensor] = []
        for l in range(layer + 1):
            if l == n_layers:
                parts.append(self[("resid_post", n_layers - 1)])
                labels.append("final_post")
                continue
            parts.append(self[("resid_pre", l)])
            labels.append(f"{l}_pre")
            if (incl_mid and l < layer) or (mlp_input and l == layer):
                parts.append(self[("resid_mid", l)])
                labels.append(f"{l}_mid")
        stack = torch.stack([pos_slice.apply(c, dim=-2) for c in parts], dim=0)
        if apply_ln:
            stack = self.apply_ln_to_stack(stack, layer, pos_slice=pos_slice, mlp_input=mlp_input)
        return (stack, labels) if return_labels else stack

    def decompose_resid(self, layer: Optional[int] = None, mlp_input: bool = False, mode: str = "all",
                        apply_ln: bool = False, pos_slice: Union[Slice, SliceInput] = None,
                        incl_embeds: bool = True, return_labels: bool = False):
        """Per-component contributions (embed, pos_embed, attn_out / mlp_out of each layer) to the
        residual stream at the input of ``layer`` (activation_cache.py:294-386)."""
        if not isinstance(pos_slice, Slice):
            pos_slice = Slice(pos_slice)
        n_layers = self.model.cfg.n_layers
        if layer is None or layer == -1:
            layer = n_layers
        assert isinstance(layer, int)
        incl_attn = mode != "mlp"
        incl_mlp = mode != "attn" and not self.model.cfg.attn_only
        parts: List[torch.Tensor] = []
        labels: List[str] = []
        if incl_embeds:
            if self.has_embed:
                parts.append(self["hook_embed"])
                labels.append("embed")
            if self.has_pos_embed:
                parts.append(self["hook_pos_embed"])
                labels.append("pos_embed")
        for l in range(layer):
            if incl_attn:
                parts.append(self[("attn_out", l)])
                labels.append(f"{l}_attn_out")
            if incl_mlp:
                parts.append(self[("mlp_out", l)])
                labels.append(f"{l}_mlp_out")
        if mlp_input and incl_attn:
            parts.append(self[("attn_out", layer)])
            labels.append(f"{layer}_attn_out")
        stack = torch.stack([pos_slice.apply(c, dim=-2) for c in parts], dim=0)
        if apply_ln:
            stack = self.apply_ln_to_stack(stack, layer, pos_slice=pos_slice, mlp_input=mlp_input)
        return (stack, labels) if return_labels else stack

    def stack_activation(self, activation_name: str, layer: int = -1, sublayer_type: Optional[str] = None
                         ) -> torch.Tensor:
        """Stack one activation over layers [0, layer) (activation_cache.py:492-521)."""
        if layer is None or layer == -1:
            layer = self.model.cfg.n_layers
        return torch.stack([self[(activation_name, l, sublayer_type)] for l in range(layer)], dim=0)

    def compute_head_results(self) -> None:
        """Adds blocks.l.attn.hook_result = z[..., h, :] @ W_O[h] for every layer when the forward ran
        with use_attn_result=False (activation_cache.py:468-490)."""
        if "blocks.0.attn.hook_result" in self.cache_dict:
            logging.warning("Tried to compute head results when they were already cached")
            return
        for l in range(self.model.cfg.n_layers):
            z = self[("z", l, "attn")]
            self.cache_dict[f"blocks.{l}.attn.hook_result"] = torch.einsum(
                "...he,hed->...hd", z, self.model.blocks[l].attn.W_O)

    def stack_head_results(self, layer: int = -1, return_labels: bool = False, incl_remainder: bool = False,
                           pos_slice: Union[Slice, SliceInput] = None, apply_ln: bool = False):
        """Per-head contributions to the residual stream up to ``layer``
        (activation_cache.py:388-466)."""
        if not isinstance(pos_slice, Slice):
            pos_slice = Slice(pos_slice)
        cfg = self.model.cfg
        if layer is None or layer == -1:
            layer = cfg.n_layers
        if "blocks.0.attn.hook_result" not in self.cache_dict:
            logging.warning("Tried to stack head results when they weren't cached. Computing head results now")
            self.compute_head_results()
        parts: List[torch.Tensor] = []
        labels: List[str] = []
        for l in range(layer):
            res = pos_slice.apply(self[("result", l, "attn")], dim=-3)     # [..., pos, head, d_model]
            labels.extend(f"L{l}H{h}" for h in range(cfg.n_heads))
            parts.append(res.movedim(-2, 0))                                 # [head, ..., pos, d_model]
        if parts:
            stack = torch.cat(parts, dim=0)
        else:
            ref = pos_slice.apply(self[("resid_post", layer - 1)], dim=-2)
            stack = torch.zeros((0, *ref.shape), dtype=ref.dtype, device=ref.device)
        if incl_remainder:
            remainder = pos_slice.apply(self[("resid_post", layer - 1)], dim=-2) - stack.sum(dim=0)
            stack = torch.cat([stack, remainder[None]], dim=0)
            labels.append("remainder")
        if apply_ln:
            stack = self.apply_ln_to_stack(stack, layer, pos_slice=pos_slice)
        return (stack, labels) if return_labels else stack

    def get_neuron_results(self, layer: int, neuron_slice: Union[Slice, SliceInput] = None,
                           pos_slice: Union[Slice, SliceInput] = None) -> torch.Tensor:
        """What every neuron of ``layer``'s MLP writes into the residual stream: ``post[..., n, None] * W_out[n]`` ->
        [..., pos, neurons, d_model]; not cached (activation_cache.py:523-562).  The reference accepts only ``Slice`` objects
        here (its ``isinstance(x, SliceInput)`` on a subscripted Union raises TypeError for anything else); plain slice inputs
        are accepted as well."""
        if not isinstance(neuron_slice, Slice):
            neuron_slice = Slice(neuron_slice)
        if not isinstance(pos_slice, Slice):
            pos_slice = Slice(pos_slice)
        neuron_acts = self[("post", layer, "mlp")]
        W_out = self.model.blocks[layer].mlp.W_out
        # (order matters: a position slice may collapse its dimension, so it is applied while the position is still at -2)
        neuron_acts = pos_slice.apply(neuron_acts, dim=-2)
        neuron_acts = neuron_slice.apply(neuron_acts, dim=-1)
        W_out = neuron_slice.apply(W_out, dim=0)
        return neuron_acts[..., None] * W_out

    def stack_neuron_results(self, layer: int, pos_slice: Union[Slice, SliceInput] = None,
                             neuron_slice: Union[Slice, SliceInput] = None, return_labels: bool = False,
                             incl_remainder: bool = False, apply_ln: bool = False):
        """Every neuron's contribution to the residual stream entering ``layer`` (labels "L{l}N{n}"), optionally with the rest of
        the stream as a last component (activation_cache.py:564-654, including its corner cases: with no layers below and
        ``incl_remainder`` the result is a one-element LIST, as there)."""
        if layer is None or layer == -1:
            layer = self.model.cfg.n_layers
        if not isinstance(neuron_slice, Slice):
            neuron_slice = Slice(neuron_slice)
        if not isinstance(pos_slice, Slice):
            pos_slice = Slice(pos_slice)
        neuron_labels = neuron_slice.apply(torch.arange(self.model.cfg.d_mlp), dim=0)
        if neuron_labels.ndim == 0:
            neuron_labels = neuron_labels[None]
        parts: List[torch.Tensor] = []
        labels: List[str] = []
        for l in range(layer):
            parts.append(self.get_neuron_results(l, pos_slice=pos_slice, neuron_slice=neuron_slice))
            labels.extend(f"L{l}N{int(h)}" for h in neuron_labels)
        if parts:
            components = torch.cat(parts, dim=-2).movedim(-2, 0)          # [neurons of all layers, ..., d_model]
            if incl_remainder:
                remainder = self[("resid_post", layer - 1)] - components.sum(dim=0)
                components = torch.cat([components, remainder[None]], dim=0)
                labels.append("remainder")
        elif incl_remainder:
            components = [pos_slice.apply(self[("resid_post", layer - 1)], dim=-2)]
        else:
            components = torch.zeros(0, *pos_slice.apply(self["hook_embed"], dim=-2).shape, device=self.model.cfg.device)
        if apply_ln:
            components = self.apply_ln_to_stack(components, layer, pos_slice=pos_slice)
        return (components, labels) if return_labels else components

    def get_full_resid_decomposition(self, layer: Optional[int] = None, mlp_input: bool = False, expand_neurons: bool = True,
                                     apply_ln: bool = False, pos_slice: Union[Slice, SliceInput] = None,
                                     return_labels: bool = False):
        """The residual stream entering ``layer`` as [every head's result | every neuron's result (or the MLP outputs) | embed |
        pos_embed | accumulated biases] (activation_cache.py:737-826).  As in the reference, ``hook_embed`` holds the patch tokens
        only ([batch, tokens - 1, d_model]: the CLS row is added afterwards, base_vit.py:169-181), so for a ViT with a CLS token
        the concatenation only goes through when ``pos_slice`` collapses the position dimension."""
        if layer is None or layer == -1:
            layer = self.model.cfg.n_layers
        if not isinstance(pos_slice, Slice):
            pos_slice = Slice(pos_slice)
        head_stack, labels = self.stack_head_results(layer + (1 if mlp_input else 0), pos_slice=pos_slice, return_labels=True)
        components = [head_stack]
        if not self.model.cfg.attn_only and layer > 0:
            if expand_neurons:
                neuron_stack, neuron_labels = self.stack_neuron_results(layer, pos_slice=pos_slice, return_labels=True)
                labels.extend(neuron_labels)
                components.append(neuron_stack)
            else:
                mlp_stack, mlp_labels = self.decompose_resid(layer, mlp_input=mlp_input, pos_slice=pos_slice, incl_embeds=False,
                                                             mode="mlp", return_labels=True)
                labels.extend(mlp_labels)
                components.append(mlp_stack)
        if self.has_embed:
            labels.append("embed")
            components.append(pos_slice.apply(self["embed"], -2)[None])
        if self.has_pos_embed:
            labels.append("pos_embed")
            components.append(pos_slice.apply(self["pos_embed"], -2)[None])
        # (without the neuron expansion the MLP biases are already inside the MLP outputs)
        bias = self.model.accumulated_bias(layer, mlp_input, include_mlp_biases=expand_neurons)
        labels.append("bias")
        components.append(bias.expand((1,) + head_stack.shape[1:]))
        residual_stack = torch.cat(components, dim=0)
        if apply_ln:
            residual_stack = self.apply_ln_to_stack(residual_stack, layer, pos_slice=pos_slice, mlp_input=mlp_input)
        return (residual_stack, labels) if return_labels else residual_stack

    def apply_ln_to_stack(self, residual_stack: torch.Tensor, layer: Optional[int] = None, mlp_input: bool = False,
                          pos_slice: Union[Slice, SliceInput] = None, batch_slice: Union[Slice, SliceInput] = None,
                          has_batch_dim: bool = True) -> torch.Tensor:
        """Centre each component and divide by the cached LayerNorm scale of the layer the stack feeds
        into (ln_final for layer == n_layers) -- activation_cache.py:656-735."""
        cfg = self.model.cfg
        if cfg.normalization_type not in ("LN", "LNPre"):
            return residual_stack
        if not isinstance(pos_slice, Slice):
            pos_slice = Slice(pos_slice)
        if not isinstance(batch_slice, Slice):
            batch_slice = Slice(batch_slice)
        if layer is None or layer == -1:
            layer = cfg.n_layers
        if has_batch_dim and not self.has_batch_dim:
            residual_stack = residual_stack
        residual_stack = residual_stack - residual_stack.mean(dim=-1, keepdim=True)
        if layer == cfg.n_layers:
            scale = self["ln_final.hook_scale"]
        else:
            scale = self[f"blocks.{layer}.ln{2 if mlp_input else 1}.hook_scale"]
        scale = pos_slice.apply(scale, dim=-2)
        if self.has_batch_dim:
            scale = batch_slice.apply(scale)
        return residual_stack / scale
