"""``HookedSAEViT``: a ``HookedViT`` that keeps sparse autoencoders spliced in place of HookPoints
(/root/reference/src/vit_prisma/models/base_vit.py:827-1086; the splice itself: sae/sae_utils.py:214-228).

``add_sae`` replaces the HookPoint module at ``sae.cfg.hook_point`` by the SAE (``cfg.return_out_only`` makes its forward
return the reconstruction alone, sae/sae.py:631-635), so every later forward / ``run_with_cache`` sees the reconstruction
downstream and the SAE's own HookPoints (``<hook_point>.hook_sae_in`` / ``hook_hidden_pre`` / ``hook_hidden_post`` /
``hook_sae_out``) in the cache; ``reset_saes`` puts HookPoints (or the previously attached SAEs) back.

On a GPU the forward stays on the HIP plan where the splice sits on a block's HookPoint (``hook_resid_pre`` of blocks >= 1,
``hook_attn_out``, ``hook_resid_mid``, ``hook_mlp_out``, ``hook_resid_post``, ``attn.hook_q / k / v / z / ...``, ``mlp.hook_pre /
hook_post``) and the SAE computes in the model's dtype: the plan is split there exactly as for a forward hook, the SAE is called on
the tapped tensor (on its own HIP engine unless one of its HookPoints is hooked or cached) and the block resumes from its output;
the SAE's HookPoints take the replaced point's place in the cache.  A splice on a block's LayerNorm point or on block 0's entry
sends that block to its own module (the others stay on the plan); on the embedding / final stage that stage runs on the model's own
modules (as for a hook there); in another dtype than the model's the call runs on the PyTorch path (``native_fallback_reason`` says
so).

One deliberate deviation: the reference's ``saes()`` context reads ``sae.cfg.hook_name`` (:1074, :1075), a field its SAE config
does not have (``hook_point`` is the one ``add_sae`` uses, :862) -- the temporary-attachment entry points raise AttributeError
there.  Here they work, keyed by ``hook_point``.
"""
from __future__ import annotations

import logging
from contextlib import contextmanager
from typing import Any, Dict, List, Optional, Sequence, Union

from torch import nn

from .hook_points import HookPoint
from .vit import HookedViT


def _parent_and_leaf(root: nn.Module, path: str):
    obj: Any = root
    parts = path.split(".")
    for part in parts[:-1]:
        obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
    return obj, parts[-1]


def _act_name(sae) -> str:
    return getattr(sae.cfg, "hook_name", None) or sae.cfg.hook_point


class HookedSAEViT(HookedViT):
    def __init__(self, *model_args: Any, **model_kwargs: Any):
        super().__init__(*model_args, **model_kwargs)
        self.acts_to_saes: Dict[str, nn.Module] = {}

    # ---- permanent attachment ----------------------------------------------------------------------------------------------
    def add_sae(self, sae: nn.Module, use_error_term: Optional[bool] = None) -> None:
        """Attach ``sae`` at ``sae.cfg.hook_point`` until ``reset_saes`` (overwrites an SAE already attached there); an unknown
        hook point is skipped with a warning (base_vit.py:850-876)."""
        act_name = sae.cfg.hook_point
        if act_name not in self.acts_to_saes and act_name not in self.hook_dict:
            logging.warning(f"No hook found for {act_name}. Skipping. Check model.hook_dict for available hooks.")
            return
        if use_error_term is not None:
            if not hasattr(sae, "_original_use_error_term"):
                sae._original_use_error_term = getattr(sae, "use_error_term", None)
            sae.use_error_term = use_error_term
        sae.cfg.return_out_only = True
        self.acts_to_saes[act_name] = sae
        parent, leaf = _parent_and_leaf(self, act_name)
        setattr(parent, leaf, sae)
        self.setup()

    def _reset_sae(self, act_name: str, prev_sae: Optional[nn.Module] = None) -> None:
        if act_name not in self.acts_to_saes:
            logging.warning(f"No SAE is attached to {act_name}. There's nothing to reset.")
            return
        current = self.acts_to_saes[act_name]
        if hasattr(current, "_original_use_error_term"):
            current.use_error_term = current._original_use_error_term
            delattr(current, "_original_use_error_term")
        parent, leaf = _parent_and_leaf(self, act_name)
        if prev_sae is not None:
            setattr(parent, leaf, prev_sae)
            self.acts_to_saes[act_name] = prev_sae
        else:
            hp = self._original_hook_points.get(act_name) or HookPoint()
            # the reference installs a FRESH HookPoint() here (base_vit.py:903): hooks that sat on the point before the splice --
            # permanent ones included -- are gone afterwards.  The original object comes back (the tree the HIP plan was built
            # for), stripped the same way.
            hp.remove_hooks("fwd", including_permanent=True)
            hp.remove_hooks("bwd", including_permanent=True)
            hp.clear_context()
            setattr(parent, leaf, hp)
            del self.acts_to_saes[act_name]

    def reset_saes(self, act_names: Optional[Union[str, Sequence[str]]] = None,
                   prev_saes: Optional[Sequence[Optional[nn.Module]]] = None) -> None:
        """Detach the SAEs at ``act_names`` (all of them by default), optionally putting ``prev_saes`` back (base_vit.py:908-936)."""
        if isinstance(act_names, str):
            act_names = [act_names]
        elif act_names is None:
            act_names = list(self.acts_to_saes.keys())
        if prev_saes:
            if len(act_names) != len(prev_saes):
                raise ValueError("act_names and prev_saes must have the same length")
        else:
            prev_saes = [None] * len(act_names)
        for name, prev in zip(act_names, prev_saes):
            self._reset_sae(name, prev)
        self.setup()

    def setup(self) -> None:
        super().setup()
        # the HookPoint objects this model was built with: reset_saes puts THEM back (the reference installs a fresh HookPoint(),
        # base_vit.py:903; _reset_sae strips the restored object of its hooks so that it behaves like one -- and here the restored tree is again the
        # one the HIP plan was built for)
        if not hasattr(self, "_original_hook_points"):
            self._original_hook_points = dict(self.hook_dict)

    # ---- temporary attachment ------------------------------------------------------------------------------------------------
    @contextmanager
    def saes(self, saes: Union[nn.Module, List[nn.Module]] = [], reset_saes_end: bool = True, use_error_term: Optional[bool] = None):
        """Attach ``saes`` for the duration of the context; on exit the SAEs attached before come back (base_vit.py:1046-1086)."""
        if isinstance(saes, nn.Module):
            saes = [saes]
        names: List[str] = []
        prev: List[Optional[nn.Module]] = []
        try:
            for sae in saes:
                names.append(_act_name(sae))
                prev.append(self.acts_to_saes.get(_act_name(sae)))
                self.add_sae(sae, use_error_term=use_error_term)
            yield self
        finally:
            if reset_saes_end:
                self.reset_saes(names, prev)

    def run_with_saes(self, *model_args: Any, saes: Union[nn.Module, List[nn.Module]] = [], reset_saes_end: bool = True,
                      use_error_term: Optional[bool] = None, **model_kwargs: Any):
        with self.saes(saes=saes, reset_saes_end=reset_saes_end, use_error_term=use_error_term):
            return self(*model_args, **model_kwargs)

    def run_with_cache_with_saes(self, *model_args: Any, saes: Union[nn.Module, List[nn.Module]] = [], reset_saes_end: bool = True,
                                 use_error_term: Optional[bool] = None, return_cache_object: bool = True,
                                 remove_batch_dim: bool = False, **kwargs: Any):
        with self.saes(saes=saes, reset_saes_end=reset_saes_end, use_error_term=use_error_term):
            return self.run_with_cache(*model_args, return_cache_object=return_cache_object, remove_batch_dim=remove_batch_dim, **kwargs)

    def run_with_hooks_with_saes(self, *model_args: Any, saes: Union[nn.Module, List[nn.Module]] = [], reset_saes_end: bool = True,
                                 fwd_hooks: list = [], bwd_hooks: list = [], reset_hooks_end: bool = True,
                                 clear_contexts: bool = False, **model_kwargs: Any):
        with self.saes(saes=saes, reset_saes_end=reset_saes_end):
            return self.run_with_hooks(*model_args, fwd_hooks=fwd_hooks, bwd_hooks=bwd_hooks, reset_hooks_end=reset_hooks_end,
                                       clear_contexts=clear_contexts, **model_kwargs)
