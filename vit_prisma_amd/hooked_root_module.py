"""HookedRootModule: names every sub-module, owns ``hook_dict`` / ``mod_dict`` and implements the
``hooks()`` context, ``run_with_hooks`` and the generic (PyTorch-hook based) ``run_with_cache``.

Same public surface and semantics as
/root/reference/src/vit_prisma/prisma_tools/hooked_root_module.py:22-332; HookedViT overrides
``run_with_cache`` with a dispatcher that sends pure-caching calls to the native HIP plan and
everything else here.
"""
from __future__ import annotations

import logging
from contextlib import contextmanager
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch.nn as nn

from .hook_points import HookPoint

NamesFilter = Optional[Union[Callable[[str], bool], Sequence[str]]]
HookSpec = Tuple[Union[str, Callable[[str], bool]], Callable]


def names_filter_to_fn(names_filter: NamesFilter) -> Callable[[str], bool]:
    """None -> everything; str -> equality; list -> membership; anything else is called
    (hooked_root_module.py:301-308 -- note a tuple is NOT treated as a list there either)."""
    if names_filter is None:
        return lambda name: True
    if type(names_filter) == str:  # noqa: E721  (exact type checks, as in the reference)
        wanted = names_filter
        return lambda name: name == wanted
    if type(names_filter) == list:  # noqa: E721
        members = names_filter
        return lambda name: name in members
    return names_filter


class HookedRootModule(nn.Module):
    def __init__(self, *args):
        super().__init__()
        self.is_caching = False
        self.context_level = 0

    def setup(self) -> None:
        """Call at the end of the subclass ``__init__``: stamps ``module.name`` on every sub-module
        and collects the HookPoints (hooked_root_module.py:30-47)."""
        self.mod_dict: Dict[str, nn.Module] = {}
        self.hook_dict: Dict[str, HookPoint] = {}
        for name, module in self.named_modules():
            if not name:
                continue
            module.name = name
            self.mod_dict[name] = module
            if isinstance(module, HookPoint):
                self.hook_dict[name] = module

    # -- bulk operations ----------------------------------------------------------------------
    def hook_points(self):
        return self.hook_dict.values()

    def remove_all_hook_fns(self, dir: str = "both", including_permanent: bool = False, level=None) -> None:
        for hp in self.hook_points():
            hp.remove_hooks(dir, including_permanent, level)

    def clear_context(self) -> None:
        for hp in self.hook_points():
            hp.clear_context()

    def reset_hooks(self, clear_contexts: bool = True, direction: str = "both",
                    including_permanent: bool = False, level=None) -> None:
        if clear_contexts:
            self.clear_context()
        self.remove_all_hook_fns(direction, including_permanent, level)
        self.is_caching = False

    # -- adding hooks -------------------------------------------------------------------------
    def check_hooks_to_add(self, hook_point, hook_point_name, hook, dir="fwd", is_permanent=False,
                           prepend=False) -> None:
        """Subclasses veto hooks that cannot fire with the current flags (base_vit.py:695-719)."""

    def check_and_add_hook(self, hook_point, hook_point_name, hook, dir="fwd", is_permanent=False,
                           level=None, prepend=False) -> None:
        self.check_hooks_to_add(hook_point, hook_point_name, hook, dir=dir, is_permanent=is_permanent,
                                prepend=prepend)
        hook_point.add_hook(hook, dir=dir, is_permanent=is_permanent, level=level, prepend=prepend)

    def add_hook(self, name, hook, dir="fwd", is_permanent=False, level=None, prepend=False) -> None:
        if type(name) == str:  # noqa: E721
            self.check_and_add_hook(self.mod_dict[name], name, hook, dir=dir, is_permanent=is_permanent,
                                    level=level, prepend=prepend)
            return
        for hp_name, hp in self.hook_dict.items():
            if name(hp_name):
                self.check_and_add_hook(hp, hp_name, hook, dir=dir, is_permanent=is_permanent,
                                        level=level, prepend=prepend)

    def add_perma_hook(self, name, hook, dir="fwd") -> None:
        self.add_hook(name, hook, dir, is_permanent=True)

    def _attach(self, specs: List[HookSpec], dir: str) -> None:
        # NB: deliberately bypasses check_hooks_to_add, like hooked_root_module.py:145-165
        for name, hook in specs:
            if type(name) == str:  # noqa: E721
                self.mod_dict[name].add_hook(hook, dir=dir, level=self.context_level)
            else:
                for hp_name, hp in self.hook_dict.items():
                    if name(hp_name):
                        hp.add_hook(hook, dir=dir, level=self.context_level)

    @contextmanager
    def hooks(self, fwd_hooks: List[HookSpec] = [], bwd_hooks: List[HookSpec] = [],
              reset_hooks_end: bool = True, clear_contexts: bool = True):
        """Hooks live for the duration of the ``with`` block; on exit (also on error) only the hooks
        registered at this nesting level are removed (hooked_root_module.py:136-174)."""
        try:
            self.context_level += 1
            self._attach(fwd_hooks, "fwd")
            self._attach(bwd_hooks, "bwd")
            yield self
        finally:
            if reset_hooks_end:
                self.reset_hooks(clear_contexts=clear_contexts, including_permanent=False,
                                 level=self.context_level)
            self.context_level -= 1

    def run_with_hooks(self, *model_args, fwd_hooks: List[HookSpec] = [], bwd_hooks: List[HookSpec] = [],
                       reset_hooks_end: bool = True, clear_contexts: bool = False):
        if len(bwd_hooks) > 0 and reset_hooks_end:
            logging.warning("WARNING: Hooks will be reset at the end of run_with_hooks. This removes the "
                            "backward hooks before a backward pass can occur.")
        with self.hooks(fwd_hooks, bwd_hooks, reset_hooks_end, clear_contexts) as hooked:
            return hooked.forward(*model_args)

    # -- caching ------------------------------------------------------------------------------
    def _cache_writers(self, cache: dict, device, remove_batch_dim: bool):
        def save_fwd(tensor, hook):
            t = tensor.detach().to(device)
            cache[hook.name] = t[0] if remove_batch_dim else t

        def save_bwd(tensor, hook):
            t = tensor.detach().to(device)
            cache[hook.name + "_grad"] = t[0] if remove_batch_dim else t

        return save_fwd, save_bwd

    def get_caching_hooks(self, names_filter: NamesFilter = None, incl_bwd: bool = False, device=None,
                          remove_batch_dim: bool = False, cache: Optional[dict] = None
                          ) -> Tuple[dict, list, list]:
        """(cache, fwd_hooks, bwd_hooks) for every HookPoint passing the filter
        (hooked_root_module.py:289-332)."""
        cache = {} if cache is None else cache
        keep = names_filter_to_fn(names_filter)
        self.is_caching = True
        save_fwd, save_bwd = self._cache_writers(cache, device, remove_batch_dim)
        fwd, bwd = [], []
        for name in self.hook_dict:
            if keep(name):
                fwd.append((name, save_fwd))
                if incl_bwd:
                    bwd.append((name, save_bwd))
        return cache, fwd, bwd

    def add_caching_hooks(self, names_filter: NamesFilter = None, incl_bwd: bool = False, device=None,
                          remove_batch_dim: bool = False, cache: Optional[dict] = None) -> dict:
        """Persistent variant (hooked_root_module.py:212-253)."""
        cache = {} if cache is None else cache
        keep = names_filter_to_fn(names_filter)
        self.is_caching = True
        save_fwd, save_bwd = self._cache_writers(cache, device, remove_batch_dim)
        for name, hp in self.hook_dict.items():
            if keep(name):
                hp.add_hook(save_fwd, dir="fwd")
                if incl_bwd:
                    hp.add_hook(save_bwd, dir="bwd")
        return cache

    def run_with_cache(self, *model_args, names_filter: NamesFilter = None, device=None,
                       remove_batch_dim: bool = False, incl_bwd: bool = False, reset_hooks_end: bool = True,
                       clear_contexts: bool = False, fwd_hooks: List[HookSpec] = [],
                       bwd_hooks: List[HookSpec] = [], **model_kwargs):
        """Generic PyTorch-hook implementation (hooked_root_module.py:255-287).  User fwd_hooks are
        registered BEFORE the caching hooks, so the cache holds post-user-hook values."""
        cache, cache_fwd, cache_bwd = self.get_caching_hooks(names_filter, incl_bwd, device,
                                                              remove_batch_dim=remove_batch_dim)
        with self.hooks(fwd_hooks=fwd_hooks + cache_fwd, bwd_hooks=bwd_hooks + cache_bwd,
                        reset_hooks_end=reset_hooks_end, clear_contexts=clear_contexts):
            model_out = self(*model_args, **model_kwargs)
            if incl_bwd or bwd_hooks:
                model_out.backward()
        return model_out, cache
