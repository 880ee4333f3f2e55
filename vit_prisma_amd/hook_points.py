"""HookPoint / LensHandle -- the hook runtime's leaf objects.

API-compatible with /root/reference/src/vit_prisma/prisma_tools/hook_point.py:16-112 and
lens_handle.py:17-28: an identity ``nn.Module`` that user callbacks ``hook(tensor, hook=<HookPoint>)``
attach to; a non-None return value replaces the activation (nn.Module forward-hook semantics).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import torch.nn as nn
import torch.utils.hooks as torch_hooks


@dataclass
class LensHandle:
    """Bookkeeping for one registered hook: the torch handle, whether it survives ``reset_hooks``
    and the ``hooks()`` context depth it was registered at (lens_handle.py:17-28)."""
    hook: torch_hooks.RemovableHandle
    is_permanent: bool = False
    context_level: Optional[int] = None


class HookPoint(nn.Module):
    def __init__(self):
        super().__init__()
        self.fwd_hooks: List[LensHandle] = []
        self.bwd_hooks: List[LensHandle] = []
        self.ctx: dict = {}
        self.name: Optional[str] = None   # filled in by HookedRootModule.setup()

    # -- registration -------------------------------------------------------------------------
    def add_hook(self, hook: Callable, dir: str = "fwd", is_permanent: bool = False,
                 level: Optional[int] = None, prepend: bool = False) -> None:
        if dir not in ("fwd", "bwd"):
            raise ValueError(f"Invalid dir {dir}. dir must be 'fwd' or 'bwd'")
        point = self
        if dir == "fwd":
            def wrapped(module, module_input, module_output):
                return hook(module_output, hook=point)
            registry, handles = self._forward_hooks, self.fwd_hooks
            raw = self.register_forward_hook(wrapped)
        else:
            def wrapped(module, module_input, module_output):
                return hook(module_output[0], hook=point)
            registry, handles = self._backward_hooks, self.bwd_hooks
            raw = self.register_backward_hook(wrapped)
        wrapped.__name__ = repr(hook)
        handle = LensHandle(raw, is_permanent, level)
        if prepend:
            # run before everything already registered (hook_point.py:54-56, pinned by the
            # reference's tests/test_hooks.py:193-231)
            registry.move_to_end(raw.id, last=False)
            handles.insert(0, handle)
        else:
            handles.append(handle)

    def add_perma_hook(self, hook: Callable, dir: str = "fwd") -> None:
        self.add_hook(hook, dir=dir, is_permanent=True)

    def remove_hooks(self, dir: str = "fwd", including_permanent: bool = False,
                     level: Optional[int] = None) -> None:
        if dir not in ("fwd", "bwd", "both"):
            raise ValueError(f"Invalid direction {dir}. dir must be 'fwd', 'bwd', or 'both'")

        def prune(handles: List[LensHandle]) -> List[LensHandle]:
            kept = []
            for h in handles:
                drop = including_permanent or (
                    not h.is_permanent and (level is None or h.context_level == level))
                if drop:
                    h.hook.remove()
                else:
                    kept.append(h)
            return kept

        # NB: like the reference (hook_point.py:93-96) "both" only prunes the forward list
        if dir in ("fwd", "both"):
            self.fwd_hooks = prune(self.fwd_hooks)
        elif dir == "bwd":
            self.bwd_hooks = prune(self.bwd_hooks)

    def has_hooks(self) -> bool:
        return bool(self._forward_hooks) or bool(self._backward_hooks)

    def clear_context(self) -> None:
        self.ctx = {}

    def forward(self, x):
        return x

    def layer(self) -> int:
        """Block index for names of the form 'blocks.{layer}....' (hook_point.py:107-112)."""
        return int(self.name.split(".")[1])
