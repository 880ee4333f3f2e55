"""Learning-rate schedules of the SAE trainer as LambdaLR multipliers
(/root/reference/src/vit_prisma/sae/training/get_scheduler.py:17-92).  NB: ``lr_end`` of the warm-up
cosine schedule is used as a *multiplier* floor there (the trainer passes cfg.lr / 10); kept as is."""
from __future__ import annotations

import math
from typing import Any, Callable, Optional

import torch.optim as optim
import torch.optim.lr_scheduler as lr_scheduler


def schedule_lambda(name: Optional[str], **kw: Any) -> Optional[Callable[[int], float]]:
    """The multiplier as a plain function of the step (None for the torch-native schedulers)."""
    name = "constant" if name is None else name.lower()
    warm = kw.get("warm_up_steps", 0)
    total = kw.get("training_steps")
    if name == "constant":
        return lambda step: 1.0
    if name == "constantwithwarmup":
        return lambda step: min(1.0, (step + 1) / warm)
    if name == "linearwarmupdecay":
        assert total is not None, "training_steps must be provided"
        return lambda step: (step + 1) / warm if step < warm else (total - step) / (total - warm)
    if name == "cosineannealingwarmup":
        assert total is not None, "training_steps must be provided"
        floor = kw.get("lr_end", 0)

        def fn(step: int) -> float:
            if step < warm:
                return (step + 1) / warm
            progress = (step - warm) / (total - warm)
            return floor + 0.5 * (1 - floor) * (1 + math.cos(math.pi * progress))
        return fn
    return None


def get_scheduler(scheduler_name: Optional[str], optimizer: optim.Optimizer, **kwargs: Any):
    fn = schedule_lambda(scheduler_name, **kwargs)
    if fn is not None:
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=fn)
    name = scheduler_name.lower()
    total = kwargs.get("training_steps")
    if name == "cosineannealing":
        assert total is not None, "training_steps must be provided"
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=total, eta_min=kwargs.get("lr_end", 0))
    if name == "cosineannealingwarmrestarts":
        return lr_scheduler.CosineAnnealingWarmRestarts(optimizer, T_0=total // kwargs.get("num_cycles", 1),
                                                        eta_min=kwargs.get("lr_end", 0))
    raise ValueError(f"Unsupported scheduler: {scheduler_name}")
