"""Weiszfeld iterations for the geometric median used to initialise ``b_dec`` (one-time host-side op;
same algorithm and stopping rule as
/root/reference/src/vit_prisma/sae/training/geometric_median.py:23-86)."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch


def _weighted_mean(points: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    return (points * (weights / weights.sum()).view(-1, 1)).sum(dim=0)


def compute_geometric_median(points: torch.Tensor, weights: Optional[torch.Tensor] = None, eps: float = 1e-6,
                             maxiter: int = 100, ftol: float = 1e-20, do_log: bool = False) -> SimpleNamespace:
    with torch.no_grad():
        w0 = torch.ones(points.shape[0], device=points.device) if weights is None else weights
        w = w0
        median = _weighted_mean(points, w0)

        def objective(m: torch.Tensor) -> torch.Tensor:
            return (torch.linalg.norm(points - m.view(1, -1), dim=1) * w0).sum()

        value = objective(median)
        logs = [value] if do_log else None
        converged = False
        for _ in range(maxiter):
            prev = value
            dist = torch.linalg.norm(points - median.view(1, -1), dim=1)
            w = w0 / torch.clamp(dist, min=eps)
            median = _weighted_mean(points, w)
            value = objective(median)
            if logs is not None:
                logs.append(value)
            if abs(prev - value) <= ftol * value:
                converged = True
                break
    return SimpleNamespace(median=_weighted_mean(points, w), new_weights=w, logs=logs,
                           termination="function value converged within tolerance" if converged
                           else "maximum iterations reached")
