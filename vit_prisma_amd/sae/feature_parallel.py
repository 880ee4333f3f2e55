"""Feature-parallel (tensor-parallel) top-k SAE train step -- new functionality.

Data parallelism over tokens moves parameter-sized collectives every step (302 MB for 768 -> 24576: the reduce-scatter of the
gradient rows and the all-gather of the updated parameter rows).  Here the FEATURES are sharded instead: rank r owns
``W_enc[:, lo:hi]``, ``W_dec[lo:hi]``, ``b_enc[lo:hi]`` and their Adam state for good, every rank sees all tokens of the
batch, and what crosses the links per step is token-sized:

    all-gather   the ranks' token batches                       N x d_in floats          (when tokens are harvested per rank)
    all-gather   each rank's k candidates per token             N x k x (4 + 4) bytes per rank
    all-reduce   the partial reconstructions                    N x d_in floats
    all-reduce   gb_dec's encoder term | clip-norm term | l0    d_in + 2 floats
    all-gather   fire counts (statistics)                       d_sae floats

The engine (``NativeSAE`` over the shard, or the CPU twin of the tests) does the arithmetic; this file is the choreography:

    encode_topk          k local candidates per token (the shard's own top-k)
    global top-k         candidates of all ranks ranked by (value desc, global feature index asc); the local candidates that
                         lose get value 0 -- a pair with value <= 0 is a hole in every kernel, exactly like the reference's
                         ReLU behind its top-k (P/sae/sae.py:795-810)
    tp_partial           this shard's part of the reconstruction          -> all-reduce
    tp_finish            LN-out, loss, dY, dh, CSR, sparse backward for the shard's features
    clip norm            per-rank sums of squares, one scalar all-reduce (clip_grad_norm_ is over ALL parameters)
    apply                clip -> project -> Adam on the shard; b_dec identically on every rank

The result equals the single-process reference step up to fp32 summation order (the reconstruction is a sum of per-rank
partial sums): ``tests/test_feature_parallel_cpu.py`` (gloo, world 2 and 4, against the oracle).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch


def shard_range(d_sae: int, rank: int, world: int) -> Tuple[int, int]:
    if d_sae % world:
        raise ValueError(f"feature parallelism needs d_sae ({d_sae}) divisible by the world size ({world})")
    per = d_sae // world
    return rank * per, (rank + 1) * per


def shard_tensors(W_enc: torch.Tensor, W_dec: torch.Tensor, b_enc: torch.Tensor, b_dec: torch.Tensor, rank: int,
                  world: int) -> Dict[str, torch.Tensor]:
    """Contiguous copies of this rank's feature shard (+ the replicated b_dec)."""
    lo, hi = shard_range(W_dec.shape[0], rank, world)
    return dict(W_enc=W_enc[:, lo:hi].contiguous(), W_dec=W_dec[lo:hi].contiguous(), b_enc=b_enc[lo:hi].contiguous(),
                b_dec=b_dec.clone().contiguous())


class FeatureParallelSAE:
    """One rank of the feature-parallel step.  ``make_engine(W_enc, W_dec, b_enc, b_dec)`` builds the engine over the shard
    tensors (``NativeSAE`` on a GPU; the oracle twin in the CPU tests)."""

    def __init__(self, W_enc: torch.Tensor, W_dec: torch.Tensor, b_enc: torch.Tensor, b_dec: torch.Tensor, k: int,
                 make_engine: Callable[..., object], dist=None, rank: int = 0, world: int = 1):
        self.dist, self.rank, self.world = dist, rank, world
        self.k = int(k)
        self.d_in, self.d_sae = W_enc.shape
        self.lo, self.hi = shard_range(self.d_sae, rank, world)
        self.shard = shard_tensors(W_enc, W_dec, b_enc, b_dec, rank, world)
        self.engine = make_engine(self.shard["W_enc"], self.shard["W_dec"], self.shard["b_enc"], self.shard["b_dec"])
        dev = W_enc.device
        self.fire_count = torch.zeros(self.d_sae, dtype=torch.float32, device=dev)     # of the last step, all features
        self.loss = self.l0 = None

    # ---- collectives (no-ops in a single process) -----------------------------------------------------------------
    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """[world, *t.shape]"""
        if self.world == 1:
            return t.unsqueeze(0)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1))
        return out

    def _all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            self.dist.all_reduce(t)
        return t

    def gather_tokens(self, x_local: torch.Tensor) -> torch.Tensor:
        """Every rank's token batch (harvested from its own images), in rank order: the global batch of the step."""
        return self._all_gather(x_local.contiguous()).reshape(-1, x_local.shape[-1])

    # ---- the step -------------------------------------------------------------------------------------------------
    def global_topk_mask(self, idx_local: torch.Tensor, val_local: torch.Tensor) -> torch.Tensor:
        """Which of this rank's candidates are among the k largest of ALL ranks' candidates of their token.  Ranking key:
        value descending, then global feature index ascending (torch.topk's order on the dense row, as the oracle)."""
        k, W = self.k, self.world
        if W == 1:
            return torch.ones_like(val_local, dtype=torch.bool)
        vals = self._all_gather(val_local)                                             # [W, N, k]
        gidx = self._all_gather(idx_local.to(torch.int32) + self.lo)                   # global feature indices
        n = val_local.shape[0]
        v = vals.permute(1, 0, 2).reshape(n, W * k)
        g = gidx.permute(1, 0, 2).reshape(n, W * k).to(torch.int64)
        o1 = torch.argsort(g, dim=1, stable=True)                                      # index ascending ...
        v1 = torch.gather(v, 1, o1)
        o2 = torch.argsort(v1, dim=1, descending=True, stable=True)                    # ... then value descending (stable)
        order = torch.gather(o1, 1, o2)                                                # positions in rank-major order, best first
        keep = torch.zeros(n, W * k, dtype=torch.bool, device=v.device)
        keep.scatter_(1, order[:, :k], True)
        return keep[:, self.rank * k:(self.rank + 1) * k]

    def step(self, x: torch.Tensor, lr: float, max_grad_norm: Optional[float] = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
        """One train step on the GLOBAL token batch x [N, d_in] (identical on every rank).  Returns (loss, l0) as device
        scalars; ``fire_count`` holds the step's firing counts of all features on every rank."""
        eng, W = self.engine, self.world
        n = x.shape[0]
        idx, val = eng.encode_topk(x)[:2]
        idx = idx[:n].contiguous()
        keep = self.global_topk_mask(idx, val[:n])
        val_kept = torch.where(keep, val[:n], torch.zeros_like(val[:n])).contiguous()
        pre_sum = self._all_reduce(eng.tp_partial(idx, val_kept, renorm_decoder=True))
        eng.tp_finish(x, pre_sum, idx, val_kept, n_global=n, enc_term_only=self.rank != 0)
        self._all_reduce(eng.g["b_dec"])                                               # colsum(dY) once + every rank's encoder term
        eng.grad_sqnorm_rows(0, self.hi - self.lo, include_b_dec=self.rank == 0)
        small = torch.stack([eng.scalars[3], eng.scalars[2]])                          # clip-norm term | kept pairs per token
        self._all_reduce(small)
        eng.scalars[3] = small[0]
        eng.apply(lr, max_grad_norm)
        fire = self._all_gather(eng.fire_count[:self.hi - self.lo].contiguous())
        self.fire_count.copy_(fire.reshape(-1))
        self.loss, self.l0 = eng.scalars[0].clone(), small[1].clone()
        return self.loss, self.l0

    # ---- parameters back in the module's layout --------------------------------------------------------------------
    def gather_parameters(self) -> Dict[str, torch.Tensor]:
        """Full-size W_enc, W_dec, b_enc, b_dec (every rank gets all of them): checkpoints, evaluation."""
        P = self.engine.params
        W_dec = self._all_gather(P["W_dec"]).reshape(self.d_sae, self.d_in)
        b_enc = self._all_gather(P["b_enc"]).reshape(self.d_sae)
        W_encT = self._all_gather(P["W_enc"].t().contiguous()).reshape(self.d_sae, self.d_in)
        return dict(W_enc=W_encT.t().contiguous(), W_dec=W_dec, b_enc=b_enc, b_dec=P["b_dec"].clone())
