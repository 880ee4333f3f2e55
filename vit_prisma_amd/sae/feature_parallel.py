"""Feature-parallel (tensor-parallel) top-k SAE train step -- new functionality.

Data parallelism over tokens moves parameter-sized collectives every step (302 MB for 768 -> 24576: the reduce-scatter of the
gradient rows and the all-gather of the updated parameter rows).  Here the FEATURES are sharded instead: rank r owns
``W_enc[:, lo:hi]``, ``W_dec[lo:hi]``, ``b_enc[lo:hi]`` and their Adam state for good, every rank sees all tokens of the
batch, and what crosses the links per step is token-sized -- three collectives (four when the tokens are harvested per rank):

    all-gather   the ranks' token batches                       N x d_in floats          (when tokens are harvested per rank)
    all-gather   each rank's k candidates per token             N x k x (4 + 4) bytes per rank, ONE buffer (values | indices)
    all-reduce   the partial reconstructions                    N x d_in floats
    all-reduce   ONE bucket: gb_dec's terms | clip-norm rows term | kept pairs per token | firing counts of all features
                                                                d_in + 4 + d_sae floats

The engine (``NativeSAE`` over the shard, or the CPU twin of the tests) does the arithmetic, in place in the two exchange
buffers (``tp_bind``); this file is the choreography, cut into four phases at the collectives:

    phase_encode     k local candidates per token (the shard's own top-k)                       -> all-gather of ``pack``
    phase_partial    global top-k (``tp_merge``: candidates of all ranks ranked by value desc, global feature index asc; the
                     local candidates that lose get value 0 -- a pair with value <= 0 is a hole in every kernel, exactly like
                     the reference's ReLU behind its top-k, P/sae/sae.py:795-810); this shard's part of the reconstruction
                                                                                                -> all-reduce of the partials
    phase_finish     LN-out, loss, dY, dh, CSR, sparse backward for the shard's features; the bucket
                                                                                                -> all-reduce of the bucket
    phase_apply      clip norm of the GLOBAL gradient (clip_grad_norm_ is over ALL parameters), clip -> project -> Adam on the
                     shard; b_dec identically on every rank

``step`` chains them with the process group's collectives; ``simulate_step`` runs W ranks of ONE process in lockstep with the
exchanges done by hand (single-GPU tests of world 4 / 8 on the real kernels, and tools/tp_shard_times.py: per-rank kernel
time without a second GPU).

The result equals the single-process reference step up to fp32 summation order (the reconstruction is a sum of per-rank
partial sums): ``tests/test_feature_parallel_cpu.py`` (gloo, world 2 and 4, against the oracle).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

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
    tensors (``NativeSAE`` on a GPU; the oracle twin in the CPU tests); its ``max_tokens`` is the global batch."""

    def __init__(self, W_enc: torch.Tensor, W_dec: torch.Tensor, b_enc: torch.Tensor, b_dec: torch.Tensor, k: int,
                 make_engine: Callable[..., object], dist=None, rank: int = 0, world: int = 1):
        self.dist, self.rank, self.world = dist, rank, world
        self.k = int(k)
        self.d_in, self.d_sae = W_enc.shape
        self.lo, self.hi = shard_range(self.d_sae, rank, world)
        self.shard = shard_tensors(W_enc, W_dec, b_enc, b_dec, rank, world)
        self.engine = eng = make_engine(self.shard["W_enc"], self.shard["W_dec"], self.shard["b_enc"], self.shard["b_dec"])
        # the shard is this object's private copy and the kernels train its transposed master: the [d_in, shard] layout is
        # rewritten only when the parameters are gathered (NativeSAE.lazy_w_enc)
        eng.lazy_w_enc = True
        dev = W_enc.device
        # the two exchange buffers (the engine writes into them in place)
        self.pack = torch.zeros(2, eng.max_tokens, self.k, dtype=torch.int32, device=dev)       # candidate values (bits) | local indices
        self.bucket = torch.zeros(self.d_in + 4 + self.d_sae, dtype=torch.float32, device=dev)  # gb_dec | 2 scalars | pad | fire counts
        eng.tp_bind(self.pack, self.bucket, self.lo, self.d_sae)
        self.fire_count = self.bucket[self.d_in + 4:]              # of the last step, all features, on every rank
        self.loss = self.l0 = None
        self._n = 0
        self._idx = self._val_kept = None

    # ---- collectives (identities in a single process) ---------------------------------------------------------------
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

    # ---- the four phases ----------------------------------------------------------------------------------------------
    def phase_encode(self, x: torch.Tensor) -> torch.Tensor:
        """-> this rank's candidates [2, n, k] int32 (values as float bits | local feature indices): what is all-gathered."""
        eng = self.engine
        n = self._n = x.shape[0]
        eng.encode_topk(x, want_ln_stats=False)                     # (lands in self.pack: tp_bind)
        return self.pack if n == eng.max_tokens else self.pack[:, :n].contiguous()

    def phase_partial(self, gathered: torch.Tensor) -> torch.Tensor:
        """gathered [world, 2, n, k] -> this shard's partial reconstruction [n, d_in] of the pairs that made the global top-k."""
        eng, n = self.engine, self._n
        self._idx = self.pack[1, :n]
        self._val_kept = eng.tp_merge(gathered, self.world, self.rank, n)
        return eng.tp_partial(self._idx, self._val_kept, renorm_decoder=True)

    def phase_finish(self, x: torch.Tensor, pre_sum: torch.Tensor) -> torch.Tensor:
        """pre_sum = the partial reconstructions summed over the ranks -> the bucket to all-reduce."""
        eng = self.engine
        eng.tp_finish(x, pre_sum, self._idx, self._val_kept, n_global=self._n, enc_term_only=self.rank != 0)
        eng.tp_bucket_pack(self.bucket)                             # colsum(dY) once (rank 0) + every rank's encoder term; scalars; fire
        return self.bucket

    def phase_apply(self, lr: float, max_grad_norm: Optional[float] = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
        eng = self.engine
        eng.tp_bucket_unpack(self.bucket)                           # scalars[3] = global clip-norm term, scalars[2] = l0
        eng.apply(lr, max_grad_norm)
        self.loss, self.l0 = eng.scalars[0].clone(), eng.scalars[2].clone()
        return self.loss, self.l0

    def step(self, x: torch.Tensor, lr: float, max_grad_norm: Optional[float] = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
        """One train step on the GLOBAL token batch x [N, d_in] (identical on every rank).  Returns (loss, l0) as device
        scalars; ``fire_count`` holds the step's firing counts of all features on every rank."""
        gathered = self._all_gather(self.phase_encode(x))
        pre_sum = self._all_reduce(self.phase_partial(gathered))
        self._all_reduce(self.phase_finish(x, pre_sum))
        return self.phase_apply(lr, max_grad_norm)

    # ---- parameters back in the module's layout --------------------------------------------------------------------
    def gather_parameters(self) -> Dict[str, torch.Tensor]:
        """Full-size W_enc, W_dec, b_enc, b_dec (every rank gets all of them): checkpoints, evaluation."""
        self.engine.materialize_w_enc()
        P = self.engine.params
        W_dec = self._all_gather(P["W_dec"]).reshape(self.d_sae, self.d_in)
        b_enc = self._all_gather(P["b_enc"]).reshape(self.d_sae)
        W_encT = self._all_gather(P["W_enc"].t().contiguous()).reshape(self.d_sae, self.d_in)
        return dict(W_enc=W_encT.t().contiguous(), W_dec=W_dec, b_enc=b_enc, b_dec=P["b_dec"].clone())


def simulate_step(ranks: List[FeatureParallelSAE], x: torch.Tensor, lr: float, max_grad_norm: Optional[float] = 1.0,
                  on_phase: Optional[Callable[[str, int], None]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ranks of a world in ONE process, in lockstep: each phase runs for every rank, the collective in between is done
    by hand (stack / sum / copy).  ``ranks[r]`` was built with ``rank=r, world=len(ranks), dist=None``.  ``on_phase(name, r)``
    is called before and after every rank's phase (timing hooks of tools/tp_shard_times.py)."""
    W = len(ranks)

    def run(name, r, fn):
        if on_phase:
            on_phase(name + ":begin", r)
        out = fn()
        if on_phase:
            on_phase(name + ":end", r)
        return out

    packs = [run("encode", r, lambda fp=fp: fp.phase_encode(x)) for r, fp in enumerate(ranks)]
    gathered = torch.stack(packs).contiguous()                                          # all-gather
    partials = [run("partial", r, lambda fp=fp: fp.phase_partial(gathered)) for r, fp in enumerate(ranks)]
    pre_sum = partials[0].clone()
    for p in partials[1:]:                                                              # all-reduce (rank order)
        pre_sum += p
    buckets = [run("finish", r, lambda fp=fp: fp.phase_finish(x, pre_sum)) for r, fp in enumerate(ranks)]
    total = buckets[0].clone()
    for b in buckets[1:]:
        total += b
    for b in buckets:
        b.copy_(total)
    out = [run("apply", r, lambda fp=fp: fp.phase_apply(lr, max_grad_norm)) for r, fp in enumerate(ranks)]
    assert W == len(out)
    return out[0]


def gather_parameters_local(ranks: List[FeatureParallelSAE]) -> Dict[str, torch.Tensor]:
    """The full-size parameters of a simulated world (see ``simulate_step``)."""
    for fp in ranks:
        fp.engine.materialize_w_enc()
    P = [fp.engine.params for fp in ranks]
    return dict(W_enc=torch.cat([p["W_enc"] for p in P], dim=1), W_dec=torch.cat([p["W_dec"] for p in P], dim=0),
                b_enc=torch.cat([p["b_enc"] for p in P]), b_dec=P[0]["b_dec"].clone())
