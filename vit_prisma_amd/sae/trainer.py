"""VisionSAETrainer: the SAE training loop behind the reference's trainer API.

Same constructor / ``train_step`` / ``run`` contract as
/root/reference/src/vit_prisma/sae/train_sae.py:61-861 (signature ``(cfg, model, dataset,
eval_dataset=None)``, ``train_step`` argument names and 7-tuple, ``run() -> sae``).  The order of
operations inside a step is the reference's (:278-411): renorm decoder -> zero_grad -> forward ->
firing statistics -> backward -> clip_grad_norm_ -> remove parallel gradient -> Adam -> scheduler.

On an MI355X (fp32 standard SAE with layer_norm / no input normalisation, no ghost grads) the step runs on
``NativeSAE`` (HIP kernels) directly on the module's parameter storage -- the k-sparse step for top-k, the
dense fused step (exact fp32 MFMA GEMMs with ReLU / L1 / loss / gate epilogues) for ReLU + L1, ghost gradients
included; Adam moments live in the engine.  Everything else (gated, transcoder, tanh-relu, top-k with ghost grads,
CPU) takes the PyTorch path below, which is the reference algorithm verbatim.

Data parallel (new functionality, SURVEY.md section 8e -- the reference is single-process): one process
per GPU, each with its share of the global token batch and a 1/W shard of the OPTIMIZER, by feature
(rows j in [r d_sae/W, (r+1) d_sae/W) of W_enc^T, W_dec, b_enc -- contiguous slices of the flat buffers):

    all-reduce     768 floats            global batch mean for the loss normaliser (sae.py:145), before the forward
    reduce-scatter gW_enc^T, gW_dec, gb_enc   every rank receives the summed gradient rows of ITS features
    all-reduce     gb_dec | fire counts | loss, mse, l0     one small bucket (768 + d_sae + 3 floats)
    all-reduce     1 float               sum of the ranks' squared-norm terms: the clip norm is the GLOBAL gradient
                                         norm exactly as in a single process (train_sae.py:394-397)
    (local)        clip -> project -> Adam on the rank's rows only: the 1.06 GB / step optimizer pass shrinks by W
    all-gather     W_enc^T, W_dec, b_enc rows    asynchronous: they ride RCCL's stream while this rank already harvests
                                         the next batch; waited for at the start of the next step

b_dec (768 floats) is updated redundantly and identically on every rank.  When d_sae is not divisible by 4 W the
trainer falls back to one all-reduce of the flat gradient buffer and a replicated optimizer.
"""
from __future__ import annotations

import os
from typing import Any, List, Optional

import torch
from torch.optim import Adam

from .config import VisionModelSAERunnerConfig
from .get_scheduler import get_scheduler
from .sae import StandardSparseAutoencoder
from .store import VisionActivationsStore, _dist_info


def _wandb():
    try:
        import wandb  # type: ignore
        return wandb
    except Exception:
        return None


class VisionSAETrainer:
    def __init__(self, cfg: VisionModelSAERunnerConfig, model, dataset, eval_dataset=None, sparse_coder=None,
                 activations_store=None):
        self.cfg = cfg
        self.is_transcoder = bool(getattr(cfg, "is_transcoder", False))
        self.model = model
        self.dataset = dataset
        self.eval_dataset = eval_dataset
        self.bad_run_check = cfg.min_l0 is not None and cfg.min_explained_variance is not None
        torch.manual_seed(cfg.seed)
        if sparse_coder is None:                          # train_sae.py:72-81
            if self.is_transcoder:
                from .variants import Transcoder
                sparse_coder = Transcoder(cfg)
            elif cfg.architecture == "gated":
                from .variants import GatedSparseAutoencoder
                sparse_coder = GatedSparseAutoencoder(cfg)
            elif cfg.architecture in ("standard", "vanilla"):
                sparse_coder = StandardSparseAutoencoder(cfg)
            else:
                raise ValueError(f"Loading of {cfg.architecture} not supported")
        self.sparse_coder = sparse_coder
        self.sae = self.sparse_coder                      # legacy alias
        self.activations_store = activations_store
        if self.activations_store is None and cfg.use_cached_activations and not self.is_transcoder:
            from .store import CacheVisionActivationStore
            self.activations_store = CacheVisionActivationStore(cfg, *_dist_info())     # train_sae.py:138-139
        if self.activations_store is None and dataset is not None:
            self.activations_store = VisionActivationsStore(cfg, model, dataset, eval_dataset=eval_dataset,
                                                            num_workers=0)
        self.checkpoint_thresholds = self.get_checkpoint_thresholds()
        self._engine = None
        self._native_pref: Optional[bool] = None            # None = auto, True = native or raise, False = PyTorch path
        self._pending = []                                  # in-flight parameter all-gathers of the sharded optimizer
        self._small = None
        self.rank, self.world = _dist_info()
        self._feature_parallel: Optional[bool] = None       # multi-rank native top-k step: None = auto (feature parallel when d_sae
                                                            # divides by the world size), True / False = forced
        self._force_dist = False                            # see _mr
        self._fp = None                                     # FeatureParallelSAE (this rank's shard engine + choreography)
        self._fp_dirty = False                              # the module's parameters lag behind the shards

    @property
    def _mr(self) -> bool:
        """Do the steps take their multi-rank form (collectives, sharded optimizer / feature shards)?  With more than one rank,
        or when ``force_distributed_paths`` asked for it in a process group of ONE rank -- the same calls on the same backend
        (the in-place reduce_scatter_tensor / all_gather_into_tensor / packed buckets on RCCL), which is how a one-GPU box
        executes the code an 8-GPU job runs."""
        return self.world > 1 or self._force_dist

    def force_distributed_paths(self, flag: bool = True) -> "VisionSAETrainer":
        import torch.distributed as dist
        if flag and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("force_distributed_paths needs an initialised torch.distributed process group")
        self._force_dist = bool(flag)
        return self

    def use_native(self, flag: Optional[bool]) -> "VisionSAETrainer":
        """True: the fused HIP step or an error; False: always the PyTorch path; None (default): native when supported."""
        self._native_pref = flag
        return self

    def use_feature_parallel(self, flag: Optional[bool] = True, drop_replicas: bool = False) -> "VisionSAETrainer":
        """How multi-rank native top-k steps are sharded.  True: by FEATURE (sae/feature_parallel.py: every rank keeps
        d_sae / world features and their optimizer state for good, sees the whole token batch, and only token-sized
        collectives cross the links); False: by token, with the optimizer sharded by feature (reduce-scatter of the gradient
        rows, all-gather of the updated parameter rows: 302 MB per 4096-token step at 768 -> 24576); None (default): feature
        parallel whenever d_sae is divisible by the world size -- by the measured per-rank phase times it is the faster of
        the two at every world size for the reference's 4096-token batch.  In the feature-parallel mode
        the module's own parameters are refreshed only by ``sync_parameters()`` (``checkpoint`` and the end of ``run`` call
        it).  drop_replicas: while the shards train, the module's two matrices (W_enc, W_dec: 2 x d_in x d_sae floats that no
        kernel reads in this mode) are released on every rank -- ``sae.W_enc`` / ``sae.W_dec`` are EMPTY between
        ``sync_parameters()`` calls, which allocate and fill them again (and the next step releases them again).  New
        functionality (the reference is single-process)."""
        self._feature_parallel = None if flag is None else bool(flag)
        self._fp_drop = bool(drop_replicas)
        return self

    def _use_tp(self, sae) -> bool:
        if not self._mr or self._feature_parallel is False:
            return False
        if self._feature_parallel is None:
            # auto: only where the feature-parallel kernels can run (pv_sae_tp_merge ranks the candidates of at most 8 ranks,
            # a shard must hold at least k features and a whole number of 4-feature groups); anything else -- two nodes, small
            # or odd shards -- takes the token-sharded step, which has no such limits.  use_feature_parallel(True) keeps the
            # hard error.
            d_sae, k = int(sae.cfg.d_sae), int((sae.cfg.activation_fn_kwargs or {}).get("k", 1))
            shard = d_sae // self.world
            return d_sae % self.world == 0 and self.world <= 8 and shard >= k and shard % 4 == 0 and k <= 64      # (pv_sae_tp_merge ranks <= 8 x 64)
        return True

    # ---- bookkeeping ------------------------------------------------------------------------------
    def get_checkpoint_thresholds(self) -> List[int]:
        if self.cfg.n_checkpoints > 0:
            t = self.cfg.total_training_tokens
            return list(range(0, t, max(t // self.cfg.n_checkpoints, 1)))[1:]
        return []

    def initialize_training_variables(self):
        dev = self.cfg.device
        act_freq_scores = torch.zeros(int(self.cfg.d_sae), device=dev)
        n_forward_passes_since_fired = torch.zeros(int(self.cfg.d_sae), device=dev)
        n_frac_active_tokens = 0
        optimizer = Adam(self.sparse_coder.parameters(), lr=self.cfg.lr)
        scheduler = get_scheduler(self.cfg.lr_scheduler_name, optimizer=optimizer,
                                  warm_up_steps=self.cfg.lr_warm_up_steps,
                                  training_steps=self.cfg.total_training_steps, lr_end=self.cfg.lr / 10)
        return act_freq_scores, n_forward_passes_since_fired, n_frac_active_tokens, optimizer, scheduler

    def initialize_geometric_medians(self):
        cfg = self.sparse_coder.cfg
        layers = cfg.hook_point_layer if isinstance(cfg.hook_point_layer, list) else [cfg.hook_point_layer]
        lid = layers.index(cfg.hook_point_layer) if not isinstance(cfg.hook_point_layer, list) else 0
        medians = {}
        if cfg.b_dec_init_method == "geometric_median":
            from .geometric_median import compute_geometric_median
            acts = self.activations_store.storage_buffer.detach()[:, lid, :]
            medians[lid] = compute_geometric_median(acts, maxiter=200).median
            out_median = None
            if self.is_transcoder:
                acts_out = self.activations_store.storage_buffer_out.detach()[:, lid, :]
                out_median = compute_geometric_median(acts_out, maxiter=200).median
            if self.world > 1:
                # every rank harvested different images: rank 0's median is THE initial b_dec (replicas must start equal)
                import torch.distributed as dist
                med = medians[lid].to(cfg.device).contiguous()
                dist.broadcast(med, src=0)
                medians[lid] = med
                if out_median is not None:
                    out_median = out_median.to(cfg.device).contiguous()
                    dist.broadcast(out_median, src=0)
            self.sparse_coder.initialize_b_dec_with_precalculated(medians[lid], out_median)
        elif cfg.b_dec_init_method == "mean":
            acts = self.activations_store.storage_buffer.detach()[:, lid, :]
            if self.world > 1:
                import torch.distributed as dist
                m = acts.float().mean(dim=0)
                dist.all_reduce(m)
                self.sparse_coder.b_dec.data = (m / self.world).to(self.sparse_coder.dtype)
            else:
                self.sparse_coder.initialize_b_dec_with_mean(acts)
        self.sparse_coder.train()
        return medians

    # ---- native engine ----------------------------------------------------------------------------
    def _native_kind(self, sae, x: torch.Tensor) -> Optional[str]:
        """Which fused HIP step serves this SAE: "topk" (k-sparse step, sae.hip), "relu" (dense ReLU + L1 step,
        sae_dense.hip), "gated" (the gated SAE's step: sparse where the batch allows it, sae.hip / sae_dense.hip) or None (PyTorch path: the top-k
        gated form / other activations / d_out != d_in / CPU)."""
        cfg = sae.cfg
        from .variants import Transcoder
        # a Transcoder (sae/transcoder.py) of equal input and output width runs on the same two steps (pv_sae_transcoder):
        # no ghost gradients; with a process group the tokens are sharded and the optimizer replicated (one all-reduce of the flat
        # gradient buffer: _native_dense_step for ReLU, _native_dp_step's transcoder branch for top-k)
        # ... and one between hook points of DIFFERENT width (no skip connection: the reference's needs d_out == d_in) on the same steps with
        # every row padded to the wider of the two (pv_sae_transcoder.d_in_true / d_out_true)
        d_out = int(getattr(cfg, "d_out", cfg.d_in))
        # ghost gradients on a transcoder (round 6; transcoder.py:82-86: the ghost term sees the INPUT activation): equal widths, one process
        tc_ghost_ok = not cfg.use_ghost_grads or (d_out == int(cfg.d_in) and not self._mr and cfg.d_in % 8 == 0)
        is_tc = (isinstance(sae, Transcoder) and tc_ghost_ok and getattr(self, "_target", None) is not None
                 and (d_out == int(cfg.d_in) or (sae._parameters.get("W_skip") is None and d_out % 8 == 0 and d_out <= 1280
                                                 and cfg.d_in % 8 == 0)))
        from .variants import GatedSparseAutoencoder
        # a GatedSparseAutoencoder (sae.py:648-792) with the ReLU magnitude path has its own step (pv_sae_gated_step_sparse: the open gates
        # as per-token lists where the batch allows it, the dense GEMMs of pv_sae_gated_step otherwise -- decided on the GPU)
        # ... and the top-k form (TopK on the magnitudes and on the gate activations) its own k-sparse step (pv_sae_gated_topk_step)
        is_gated = (isinstance(sae, GatedSparseAutoencoder) and cfg.d_in % 8 == 0 and cfg.d_sae % 8 == 0
                    and (cfg.activation_fn_str == "relu"
                         or (cfg.activation_fn_str == "topk" and 1 <= cfg.activation_fn_kwargs.get("k", 0) <= 256)))
        common = (x.is_cuda and (isinstance(sae, StandardSparseAutoencoder) or is_tc or is_gated) and cfg.dtype == torch.float32
                  and cfg.normalize_activations in ("layer_norm", "constant_norm_rescale", "none", None)
                  and all(p.is_cuda and p.dtype == torch.float32 for p in sae._parameters.values() if p is not None)   # (not .parameters(): no sync of lazily kept layouts)
                  and cfg.d_in % 4 == 0 and cfg.d_in <= 1280 and cfg.d_sae % 4 == 0 and cfg.d_sae <= 65536
                  and self._native_pref is not False)
        if not common:
            return None
        if is_gated:
            return "gated"
        if (cfg.activation_fn_str == "topk" and 1 <= cfg.activation_fn_kwargs.get("k", 0) <= 256
                # ghost gradients on top-k (pv_sae_topk_ghost): plain SAE, d_in a multiple of 8
                and (not cfg.use_ghost_grads or cfg.d_in % 8 == 0)):
            return "topk"
        # ReLU + L1; since round 6 also "tanh-relu" (sae.py:823-830) and lp_norm > 1 in the sparsity term (sae.py:617) -- both on the dense
        # GEMMs' epilogues (the sparse form of the step serves the plain one)
        if (cfg.activation_fn_str in ("relu", "tanh-relu") and float(getattr(sae, "lp_norm", 1)) >= 1.0
                and cfg.d_in % 8 == 0 and cfg.d_sae % 8 == 0):
            return "relu"
        return None

    def _native_ok(self, sae, x: torch.Tensor) -> bool:
        return self._native_kind(sae, x) is not None

    def _native_why_not(self, sae) -> str:
        """Best-effort diagnosis for the one-time fallback warning (the limits of ``_native_kind``)."""
        cfg = sae.cfg
        why = []
        if cfg.dtype != torch.float32:
            why.append(f"dtype {cfg.dtype} (the fused steps keep fp32 master weights)")
        if cfg.d_in > 1280 or cfg.d_in % 4:
            why.append(f"d_in = {cfg.d_in} (supported: multiples of 4 up to 1280 = ViT-H/14)")
        if cfg.d_sae > 65536 or cfg.d_sae % 4:
            why.append(f"d_sae = {cfg.d_sae} (supported: multiples of 4 up to 65536)")
        if cfg.activation_fn_str == "topk" and not 1 <= cfg.activation_fn_kwargs.get("k", 0) <= 256:
            why.append(f"k = {cfg.activation_fn_kwargs.get('k')} (supported: 1..256)")
        if getattr(cfg, "is_transcoder", False) and cfg.use_ghost_grads and (self._mr or int(getattr(cfg, "d_out", cfg.d_in)) != int(cfg.d_in)):
            why.append("ghost gradients on a transcoder with tokens sharded over ranks / d_out != d_in")
        if cfg.activation_fn_str not in ("topk", "relu", "tanh-relu"):
            why.append(f"activation {cfg.activation_fn_str!r}")
        if float(getattr(sae, "lp_norm", 1)) < 1.0:
            why.append(f"lp_norm = {sae.lp_norm} (p < 1 has no finite gradient at zero activations: supported p >= 1)")
        if getattr(cfg, "is_transcoder", False) and int(getattr(cfg, "d_out", cfg.d_in)) != int(cfg.d_in):
            why.append("a transcoder with d_out != d_in and the skip connection / widths that are not multiples of 8 / d_out > 1280")
        if cfg.normalize_activations not in ("layer_norm", "constant_norm_rescale", "none", None):
            why.append(f"normalize_activations = {cfg.normalize_activations!r}")
        return "; ".join(why) or "a parameter is not a contiguous fp32 CUDA tensor"

    def _get_engine(self, sae, n_tokens: int):
        from .native_sae import NativeSAE
        eng = self._engine
        tc_names = tuple(n for n in ("b_dec_out", "W_skip") if sae._parameters.get(n) is not None)
        stale = eng is not None and (any(eng.params[n].data_ptr() != sae._parameters[n].data_ptr()
                                         for n in ("W_enc", "W_dec", "b_enc", "b_dec") + tc_names
                                         + tuple(n for n in ("b_gate", "r_mag", "b_mag") if n in sae._parameters))     # e.g. b_dec.data re-bound by an init
                                     or tuple(n for n in ("b_dec_out", "W_skip") if n in eng.params) != tc_names)
        if eng is None or eng.max_tokens < n_tokens or stale:
            if self._mr and eng is None:
                # replicas must start from identical parameters (a per-rank b_dec initialisation, a different seed or a
                # caller-supplied module would otherwise never converge: every rank applies the same summed gradient).
                # Only at the FIRST creation, which every rank reaches in its first step: a later re-creation (a larger
                # batch, a re-bound parameter) is a per-rank event and must not contain a collective.
                import torch.distributed as dist
                for p in sae.parameters():
                    if p.data.is_contiguous():
                        dist.broadcast(p.data, src=0)
                    else:                                        # (a view of an earlier engine's padded storage: unequal-width transcoder)
                        tmp = p.data.contiguous()
                        dist.broadcast(tmp, src=0)
                        p.data.copy_(tmp)
            old = eng
            if old is not None:
                old.materialize_w_enc()                          # (the new engine derives its shadows from the parameter)
            P_ = dict(sae._parameters)
            tcw = None
            if self.is_transcoder and int(getattr(sae.cfg, "d_out", sae.cfg.d_in)) != int(sae.cfg.d_in):
                # rows padded to D = max(d_in, d_out): the engine owns the padded storage, the module's parameters become views of it
                # (W_enc / b_dec: leading rows; W_dec / b_dec_out: leading columns), so both sides see every update
                d_i, d_o = int(sae.cfg.d_in), int(sae.cfg.d_out)
                D = max(d_i, d_o)
                with torch.no_grad():
                    def padded(p, shape, sl):
                        buf = torch.zeros(shape, dtype=torch.float32, device=p.device)
                        buf[sl].copy_(p.data)
                        p.data = buf[sl]
                        return buf
                    P_["W_enc"] = padded(P_["W_enc"], (D, sae.cfg.d_sae), (slice(0, d_i),))
                    P_["b_dec"] = padded(P_["b_dec"], (D,), (slice(0, d_i),))
                    P_["W_dec"] = padded(P_["W_dec"], (sae.cfg.d_sae, D), (slice(None), slice(0, d_o)))
                    P_["b_dec_out"] = padded(P_["b_dec_out"], (D,), (slice(0, d_o),))
                tcw = (d_i, d_o)
            eng = NativeSAE(P_["W_enc"], P_["W_dec"], P_["b_enc"], P_["b_dec"],
                            k=sae.cfg.activation_fn_kwargs.get("k", 1),        # (the dense ReLU + L1 step has no k)
                            layer_norm=sae.cfg.normalize_activations,
                            max_tokens=max(n_tokens, self.cfg.train_batch_size // self.world),
                            **{n: P_[n] for n in tc_names}, **({"tc_widths": tcw} if tcw else {}),
                            **({"gated": {n: P_[n] for n in ("b_gate", "r_mag", "b_mag")},
                                "gated_topk": sae.cfg.activation_fn_str == "topk"} if "b_gate" in P_ else
                               {"activation": sae.cfg.activation_fn_str,
                                "lp_norm": float(getattr(sae, "lp_norm", 1)) if sae.cfg.activation_fn_str != "topk" else 1.0}))
            if old is not None and old.n_flat == eng.n_flat:             # keep the optimizer state across a re-bind
                eng.flat_m.copy_(old.flat_m)
                eng.flat_v.copy_(old.flat_v)
                eng.adam_step = old.adam_step
            if not self._mr and isinstance(sae, StandardSparseAutoencoder):
                # one process: nobody but the kernels reads W_enc between steps -- they read its transposed master -- so the
                # parameter's own layout is rewritten only when somebody asks for it (sae.W_enc, state_dict(), parameters())
                eng.lazy_w_enc = True
                object.__setattr__(sae, "_native_sync_fn", eng.materialize_w_enc)
            self._engine = eng
        return eng

    # ---- data-parallel plumbing ----------------------------------------------------------------------
    def _shard(self, d_sae: int):
        """This rank's feature rows, or None when the optimizer cannot be sharded evenly on 16-byte boundaries."""
        if not self._mr or d_sae % (4 * self.world) != 0:
            return None
        n = d_sae // self.world
        return self.rank * n, (self.rank + 1) * n

    def sync_parameters(self) -> None:
        """Data parallel only: make the module's parameters complete on this rank.  ``train_step`` returns while the
        all-gather of the other ranks' updated rows is still in flight (it overlaps the next harvest); the next
        ``train_step``, ``checkpoint`` and the end of ``run`` wait for it themselves -- call this before reading
        ``sae.W_enc`` / ``W_dec`` / ``b_enc`` in between."""
        self._dp_flush(materialize=True)

    def _dp_flush(self, materialize: bool = False) -> None:
        """Wait for the parameter all-gathers of the previous step and rebuild the other ranks' rows of W_enc / W_enc16T.
        materialize (sync_parameters / checkpoint / end of run; NOT the per-step call): also bring the module's parameters up to
        date with what the engines train -- the feature-parallel shards, the lazily kept layout of W_enc."""
        if materialize and self._engine is not None:
            self._engine.materialize_w_enc()                     # (single process: the lazily kept parameter layout of W_enc)
        if materialize and self._fp is not None and self._fp_dirty:         # feature parallel: gather the shards into the module
            P = self._fp.gather_parameters()
            with torch.no_grad():
                for n in ("W_enc", "W_dec", "b_enc", "b_dec"):
                    p = getattr(self.sparse_coder, n)
                    if p.shape != P[n].shape:                       # (released by drop_replicas: a fresh allocation)
                        p.data = torch.empty_like(P[n])
                    p.copy_(P[n])                                   # (not through .data: the version counter must move)
            eng = getattr(self.sparse_coder, "_engine", None)       # the module's own inference engine re-derives its shadows
            if eng is not None:
                eng.invalidate()
            self._fp_dirty = False
        if not self._pending:
            return
        for w in self._pending:
            w.wait()
        self._pending = []
        eng = self._engine
        j_lo, j_hi = self._shard(eng.d_sae)
        eng.sync_shadows(from_transposed=True, j_lo=0, j_hi=j_lo)
        eng.sync_shadows(from_transposed=True, j_lo=j_hi, j_hi=eng.d_sae)

    @staticmethod
    def _reduce_scatter_rows(dist, seg: torch.Tensor, j_lo: int, j_hi: int) -> None:
        """seg[j_lo:j_hi] <- sum over ranks; the other rows are left undefined."""
        mine = seg[j_lo:j_hi]
        if dist.get_backend() == "nccl":
            dist.reduce_scatter_tensor(mine, seg)                   # in place: output = this rank's chunk of the input
        else:
            out = torch.empty_like(mine)                            # gloo (tests): no aliasing guarantees
            dist.reduce_scatter_tensor(out, seg)
            mine.copy_(out)

    # ---- one step ---------------------------------------------------------------------------------
    def train_step(self, sparse_autoencoder, optimizer, scheduler, act_freq_scores, n_forward_passes_since_fired,
                   n_frac_active_tokens, layer_acts, n_training_steps, n_training_tokens):
        hp = sparse_autoencoder.cfg
        layers = hp.hook_point_layer if isinstance(hp.hook_point_layer, list) else [hp.hook_point_layer]
        layer_id = 0 if isinstance(hp.hook_point_layer, list) else layers.index(hp.hook_point_layer)
        self._target = None
        if self.is_transcoder:                                                   # train_sae.py:299-301
            sae_in, self._target = layer_acts[:, 0, :], layer_acts[:, 1, :]
        else:
            sae_in = layer_acts[:, layer_id, :]
        sparse_autoencoder.train()

        if (n_training_steps + 1) % self.cfg.feature_sampling_window == 0:     # train_sae.py:310-326
            feature_sparsity = act_freq_scores / n_frac_active_tokens
            self._log_feature_sparsity(feature_sparsity, n_training_steps)
            act_freq_scores = torch.zeros(hp.d_sae, device=hp.device)
            n_frac_active_tokens = 0

        native = self._native_ok(sparse_autoencoder, sae_in)
        if self._native_pref is True and not native:
            from .._native import NativeError
            raise NativeError("use_native(True): this SAE configuration / input is not served by the fused HIP step "
                              f"(activation {hp.activation_fn_str!r}, architecture {getattr(hp, 'architecture', 'standard')!r}, "
                              f"device {sae_in.device})")
        if not native and sae_in.is_cuda and self._native_pref is None and not getattr(self, "_warned_torch_path", False):
            # auto mode on a GPU: say ONCE why this run is not on the fused kernels (a silent PyTorch path is ~1000 x slower)
            self._warned_torch_path = True
            import warnings
            warnings.warn("vit_prisma_amd: VisionSAETrainer trains this SAE on the PyTorch path, not on the MI355X kernels: "
                          + self._native_why_not(sparse_autoencoder), stacklevel=2)
        if native:
            loss, mse_loss, l1_loss, l0 = self._native_step(sparse_autoencoder, optimizer, scheduler, sae_in,
                                                            act_freq_scores, n_forward_passes_since_fired)
        else:
            loss, mse_loss, l1_loss, l0 = self._torch_step(sparse_autoencoder, optimizer, scheduler, sae_in,
                                                           act_freq_scores, n_forward_passes_since_fired)
        n_frac_active_tokens += sae_in.shape[0] * self.world
        self.last_step_native = native
        return loss, mse_loss, l1_loss, l0, act_freq_scores, n_forward_passes_since_fired, n_frac_active_tokens

    def _native_step(self, sae, optimizer, scheduler, x, act_freq_scores, n_since_fired):
        lr = optimizer.param_groups[0]["lr"]
        kind = self._native_kind(sae, x)
        if kind in ("relu", "gated"):
            return self._native_dense_step(sae, optimizer, scheduler, x, lr, act_freq_scores, n_since_fired, gated=kind == "gated")
        ghost_mr = bool(sae.cfg.use_ghost_grads) and sae.training
        if self._use_tp(sae) and not self.is_transcoder and not ghost_mr:      # (the feature-parallel step serves the plain SAE, no ghost term)
            return self._native_tp_step(sae, optimizer, scheduler, x, lr, act_freq_scores, n_since_fired)
        self._dp_flush()                                        # parameters of the previous step must have landed
        eng = self._get_engine(sae, x.shape[0])
        # statistics tensors are the caller's: the kernels update them in place
        eng.act_freq_scores = act_freq_scores
        eng.n_fwd_since_fired = n_since_fired
        ghost = bool(sae.cfg.use_ghost_grads) and sae.training
        if not self._mr and ghost:
            # top-k + ghost gradients (sae.py:151-179; the mask of train_sae.py:330-332 is taken BEFORE this step's statistics): the
            # k-sparse step with complete gradient buffers over a decoder renormalised in place, then the ghost term's additions
            dead = n_since_fired > sae.cfg.dead_feature_window
            eng.renorm_decoder()
            eng.step(x, update_stats=True, renorm_decoder=False, sparse_grads=False, want_out=True,
                     target=self._target if eng.transcoder else None)
            eng.topk_ghost(x, dead)
            eng.grad_sqnorm()
            eng.apply(lr, self.cfg.max_grad_norm)
        elif not self._mr:
            # set_decoder_norm_to_unit_norm is part of the step; one process = nobody but the step's own apply reads the
            # gradient buffers, so the rows of features that kept no token are neither zeroed nor read (PV_SAE_SPARSE_GRADS)
            eng.step(x, update_stats=True, renorm_decoder=True, sparse_grads=True, target=self._target if eng.transcoder else None,
                     fused_sqnorm=True)
            eng.grad_sqnorm(from_step=True)                     # clip_grad_norm_ (the gradient is as the step wrote it)
            eng.apply(lr, self.cfg.max_grad_norm)
        else:
            self._native_dp_step(eng, sae, x, lr, act_freq_scores, n_since_fired)
        optimizer._opt_called = True                            # the native apply IS the optimizer step
        scheduler.step()
        self._invalidate_inference_engine(sae)
        sc = eng.scalars.clone()
        return sc[0], sc[1], None, sc[2]

    def _native_dense_step(self, sae, optimizer, scheduler, x, lr, act_freq_scores, n_since_fired, gated: bool = False):
        """ReLU + L1 (sae.py:617-626; also a ReLU Transcoder) on pv_sae_relu_step / pv_sae_dense_step, or a Gated SAE on
        pv_sae_gated_step_sparse.  Multi-rank: tokens sharded, ONE all-reduce of the flat gradient buffer and a replicated optimizer (the
        step is 6-11 ms of fp32 GEMMs: the 151 MB are not what bounds it), statistics and losses over the global batch like every
        other path."""
        self._dp_flush()
        eng = self._get_engine(sae, x.shape[0])
        eng.act_freq_scores = act_freq_scores
        eng.n_fwd_since_fired = n_since_fired
        l1 = float(sae.l1_coefficient)
        target = self._target if eng.transcoder else None

        def run(**kw):
            if gated and getattr(eng, "gated_topk", False):
                eng.gated_topk_step(x, **kw)
            elif gated:
                eng.gated_step(x, l1, **kw)
            elif kw.get("dead_mask") is not None:                 # ghost gradients: exp(hidden_pre) of the dead columns is a dense quantity
                eng.dense_step(x, l1, renorm_decoder=True, target=target, **kw)
            else:
                # sparse where the batch allows it, the dense GEMMs otherwise: the GPU decides (NativeSAE.relu_step)
                kw.pop("dead_mask", None)
                eng.relu_step(x, l1, renorm_decoder=True, target=target, sparse_grads=not self._mr, **kw)

        if not self._mr:
            dead = None
            if sae.cfg.use_ghost_grads and sae.training and not gated:        # train_sae.py:330-332 (the mask is taken BEFORE this step's statistics)
                dead = n_since_fired > sae.cfg.dead_feature_window
            run(update_stats=True, **({} if gated else {"dead_mask": dead}))
        else:
            import torch.distributed as dist
            W = self.world
            n_global = x.shape[0] * W
            bm = (target if target is not None else x).float().sum(dim=0)
            dist.all_reduce(bm)                                 # global batch mean of what the loss is taken against (sae.py:145)
            if sae.cfg.use_ghost_grads and sae.training and not gated and not eng.transcoder:
                # ghost gradients (sae.py:151-179) need the residual's column mean and the mse loss of the WHOLE batch before the ghost
                # term is formed: a pass without it gives this rank's share of both (one small all-reduce), the step proper follows
                dead = n_since_fired > sae.cfg.dead_feature_window
                eng.dense_step(x, l1, batch_mean=bm / n_global, n_global=n_global, update_stats=False, want_out=True, renorm_decoder=True)
                gl = torch.cat([(eng.sae_out[:x.shape[0]] - x).sum(dim=0), eng.scalars[1:2]])
                dist.all_reduce(gl)
                d_in_ = eng.d_in
                eng.dense_step(x, l1, batch_mean=bm / n_global, n_global=n_global, update_stats=False, renorm_decoder=True,
                               dead_mask=dead, ghost_global=(gl[:d_in_] / n_global, gl[d_in_:], n_global))
            else:
                run(batch_mean=bm / n_global, n_global=n_global, update_stats=False)
            d_sae = eng.d_sae
            if self._small is None or self._small.numel() != d_sae + 7:
                self._small = torch.empty(d_sae + 7, dtype=torch.float32, device=x.device)
            small = self._small
            small[:d_sae].copy_(eng.fire_count)
            small[d_sae:].copy_(eng.scalars[:7])
            dist.all_reduce(eng.flat_g)                         # every rank's share of the gradient (scaled by 1 / N_global already)
            dist.all_reduce(small)                              # fire counts | loss, mse, l0, -, l1, ghost, aux in one bucket
            fire = small[:d_sae]
            eng.scalars[:7].copy_(small[d_sae:])
            eng.scalars[2] /= W                                 # l0 is a mean over tokens
            n_since_fired += 1                                  # train_sae.py:356-361 on the global batch
            n_since_fired[fire > 0] = 0
            act_freq_scores += fire
        # clip_grad_norm_ over the whole (summed) gradient; single-process relu_step: from the per-feature terms the step left
        eng.grad_sqnorm(from_step=not self._mr)
        eng.apply(lr, self.cfg.max_grad_norm)
        optimizer._opt_called = True
        scheduler.step()
        self._invalidate_inference_engine(sae)
        sc = eng.scalars.clone()
        return sc[0], sc[1], sc[4], sc[2]

    @staticmethod
    def _invalidate_inference_engine(sae) -> None:
        """The training kernels update the parameters through raw pointers (no version bump): the module's own inference
        engine (StandardSparseAutoencoder.forward / encode on the HIP path) must re-derive its shadows before its next use."""
        inf = sae.__dict__.get("_engine")
        if inf is not None:
            inf.invalidate()

    def _make_shard_engine(self, sae, max_tokens: int):
        """Engine over one rank's feature shard (tests substitute the CPU twin)."""
        from .native_sae import NativeSAE
        k, ln = sae.cfg.activation_fn_kwargs["k"], sae.cfg.normalize_activations
        return lambda We, Wd, be, bd: NativeSAE(We, Wd, be, bd, k=k, layer_norm=ln, max_tokens=max_tokens)

    def _native_tp_step(self, sae, optimizer, scheduler, x, lr, act_freq_scores, n_since_fired):
        """Feature-parallel step (sae/feature_parallel.py): every rank brings its tokens, all ranks see the global batch,
        each owns d_sae / world features for good."""
        import torch.distributed as dist
        from .feature_parallel import FeatureParallelSAE
        n_global = x.shape[0] * self.world
        if self._fp is None:
            for p in sae.parameters():                          # shards are cut from identical replicas
                dist.broadcast(p.data, src=0)
            self._fp = FeatureParallelSAE(sae.W_enc.data, sae.W_dec.data, sae.b_enc.data, sae.b_dec.data,
                                          sae.cfg.activation_fn_kwargs["k"], self._make_shard_engine(sae, n_global),
                                          dist=dist, rank=self.rank, world=self.world)
        fp = self._fp
        if getattr(self, "_fp_drop", False):                    # the shards are the truth from here on: release the replicas
            for n in ("W_enc", "W_dec"):
                p = getattr(sae, n)
                if p.numel():
                    p.data = torch.empty(0, dtype=p.dtype, device=p.device)
        loss, l0 = fp.step(fp.gather_tokens(x), lr, self.cfg.max_grad_norm)
        self._fp_dirty = True
        n_since_fired += 1                                      # train_sae.py:356-361 on the global batch
        n_since_fired[fp.fire_count > 0] = 0
        act_freq_scores += fp.fire_count
        optimizer._opt_called = True
        scheduler.step()
        return loss, loss, None, l0

    def _native_dp_step(self, eng, sae, x, lr, act_freq_scores, n_since_fired):
        import torch.distributed as dist
        W = self.world
        n_global = x.shape[0] * W
        d_in, d_sae = eng.d_in, eng.d_sae
        if eng.transcoder:
            # a top-k Transcoder (b_dec_out, W_skip beside the four tensors the row shards know): tokens sharded, the mean of the
            # TARGET all-reduced (its loss normaliser, transcoder.py:78), ONE all-reduce of the whole flat gradient buffer and a
            # replicated optimizer, as the dense step's multi-rank form
            bm = self._target.float().sum(dim=0)
            dist.all_reduce(bm)
            eng.step(x, batch_mean=bm / n_global, n_global=n_global, update_stats=False, renorm_decoder=True, target=self._target)
            if self._small is None or self._small.numel() != d_sae + 3:
                self._small = torch.empty(d_sae + 3, dtype=torch.float32, device=x.device)
            small = self._small
            small[:d_sae].copy_(eng.fire_count)
            small[d_sae:].copy_(eng.scalars[:3])
            dist.all_reduce(eng.flat_g)
            dist.all_reduce(small)                              # fire counts | loss, mse, l0 in one bucket
            fire = small[:d_sae]
            eng.scalars[:3].copy_(small[d_sae:])
            eng.scalars[2] /= W
            n_since_fired += 1
            n_since_fired[fire > 0] = 0
            act_freq_scores += fire
            eng.grad_sqnorm()
            eng.apply(lr, self.cfg.max_grad_norm)
            return
        bm = x.float().sum(dim=0)
        dist.all_reduce(bm)                                     # global batch mean (sae.py:145)
        if sae.cfg.use_ghost_grads and sae.training:
            # top-k + ghost gradients (sae.py:151-179): the ghost term normalises by the residual's column mean and rescales by the mse
            # loss of the WHOLE batch -- one more small all-reduce between the step and pv_sae_topk_ghost; the decoder renormalised in
            # place (the ghost term reads W_dec as it lies), complete gradient buffers as every multi-rank step has them
            dead = n_since_fired > sae.cfg.dead_feature_window
            eng.renorm_decoder()
            eng.step(x, batch_mean=bm / n_global, n_global=n_global, update_stats=False, renorm_decoder=False, want_out=True)
            gl = torch.cat([(eng.sae_out[:x.shape[0]] - x).sum(dim=0), eng.scalars[1:2]])
            dist.all_reduce(gl)
            eng.topk_ghost(x, dead, ghost_global=(gl[:d_in] / n_global, gl[d_in:], n_global))
        else:
            eng.step(x, batch_mean=bm / n_global, n_global=n_global, update_stats=False, renorm_decoder=True)
        shard = self._shard(eng.d_sae)
        if self._small is None or self._small.numel() != d_in + d_sae + 3:
            self._small = torch.empty(d_in + d_sae + 3, dtype=torch.float32, device=x.device)
        small = self._small
        small[:d_in].copy_(eng._g["b_dec"])
        small[d_in:d_in + d_sae].copy_(eng.fire_count)
        small[d_in + d_sae:].copy_(eng.scalars[:3])
        if shard is None:
            dist.all_reduce(eng.flat_g[:eng.n_flat - d_in])      # replicated optimizer: every rank needs every row
        else:
            for name in ("W_encT", "W_dec", "b_enc"):
                self._reduce_scatter_rows(dist, eng._g[name], *shard)
        dist.all_reduce(small)                                  # gb_dec | fire counts | loss, mse, l0 in one bucket
        eng._g["b_dec"].copy_(small[:d_in])
        fire = small[d_in:d_in + d_sae]
        eng.scalars[:3].copy_(small[d_in + d_sae:])
        eng.scalars[2] /= W                                     # l0 is a mean over tokens
        n_since_fired += 1                                      # train_sae.py:356-361 on the global batch
        n_since_fired[fire > 0] = 0
        act_freq_scores += fire
        if shard is None:
            eng.grad_sqnorm()
            eng.apply(lr, self.cfg.max_grad_norm)
            return
        j_lo, j_hi = shard
        eng.grad_sqnorm_rows(j_lo, j_hi, include_b_dec=self.rank == 0)
        dist.all_reduce(eng.scalars[3:4])                       # the clip norm is over the GLOBAL gradient
        eng.apply(lr, self.cfg.max_grad_norm, j_lo, j_hi)       # this rank's rows (+ b_dec, identically everywhere)
        self._pending = [dist.all_gather_into_tensor(full, full[j_lo:j_hi], async_op=True)
                         for full in (eng.W_encT, eng.params["W_dec"], eng.params["b_enc"])]

    def _torch_step(self, sae, optimizer, scheduler, x, act_freq_scores, n_since_fired):
        """The reference algorithm on PyTorch autograd (CPU, ReLU/L1, ghost grads, ...)."""
        sae.set_decoder_norm_to_unit_norm()
        optimizer.zero_grad()
        dead = (n_since_fired > sae.cfg.dead_feature_window).bool()
        if self.is_transcoder:
            sae_out, feature_acts, loss, mse_loss, l1_loss, ghost, aux = sae(x, self._target, dead)
        else:
            sae_out, feature_acts, loss, mse_loss, l1_loss, ghost, aux = sae(x, dead)
        with torch.no_grad():
            fire = (feature_acts.abs() > 0).float().sum(0)
            pos = (feature_acts > 0).float().sum(-2)
            l0 = (feature_acts > 0).float().sum(-1).mean()
            if self.world > 1:                                   # statistics are over the GLOBAL batch, like the native path
                import torch.distributed as dist
                dist.all_reduce(fire)
                dist.all_reduce(pos)
                dist.all_reduce(l0)
                l0 = l0 / self.world
            n_since_fired += 1
            n_since_fired[pos > 0] = 0
            act_freq_scores += fire
        loss.backward()
        if self.world > 1:
            import torch.distributed as dist
            for p in sae.parameters():
                dist.all_reduce(p.grad)
                p.grad /= self.world
        if self.cfg.max_grad_norm:
            torch.nn.utils.clip_grad_norm_(sae.parameters(), max_norm=self.cfg.max_grad_norm)
        sae.remove_gradient_parallel_to_decoder_directions()
        optimizer.step()
        scheduler.step()
        return loss, mse_loss, l1_loss, l0

    # ---- logging / checkpoints --------------------------------------------------------------------
    def _log_feature_sparsity(self, feature_sparsity: torch.Tensor, n_training_steps: int) -> None:
        wb = _wandb()
        if self.cfg.log_to_wandb and wb is not None and self.rank == 0:
            log_sp = torch.log10(feature_sparsity + 1e-10).detach().cpu()
            wb.log({"metrics/mean_log10_feature_sparsity": log_sp.mean().item(),
                    "sparsity/below_1e-5": (feature_sparsity < 1e-5).float().mean().item(),
                    "sparsity/below_1e-6": (feature_sparsity < 1e-6).float().mean().item()}, step=n_training_steps)

    def checkpoint(self, sae, n_training_tokens, act_freq_scores, n_frac_active_tokens):
        self._dp_flush(materialize=True)
        if self.rank != 0:
            return
        folder = self.cfg.checkpoint_path
        os.makedirs(folder, exist_ok=True)
        self.cfg.save_config(os.path.join(folder, "config.json"))
        sae.set_decoder_norm_to_unit_norm()
        if self._engine is not None:                             # W_dec was edited between steps: 1 / ||row|| of the last apply is stale
            self._engine.invalidate()
        path = os.path.join(folder, f"n_images_{n_training_tokens // max(self.cfg.context_size, 1)}.pt")
        sae.save_model(path)
        sparsity = torch.log10(act_freq_scores / max(n_frac_active_tokens, 1) + 1e-10).detach().cpu()
        torch.save(sparsity, path.replace(".pt", "_log_feature_sparsity.pt"))

    # ---- the loop (train_sae.py:772-861) ----------------------------------------------------------
    def run(self):
        cfg = self.cfg
        wb = _wandb()
        if cfg.log_to_wandb and wb is None and self.rank == 0:
            print("[vit_prisma_amd] wandb is not installed: log_to_wandb ignored")
        (act_freq_scores, n_since_fired, n_frac_active_tokens, optimizer, scheduler) = self.initialize_training_variables()
        self.initialize_geometric_medians()
        n_training_steps = 0
        n_training_tokens = 0
        try:
            from tqdm import tqdm
            pbar = tqdm(total=cfg.total_training_tokens, desc="Training SAE", mininterval=20, disable=self.rank != 0)
        except Exception:
            pbar = None
        loss = mse_loss = l0 = None
        while n_training_tokens < cfg.total_training_tokens:
            layer_acts = self.activations_store.next_batch()
            (loss, mse_loss, l1_loss, l0, act_freq_scores, n_since_fired, n_frac_active_tokens) = self.train_step(
                sparse_autoencoder=self.sparse_coder, optimizer=optimizer, scheduler=scheduler, layer_acts=layer_acts,
                n_training_steps=n_training_steps, n_training_tokens=n_training_tokens, act_freq_scores=act_freq_scores,
                n_forward_passes_since_fired=n_since_fired, n_frac_active_tokens=n_frac_active_tokens)
            n_training_steps += 1
            n_training_tokens += layer_acts.shape[0] * self.world
            if self.checkpoint_thresholds and n_training_tokens > self.checkpoint_thresholds[0]:
                self.checkpoint(self.sparse_coder, n_training_tokens, act_freq_scores, n_frac_active_tokens)
                self.checkpoint_thresholds.pop(0)
            if pbar is not None:
                pbar.update(layer_acts.shape[0] * self.world)
                if n_training_steps % 50 == 0:        # .item() syncs the stream: keep it off the hot loop
                    pbar.set_description(f"Training SAE: Loss: {float(loss):.4f}, MSE Loss: {float(mse_loss):.4f}, "
                                         f"L0: {float(l0):.4f}", refresh=False)
        self._dp_flush(materialize=True)
        if cfg.n_checkpoints:
            self.checkpoint(self.sparse_coder, n_training_tokens, act_freq_scores, n_frac_active_tokens)
        if pbar is not None:
            pbar.close()
        self.final_stats = dict(loss=None if loss is None else float(loss), n_training_steps=n_training_steps,
                                n_training_tokens=n_training_tokens)
        return self.sparse_coder
