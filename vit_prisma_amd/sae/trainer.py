"""VisionSAETrainer: the SAE training loop behind the reference's trainer API.

Same constructor / ``train_step`` / ``run`` contract as
/root/reference/src/vit_prisma/sae/train_sae.py:61-861 (signature ``(cfg, model, dataset,
eval_dataset=None)``, ``train_step`` argument names and 7-tuple, ``run() -> sae``).  The order of
operations inside a step is the reference's (:278-411): renorm decoder -> zero_grad -> forward ->
firing statistics -> backward -> clip_grad_norm_ -> remove parallel gradient -> Adam -> scheduler.

On an MI355X (fp32 top-k standard SAE with layer_norm / no input normalisation, no ghost grads) the
step runs on ``NativeSAE`` (HIP kernels) directly on the module's parameter storage; Adam moments live
in the engine.  Everything else (ReLU+L1, gated, ghost grads, CPU) takes the PyTorch path below, which
is the reference algorithm verbatim.

Data parallel (new functionality, SURVEY.md section 8e -- the reference is single-process): one process
per GPU; per step ONE 768-float all-reduce (global batch mean for the loss normaliser) and ONE RCCL
all-reduce of the flat 37.77 M-float gradient buffer, issued before the clip norm, which is therefore
the GLOBAL gradient norm exactly as in a single process at the global batch size.
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
        if self.is_transcoder or cfg.architecture != "standard":
            raise NotImplementedError("only the standard SAE architecture is on the MI355X hot path")
        self.model = model
        self.dataset = dataset
        self.eval_dataset = eval_dataset
        self.bad_run_check = cfg.min_l0 is not None and cfg.min_explained_variance is not None
        torch.manual_seed(cfg.seed)
        self.sparse_coder = sparse_coder if sparse_coder is not None else StandardSparseAutoencoder(cfg)
        self.sae = self.sparse_coder                      # legacy alias
        self.activations_store = activations_store
        if self.activations_store is None and dataset is not None:
            self.activations_store = VisionActivationsStore(cfg, model, dataset, eval_dataset=eval_dataset,
                                                            num_workers=0)
        self.checkpoint_thresholds = self.get_checkpoint_thresholds()
        self._engine = None
        self.rank, self.world = _dist_info()

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
            self.sparse_coder.initialize_b_dec_with_precalculated(medians[lid])
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
    def _native_ok(self, sae, x: torch.Tensor) -> bool:
        cfg = sae.cfg
        return (x.is_cuda and isinstance(sae, StandardSparseAutoencoder) and cfg.activation_fn_str == "topk"
                and cfg.dtype == torch.float32 and not cfg.use_ghost_grads
                and cfg.normalize_activations in ("layer_norm", "none", None)
                and all(p.is_cuda and p.dtype == torch.float32 for p in sae.parameters())
                and cfg.d_in % 4 == 0 and cfg.d_in <= 1024 and cfg.d_sae % 4 == 0 and cfg.d_sae <= 32768
                and 1 <= cfg.activation_fn_kwargs.get("k", 0) <= 64
                and os.environ.get("PV_SAE_NATIVE", "1") != "0")

    def _get_engine(self, sae, n_tokens: int):
        from .native_sae import NativeSAE
        eng = self._engine
        if eng is None or eng.max_tokens < n_tokens or eng.params["W_enc"].data_ptr() != sae.W_enc.data_ptr():
            eng = NativeSAE(sae.W_enc.data, sae.W_dec.data, sae.b_enc.data, sae.b_dec.data,
                            k=sae.cfg.activation_fn_kwargs["k"], layer_norm=sae.cfg.normalize_activations == "layer_norm",
                            max_tokens=max(n_tokens, self.cfg.train_batch_size // self.world))
            self._engine = eng
        return eng

    # ---- one step ---------------------------------------------------------------------------------
    def train_step(self, sparse_autoencoder, optimizer, scheduler, act_freq_scores, n_forward_passes_since_fired,
                   n_frac_active_tokens, layer_acts, n_training_steps, n_training_tokens):
        hp = sparse_autoencoder.cfg
        layers = hp.hook_point_layer if isinstance(hp.hook_point_layer, list) else [hp.hook_point_layer]
        layer_id = 0 if isinstance(hp.hook_point_layer, list) else layers.index(hp.hook_point_layer)
        sae_in = layer_acts[:, layer_id, :]
        sparse_autoencoder.train()

        if (n_training_steps + 1) % self.cfg.feature_sampling_window == 0:     # train_sae.py:310-326
            feature_sparsity = act_freq_scores / n_frac_active_tokens
            self._log_feature_sparsity(feature_sparsity, n_training_steps)
            act_freq_scores = torch.zeros(hp.d_sae, device=hp.device)
            n_frac_active_tokens = 0

        if self._native_ok(sparse_autoencoder, sae_in):
            loss, mse_loss, l1_loss, l0 = self._native_step(sparse_autoencoder, optimizer, scheduler, sae_in,
                                                            act_freq_scores, n_forward_passes_since_fired)
        else:
            loss, mse_loss, l1_loss, l0 = self._torch_step(sparse_autoencoder, optimizer, scheduler, sae_in,
                                                           act_freq_scores, n_forward_passes_since_fired)
        n_frac_active_tokens += sae_in.shape[0] * self.world
        self.last_step_native = self._native_ok(sparse_autoencoder, sae_in)
        return loss, mse_loss, l1_loss, l0, act_freq_scores, n_forward_passes_since_fired, n_frac_active_tokens

    def _native_step(self, sae, optimizer, scheduler, x, act_freq_scores, n_since_fired):
        eng = self._get_engine(sae, x.shape[0])
        lr = optimizer.param_groups[0]["lr"]
        # statistics tensors are the caller's: the kernels update them in place
        eng.act_freq_scores = act_freq_scores
        eng.n_fwd_since_fired = n_since_fired
        eng.renorm_decoder()                                    # set_decoder_norm_to_unit_norm
        if self.world == 1:
            eng.step(x, update_stats=True)
        else:
            import torch.distributed as dist
            n_global = x.shape[0] * self.world
            bm = x.float().sum(dim=0)
            dist.all_reduce(bm)                                 # global batch mean (sae.py:145)
            eng.step(x, batch_mean=bm / n_global, n_global=n_global, update_stats=False)
            dist.all_reduce(eng.flat_g)                         # one RCCL all-reduce of all four gradients
            dist.all_reduce(eng.fire_count)
            dist.all_reduce(eng.scalars[:3])
            eng.scalars[2] /= self.world                        # l0 is a mean over tokens
            fired = eng.fire_count > 0                          # train_sae.py:356-361 on the global batch
            n_since_fired += 1
            n_since_fired[fired] = 0
            act_freq_scores += eng.fire_count
        eng.grad_sqnorm()                                       # clip_grad_norm_ over the (global) gradient
        eng.apply(lr, self.cfg.max_grad_norm)
        optimizer._opt_called = True                            # the native apply IS the optimizer step
        scheduler.step()
        sc = eng.scalars.clone()
        return sc[0], sc[1], None, sc[2]

    def _torch_step(self, sae, optimizer, scheduler, x, act_freq_scores, n_since_fired):
        """The reference algorithm on PyTorch autograd (CPU, ReLU/L1, ghost grads, ...)."""
        sae.set_decoder_norm_to_unit_norm()
        optimizer.zero_grad()
        dead = (n_since_fired > sae.cfg.dead_feature_window).bool()
        sae_out, feature_acts, loss, mse_loss, l1_loss, ghost, aux = sae(x, dead)
        with torch.no_grad():
            did_fire = (feature_acts > 0).float().sum(-2) > 0
            n_since_fired += 1
            n_since_fired[did_fire] = 0
            act_freq_scores += (feature_acts.abs() > 0).float().sum(0)
            l0 = (feature_acts > 0).float().sum(-1).mean()
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
        if self.rank != 0:
            return
        folder = self.cfg.checkpoint_path
        os.makedirs(folder, exist_ok=True)
        self.cfg.save_config(os.path.join(folder, "config.json"))
        sae.set_decoder_norm_to_unit_norm()
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
        if cfg.n_checkpoints:
            self.checkpoint(self.sparse_coder, n_training_tokens, act_freq_scores, n_frac_active_tokens)
        if pbar is not None:
            pbar.close()
        self.final_stats = dict(loss=None if loss is None else float(loss), n_training_steps=n_training_steps,
                                n_training_tokens=n_training_tokens)
        return self.sparse_coder
