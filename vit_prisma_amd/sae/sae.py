"""Sparse autoencoders behind the reference's module API.

Drop-in for /root/reference/src/vit_prisma/sae/sae.py: ``SparseAutoencoder`` (:29-533),
``StandardSparseAutoencoder`` (:535-645), ``TopK`` (:795-810), ``get_activation_fn`` (:813-839):
same parameter names / layouts (``W_enc [d_in,d_sae]``, ``W_dec [d_sae,d_in]``, ``b_enc``, ``b_dec``),
same HookPoints, same 7-tuple from ``forward``.

The module's ``forward`` is the faithful PyTorch implementation (hooks, autograd, every activation /
normalisation variant).  Training on an MI355X does not go through it: ``VisionSAETrainer`` drives
``native_sae.NativeSAE`` (HIP kernels) directly on this module's parameter storage.
"""
from __future__ import annotations

import logging
import math
import os
import pickle
from abc import ABC, abstractmethod
from typing import Any, Callable, Optional

import torch
from torch import nn

from ..hook_points import HookPoint
from ..hooked_root_module import HookedRootModule
from .config import VisionModelSAERunnerConfig


def _compat_pickle():
    """A ``pickle``-shaped module whose Unpickler resolves the reference's module paths (``vit_prisma.sae.config`` ...)
    to this package's classes -- what ``vit_prisma_amd.install_as("vit_prisma")`` does process-wide, scoped to one load."""
    import types
    remap = {"vit_prisma.sae.config": "vit_prisma_amd.sae.config", "vit_prisma.sae.sae": "vit_prisma_amd.sae.sae",
             "vit_prisma.sae.transcoder": "vit_prisma_amd.sae.variants",
             "vit_prisma.configs.HookedViTConfig": "vit_prisma_amd.configs"}

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module in remap and (module not in __import__("sys").modules or module.startswith("vit_prisma.")):
                import importlib
                tgt = importlib.import_module(remap[module])
                if hasattr(tgt, name):
                    return getattr(tgt, name)
                if module == "vit_prisma.sae.sae":                     # GatedSparseAutoencoder lives in variants here
                    from . import variants
                    if hasattr(variants, name):
                        return getattr(variants, name)
            return super().find_class(module, name)

    mod = types.ModuleType("pv_compat_pickle")
    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    mod.__name__ = "pickle"
    for n in ("Pickler", "dump", "dumps", "loads", "HIGHEST_PROTOCOL", "DEFAULT_PROTOCOL", "PickleError", "UnpicklingError"):
        setattr(mod, n, getattr(pickle, n))
    return mod


class TopK(nn.Module):
    """Keep the k largest pre-activations per row, apply ``postact_fn`` (ReLU), zero the rest."""

    def __init__(self, k: int, postact_fn: Callable[[torch.Tensor], torch.Tensor] = nn.ReLU()):
        super().__init__()
        self.k = k
        self.postact_fn = postact_fn

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        top = torch.topk(x, k=self.k, dim=-1)
        out = torch.zeros_like(x)
        out.scatter_(-1, top.indices, self.postact_fn(top.values))
        return out


def get_activation_fn(activation_fn: str, **kwargs: Any) -> Callable[[torch.Tensor], torch.Tensor]:
    if activation_fn == "relu":
        return nn.ReLU()
    if activation_fn == "tanh-relu":
        return lambda x: torch.tanh(torch.relu(x))
    if activation_fn == "topk":
        assert "k" in kwargs, "TopK activation function requires a k value."
        return TopK(kwargs.get("k", 64), kwargs.get("postact_fn", nn.ReLU()))
    raise ValueError(f"Unknown activation function: {activation_fn}")


class SparseAutoencoder(HookedRootModule, ABC):
    def __init__(self, cfg: VisionModelSAERunnerConfig):
        super().__init__()
        self.cfg = cfg
        self.d_in = cfg.d_in
        if not isinstance(self.d_in, int):
            raise ValueError(f"d_in must be an int but was {self.d_in}; {type(self.d_in)}")
        assert cfg.d_sae is not None
        self.d_sae = cfg.d_sae
        self.l1_coefficient = cfg.l1_coefficient
        self.lp_norm = cfg.lp_norm
        self.dtype = cfg.dtype
        self.device = cfg.device
        self.initialization_method = cfg.initialization_method
        self.zero_loss = torch.tensor(0.0, dtype=self.dtype, device=self.device)
        self.initialize_sae_weights()
        self.hook_sae_in = HookPoint()
        self.hook_hidden_pre = HookPoint()
        self.hook_hidden_post = HookPoint()
        self.hook_sae_out = HookPoint()
        self.activation_fn = get_activation_fn(cfg.activation_fn_str, **cfg.activation_fn_kwargs)
        self.setup()

    # ---- run-time input normalisation (sae.py:59-96) --------------------------------------------
    def run_time_activation_norm_fn_in(self, x: torch.Tensor) -> torch.Tensor:
        mode = self.cfg.normalize_activations
        if mode == "constant_norm_rescale":
            self.x_norm_coeff = (self.cfg.d_in ** 0.5) / x.norm(dim=-1, keepdim=True)
            return x * self.x_norm_coeff
        if mode == "layer_norm":
            mu = x.mean(dim=-1, keepdim=True)
            x = x - mu
            std = x.std(dim=-1, keepdim=True)          # unbiased
            self.ln_mu, self.ln_std = mu, std
            return x / (std + 1e-5)
        return x

    def run_time_activation_norm_fn_out(self, x: torch.Tensor) -> torch.Tensor:
        mode = self.cfg.normalize_activations
        if mode == "constant_norm_rescale":
            x = x / self.x_norm_coeff
            del self.x_norm_coeff
            return x
        if mode == "layer_norm":
            return x * self.ln_std + self.ln_mu
        return x

    def initialize_weights(self, out_features: int, in_features: int) -> torch.Tensor:
        """Kaiming-uniform rows normalised to unit L2 norm (sae.py:104-130)."""
        w = torch.empty(out_features, in_features, dtype=self.dtype, device=self.device)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        with torch.no_grad():
            w /= torch.norm(w, dim=1, keepdim=True)
        return w

    @abstractmethod
    def encode(self, x: torch.Tensor): ...

    @abstractmethod
    def decode(self, features: torch.Tensor): ...

    @abstractmethod
    def initialize_sae_weights(self): ...

    @abstractmethod
    def forward(self, x: torch.Tensor, dead_neuron_mask: torch.Tensor = None): ...

    # ---- losses ---------------------------------------------------------------------------------
    def _compute_mse_loss(self, x: torch.Tensor, sae_out: torch.Tensor) -> torch.Tensor:
        """Per-token squared error normalised by ||x - mean_batch(x)||_2, averaged over everything
        (sae.py:144-149)."""
        centred = x - x.mean(dim=0, keepdim=True)
        err = torch.nn.functional.mse_loss(sae_out, x.detach(), reduction="none")
        return (err / torch.norm(centred, p=2, dim=-1, keepdim=True)).mean()

    def _compute_ghost_residual_loss(self, x, sae_out, hidden_pre, dead_neuron_mask) -> torch.Tensor:
        """Ghost gradients (sae.py:151-179); PyTorch path only."""
        residual = x - sae_out
        residual_centred = residual - residual.mean(dim=0, keepdim=True)
        l2_resid = torch.norm(residual, dim=-1)
        dead_acts = torch.exp(hidden_pre[:, dead_neuron_mask])
        ghost_out = dead_acts @ self.W_dec[dead_neuron_mask, :]
        scale = l2_resid / (1e-6 + torch.norm(ghost_out, dim=-1) * 2)
        ghost_out = ghost_out * scale[:, None].detach()
        ghost = torch.pow(ghost_out - residual.detach().float(), 2) / (residual_centred.detach() ** 2).sum(dim=-1, keepdim=True).sqrt()
        rescale = (self._compute_mse_loss(x, sae_out) / (ghost + 1e-6)).detach()
        return (rescale * ghost).mean()

    # ---- decoder bias initialisation (sae.py:181-242) -------------------------------------------
    @torch.no_grad()
    def initialize_b_dec_with_precalculated(self, origin: torch.Tensor, transcoder_dec_b: torch.Tensor = None):
        self.b_dec.data = origin.clone().detach().to(dtype=self.dtype, device=self.device)

    @torch.no_grad()
    def initialize_b_dec_with_mean(self, all_activations: torch.Tensor):
        self.b_dec.data = all_activations.mean(dim=0).to(self.dtype).to(self.device)

    @torch.no_grad()
    def initialize_b_dec_with_geometric_median(self, all_activations: torch.Tensor):
        """b_dec <- the geometric median of the activations (sae.py:210-226; Weiszfeld, at most 100 iterations)."""
        from .geometric_median import compute_geometric_median
        self.initialize_b_dec_with_precalculated(compute_geometric_median(all_activations, maxiter=100).median)

    @torch.no_grad()
    def initialize_b_dec(self, all_activations: torch.Tensor):
        method = self.cfg.b_dec_init_method
        if method == "geometric_median":
            self.initialize_b_dec_with_geometric_median(all_activations)
        elif method == "mean":
            self.initialize_b_dec_with_mean(all_activations)
        elif method != "zeros":
            raise ValueError(f"Unexpected b_dec_init_method: {method}")

    # ---- decoder constraints (sae.py:275-297) ---------------------------------------------------
    @torch.no_grad()
    def set_decoder_norm_to_unit_norm(self):
        # in place on the Parameter itself (not through ``.data``): the version counter moves, which is how the native
        # engines notice an edit made behind their back (sae/native_sae.py: _ensure_shadows, the deferred-renorm key)
        self.W_dec.div_(torch.norm(self.W_dec, dim=1, keepdim=True))

    @torch.no_grad()
    def remove_gradient_parallel_to_decoder_directions(self):
        parallel = (self.W_dec.grad * self.W_dec.data).sum(dim=1, keepdim=True)
        self.W_dec.grad -= parallel * self.W_dec.data

    # ---- persistence (sae.py:299-320, 410-528: the pickled {"cfg", "state_dict"} format) --------
    def save_model(self, path: str):
        folder = os.path.dirname(path)
        if folder:
            os.makedirs(folder, exist_ok=True)
        # (a tensor that is a VIEW of larger storage -- the parameters of a transcoder of unequal widths are views of the engine's padded
        # buffers, trainer.py:_get_engine -- would be pickled with its whole storage and come back non-contiguous: saved as its own copy)
        blob = {"cfg": self.cfg, "state_dict": {k: (v if v.is_contiguous() and v.untyped_storage().nbytes() == v.numel() * v.element_size()
                                                    else v.contiguous().clone()) for k, v in self.state_dict().items()}}
        if path.endswith(".pt"):
            torch.save(blob, path)
        elif path.endswith(".pkl.gz"):
            import gzip
            with gzip.open(path, "wb") as f:
                pickle.dump(blob, f)
        else:
            raise ValueError(f"Unexpected file extension: {path}, supported extensions are .pt and .pkl.gz")
        logging.info(f"Saved model to {path}")

    @classmethod
    def load_from_pretrained(cls, weights_path, current_cfg=None, config_path=None):
        """Same signature and resolution order as the reference (sae.py:410-528): ``weights_path`` is a ``.pt`` /
        ``.pkl`` / ``.pkl.gz`` file holding either the legacy ``{"cfg", "state_dict"}`` blob (its pickled config is
        used when ``config_path`` is None) or a bare state dict (the config is then ``config.json`` next to the weights,
        as in the reference -- an explicit ``config_path`` is honoured first); ``current_cfg`` (a mapping or a config
        object) overrides every matching attribute; the class is picked from the config (transcoder / standard / gated).
        Checkpoints written by the reference pickle ``vit_prisma.sae.config.VisionModelSAERunnerConfig``: the unpickler
        maps ``vit_prisma.*`` onto this package, so they load without the reference installed."""
        if not os.path.isfile(weights_path):
            raise FileNotFoundError(f"No weights file found at: {weights_path}")
        try:
            if weights_path.endswith(".pt"):
                blob = torch.load(weights_path, map_location="cpu", weights_only=False, pickle_module=_compat_pickle())
            elif weights_path.endswith(".pkl.gz"):
                import gzip
                with gzip.open(weights_path, "rb") as f:
                    blob = _compat_pickle().Unpickler(f).load()
            elif weights_path.endswith(".pkl"):
                with open(weights_path, "rb") as f:
                    blob = _compat_pickle().Unpickler(f).load()
            else:
                raise ValueError(f"Unexpected file extension: {weights_path}")
        except (ValueError, FileNotFoundError):
            raise
        except Exception as e:
            raise IOError(f"Error loading the state dictionary from {weights_path}: {e}")
        is_legacy = isinstance(blob, dict) and "cfg" in blob and "state_dict" in blob
        if is_legacy and config_path is None:
            loaded_cfg, weights = blob["cfg"], blob["state_dict"]
        else:
            if config_path is None or not os.path.isfile(config_path):
                config_path = os.path.join(os.path.dirname(weights_path), "config.json")
            if not os.path.isfile(config_path):
                raise FileNotFoundError(f"No config file found at {config_path} and no legacy format detected")
            loaded_cfg = VisionModelSAERunnerConfig.load_config(config_path)
            weights = blob["state_dict"] if is_legacy else blob
        if not hasattr(loaded_cfg, "activation_fn_kwargs"):             # very old pickles (sae.py:486-497)
            slope = {"negative_slope": 0.01} if getattr(loaded_cfg, "activation_fn_str", "relu") == "leaky_relu" else {}
            loaded_cfg.activation_fn_kwargs = slope
        if current_cfg is not None:
            items = current_cfg.items() if hasattr(current_cfg, "items") else vars(current_cfg).items()
            for key, value in items:
                if hasattr(loaded_cfg, key):
                    try:
                        setattr(loaded_cfg, key, value)
                    except AttributeError:                              # read-only property of the config
                        pass
        if getattr(loaded_cfg, "is_transcoder", False):
            from .variants import Transcoder
            model_cls = Transcoder
        elif getattr(loaded_cfg, "architecture", "standard") in ("standard", "vanilla"):
            model_cls = StandardSparseAutoencoder
        elif loaded_cfg.architecture == "gated":
            from .variants import GatedSparseAutoencoder
            model_cls = GatedSparseAutoencoder
        else:
            raise ValueError(f"Unsupported architecture type: {loaded_cfg.architecture}")
        inst = model_cls(loaded_cfg)
        inst.load_state_dict(weights)
        return inst

    def get_name(self) -> str:
        return f"sparse_autoencoder_{self.cfg.model_name}_{self.cfg.hook_point}_{self.cfg.d_sae}"


class StandardSparseAutoencoder(SparseAutoencoder):
    def initialize_sae_weights(self):
        self.W_dec = nn.Parameter(self.initialize_weights(self.d_sae, self.d_in))
        if self.initialization_method == "independent":
            self.W_enc = nn.Parameter(self.initialize_weights(self.d_in, self.d_sae))
        elif self.initialization_method == "encoder_transpose_decoder":
            self.W_enc = nn.Parameter(self.W_dec.data.t().clone())
        else:
            raise ValueError(f"Unknown initialization method: {self.initialization_method}")
        self.b_enc = nn.Parameter(torch.zeros(self.d_sae, dtype=self.dtype, device=self.device))
        self.b_dec = nn.Parameter(torch.zeros(self.d_in, dtype=self.dtype, device=self.device))

    # ---- native inference path (SAE substitution / evals run the SAE inside a ViT hook) ----------
    def use_native(self, flag: Optional[bool]) -> "StandardSparseAutoencoder":
        """None (default): ``forward`` / ``encode`` run on the HIP kernels whenever that is observationally identical
        (see ``native_fallback_reason``); True: or raise; False: always PyTorch."""
        self._native_pref = flag
        return self

    # ---- parameters a native training engine keeps in its own layout -------------------------------------------------
    # VisionSAETrainer's single-process engine trains W_enc in its transposed fp32 master and rewrites the parameter's own
    # [d_in, d_sae] layout only on demand (NativeSAE.lazy_w_enc).  Every way of reaching the parameter goes through one of
    # these, so a reader always sees current values.
    def _native_sync(self) -> None:
        fn = self.__dict__.get("_native_sync_fn")
        if fn is not None:
            fn()

    def __getattr__(self, name: str):
        if name == "W_enc":
            self._native_sync()
        return super().__getattr__(name)

    def state_dict(self, *args, **kwargs):
        self._native_sync()
        return super().state_dict(*args, **kwargs)

    def named_parameters(self, *args, **kwargs):
        self._native_sync()
        return super().named_parameters(*args, **kwargs)

    # the paths that read ``self._parameters`` directly: .to() / .half() / .cpu() (Module._apply), copy.deepcopy, pickling
    def _apply(self, fn, *args, **kwargs):
        self._native_sync()
        return super()._apply(fn, *args, **kwargs)

    def __deepcopy__(self, memo):
        self._native_sync()
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_native_sync_fn":                           # (the copy is not the module the engine trains)
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        self._native_sync()
        state = dict(self.__dict__)
        state.pop("_native_sync_fn", None)
        return state

    def _native_reason(self, x: torch.Tensor, need_hidden_pre: bool = False) -> Optional[str]:
        cfg = self.cfg
        if getattr(self, "_native_pref", None) is False:
            return "disabled by use_native(False)"
        if not x.is_cuda:
            return "input is not on a GPU"
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return "autograd is recording (the native forward has no backward; wrap the call in torch.no_grad())"
        if need_hidden_pre:
            return "hidden_pre [N, d_sae] is never materialised by the native encoder"
        if cfg.activation_fn_str != "topk" or not isinstance(self.activation_fn, TopK) or not isinstance(self.activation_fn.postact_fn, nn.ReLU):
            return f"activation {cfg.activation_fn_str!r} (the native encoder is the top-k one)"
        if cfg.normalize_activations not in ("layer_norm", "constant_norm_rescale", "none", None):
            return f"normalize_activations={cfg.normalize_activations!r}"
        if self.dtype != torch.float32 or any(p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() for p in self.parameters()):
            return "parameters are not contiguous fp32 CUDA tensors"
        k = cfg.activation_fn_kwargs.get("k", 0)
        if not (cfg.d_in % 4 == 0 and cfg.d_in <= 1280 and cfg.d_sae % 4 == 0 and cfg.d_sae <= 65536 and 1 <= k <= 256):
            return "shape outside the native plan's limits"
        if self.is_caching or any(hp.fwd_hooks or hp.bwd_hooks for hp in (self.hook_sae_in, self.hook_hidden_pre, self.hook_hidden_post, self.hook_sae_out)):
            return "hooks on the SAE's own hook points"
        return None

    def _native_engine(self, n_tokens: int):
        from .native_sae import NativeSAE
        eng = getattr(self, "_engine", None)
        stale = eng is not None and any(eng.params[n].data_ptr() != getattr(self, n).data_ptr() for n in ("W_enc", "W_dec", "b_enc", "b_dec"))
        if eng is None or stale or eng.max_tokens < n_tokens:
            eng = NativeSAE(self.W_enc, self.W_dec, self.b_enc, self.b_dec, k=self.cfg.activation_fn_kwargs["k"],
                            layer_norm=self.cfg.normalize_activations, max_tokens=max(n_tokens, 4096), inference=True)
            object.__setattr__(self, "_engine", eng)          # (not a submodule / parameter)
        return eng

    def _native_dispatch(self, x: torch.Tensor, need_hidden_pre: bool = False) -> bool:
        why = self._native_reason(x, need_hidden_pre)
        self.native_fallback_reason = why
        self.last_run_native = why is None
        if why is not None and getattr(self, "_native_pref", None) is True:
            from .._native import NativeError
            raise NativeError(f"use_native(True): {why}")
        return why is None

    def _dense_acts(self, idx: torch.Tensor, val: torch.Tensor, lead_shape) -> torch.Tensor:
        acts = torch.zeros(idx.shape[0], self.d_sae, dtype=self.dtype, device=idx.device)
        acts.scatter_(1, idx.long(), val)                     # TopK.forward's dense output (sae.py:808-809)
        return acts.view(*lead_shape, self.d_sae)

    def encode(self, x: torch.Tensor, return_hidden_pre: bool = False):
        if self._native_dispatch(x, need_hidden_pre=return_hidden_pre):
            xf = x.to(self.dtype)
            sae_in = self.run_time_activation_norm_fn_in(xf) - self.b_dec          # (cheap elementwise; also sets ln_mu / ln_std for decode)
            flat = xf.reshape(-1, self.d_in)
            idx, val, _, _ = self._native_engine(flat.shape[0]).encode_topk(flat)
            return sae_in, self._dense_acts(idx, val, x.shape[:-1])
        x = x.to(self.dtype)
        sae_in = self.hook_sae_in(self.run_time_activation_norm_fn_in(x) - self.b_dec)
        hidden_pre = self.hook_hidden_pre(sae_in @ self.W_enc + self.b_enc)
        feature_acts = self.hook_hidden_post(self.activation_fn(hidden_pre))
        if return_hidden_pre:
            return sae_in, feature_acts, hidden_pre
        return sae_in, feature_acts

    def decode(self, features: torch.Tensor) -> torch.Tensor:
        sae_out = self.hook_sae_out(features @ self.W_dec + self.b_dec)
        return self.run_time_activation_norm_fn_out(sae_out)

    def forward(self, x: torch.Tensor, dead_neuron_mask: torch.Tensor = None, *args, **kwargs):
        ghost_wanted = self.cfg.use_ghost_grads and self.training and dead_neuron_mask is not None
        if not ghost_wanted and self._native_dispatch(x):
            # top-k encode (filtered fp16 MFMA + exact re-scoring) + sparse decode + LN-out on the HIP kernels: the
            # [N, d_sae] pre-activations are never formed; the dense feature_acts the API returns is scattered on demand
            xf = x.to(self.dtype)
            flat = xf.reshape(-1, self.d_in)
            out, idx, val = self._native_engine(flat.shape[0]).forward(flat)
            sae_out = out.clone().view(xf.shape)
            if getattr(self.cfg, "return_out_only", False):
                return sae_out
            feature_acts = self._dense_acts(idx, val, x.shape[:-1])
            mse_loss = self._compute_mse_loss(xf, sae_out)
            return sae_out, feature_acts, mse_loss, mse_loss, None, self.zero_loss, torch.tensor(0.0)
        _, feature_acts, hidden_pre = self.encode(x, return_hidden_pre=True)
        sae_out = self.decode(feature_acts)
        mse_loss = self._compute_mse_loss(x, sae_out)
        if self.cfg.use_ghost_grads and self.training and dead_neuron_mask is not None:
            ghost = self._compute_ghost_residual_loss(x, sae_out, hidden_pre, dead_neuron_mask)
        else:
            ghost = self.zero_loss
        sparsity = feature_acts.norm(p=self.lp_norm, dim=1).mean(dim=(0,))
        l1_loss = self.l1_coefficient * sparsity if self.cfg.activation_fn_str != "topk" else None
        loss = mse_loss + (l1_loss if l1_loss is not None else 0) + ghost
        aux = torch.tensor(0.0)
        if getattr(self.cfg, "return_out_only", False):
            return sae_out
        return sae_out, feature_acts, loss, mse_loss, l1_loss, ghost, aux
