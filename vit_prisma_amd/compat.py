"""``import vit_prisma`` compatibility: run existing Prisma notebooks / scripts / checkpoints against this build.

    import vit_prisma_amd
    vit_prisma_amd.install_as("vit_prisma")
    from vit_prisma.models.base_vit import HookedViT           # -> vit_prisma_amd.HookedViT
    from vit_prisma.sae.config import VisionModelSAERunnerConfig

Registers ``sys.modules`` aliases for the reference's import paths that sit on the hot path
(/root/reference/src/vit_prisma/{models/base_vit (HookedViT, HookedSAEViT), models/model_loader (offline form), models/layers/*, configs/HookedViTConfig,
prisma_tools/{hook_point, hooked_root_module, activation_cache, lens_handle}, sae/{config, sae,
train_sae}, sae/training/{activations_store, geometric_median, get_scheduler}}.py).  Reference SAE
checkpoints pickle ``vit_prisma.sae.config.VisionModelSAERunnerConfig`` inside the ``.pt``
(sae.py:299-320): with the alias in place ``torch.load`` resolves it to this package's class, so
``StandardSparseAutoencoder.load_from_pretrained`` reads them unchanged.

Nothing is aliased when a real ``vit_prisma`` package is already imported (``force=True`` overrides).
"""
from __future__ import annotations

import importlib
import importlib.machinery
import sys
import types
from typing import Dict, Iterable


def _shell(name: str, attrs: Dict[str, object], is_pkg: bool) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=is_pkg)
    if is_pkg:
        m.__path__ = []          # a package with no files behind it: submodules come from sys.modules only
    m._pv_alias = True
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _public(mod: types.ModuleType, names: Iterable[str] = ()) -> Dict[str, object]:
    names = list(names) or [n for n in dir(mod) if not n.startswith("_")]
    return {n: getattr(mod, n) for n in names if hasattr(mod, n)}


def install_as(name: str = "vit_prisma", force: bool = False) -> None:
    existing = sys.modules.get(name)
    if existing is not None and not getattr(existing, "_pv_alias", False) and not force:
        raise RuntimeError(f"a real '{name}' package is already imported; pass force=True to shadow it")
    import vit_prisma_amd as A
    from . import activation_cache, configs, hook_points, hooked_root_module, model_loader, utils as U, vit
    from .sae import config as sae_config, geometric_median, get_scheduler, sae as sae_mod, store, trainer, variants

    layers = dict(
        attention=_public(vit, ["Attention"]), mlp=_public(vit, ["MLP"]), head=_public(vit, ["Head"]),
        layer_norm=_public(vit, ["LayerNorm", "LayerNormPre"]), patch_embedding=_public(vit, ["PatchEmbedding"]),
        position_embedding=_public(vit, ["PosEmbedding"]), transformer_block=_public(vit, ["TransformerBlock"]))
    tree = {
        "": (True, _public(A, A.__all__)),
        "models": (True, {}),
        "models.base_vit": (False, {**_public(vit), "HookedSAEViT": A.HookedSAEViT}),
        "models.model_loader": (False, _public(model_loader, ["load_hooked_model", "load_config", "list_available_models"])),
        "models.layers": (True, {}),
        **{f"models.layers.{k}": (False, v) for k, v in layers.items()},
        "configs": (True, _public(configs, ["HookedViTConfig"])),
        "configs.HookedViTConfig": (False, _public(configs)),
        "prisma_tools": (True, {}),
        "prisma_tools.hook_point": (False, _public(hook_points, ["HookPoint"])),
        "prisma_tools.lens_handle": (False, _public(hook_points, ["LensHandle"])),
        "prisma_tools.hooked_root_module": (False, _public(hooked_root_module)),
        "prisma_tools.activation_cache": (False, _public(activation_cache)),
        "utils": (True, {}),
        "utils.prisma_utils": (False, _public(U)),
        "sae": (True, {**_public(A.sae, A.sae.__all__)}),
        "sae.config": (False, _public(sae_config)),
        "sae.sae": (False, {**_public(sae_mod), **_public(variants, ["GatedSparseAutoencoder"])}),
        "sae.transcoder": (False, _public(variants, ["Transcoder"])),
        "sae.train_sae": (False, _public(trainer)),
        "sae.training": (True, {}),
        "sae.training.activations_store": (False, _public(store)),
        "sae.training.geometric_median": (False, _public(geometric_median)),
        "sae.training.get_scheduler": (False, _public(get_scheduler)),
    }
    mods = {}
    for rel, (is_pkg, attrs) in tree.items():
        full = name if rel == "" else f"{name}.{rel}"
        mods[rel] = _shell(full, attrs, is_pkg)
    for rel, m in mods.items():          # parents carry their children as attributes, like real packages
        if rel:
            parent, _, leaf = rel.rpartition(".")
            setattr(mods[parent], leaf, m)
    for rel, m in mods.items():
        sys.modules[name if rel == "" else f"{name}.{rel}"] = m
    importlib.invalidate_caches()


def uninstall(name: str = "vit_prisma") -> None:
    for k in [k for k, m in sys.modules.items() if (k == name or k.startswith(name + ".")) and getattr(m, "_pv_alias", False)]:
        del sys.modules[k]
