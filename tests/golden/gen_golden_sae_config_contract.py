"""The reference's VisionModelSAERunnerConfig (sae/config.py) as a CONTRACT (build container only): for several keyword sets, every dataclass
field after __post_init__, every property, and what assigning through its setters (device / dtype / hook_point) leaves behind.

    python tests/golden/gen_golden_sae_config_contract.py     ->  tests/golden/sae_config_contract.json"""
import dataclasses
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

CASES = {
    "defaults_cpu": dict(_device="cpu", log_to_wandb=False, verbose=False),
    "topk_b32": dict(_device="cpu", log_to_wandb=False, verbose=False, hook_point_layer=6, layer_subtype="hook_resid_post", d_in=768,
                     expansion_factor=32, activation_fn_str="topk", activation_fn_kwargs={"k": 32}, train_batch_size=4096,
                     context_size=50, normalize_activations="layer_norm"),
    "cls_only": dict(_device="cpu", log_to_wandb=False, verbose=False, cls_token_only=True, context_size=50, store_batch_size=32,
                     n_batches_in_buffer=20, num_epochs=2),
    "patches_only": dict(_device="cpu", log_to_wandb=False, verbose=False, use_patches_only=True, context_size=50),
    "transcoder": dict(_device="cpu", log_to_wandb=False, verbose=False, is_transcoder=True, d_out=640, out_hook_point_layer=4,
                       layer_out_subtype="hook_mlp_out", hook_point_layer=4, layer_subtype="hook_resid_mid"),
    "gated_bf16": dict(_device="cpu", _dtype="bfloat16", log_to_wandb=False, verbose=False, architecture="gated", l1_coefficient=2e-3),
}
PROPS = ("device", "dtype", "hook_point", "out_hook_point", "tokens_per_buffer", "total_training_tokens", "total_training_steps",
         "total_training_images")


def plain(v):
    if isinstance(v, (torch.dtype, torch.device)):
        return str(v)
    if dataclasses.is_dataclass(v):
        return "<dataclass>"
    if isinstance(v, (list, tuple)):
        return [plain(t) for t in v]
    if isinstance(v, dict):
        return {str(k): plain(t) for k, t in v.items()}
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    return repr(type(v))


def snapshot(cfg):
    out = {"fields": {}, "props": {}}
    for f in dataclasses.fields(cfg):
        out["fields"][f.name] = plain(getattr(cfg, f.name, "<unset>"))
    for p in PROPS:
        try:
            out["props"][p] = plain(getattr(cfg, p))
        except Exception as e:                                     # (a property that raises for this keyword set is part of the contract)
            out["props"][p] = f"raises {type(e).__name__}"
    return out


def contract(Cfg):
    res = {}
    for tag, kw in CASES.items():
        cfg = Cfg(**kw)
        snap = snapshot(cfg)
        # the setters
        cfg.hook_point = "hook_embed"
        cfg.dtype = "float16"
        cfg.device = "cpu"
        snap["after_setters"] = {p: plain(getattr(cfg, p)) for p in ("hook_point", "dtype", "device")}
        res[tag] = snap
    return res


if __name__ == "__main__":
    from gen_golden_sae import ref_trainer_classes
    Cfg, _, _ = ref_trainer_classes()
    res = contract(Cfg)
    with open(os.path.join(HERE, "sae_config_contract.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print({k: (len(v["fields"]), v["props"]) for k, v in res.items()})
