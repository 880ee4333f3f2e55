"""Golden runs of the reference with forward hooks ON its flag-gated HookPoints (build container only): the tiny model, fp32, the flag
sets of gen_golden_vit_flags.py, hooks that EDIT `attn.hook_result` (per-head result, use_attn_result), `hook_mlp_in` (use_hook_mlp_in),
`hook_attn_in` / `hook_q_input` (the per-head block inputs, use_attn_in / use_split_qkv_input) and an ordinary point beside them.

    python tests/golden/gen_golden_vit_flag_hooks.py     ->  tests/golden/vit_tiny_flag_hooks.npz

Per case: the output, the key order and every cache tensor of `run_with_cache` under `model.hooks(fwd_hooks=...)`."""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from vit_prisma_amd.synth import synth_images

# (hook functions are pure functions of the tensor: the test re-creates them from this table)
EDITS = {"half": lambda t, hook: t * 0.5, "plus1": lambda t, hook: t + 1.0, "double": lambda t, hook: t * 2.0,
         "zero_head1": lambda t, hook: torch.cat([t[:, :, :1], torch.zeros_like(t[:, :, 1:2]), t[:, :, 2:]], dim=2)}
CASES = {
    "result_mlp": (dict(use_attn_result=True, use_hook_mlp_in=True),
                   [("blocks.0.attn.hook_result", "half"), ("blocks.1.hook_mlp_in", "plus1"), ("blocks.1.attn.hook_z", "double")]),
    "all": (dict(use_attn_result=True, use_split_qkv_input=True, use_attn_in=True, use_hook_mlp_in=True),
            [("blocks.0.attn.hook_result", "half"), ("blocks.1.hook_mlp_in", "plus1"), ("blocks.1.attn.hook_z", "double")]),
    "attn_in": (dict(use_attn_in=True), [("blocks.1.hook_attn_in", "zero_head1"), ("blocks.0.hook_resid_mid", "plus1")]),
    "split": (dict(use_split_qkv_input=True), [("blocks.0.hook_q_input", "zero_head1"), ("blocks.1.hook_v_input", "half")]),
}

if __name__ == "__main__":
    from gen_golden_vit import build_reference_model
    blob = {}
    for tag, (flags, hooks) in CASES.items():
        model, arch = build_reference_model("tiny")
        for k, v in flags.items():
            setattr(model.cfg, k, v)
        x = torch.from_numpy(synth_images(arch, 2, 1))
        with torch.no_grad(), model.hooks(fwd_hooks=[(n, EDITS[e]) for n, e in hooks]):
            out, cache = model.run_with_cache(x)
        blob[f"{tag}::__out__"] = out.numpy()
        blob[f"{tag}::__keys__"] = np.array(list(cache.cache_dict.keys()))
        blob[f"{tag}::__hooks__"] = np.array([f"{n}={e}" for n, e in hooks])
        for k, v in cache.cache_dict.items():
            blob[f"{tag}::{k}"] = np.ascontiguousarray(v.numpy())
        print(tag, len(cache.cache_dict), float(out.abs().sum()))
    np.savez_compressed(os.path.join(HERE, "vit_tiny_flag_hooks.npz"), **blob)
    print(os.path.getsize(os.path.join(HERE, "vit_tiny_flag_hooks.npz")) // 1024, "kB")
