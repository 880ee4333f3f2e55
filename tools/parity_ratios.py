"""Per-key error of the native bf16 path at the configurations bench.py reports, against the fp32 numpy oracle,
as a ratio of the reference's own bf16-vs-fp32 error for the same images (tests/golden/vit_*_budget_sub*.json).
Diagnostic twin of tests/test_native_vit_gpu.py::test_bf16_bs512_* / test_bf16_l14_bs128_*: prints the worst
ratios and writes everything to gpurun_out/parity_ratios.json.      python tools/parity_ratios.py [b32] [l14]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.vit_oracle import vit_forward  # noqa: E402
from vit_prisma_amd import HookedViT, HookedViTConfig  # noqa: E402
from vit_prisma_amd.synth import ARCHS, synth_images, synth_vit_state  # noqa: E402
from conftest import GOLDEN, rel_fro  # noqa: E402


def build(arch_name):
    arch = ARCHS[arch_name]
    model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
    sd = synth_vit_state(arch, 0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return model.to(torch.bfloat16).cuda().eval().use_native(True), arch, sd


def ratios(cache, c_ref, budget, sub):
    out = {}
    for k, ref in c_ref.items():
        got = cache[k][sub].float().cpu().numpy()
        err = rel_fro(got, ref)
        out[k] = {"err": err, "budget": budget[k]["rel_fro"], "ratio": err / max(budget[k]["rel_fro"], 1e-30)}
    return out


def main():
    what = sys.argv[1:] or ["b32", "l14"]
    res = {}
    if "b32" in what:
        with open(os.path.join(GOLDEN, "vit_b32_bf16_budget_sub512.json")) as f:
            G = json.load(f)
        sub = G["images"]
        model, arch, sd = build("clip-vit-b32")
        imgs = synth_images(arch, 512, G["seed"])
        t0 = time.time()
        o_ref, c_ref = vit_forward(sd, arch, imgs[sub])
        print("oracle 16 images", time.time() - t0, "s", flush=True)
        x = torch.from_numpy(imgs).cuda().bfloat16()
        with torch.no_grad():
            out, cache = model.run_with_cache(x)
        torch.cuda.synchronize()
        assert model.last_run_native
        r = ratios(cache, c_ref, G["budget"], sub)
        r["__out__"] = {"err": rel_fro(out[sub].float().cpu().numpy(), o_ref), "budget": G["budget"]["__out__"]["rel_fro"]}
        r["__out__"]["ratio"] = r["__out__"]["err"] / r["__out__"]["budget"]
        res["b32_bs512"] = r
        del cache
        h_ref, hc_ref = vit_forward(sd, arch, imgs[sub], stop_at_layer=7, names_filter=["blocks.6.hook_resid_post"])
        with torch.no_grad():
            hout, hcache = model.run_with_cache(x, stop_at_layer=7, names_filter=["blocks.6.hook_resid_post"])
        res["b32_bs512_harvest"] = ratios(hcache, hc_ref, G["harvest"], sub)
        # bs = 4 (the round-1 test) for comparison
        with open(os.path.join(GOLDEN, "vit_b32_bf16_budget.json")) as f:
            G4 = json.load(f)["budget"]
        i4 = synth_images(arch, 4, 1)
        o4, c4 = vit_forward(sd, arch, i4)
        with torch.no_grad():
            _, cache4 = model.run_with_cache(torch.from_numpy(i4).cuda().bfloat16())
        res["b32_bs4"] = ratios(cache4, c4, G4, list(range(4)))
        del model, cache4, hcache, x
        torch.cuda.empty_cache()
    if "l14" in what and os.path.exists(os.path.join(GOLDEN, "vit_l14_bf16_budget_sub128.json")):
        with open(os.path.join(GOLDEN, "vit_l14_bf16_budget_sub128.json")) as f:
            G = json.load(f)
        sub = G["images"]
        model, arch, sd = build("clip-vit-l14-336")
        imgs = synth_images(arch, 128, G["seed"])
        want = [f"blocks.{l}.attn.hook_pattern" for l in (0, 23)]
        t0 = time.time()
        o_ref, c_ref = vit_forward(sd, arch, imgs[sub], names_filter=want)
        print("oracle l14 2 images", time.time() - t0, "s", flush=True)
        x = torch.from_numpy(imgs).cuda().bfloat16()
        with torch.no_grad():
            out, cache = model.run_with_cache(x, names_filter=lambda n: n.endswith("attn.hook_pattern"))
        torch.cuda.synchronize()
        assert model.last_run_native and len(cache) == 24
        res["l14_bs128"] = ratios({k: cache[k] for k in want}, c_ref, G["budget"], sub)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_ratios.json"), "w") as f:
        json.dump(res, f, indent=1)
    for cfg, r in res.items():
        worst = sorted(r.items(), key=lambda kv: -kv[1]["ratio"])[:12]
        print(cfg, "keys", len(r), "n>1.0:", sum(v["ratio"] > 1.0 for v in r.values()))
        for k, v in worst:
            print(f"   {k:42s} err {v['err']:.3e} budget {v['budget']:.3e} ratio {v['ratio']:.3f}")


if __name__ == "__main__":
    main()
