"""HookedSAEViT (SAEs spliced in place of HookPoints) against the reference's own run of its class
(tests/golden/gen_golden_sae_vit.py -> sae_vit_tiny.npz): outputs, cache keys and the tensors around the splice with one SAE
attached, two, one removed, all removed; the temporary-attachment entry points; the module tree after reset_saes is the one the
model was built with (so the HIP plan applies again)."""
import os
import numpy as np
import pytest
import torch

from vit_prisma_amd import HookedSAEViT, HookedViTConfig
from vit_prisma_amd.sae import StandardSparseAutoencoder, VisionModelSAERunnerConfig
from vit_prisma_amd.synth import ARCHS, synth_images, synth_sae_state, synth_vit_state

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def build():
    arch = ARCHS["tiny"]
    model = HookedSAEViT(HookedViTConfig(**arch, dtype=torch.float32, device="cpu"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model.eval()
    return model, arch, torch.from_numpy(synth_images(arch, 2, 1))


def make_sae(arch, layer, subtype, act, kw, seed):
    cfg = VisionModelSAERunnerConfig(hook_point_layer=layer, layer_subtype=subtype, d_in=arch["d_model"], expansion_factor=4,
                                     activation_fn_str=act, activation_fn_kwargs=kw, normalize_activations="layer_norm",
                                     initialization_method="independent", b_dec_init_method="mean", _device="cpu", _dtype="float32",
                                     log_to_wandb=False, use_ghost_grads=False, verbose=False)
    sae = StandardSparseAutoencoder(cfg)
    with torch.no_grad():
        for name, val in synth_sae_state(arch["d_model"], arch["d_model"] * 4, seed=seed).items():
            getattr(sae, name).copy_(torch.from_numpy(val))
    sae.eval()
    return sae


def check(model, x, G, tag):
    with torch.no_grad():
        out, cache = model.run_with_cache(x)
    assert list(cache.cache_dict.keys()) == [str(k) for k in G[f"{tag}_keys"]], tag
    assert rel(out.numpy(), G[f"{tag}_out"]) < TOL, tag
    for key in G.files:
        if key.startswith(tag + "::"):
            name = key.split("::", 1)[1]
            assert cache[name].shape == G[key].shape and rel(cache[name].numpy(), G[key]) < TOL, key
    return out


def test_spliced_saes_match_the_reference_run():
    G = np.load(os.path.join(GOLDEN, "sae_vit_tiny.npz"))
    model, arch, x = build()
    tree0 = [(n, id(m)) for n, m in model.named_modules()]
    check(model, x, G, "plain")
    a = make_sae(arch, 0, "hook_resid_post", "relu", {}, 3)
    b = make_sae(arch, 1, "hook_mlp_out", "topk", {"k": 8}, 4)
    assert a.cfg.hook_point == str(G["hook_point_a"]) and b.cfg.hook_point == str(G["hook_point_b"])
    model.add_sae(a)
    assert model.acts_to_saes == {a.cfg.hook_point: a} and a.cfg.return_out_only is True
    check(model, x, G, "one")
    with torch.no_grad():
        assert rel(model(x).numpy(), G["one_forward"]) < TOL
    model.add_sae(b)
    check(model, x, G, "two")
    model.reset_saes(a.cfg.hook_point)
    assert list(model.acts_to_saes) == [b.cfg.hook_point]
    check(model, x, G, "only_b")
    model.reset_saes()
    assert model.acts_to_saes == {}
    check(model, x, G, "reset")
    assert [(n, id(m)) for n, m in model.named_modules()] == tree0            # the tree the HIP plan was built for is back
    assert model._native_reason((x,), {}) in (None, "input is not on a GPU")


def test_temporary_attachment_and_unknown_hook_point(caplog):
    G = np.load(os.path.join(GOLDEN, "sae_vit_tiny.npz"))
    model, arch, x = build()
    a = make_sae(arch, 0, "hook_resid_post", "relu", {}, 3)
    b = make_sae(arch, 1, "hook_mlp_out", "topk", {"k": 8}, 4)
    with torch.no_grad():
        assert rel(model.run_with_saes(x, saes=[a]).numpy(), G["one_forward"]) < TOL
        assert model.acts_to_saes == {}                                      # detached again
        out, cache = model.run_with_cache_with_saes(x, saes=[a, b])
        assert rel(out.numpy(), G["two_out"]) < TOL and list(cache.cache_dict.keys()) == [str(k) for k in G["two_keys"]]
        # a permanently attached SAE comes back after a temporary one at the same point
        model.add_sae(a)
        a2 = make_sae(arch, 0, "hook_resid_post", "relu", {}, 9)
        with model.saes(saes=a2):
            assert model.acts_to_saes[a.cfg.hook_point] is a2
        assert model.acts_to_saes[a.cfg.hook_point] is a
        assert rel(model(x).numpy(), G["one_forward"]) < TOL
        # hooks and SAEs together: a hook on one of the spliced SAE's own points sees its tensor, the run is the spliced run
        seen = []
        o1 = model.run_with_hooks_with_saes(x, saes=[b], fwd_hooks=[(b.cfg.hook_point + ".hook_hidden_post", lambda t, hook: seen.append(tuple(t.shape)))])
        assert seen == [(2, 17, 4 * arch["d_model"])] and list(model.acts_to_saes) == [a.cfg.hook_point]
        assert rel(o1.numpy(), G["two_out"]) < TOL
        model.reset_saes()
    bogus = make_sae(arch, 7, "hook_resid_post", "relu", {}, 1)             # the tiny model has 2 blocks
    with caplog.at_level("WARNING"):
        model.add_sae(bogus)
    assert model.acts_to_saes == {} and "No hook found" in caplog.text


def test_alias_module_exposes_the_class():
    import vit_prisma_amd
    from vit_prisma_amd.compat import uninstall
    vit_prisma_amd.install_as("vit_prisma")
    try:
        from vit_prisma.models.base_vit import HookedSAEViT as Aliased
        assert Aliased is HookedSAEViT
    finally:
        uninstall("vit_prisma")


def test_hooks_registered_before_a_splice_are_gone_after_reset():
    """reset_saes restores the ORIGINAL HookPoint object (the tree the HIP plan was built for), but like the reference's fresh
    HookPoint() (base_vit.py:903) it carries no hooks afterwards -- a permanent hook registered on the point before add_sae must
    not come back to life (round-4 advisor finding)."""
    model, arch, x = build()
    a = make_sae(arch, 0, "hook_resid_post", "relu", {}, 3)
    fired = []
    name = a.cfg.hook_point
    original = model.hook_dict[name]
    model.add_perma_hook(name, lambda t, hook: fired.append(1))
    with torch.no_grad():
        model(x)
        assert fired == [1]
        model.add_sae(a)
        model(x)                                   # (the SAE stands in the point's place: the hook does not fire)
        assert fired == [1]
        model.reset_saes()
        assert model.hook_dict[name] is original and not original.fwd_hooks and not original.has_hooks()
        model(x)
    assert fired == [1]
    assert model._tree_matches()


@pytest.mark.timeout(600)
def test_spliced_saes_at_b32_size_match_the_reference_run():
    """The package's PyTorch path of HookedSAEViT at CLIP ViT-B/32 size (bs = 4, fp32; a top-k SAE in place of blocks.6.hook_resid_post, then a
    ReLU SAE in place of blocks.3.hook_mlp_out as well) against the reference's own run of its class: key order, shapes and
    fingerprints of all 217 / 220 cache entries (tests/golden/sae_vit_b32_bs4.json; the GPU suite holds the HIP plan to the same file)."""
    import json
    from oracle.vit_oracle import fingerprint
    with open(os.path.join(GOLDEN, "sae_vit_b32_bs4.json")) as f:
        G = json.load(f)
    arch = ARCHS["clip-vit-b32"]
    model = HookedSAEViT(HookedViTConfig(**arch, dtype=torch.float32, device="cpu"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model.eval()
    x = torch.from_numpy(synth_images(arch, G["batch"], G["seed"]))
    for tag, spec in zip(("one", "two"), G["saes"]):
        cfg = VisionModelSAERunnerConfig(hook_point_layer=spec["layer"], layer_subtype=spec["subtype"], d_in=arch["d_model"], expansion_factor=4,
                                         activation_fn_str=spec["act"], activation_fn_kwargs=spec["kw"], normalize_activations="layer_norm",
                                         initialization_method="independent", b_dec_init_method="mean", _device="cpu", _dtype="float32",
                                         log_to_wandb=False, use_ghost_grads=False, verbose=False)
        sae = StandardSparseAutoencoder(cfg)
        with torch.no_grad():
            for name, val in synth_sae_state(arch["d_model"], arch["d_model"] * 4, seed=spec["seed"]).items():
                getattr(sae, name).copy_(torch.from_numpy(val))
        sae.eval()
        model.add_sae(sae)
        with torch.no_grad():
            out, cache = model.run_with_cache(x)
        assert list(cache.cache_dict.keys()) == G[tag]["keys"], tag
        for k in G[tag]["keys"]:
            want = G[tag]["cache"][k]
            got = cache[k].numpy()
            assert list(got.shape) == want["shape"], (tag, k)
            fp = fingerprint(np.ascontiguousarray(got))
            assert abs(fp["l2"] - want["l2"]) <= 1e-4 * max(want["l2"], 1e-30), (tag, k)
        assert abs(fingerprint(out.numpy())["l2"] - G[tag]["out"]["l2"]) <= 1e-4 * G[tag]["out"]["l2"]
