"""Hook-runtime semantics of HookedRootModule / HookPoint (CPU).  Coverage follows the reference's own
strategy (/root/reference/tests/test_hooks.py: attach/detach, perma hooks, nested hooks() contexts incl.
failure unwinding, reset inside a context, flag-gated hooks + cached shapes, prepend ordering) plus the
cache-key order test (/root/reference/tests/test_cache_hook_names.py) and the fine print of
SURVEY.md section 8b."""
import pytest
import torch

from vit_prisma_amd import HookedViT, HookedViTConfig

B, S, P = 2, 32, 8
T = (S // P) ** 2 + 1


@pytest.fixture()
def model():
    torch.manual_seed(0)
    cfg = HookedViTConfig(n_layers=2, d_model=16, d_head=8, d_mlp=32, n_heads=2, patch_size=P, image_size=S,
                          n_classes=5, return_type="logits")
    return HookedViT(cfg).eval()


@pytest.fixture()
def x():
    return torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1))


class Tally:
    def __init__(self):
        self.n = 0
        self.order = []

    def __call__(self, t, hook):
        self.n += 1
        self.order.append(hook.name)


def n_fwd(model, name="hook_embed"):
    return len(model.hook_dict[name].fwd_hooks)


def test_run_with_hooks_attaches_and_detaches(model, x):
    t = Tally()
    model.run_with_hooks(x, fwd_hooks=[(lambda n: n == "hook_embed", t)])
    assert t.n == 1 and all(len(hp.fwd_hooks) == 0 for hp in model.hook_dict.values())
    model.run_with_hooks(x, fwd_hooks=[("blocks.1.hook_resid_post", t)], reset_hooks_end=False)
    assert n_fwd(model, "blocks.1.hook_resid_post") == 1
    model.reset_hooks()
    assert n_fwd(model, "blocks.1.hook_resid_post") == 0


def test_permanent_hooks_survive_resets(model, x):
    t = Tally()
    model.add_perma_hook("hook_embed", t)
    model.run_with_hooks(x, fwd_hooks=[])
    model.reset_hooks()
    model.remove_all_hook_fns()
    assert n_fwd(model) == 1 and t.n == 1
    with model.hooks(fwd_hooks=[("hook_embed", t)]):
        assert n_fwd(model) == 2
        model(x)
    assert n_fwd(model) == 1 and t.n == 3
    model.remove_all_hook_fns(including_permanent=True)
    assert n_fwd(model) == 0


def test_nested_contexts_and_failure_unwinding(model, x):
    t = Tally()

    def boom(z, hook):
        raise ValueError("fail")

    with model.hooks(fwd_hooks=[("hook_embed", t)]):
        assert model.context_level == 1
        model(x)
        with model.hooks(fwd_hooks=[("hook_embed", t)]):
            assert n_fwd(model) == 2 and model.context_level == 2
            model(x)
        assert n_fwd(model) == 1 and t.n == 3
        with pytest.raises(ValueError):
            with model.hooks(fwd_hooks=[("hook_embed", boom)]):
                model(x)
        assert n_fwd(model) == 1 and model.context_level == 1      # inner level removed, outer intact
        model.run_with_cache(x)                                      # its own level, leaves ours alone
        assert n_fwd(model) == 1
        model.reset_hooks()                                          # level=None: removes everything non-permanent
        assert n_fwd(model) == 0
    assert model.context_level == 0


def test_flag_gated_hooks_and_their_cached_shapes(model, x):
    ident = lambda z, hook: z  # noqa: E731
    for name, setter in [("blocks.0.attn.hook_result", model.set_use_attn_result),
                         ("blocks.0.hook_q_input", model.set_use_split_qkv_input),
                         ("blocks.0.hook_mlp_in", model.set_use_hook_mlp_in),
                         ("blocks.0.hook_attn_in", model.set_use_attn_in)]:
        model.reset_hooks()
        setter(False)
        with pytest.raises(AssertionError):
            model.add_hook(name, ident)
        setter(True)
        model.add_hook(name, ident)
        setter(False)
    model.reset_hooks()
    d, H = model.cfg.d_model, model.cfg.n_heads
    for name, setter, shape in [("blocks.0.hook_q_input", model.set_use_split_qkv_input, (B, T, H, d)),
                                ("blocks.0.hook_attn_in", model.set_use_attn_in, (B, T, H, d)),
                                ("blocks.0.hook_mlp_in", model.set_use_hook_mlp_in, (B, T, d)),
                                ("blocks.0.attn.hook_result", model.set_use_attn_result, (B, T, H, d))]:
        setter(True)
        _, cache = model.run_with_cache(x, names_filter=lambda n: n == name)
        assert list(cache.keys()) == [name] and tuple(cache[name].shape) == shape
        setter(False)
    # per-head result sums to the fused O-projection
    model.set_use_attn_result(True)
    with torch.no_grad():
        _, c = model.run_with_cache(x)
        assert torch.allclose(c["blocks.0.attn.hook_result"].sum(2) + model.blocks[0].attn.b_O, c["blocks.0.hook_attn_out"], atol=1e-5)
    model.set_use_attn_result(False)


@pytest.mark.parametrize("zero_pos", [0, 1])
@pytest.mark.parametrize("prepend", [True, False])
def test_prepend_controls_execution_order(model, x, zero_pos, prepend):
    """Two replacing hooks on the last residual: zeros and noise.  The output equals what zeros alone give
    exactly when the zero hook runs LAST."""
    def zero(z, hook):
        return torch.zeros_like(z)

    def noise(z, hook):
        return torch.randn_like(z)

    with torch.no_grad():
        with model.hooks(fwd_hooks=[("blocks.1.hook_resid_post", zero)]):
            want = model(x[:1])
        model.reset_hooks()
        for i in range(2):
            model.add_hook("blocks.1.hook_resid_post", zero if i == zero_pos else noise, prepend=prepend)
        got = model(x[:1])
        model.reset_hooks()
    zero_runs_last = (zero_pos == 1) != prepend
    assert torch.allclose(got, want, atol=1e-6) == zero_runs_last


def test_cache_key_order_and_observe_only_points(model, x):
    with torch.no_grad():
        _, cache = model.run_with_cache(x)
    keys = list(cache.keys())
    assert keys[:4] == ["hook_embed", "hook_pos_embed", "hook_full_embed", "blocks.0.hook_resid_pre"]   # no ln_pre here
    per_block = ["hook_resid_pre", "ln1.hook_scale", "ln1.hook_normalized", "attn.hook_q", "attn.hook_k", "attn.hook_v",
                 "attn.hook_attn_scores", "attn.hook_pattern", "attn.hook_z", "hook_attn_out", "hook_resid_mid",
                 "ln2.hook_scale", "ln2.hook_normalized", "mlp.hook_pre", "mlp.hook_post", "hook_mlp_out", "hook_resid_post"]
    assert keys[3:3 + 17] == ["blocks.0." + s for s in per_block]
    assert keys[-4:] == ["ln_final.hook_scale", "ln_final.hook_normalized", "hook_ln_final", "hook_post_head_pre_normalize"]
    assert len(keys) == 3 + 17 * 2 + 4 and len(model.hook_dict) == 3 + 23 * 2 + 4       # 286-style inventory: 23 per block
    assert cache["hook_embed"].shape == (B, T - 1, 16) and cache["hook_pos_embed"].stride(0) == 0
    # hook_full_embed / hook_ln_final / hook_post_head_pre_normalize are observe-only: a replacing hook there
    # must NOT change the output; the same hook on a data-flow point must
    with torch.no_grad():
        base = model(x)
        for name in ("hook_full_embed", "hook_ln_final", "hook_post_head_pre_normalize"):
            out = model.run_with_hooks(x, fwd_hooks=[(name, lambda z, hook: torch.zeros_like(z))])
            assert torch.equal(out, base), name
        out = model.run_with_hooks(x, fwd_hooks=[("hook_embed", lambda z, hook: torch.zeros_like(z))])
        assert not torch.equal(out, base)
        # user hooks run BEFORE the caching hooks: the cache holds the post-hook value
        _, c = model.run_with_cache(x, fwd_hooks=[("blocks.0.hook_attn_out", lambda z, hook: z * 0 + 7.0)])
        assert float(c["blocks.0.hook_attn_out"].min()) == 7.0
        # stop_at_layer semantics incl. negative index; incl_bwd caches *_grad entries
        r, c = model.run_with_cache(x, stop_at_layer=-1)
        assert r.shape == (B, T, 16) and "blocks.1.hook_resid_pre" not in c and "ln_final.hook_scale" not in c
    model.zero_grad()
    out, c = model.run_with_cache(x[:1].requires_grad_(False), incl_bwd=False)
    assert not any(k.endswith("_grad") for k in c.keys())
    assert model.hook_dict["blocks.1.mlp.hook_post"].layer() == 1


def test_boundary_hook_classification_for_the_split_native_plan(model, x):
    """Which registered hooks the HIP plan can serve by splitting (HookedViT._boundary_hooks): forward hooks on the
    residual-stream points of a block -- hook_resid_pre (block >= 1), hook_attn_out, hook_resid_mid, hook_mlp_out,
    hook_resid_post -- and on every point inside it: ln1 / ln2 .hook_scale / .hook_normalized, attn.hook_q / hook_k / hook_v,
    attn.hook_attn_scores, attn.hook_pattern, attn.hook_z, mlp.hook_pre, mlp.hook_post, keyed by position (ten per block: entry |
    ln1 | q, k, v | scores | pattern | z | after the attention half | ln2 | mlp pre | mlp post); hooks on the embedding stage (incl.
    block 0's resid_pre) and on the final stage are classified under two special keys -- those stages then run on the model's own
    modules and the blocks stay on the plan; anything else (the flag-gated points, backward hooks) keeps the PyTorch path.  On CPU the
    call itself always runs in PyTorch."""
    ident = lambda t, hook: t  # noqa: E731
    assert model._boundary_hooks() == {}
    with model.hooks(fwd_hooks=[("blocks.0.hook_resid_post", ident), ("blocks.1.hook_resid_pre", ident),
                                ("blocks.1.hook_resid_post", ident), ("blocks.0.hook_mlp_out", ident),
                                ("blocks.1.hook_attn_out", ident), ("blocks.0.hook_resid_mid", ident),
                                ("blocks.0.attn.hook_z", ident), ("blocks.1.attn.hook_q", ident), ("blocks.1.attn.hook_v", ident),
                                ("blocks.1.mlp.hook_post", ident), ("blocks.0.attn.hook_pattern", ident),
                                ("blocks.1.attn.hook_attn_scores", ident), ("blocks.0.ln2.hook_scale", ident),
                                ("blocks.1.mlp.hook_pre", ident), ("blocks.1.ln1.hook_normalized", ident),
                                ("blocks.1.ln1.hook_scale", ident)]):
        bh = model._boundary_hooks()
        assert sorted(bh) == [4, 5, 6, 7, 10, 11, 12, 13, 16, 18, 19, 20]
        assert sorted(bh[4]) == ["pattern"] and sorted(bh[5]) == ["z"] and sorted(bh[6]) == ["mid"] and sorted(bh[7]) == ["ln2s"]
        assert sorted(bh[10]) == ["mlp", "post", "pre"] and sorted(bh[11]) == ["ln1n", "ln1s"] and sorted(bh[12]) == ["q", "v"]
        assert sorted(bh[13]) == ["scores"] and sorted(bh[16]) == ["attn"] and sorted(bh[18]) == ["mlppre"]
        assert sorted(bh[19]) == ["mlppost"] and sorted(bh[20]) == ["post"]
        assert bh[10]["post"] is model.hook_dict["blocks.0.hook_resid_post"]
        out = model(x)                                   # CPU input: PyTorch path, result defined by the hooks
        assert out.shape[0] == B and not model.last_run_native
    assert model._boundary_hooks() == {}
    for name, key in (("blocks.0.hook_resid_pre", model._EMBED_POS), ("hook_embed", model._EMBED_POS), ("hook_pos_embed", model._EMBED_POS),
                      ("ln_final.hook_normalized", model._FINAL_POS), ("hook_ln_final", model._FINAL_POS)):
        with model.hooks(fwd_hooks=[(name, ident)]):
            assert list(model._boundary_hooks()) == [key], name
    for name, flag in (("blocks.0.hook_mlp_in", "use_hook_mlp_in"), ("blocks.1.attn.hook_result", "use_attn_result"),
                       ("blocks.0.hook_q_input", "use_split_qkv_input")):
        with model.hooks(fwd_hooks=[(name, ident)]):
            assert model._boundary_hooks() == {}, name                  # flag off: the point is never called, the hook cannot fire
            setattr(model.cfg, flag, True)                              # flag on (round 6): served ON the plan -- hook_mlp_in / hook_result as kinds
            try:                                                        # of their own at the block's positions 7 / 6, the per-head inputs by _head_glue
                l, hp = int(name.split(".")[1]), model.hook_dict[name]
                want = {"use_hook_mlp_in": {model._NPOS * l + 7: {"mlpin": hp}}, "use_attn_result": {model._NPOS * l + 6: {"result": hp}},
                        "use_split_qkv_input": {model._HEAD_POS: {l: True}}}[flag]
                assert model._boundary_hooks() == want, name
            finally:
                setattr(model.cfg, flag, False)
    with model.hooks(bwd_hooks=[("blocks.0.hook_resid_post", ident)]):
        assert model._boundary_hooks() is None


def test_spliced_modules_disable_the_native_plan(model, x):
    """HookedSAEViT.add_sae-style surgery (a module set in place of a HookPoint, base_vit.py:850-873) changes what the
    forward computes: the dispatcher must notice and keep such a model on the PyTorch path."""
    import torch.nn as nn
    assert "module tree" not in model._native_reason((x,), {})          # (CPU input: refused for the device only)

    class Doubler(nn.Module):
        def forward(self, t):
            return 2.0 * t

    base = model(x)
    model.blocks[0].hook_resid_post = Doubler()
    model.setup()
    assert not torch.allclose(model(x), base)
    assert "module tree was modified" in model._native_reason((x,), {})
