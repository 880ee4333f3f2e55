"""The split native plan of HookedViT._run_with_cache_native (ten positions per block, hooks applied between segments, forced taps,
carried residual stream and activations) on CPU: the HIP backend is replaced by a stand-in that computes a SEGMENT
(first_block / entry_stage .. n_blocks / exit_stage, requested taps) with plain torch arithmetic on the model's own parameters and
never touches a HookPoint -- so everything the orchestration gets wrong (a hook applied twice or not at all, a wrong resume
tensor, a tap taken from the wrong segment, a cache entry not reflecting the hook) shows against the model's ordinary PyTorch hook
path.  The kernels themselves are held to the same comparison on the GPU (tests/test_native_vit_gpu.py)."""
import pytest
import torch
import torch.nn.functional as F

from vit_prisma_amd import HookedViT, HookedViTConfig

B, S, P = 3, 32, 8


def ln(mod, x, taps, prefix, want):
    x = x - x.mean(-1, keepdim=True)
    scale = (x.pow(2).mean(-1, keepdim=True) + mod.eps).sqrt()
    out = x / scale * mod.w + mod.b
    if prefix + ".hook_scale" in want:
        taps[prefix + ".hook_scale"] = scale
    if prefix + ".hook_normalized" in want:
        taps[prefix + ".hook_normalized"] = out
    return out


class SegmentBackend:
    """What NativeViT.forward does, in torch, without HookPoints."""

    def __init__(self):
        self.calls = []

    def forward(self, model, images, names, n_blocks, run_head, cache_device=None, remove_batch_dim=False, first_block=0,
                resid_in=None, entry_mid=False, exit_mid=False, entry_stage=0, exit_stage=0, act_in=()):
        es, xs = (6 if entry_mid else entry_stage), (6 if exit_mid else exit_stage)
        self.calls.append((first_block, es, n_blocks, xs, bool(run_head)))
        want, taps, cfg, m = set(names), {}, model.cfg, model

        def tap(name, t):
            if name in want:
                taps[name] = t
            return t

        if resid_in is None:
            assert first_block == 0 and es == 0
            e = tap("hook_embed", m.embed(images))
            e = torch.cat((m.cls_token.expand(images.shape[0], -1, -1), e), dim=1)
            tap("hook_pos_embed", m.pos_embed(images))
            resid = tap("hook_full_embed", e + m.pos_embed(images))
            if cfg.layer_norm_pre:
                resid = tap("hook_ln_pre", ln(m.ln_pre, resid, taps, "ln_pre", want))
        else:
            assert images is None
            resid = resid_in
        out = None
        last = n_blocks + 1 if xs else n_blocks
        for l in range(first_block, last):
            blk, pre = m.blocks[l], f"blocks.{l}."
            e_ = es if l == first_block else 0
            x_ = xs if (xs and l == n_blocks) else 99
            a = blk.attn
            if e_ < 6:
                if e_ == 0:
                    tap(pre + "hook_resid_pre", resid)
                    h = ln(blk.ln1, resid, taps, pre + "ln1", want)
                elif e_ == 1:
                    (h,) = act_in
                    assert h.dtype == torch.float32
                if x_ == 1:
                    out = h
                    break
                if e_ < 2:
                    q = tap(pre + "attn.hook_q", torch.einsum("btd,hde->bthe", h, a.W_Q) + a.b_Q)
                    k = tap(pre + "attn.hook_k", torch.einsum("btd,hde->bthe", h, a.W_K) + a.b_K)
                    v = tap(pre + "attn.hook_v", torch.einsum("btd,hde->bthe", h, a.W_V) + a.b_V)
                elif e_ == 2:
                    q, k, v = act_in
                if x_ == 2:
                    out = q
                    break
                if e_ < 3:
                    scores = tap(pre + "attn.hook_attn_scores", torch.einsum("bqhe,bkhe->bhqk", q, k) / a.attn_scale)
                elif e_ == 3:
                    scores, v = act_in
                if x_ == 3:
                    out = scores
                    break
                if e_ < 4:
                    pat = F.softmax(scores, dim=-1)
                    pat = tap(pre + "attn.hook_pattern", torch.where(torch.isnan(pat), torch.zeros_like(pat), pat))
                elif e_ == 4:
                    pat, v = act_in
                if x_ == 4:
                    out = pat
                    break
                if e_ < 5:
                    z = tap(pre + "attn.hook_z", torch.einsum("bkhe,bhqk->bqhe", v, pat))
                else:
                    (z,) = act_in
                if x_ == 5:
                    out = z
                    break
                attn_out = tap(pre + "hook_attn_out", torch.einsum("bqhe,hed->bqd", z, a.W_O) + a.b_O)
                mid = tap(pre + "hook_resid_mid", resid + attn_out)
            else:
                mid = resid
            if x_ == 6:
                out = resid = mid
                break
            mlp = blk.mlp
            if e_ < 9:
                if e_ < 7:
                    h2 = ln(blk.ln2, mid, taps, pre + "ln2", want)
                elif e_ == 7:
                    (h2,) = act_in
                    assert h2.dtype == torch.float32
                if x_ == 7:
                    out = h2
                    break
                if e_ < 8:
                    pre_act = tap(pre + "mlp.hook_pre", h2 @ mlp.W_in + mlp.b_in)
                else:
                    (pre_act,) = act_in
                if x_ == 8:
                    out = pre_act
                    break
                post = tap(pre + "mlp.hook_post", mlp.act_fn(pre_act))
            else:
                (post,) = act_in
            if x_ == 9:
                out = post
                break
            mlp_out = tap(pre + "hook_mlp_out", post @ mlp.W_out + mlp.b_out)
            resid = tap(pre + "hook_resid_post", mid + mlp_out)
        if run_head:
            x = ln(m.ln_final, resid, taps, "ln_final", want)
            tap("hook_ln_final", x)
            x = m.head(x[:, 0])
            tap("hook_post_head_pre_normalize", x)
            out = F.normalize(x, dim=-1) if cfg.normalize_output else x
        elif out is None:
            out = resid
        assert want <= set(taps), sorted(want - set(taps))
        return out, taps


def _make_model(**extra):
    torch.manual_seed(0)
    cfg = HookedViTConfig(n_layers=3, d_model=16, d_head=8, d_mlp=32, n_heads=2, patch_size=P, image_size=S, n_classes=5,
                          return_type="logits", **extra)
    m = HookedViT(cfg).eval()
    with torch.no_grad():                                 # (no parameter at its trivial initial value)
        for p_ in m.parameters():
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.1)
    backend = SegmentBackend()
    m._get_native = lambda device: backend
    m._backend = backend
    return m


@pytest.fixture()
def model():
    return _make_model()


@pytest.fixture()
def model_ln_pre():
    return _make_model(layer_norm_pre=True, normalize_output=True)


def scale_shift(t, hook):
    return t * 0.5 + 1.0


def zero_cls(t, hook):                    # in place, returns None
    t[:, 0] = 0.0


def half(t, hook):
    return t * 0.5


def kill_head_1(t, hook):
    t[:, :, 1] = 0.0


def kill_neurons(t, hook):
    t[..., ::3] = 0.0


def swap_heads(t, hook):
    return t.flip(2)


def no_cls_attention(t, hook):
    t = t.clone()
    t[..., 0] = 0.0
    return t / t.sum(-1, keepdim=True).clamp_min(1e-6)


def mask_scores(t, hook):
    t[:, 0, :, -1] = float("-inf")


def nan_row(t, hook):
    t[:, 1, 2, 3] = float("nan")


def freeze_scale(t, hook):                # "frozen LayerNorm": the scale replaced by a constant
    return torch.full_like(t, 1.25)


def shift_pre(t, hook):
    return t - 0.25


NL = 3
CASES = [
    [("blocks.0.hook_resid_post", scale_shift)],
    [(f"blocks.{NL - 1}.hook_resid_post", zero_cls)],
    [("blocks.1.hook_resid_pre", scale_shift), ("blocks.0.hook_resid_post", zero_cls)],
    [(lambda n: n.endswith("hook_resid_post"), scale_shift)],
    [("blocks.0.hook_mlp_out", scale_shift)],
    [("blocks.1.hook_resid_mid", zero_cls), ("blocks.1.hook_attn_out", scale_shift)],
    [(lambda n: n.endswith(("hook_attn_out", "hook_resid_mid", "hook_mlp_out", "hook_resid_post")) or n == "blocks.1.hook_resid_pre", scale_shift)],
    [("blocks.0.attn.hook_z", kill_head_1)],
    [("blocks.1.mlp.hook_post", kill_neurons)],
    [("blocks.0.attn.hook_q", half), ("blocks.0.attn.hook_k", kill_head_1), ("blocks.0.attn.hook_v", swap_heads)],
    [("blocks.0.attn.hook_z", half), ("blocks.1.mlp.hook_post", kill_neurons), ("blocks.1.hook_attn_out", half)],
    [(lambda n: n.endswith(("attn.hook_q", "attn.hook_z", "hook_resid_mid", "mlp.hook_post", "hook_resid_post")), half)],
    [("blocks.0.attn.hook_pattern", no_cls_attention)],
    [(f"blocks.{NL - 1}.attn.hook_attn_scores", mask_scores)],
    [("blocks.0.attn.hook_attn_scores", nan_row), ("blocks.0.attn.hook_pattern", half)],
    [("blocks.1.attn.hook_q", half), ("blocks.1.attn.hook_attn_scores", mask_scores), ("blocks.1.attn.hook_pattern", no_cls_attention),
     ("blocks.1.attn.hook_z", kill_head_1)],
    [("blocks.0.attn.hook_v", swap_heads), ("blocks.0.attn.hook_pattern", half), ("blocks.0.hook_resid_mid", half)],
    [(lambda n: n.endswith(("attn.hook_attn_scores", "attn.hook_pattern", "attn.hook_z", "mlp.hook_post", "hook_mlp_out")), half)],
    [("blocks.0.ln1.hook_scale", freeze_scale)],
    [("blocks.1.ln2.hook_normalized", half), ("blocks.1.mlp.hook_pre", shift_pre)],
    [("blocks.0.ln1.hook_scale", freeze_scale), ("blocks.0.ln1.hook_normalized", scale_shift), ("blocks.0.attn.hook_q", half),
     ("blocks.0.ln2.hook_scale", half), ("blocks.0.mlp.hook_pre", kill_neurons), ("blocks.0.mlp.hook_post", half)],
    [(lambda n: n.endswith(("ln1.hook_scale", "ln2.hook_normalized", "mlp.hook_pre")), half)],
    [(lambda n: n.startswith("blocks.") and n.split(".", 2)[2] in ("attn.hook_q", "attn.hook_k", "attn.hook_v", "attn.hook_attn_scores",
                                                                   "attn.hook_pattern", "attn.hook_z", "hook_attn_out", "hook_resid_mid",
                                                                   "mlp.hook_post", "hook_mlp_out", "hook_resid_post", "ln1.hook_scale",
                                                                   "ln1.hook_normalized", "ln2.hook_scale", "ln2.hook_normalized",
                                                                   "mlp.hook_pre"), half)],
]
FORMS = [{}, {"names_filter": lambda n: "resid" in n or n.endswith(("hook_z", "hook_pattern", "mlp.hook_post", "ln1.hook_scale", "ln2.hook_normalized"))},
         {"names_filter": lambda n: n.endswith(("hook_attn_scores", "hook_v", "hook_attn_out"))}, {"stop_at_layer": NL - 1},
         {"names_filter": [], "stop_at_layer": 1}]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_split_plan_equals_the_hook_path(model, case):
    hooks = CASES[case]
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1))
    real_reason = model._native_reason
    with torch.no_grad():
        for kw in FORMS:
            model.use_native(False)                                        # the ordinary PyTorch hook path
            w_out, w_cache = model.run_with_cache(x.clone(), fwd_hooks=hooks, **kw)
            assert not model.last_run_native
            # the split plan on the stand-in backend: same dispatch, minus the "input is on a GPU" condition
            model.use_native(True)
            model._native_reason = lambda a, k: None if model._boundary_hooks() is not None else "a hook the plan cannot be split at"
            model._backend.calls.clear()
            try:
                g_out, g_cache = model.run_with_cache(x.clone(), fwd_hooks=hooks, **kw)
            finally:
                model._native_reason = real_reason
            assert model.last_run_native and len(model._backend.calls) >= 1
            assert list(g_cache.keys()) == list(w_cache.keys())
            assert torch.allclose(g_out, w_out, atol=1e-5, equal_nan=True), kw
            for k_ in w_cache.keys():
                a, b = g_cache[k_], w_cache[k_]
                assert a.shape == b.shape and torch.allclose(a, b, atol=1e-5, equal_nan=True), (k_, kw)
            assert all(len(hp.fwd_hooks) == 0 for hp in model.hook_dict.values())


# ---- hooks on the embedding stage and on the final stage: those stages run on the model's own PyTorch modules (their HookPoints fire as
# in the reference, base_vit.py:169-185, 192-217), every block stays on the plan -- resumed at block 0 from the residual stream the
# embedding stage left, stopped before ln_final
EDGE_CASES = [
    [("hook_embed", scale_shift)],
    [("hook_pos_embed", half), ("blocks.0.hook_resid_pre", zero_cls)],
    [("hook_full_embed", zero_cls)],                                        # observe-only, but an in-place edit reaches the stream
    [("blocks.0.hook_resid_pre", scale_shift)],
    [("ln_final.hook_normalized", half)],
    [("ln_final.hook_scale", freeze_scale)],
    [("hook_ln_final", scale_shift)],                                       # observe-only: the return value is dropped
    [("hook_post_head_pre_normalize", half)],
    [("hook_embed", half), ("blocks.1.attn.hook_z", kill_head_1), ("ln_final.hook_normalized", half)],
    [("hook_embed", zero_cls), ("blocks.0.ln1.hook_scale", freeze_scale), ("blocks.2.hook_resid_post", half), ("ln_final.hook_scale", half)],
    [(lambda n: not n.startswith("blocks.") and n != "hook_pos_embed", half)],
]
EDGE_CASES_LN_PRE = [
    [("ln_pre.hook_scale", freeze_scale)],
    [("ln_pre.hook_normalized", half), ("hook_ln_pre", scale_shift)],
    [("hook_ln_pre", zero_cls), ("blocks.0.attn.hook_q", half), ("hook_post_head_pre_normalize", half)],
]


def _edge(model, hooks):
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1))
    real_reason = model._native_reason
    with torch.no_grad():
        for kw in FORMS + [{"stop_at_layer": 0}, {"names_filter": lambda n: not n.startswith("blocks.")}]:
            model.use_native(False)
            w_out, w_cache = model.run_with_cache(x.clone(), fwd_hooks=hooks, **kw)
            model.use_native(True)
            model._native_reason = lambda a, k: None if model._boundary_hooks() is not None else "a hook the plan cannot be split at"
            model._backend.calls.clear()
            try:
                g_out, g_cache = model.run_with_cache(x.clone(), fwd_hooks=hooks, **kw)
            finally:
                model._native_reason = real_reason
            assert model.last_run_native
            assert list(g_cache.keys()) == list(w_cache.keys()), kw
            assert torch.allclose(g_out, w_out, atol=1e-5), kw
            for k_ in w_cache.keys():
                a, b = g_cache[k_], w_cache[k_]
                assert a.shape == b.shape and torch.allclose(a, b, atol=1e-5), (k_, kw)
            assert all(len(hp.fwd_hooks) == 0 and len(hp._forward_hooks) == 0 for hp in model.hook_dict.values())


@pytest.mark.parametrize("case", range(len(EDGE_CASES)))
def test_embedding_and_final_stage_hooks_keep_the_blocks_on_the_plan(model, case):
    _edge(model, EDGE_CASES[case])


@pytest.mark.parametrize("case", range(len(EDGE_CASES_LN_PRE) + len(EDGE_CASES)))
def test_embedding_and_final_stage_hooks_with_ln_pre(model_ln_pre, case):
    _edge(model_ln_pre, (EDGE_CASES_LN_PRE + EDGE_CASES)[case])


# ---- flag-gated HookPoints (use_attn_in / use_split_qkv_input / use_attn_result / use_hook_mlp_in): the plan runs as always, their cache
# entries are derived from what it tapped; a forward hook ON such a point (or on ln1 while the block inputs carry a head dimension)
# takes the PyTorch path
FLAG_SETS = [dict(use_attn_result=True, use_split_qkv_input=True, use_attn_in=True, use_hook_mlp_in=True),
             dict(use_attn_result=True, use_hook_mlp_in=True), dict(use_attn_in=True), dict(use_split_qkv_input=True)]
FLAG_HOOKS = [[], [("blocks.0.hook_resid_pre", scale_shift), ("blocks.1.attn.hook_z", kill_head_1), ("blocks.1.hook_resid_mid", half)],
              [("blocks.0.attn.hook_pattern", no_cls_attention), ("blocks.2.mlp.hook_post", kill_neurons), ("hook_embed", half)]]


@pytest.mark.parametrize("flags", range(len(FLAG_SETS)))
def test_flag_gated_points_are_derived_from_the_plan(flags):
    model = _make_model(**FLAG_SETS[flags])
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1))
    real_reason = model._native_reason
    forms = FORMS + [{"names_filter": lambda n: n.endswith(("hook_result", "hook_mlp_in", "hook_attn_in", "hook_k_input", "ln1.hook_scale"))},
                     {"remove_batch_dim": True}]
    with torch.no_grad():
        for hooks in FLAG_HOOKS:
            for kw in forms:
                xi = x[:1] if kw.get("remove_batch_dim") else x
                model.use_native(False)
                w_out, w_cache = model.run_with_cache(xi.clone(), fwd_hooks=hooks, **kw)
                model.use_native(True)
                model._native_reason = lambda a, k: None if model._boundary_hooks() is not None else "a hook the plan cannot be split at"
                try:
                    g_out, g_cache = model.run_with_cache(xi.clone(), fwd_hooks=hooks, **kw)
                finally:
                    model._native_reason = real_reason
                assert model.last_run_native
                assert list(g_cache.keys()) == list(w_cache.keys()), kw
                assert torch.allclose(g_out, w_out, atol=1e-5), kw
                for k_ in w_cache.keys():
                    a, b = g_cache[k_], w_cache[k_]
                    assert a.shape == b.shape and a.dtype == b.dtype and torch.allclose(a, b, atol=1e-5), (k_, kw)
    # a hook ON a flag-gated point (or on ln1 under per-head block inputs) changes that block's forward.  Round 6: the block stays on
    # the plan all the same -- attn.hook_result / hook_mlp_in are served at the block's positions 6 / 7, hooks on the per-head inputs
    # run only the block's head (inputs, ln1, q / k / v projections) on the module's code and enter the plan at PV_STAGE_QKV.  No block
    # may be sent to its PyTorch module here:
    def no_torch_block(*a, **k):
        raise AssertionError("a block hooked on a flag-gated point was sent to its PyTorch module")

    model._torch_block_stage = no_torch_block
    def head_edit(t, hook):
        t = t.clone()
        t[:, :, 1] = t[:, :, 1] * 0.5
        return t

    F_ = FLAG_SETS[flags]
    mixed = []
    if F_.get("use_attn_result"):
        mixed.append([("blocks.1.attn.hook_result", head_edit), ("blocks.0.hook_resid_post", half)])
    if F_.get("use_split_qkv_input"):
        mixed.append([("blocks.0.hook_v_input", head_edit), ("blocks.2.hook_q_input", half), ("blocks.1.attn.hook_z", kill_head_1)])
    if F_.get("use_attn_in"):
        mixed.append([("blocks.2.hook_attn_in", head_edit), ("hook_embed", half)])
        mixed.append([("blocks.0.hook_attn_in", head_edit), ("blocks.1.hook_attn_in", half), ("blocks.1.hook_resid_pre", scale_shift),
                      ("blocks.2.hook_resid_pre", half)])
    if F_.get("use_hook_mlp_in"):
        mixed.append([("blocks.1.hook_mlp_in", scale_shift), ("blocks.1.hook_mlp_out", half), ("blocks.2.hook_resid_pre", half),
                      ("ln_final.hook_normalized", half)])
    if F_.get("use_attn_in") or F_.get("use_split_qkv_input"):
        mixed.append([("blocks.1.ln1.hook_scale", freeze_scale), ("blocks.0.ln2.hook_normalized", half)])
    with torch.no_grad():
        for hooks in mixed:
            for kw in forms:
                xi = x[:1] if kw.get("remove_batch_dim") else x
                model.use_native(False)
                w_out, w_cache = model.run_with_cache(xi.clone(), fwd_hooks=hooks, **kw)
                model.use_native(True)
                model._native_reason = lambda a, k: None if model._boundary_hooks() is not None else "a hook the plan cannot be split at"
                model._backend.calls.clear()
                try:
                    g_out, g_cache = model.run_with_cache(xi.clone(), fwd_hooks=hooks, **kw)
                finally:
                    model._native_reason = real_reason
                assert model.last_run_native, hooks
                # every block's MLP output stage (9 -> the block's end) ran on the backend: no block left the plan
                n_run = model.cfg.n_layers if kw.get("stop_at_layer") is None else len(range(model.cfg.n_layers)[:kw["stop_at_layer"]])
                done = set()
                for fb, es, nb, xs, _ in model._backend.calls:
                    done.update(range(fb, nb))                     # (a segment that exits inside block nb does not finish it)
                assert done >= set(range(n_run)), (hooks, kw, model._backend.calls)
                assert list(g_cache.keys()) == list(w_cache.keys()), (hooks, kw)
                assert torch.allclose(g_out, w_out, atol=1e-5), (hooks, kw)
                for k_ in w_cache.keys():
                    a, b = g_cache[k_], w_cache[k_]
                    assert a.shape == b.shape and a.dtype == b.dtype and torch.allclose(a, b, atol=1e-5), (k_, hooks, kw)
                assert all(len(hp.fwd_hooks) == 0 and len(hp._forward_hooks) == 0 for hp in model.hook_dict.values())


# ---- modules spliced in place of HookPoints (HookedSAEViT.add_sae): served by the plan like a hook at that point; the spliced
# module's own HookPoints take the replaced point's place in the cache
class _ToySAE(torch.nn.Module):
    """The shape of a spliced SAE: own HookPoints, returns the tensor the block continues from."""

    def __init__(self, d, hook_point, seed):
        super().__init__()
        from vit_prisma_amd.hook_points import HookPoint
        import types
        g = torch.Generator().manual_seed(seed)
        self.W_enc = torch.nn.Parameter(torch.randn(d, 3 * d, generator=g) * 0.3)
        self.W_dec = torch.nn.Parameter(torch.randn(3 * d, d, generator=g) * 0.3)
        self.hook_sae_in, self.hook_hidden_post, self.hook_sae_out = HookPoint(), HookPoint(), HookPoint()
        self.cfg = types.SimpleNamespace(hook_point=hook_point, return_out_only=False)
        self.dtype = torch.float32

    def forward(self, x):
        f = self.hook_hidden_post(torch.relu(self.hook_sae_in(x) @ self.W_enc))
        return self.hook_sae_out(f @ self.W_dec)


def _make_sae_model(**flags):
    from vit_prisma_amd import HookedSAEViT
    torch.manual_seed(0)
    cfg = HookedViTConfig(n_layers=3, d_model=16, d_head=8, d_mlp=32, n_heads=2, patch_size=P, image_size=S, n_classes=5,
                          return_type="logits", **flags)
    m = HookedSAEViT(cfg).eval()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.1)
    backend = SegmentBackend()
    m._get_native = lambda device: backend
    m._backend = backend
    return m


SPLICES = [["blocks.0.hook_resid_post"], ["blocks.1.hook_mlp_out", "blocks.2.hook_resid_mid"], ["blocks.1.hook_resid_pre", "blocks.0.attn.hook_z"],
           ["blocks.2.hook_attn_out", "blocks.2.hook_resid_post"]]


@pytest.mark.parametrize("case", range(len(SPLICES)))
def test_spliced_modules_are_served_by_the_plan(case):
    model = _make_sae_model()
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1))
    d_of = lambda name: 8 if name.endswith("hook_z") else 16  # noqa: E731
    for i, name in enumerate(SPLICES[case]):
        model.add_sae(_ToySAE(d_of(name), name, seed=i))
    real_reason = model._native_reason
    user_hooks = [[], [("blocks.0.hook_resid_mid", half), (SPLICES[case][0] + ".hook_hidden_post", kill_neurons)]]
    with torch.no_grad():
        for hooks in user_hooks:
            for kw in FORMS + [{"names_filter": lambda n: n.endswith(("hook_sae_in", "hook_sae_out", "hook_resid_post"))}]:
                model.use_native(False)
                w_out, w_cache = model.run_with_cache(x.clone(), fwd_hooks=hooks, **kw)
                assert not model.last_run_native
                model.use_native(True)
                model._native_reason = lambda a, k: None if (model._tree_matches() and model._boundary_hooks() is not None) else "no"
                model._backend.calls.clear()
                try:
                    g_out, g_cache = model.run_with_cache(x.clone(), fwd_hooks=hooks, **kw)
                finally:
                    model._native_reason = real_reason
                assert model.last_run_native and len(model._backend.calls) >= 1
                assert list(g_cache.keys()) == list(w_cache.keys()), (kw, list(g_cache.keys()), list(w_cache.keys()))
                assert torch.allclose(g_out, w_out, atol=1e-5), kw
                for k_ in w_cache.keys():
                    assert g_cache[k_].shape == w_cache[k_].shape and torch.allclose(g_cache[k_], w_cache[k_], atol=1e-5), (k_, kw)
                assert all(len(hp.fwd_hooks) == 0 for hp in model.hook_dict.values())
    # a splice the plan cannot serve at a block's point (a LayerNorm point, block 0's entry): that block on its own module; on the
    # embedding / final stage: that stage on the model's own modules (as for a hook there), every block on the plan
    model.reset_saes()
    assert model._tree_matches() and model._boundary_hooks() == {}
    with torch.no_grad():
        for name, key in (("blocks.1.ln2.hook_normalized", model._TORCH_POS), ("blocks.0.hook_resid_pre", model._TORCH_POS),
                          ("hook_embed", model._EMBED_POS), ("hook_post_head_pre_normalize", model._FINAL_POS)):
            model.add_sae(_ToySAE(5 if name.startswith("hook_post") else 16, name, seed=5))
            assert model._tree_matches() and list(model._boundary_hooks()) == [key], name
            for kw in FORMS:
                model.use_native(False)
                w_out, w_cache = model.run_with_cache(x.clone(), **kw)
                model.use_native(True)
                model._native_reason = lambda a, k: None if (model._tree_matches() and model._boundary_hooks() is not None) else "no"
                try:
                    g_out, g_cache = model.run_with_cache(x.clone(), **kw)
                finally:
                    model._native_reason = real_reason
                assert model.last_run_native and list(g_cache.keys()) == list(w_cache.keys()), (name, kw)
                assert torch.allclose(g_out, w_out, atol=1e-5)
                for k_ in w_cache.keys():
                    assert torch.allclose(g_cache[k_], w_cache[k_], atol=1e-5), (k_, name, kw)
            model.reset_saes()
    # any other edit of the tree is still a reason to leave the plan
    model.blocks[1].hook_resid_post = torch.nn.Identity()
    assert not model._tree_matches()


# a module spliced in place of a HookPoint TOGETHER with flag-gated HookPoints (round-4 advisor finding: KeyError on the spliced
# name): the splice's own HookPoints stand in the cache where the replaced point stood, the flag-gated entries are derived as
# without a splice -- and a splice on the very tensor a flag-gated point is derived from (block input / z / resid_mid) sends that
# block to its own module, whose flag-gated points then see the module's output as in the reference
SPLICE_X_FLAG = [(dict(use_attn_result=True), ["blocks.1.hook_resid_post"]),
                 (dict(use_attn_in=True), ["blocks.1.hook_resid_post", "blocks.2.hook_attn_out"]),
                 (dict(use_split_qkv_input=True), ["blocks.0.hook_mlp_out"]),
                 (dict(use_hook_mlp_in=True), ["blocks.1.hook_resid_post"]),
                 (dict(use_hook_mlp_in=True), ["blocks.1.hook_resid_mid"]),                   # the source of hook_mlp_in itself
                 (dict(use_attn_in=True, use_split_qkv_input=True), ["blocks.1.hook_resid_pre"]),  # the source of hook_attn_in / q, k, v inputs
                 (dict(use_attn_result=True), ["blocks.2.attn.hook_z", "blocks.0.hook_resid_mid"]),  # the source of attn.hook_result
                 (dict(use_attn_result=True, use_split_qkv_input=True, use_attn_in=True, use_hook_mlp_in=True),
                  ["blocks.0.hook_resid_post", "blocks.2.hook_mlp_out"])]


@pytest.mark.parametrize("case", range(len(SPLICE_X_FLAG)))
def test_spliced_modules_together_with_flag_gated_points(case):
    flags, splices = SPLICE_X_FLAG[case]
    model = _make_sae_model(**flags)
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1))
    d_of = lambda name: 8 if name.endswith("hook_z") else 16  # noqa: E731
    for i, name in enumerate(splices):
        model.add_sae(_ToySAE(d_of(name), name, seed=i))
    real_reason = model._native_reason
    forms = FORMS + [{"names_filter": lambda n: n.endswith(("hook_sae_in", "hook_sae_out", "hook_result", "hook_mlp_in", "hook_attn_in",
                                                            "hook_k_input", "hook_resid_post"))}]
    with torch.no_grad():
        for hooks in ([], [("blocks.0.hook_attn_out", half)]):
            for kw in forms:
                model.use_native(False)
                w_out, w_cache = model.run_with_cache(x.clone(), fwd_hooks=hooks, **kw)
                model.use_native(True)
                model._native_reason = lambda a, k: None if (model._tree_matches() and model._boundary_hooks() is not None) else "no"
                try:
                    g_out, g_cache = model.run_with_cache(x.clone(), fwd_hooks=hooks, **kw)
                finally:
                    model._native_reason = real_reason
                assert model.last_run_native
                assert list(g_cache.keys()) == list(w_cache.keys()), (kw, list(g_cache.keys()), list(w_cache.keys()))
                assert torch.allclose(g_out, w_out, atol=1e-5), kw
                for k_ in w_cache.keys():
                    a, b = g_cache[k_], w_cache[k_]
                    assert a.shape == b.shape and a.dtype == b.dtype and torch.allclose(a, b, atol=1e-5), (k_, kw)
    # the derived hook_mlp_in owns its storage (an in-place edit must not reach hook_resid_mid)
    if flags.get("use_hook_mlp_in") and "blocks.1.hook_resid_mid" not in splices:
        model.use_native(True)
        model._native_reason = lambda a, k: None
        try:
            with torch.no_grad():
                _, c = model.run_with_cache(x.clone())
        finally:
            model._native_reason = real_reason
        assert c["blocks.0.hook_mlp_in"].data_ptr() != c["blocks.0.hook_resid_mid"].data_ptr()
