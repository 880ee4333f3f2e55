"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of ViT-Prisma's
``HookedViT.run_with_cache`` forward in numpy.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``vit_prisma_amd``) never does; it fails loudly when the HIP
library is missing.

Parity pin: this restatement is checked against golden fixtures produced by executing the
*reference itself* in the build container (``tests/golden/gen_golden.py`` imports
/root/reference/src/vit_prisma with the stub recipe of SURVEY.md section 4 and dumps outputs);
see ``tests/test_oracle_vs_golden.py``.  The reference's own tests hold no numeric golden vector
for this path, so the executed reference is the only pin.

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/vit_prisma/).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple, Union

import numpy as np
from scipy.special import erf as _erf

Array = np.ndarray


# --------------------------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------------------------

def layer_norm(x: Array, w: Optional[Array], b: Optional[Array], eps: float) -> Tuple[Array, Array]:
    """models/layers/layer_norm.py:75-93 -- centre, scale = sqrt(mean(x^2) + eps) (population
    variance), x / scale * w + b.  Returns (normalized, scale[..., 1])."""
    x = x - x.mean(axis=-1, keepdims=True)
    scale = np.sqrt((x * x).mean(axis=-1, keepdims=True) + x.dtype.type(eps))
    y = x / scale
    if w is not None:
        y = y * w + b
    return y, scale


def gelu_erf(x: Array) -> Array:
    """models/layers/mlp.py:43-44 -- ``F.gelu`` default = exact erf form."""
    dt = x.dtype.type
    return (dt(0.5) * x * (dt(1.0) + _erf(x * dt(0.7071067811865476)))).astype(x.dtype)


def quick_gelu(x: Array) -> Array:
    """models/activation_fns.py:19 -- x * sigmoid(1.702 x)."""
    dt = x.dtype.type
    return (x / (dt(1.0) + np.exp(-dt(1.702) * x))).astype(x.dtype)


ACTIVATIONS: Dict[str, Callable[[Array], Array]] = {
    "gelu": gelu_erf,
    "quick_gelu": quick_gelu,
    "relu": lambda x: np.maximum(x, x.dtype.type(0)),
}


def softmax_lastdim(s: Array) -> Array:
    """models/layers/attention.py:148-149 -- softmax over keys, NaN -> 0."""
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    p = e / e.sum(axis=-1, keepdims=True)
    return np.where(np.isnan(p), p.dtype.type(0), p)


def patch_embed(images: Array, weight: Array, bias: Array, patch: int) -> Array:
    """models/layers/patch_embedding.py:26-32 -- Conv2d(k=p, stride=p) then flatten(2).transpose(1,2)
    == GEMM [B*P, C*p*p] x [C*p*p, d] with K ordered (c, i, j) and patches ordered (py, px)."""
    B, C, Hh, Ww = images.shape
    gy, gx = Hh // patch, Ww // patch
    cols = images[:, :, : gy * patch, : gx * patch].reshape(B, C, gy, patch, gx, patch)
    cols = cols.transpose(0, 2, 4, 1, 3, 5).reshape(B, gy * gx, C * patch * patch)
    return cols @ weight.reshape(weight.shape[0], -1).T + bias


# --------------------------------------------------------------------------------------------
# the forward with every HookPoint recorded in firing order
# --------------------------------------------------------------------------------------------

def hook_names_in_order(cfg: dict, stop_at_layer: Optional[int] = None) -> List[str]:
    """Firing order = dict insertion order of the reference cache;
    follows models/base_vit.py:152-217, models/layers/transformer_block.py:80-138,
    models/layers/attention.py:126-184, models/layers/mlp.py:65-80."""
    names = ["hook_embed", "hook_pos_embed", "hook_full_embed"]
    if cfg.get("layer_norm_pre", False):
        names += ["ln_pre.hook_scale", "ln_pre.hook_normalized", "hook_ln_pre"]
    L = cfg["n_layers"]
    layers = list(range(L))[:stop_at_layer]
    for l in layers:
        p = f"blocks.{l}."
        names += [p + "hook_resid_pre", p + "ln1.hook_scale", p + "ln1.hook_normalized",
                  p + "attn.hook_q", p + "attn.hook_k", p + "attn.hook_v",
                  p + "attn.hook_attn_scores", p + "attn.hook_pattern", p + "attn.hook_z",
                  p + "hook_attn_out", p + "hook_resid_mid", p + "ln2.hook_scale",
                  p + "ln2.hook_normalized", p + "mlp.hook_pre", p + "mlp.hook_post",
                  p + "hook_mlp_out", p + "hook_resid_post"]
    if stop_at_layer is None:
        names += ["ln_final.hook_scale", "ln_final.hook_normalized", "hook_ln_final",
                  "hook_post_head_pre_normalize"]
    return names


def vit_forward(
    sd: Dict[str, Array],
    cfg: dict,
    images: Array,
    stop_at_layer: Optional[int] = None,
    dtype=np.float32,
    names_filter: Optional[Union[str, List[str], Callable[[str], bool]]] = None,
) -> Tuple[Array, "OrderedDict[str, Array]"]:
    """HookedViT.forward (models/base_vit.py:152-217) with the caching hooks of
    prisma_tools/hooked_root_module.py:289-332 applied: returns (model_out, cache)."""
    if names_filter is None:
        keep = lambda n: True  # noqa: E731
    elif isinstance(names_filter, str):
        keep = lambda n: n == names_filter  # noqa: E731
    elif isinstance(names_filter, list):
        keep = lambda n: n in names_filter  # noqa: E731
    else:
        keep = names_filter

    cache: "OrderedDict[str, Array]" = OrderedDict()

    def tap(name: str, x: Array) -> Array:
        if keep(name):
            cache[name] = x
        return x

    P = {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}
    x = np.asarray(images, dtype=dtype)
    B = x.shape[0]
    d, H, dh = cfg["d_model"], cfg["n_heads"], cfg["d_head"]
    eps = cfg["eps"]
    act = ACTIVATIONS[cfg.get("activation_name", "gelu")]
    # attention.py:96-99: attn_scale = sqrt(d_head), applied as a division in the activation dtype
    attn_scale = dtype(np.sqrt(dh)) if cfg.get("use_attn_scale", True) else dtype(1.0)

    # base_vit.py:169 ; patch_embedding.py:29
    embed = tap("hook_embed", patch_embed(x, P["embed.proj.weight"], P["embed.proj.bias"], cfg["patch_size"]))
    if cfg.get("use_cls_token", True):
        # base_vit.py:171-175
        cls = np.broadcast_to(P["cls_token"], (B, 1, d))
        embed = np.concatenate([cls, embed], axis=1)
    T = embed.shape[1]
    # base_vit.py:177 ; position_embedding.py:32-38 (stride-0 broadcast of W_pos)
    pos = tap("hook_pos_embed", np.broadcast_to(P["pos_embed.W_pos"], (B, T, d)))
    resid = embed + pos                       # base_vit.py:179
    tap("hook_full_embed", resid)             # base_vit.py:181 (observe-only)
    if cfg.get("layer_norm_pre", False):      # base_vit.py:183-185
        y, sc = layer_norm(resid, P["ln_pre.w"], P["ln_pre.b"], eps)
        tap("ln_pre.hook_scale", sc)
        resid = tap("ln_pre.hook_normalized", y)
        resid = tap("hook_ln_pre", resid)

    for l in list(range(cfg["n_layers"]))[:stop_at_layer]:   # base_vit.py:187-188
        p = f"blocks.{l}."
        resid_pre = tap(p + "hook_resid_pre", resid)          # transformer_block.py:86
        y, sc = layer_norm(resid_pre, P[p + "ln1.w"], P[p + "ln1.b"], eps)   # :106-109 (x3, identical)
        tap(p + "ln1.hook_scale", sc)
        y = tap(p + "ln1.hook_normalized", y)
        # attention.py:186-244: x[b,t,:] @ W[h] + b[h]  -> [B,T,H,dh]; W[h] stacked head-major into one
        # [d, H*dh] matrix so the projection is a single BLAS GEMM (same arithmetic, same result layout)
        def proj(W, b):
            Wm = np.ascontiguousarray(W.transpose(1, 0, 2).reshape(d, H * dh))
            return (y.reshape(B * T, d) @ Wm).reshape(B, T, H, dh) + b
        q = tap(p + "attn.hook_q", proj(P[p + "attn.W_Q"], P[p + "attn.b_Q"]))
        k = tap(p + "attn.hook_k", proj(P[p + "attn.W_K"], P[p + "attn.b_K"]))
        v = tap(p + "attn.hook_v", proj(P[p + "attn.W_V"], P[p + "attn.b_V"]))
        # attention.py:246-265 (vision path: no mask): scores[b,h,q,k] = q . k / attn_scale
        qh = np.ascontiguousarray(q.transpose(0, 2, 1, 3))          # [B,H,T,dh]
        kh = np.ascontiguousarray(k.transpose(0, 2, 3, 1))          # [B,H,dh,T]
        scores = np.matmul(qh, kh) / attn_scale
        scores = tap(p + "attn.hook_attn_scores", scores.astype(dtype))
        pattern = tap(p + "attn.hook_pattern", softmax_lastdim(scores).astype(dtype))   # :148-150
        vh = np.ascontiguousarray(v.transpose(0, 2, 1, 3))          # [B,H,T,dh]
        z = tap(p + "attn.hook_z", np.ascontiguousarray(np.matmul(pattern, vh).transpose(0, 2, 1, 3)))   # :267-281
        # attention.py:155-167: sum_h z[:,:,h,:] @ W_O[h] + b_O  ==  [B*T, H*dh] @ [H*dh, d]
        attn_out = (z.reshape(B * T, H * dh) @ P[p + "attn.W_O"].reshape(H * dh, d)).reshape(B, T, d) + P[p + "attn.b_O"]
        attn_out = tap(p + "hook_attn_out", attn_out.astype(dtype))                    # transformer_block.py:117-119
        resid_mid = tap(p + "hook_resid_mid", resid_pre + attn_out)                    # :122-124
        y2, sc2 = layer_norm(resid_mid, P[p + "ln2.w"], P[p + "ln2.b"], eps)           # :130
        tap(p + "ln2.hook_scale", sc2)
        y2 = tap(p + "ln2.hook_normalized", y2)
        pre = tap(p + "mlp.hook_pre", y2 @ P[p + "mlp.W_in"] + P[p + "mlp.b_in"])     # mlp.py:67-70
        post = tap(p + "mlp.hook_post", act(pre))                                      # mlp.py:72
        mlp_out = tap(p + "hook_mlp_out", post @ P[p + "mlp.W_out"] + P[p + "mlp.b_out"])   # mlp.py:77-80 ; block :133
        resid = tap(p + "hook_resid_post", resid_mid + mlp_out)                        # block :134

    if stop_at_layer is not None:             # base_vit.py:189-190
        return resid, cache

    y, sc = layer_norm(resid, P["ln_final.w"], P["ln_final.b"], eps)   # base_vit.py:192
    tap("ln_final.hook_scale", sc)
    y = tap("ln_final.hook_normalized", y)
    tap("hook_ln_final", y)                   # base_vit.py:193 (observe-only)
    xcls = y[:, 0]                            # base_vit.py:199-208 (classification_type == 'cls')
    if cfg.get("return_type", "pre_logits") != "pre_logits":
        xcls = xcls @ P["head.W_H"] + P["head.b_H"]      # head.py:27-37
    tap("hook_post_head_pre_normalize", xcls)            # base_vit.py:212 (observe-only)
    if cfg.get("normalize_output", False):               # base_vit.py:214-215 ; F.normalize eps=1e-12
        nrm = np.maximum(np.sqrt((xcls * xcls).sum(axis=-1, keepdims=True)), dtype(1e-12))
        xcls = xcls / nrm
    return xcls, cache


def fingerprint(x: Array, n_samples: int = 32) -> dict:
    """Compact, order-sensitive digest of a tensor used by the golden fixtures for the
    full-size configs (whole tensors would be hundreds of MB): shape, sum, L2 norm, a position-
    weighted sum (catches permutations/transposes) and ``n_samples`` values at fixed strided flat
    indices."""
    a = np.ascontiguousarray(np.asarray(x), dtype=np.float64).reshape(-1)
    n = a.size
    idx = (np.arange(n_samples, dtype=np.int64) * 2654435761 + 12345) % max(n, 1)
    w = np.cos(np.arange(n, dtype=np.float64) * 0.61803398875)   # non-symmetric position weights
    return {
        "shape": list(np.asarray(x).shape),
        "sum": float(a.sum()),
        "l2": float(np.sqrt((a * a).sum())),
        "wsum": float((a * w).sum()),
        "idx": idx.tolist(),
        "vals": a[idx].tolist(),
    }
