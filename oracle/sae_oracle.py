"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of ViT-Prisma's SAE forward, loss,
backward and optimiser step: the top-k SAE (k given), the ReLU + L1 SAE (k = None, l1_coefficient) and the Transcoder
(``target`` given; parameters ``b_dec_out`` and optionally ``W_skip`` in P).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Parity pin: the reference's own tests hold NO numeric golden vector for the SAE path (SURVEY.md
section 4: "parity unpinned" by the reference), so this restatement is pinned against outputs of the
reference itself executed in the build container -- ``tests/golden/gen_golden_sae.py`` drives the
reference's real ``StandardSparseAutoencoder`` and ``VisionSAETrainer.train_step`` for three
consecutive steps and dumps losses, gradients and post-step parameters
(``tests/test_oracle_sae_vs_golden.py``); the ReLU + L1 form against ``tests/golden/sae_variants_steps.npz``
(``relu_l1`` / ``relu_ghost`` / ``transcoder``: the reference's own classes through its own train_step,
``tests/golden/gen_golden_sae_variants.py``).

Citations are relative to /root/reference/src/vit_prisma/.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

Array = np.ndarray
F32 = np.float32


def ln_in(x: Array, eps: float = 1e-5) -> Tuple[Array, Array, Array]:
    """sae/sae.py:78-87 -- mu = mean; x <- x - mu; std = x.std() (UNBIASED); x / (std + eps)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    std = xc.std(axis=-1, keepdims=True, ddof=1)
    return xc / (std + x.dtype.type(eps)), mu, std


def topk_mask(hidden_pre: Array, k: int) -> Tuple[Array, Array]:
    """sae/sae.py:795-810 -- torch.topk(k) -> relu -> scatter into zeros.  Returns (idx [N,k] sorted by
    value descending, vals [N,k] after relu)."""
    part = np.argpartition(-hidden_pre, k - 1, axis=-1)[:, :k]
    pv = np.take_along_axis(hidden_pre, part, axis=-1)
    order = np.argsort(-pv, axis=-1, kind="stable")
    idx = np.take_along_axis(part, order, axis=-1)
    vals = np.maximum(np.take_along_axis(hidden_pre, idx, axis=-1), hidden_pre.dtype.type(0))
    return idx, vals


def norm_mode(layer_norm) -> str:
    """cfg.normalize_activations as the oracle's functions take it: a bool (round 1-5 callers: layer_norm on / off) or the config string."""
    if layer_norm is True or layer_norm == "layer_norm":
        return "layer_norm"
    if layer_norm in (False, None, "none"):
        return "none"
    assert layer_norm == "constant_norm_rescale", layer_norm
    return layer_norm


def sae_forward(P: Dict[str, Array], x: Array, k: Optional[int], layer_norm: bool = True, batch_mean: Optional[Array] = None,
                n_global: Optional[int] = None, l1_coefficient: float = 0.0, dead_mask: Optional[Array] = None,
                target: Optional[Array] = None, ghost_global: Optional[Tuple[Array, float]] = None,
                idx: Optional[Array] = None, act: str = "relu", lp_norm: float = 1.0) -> Dict[str, Array]:
    """StandardSparseAutoencoder.forward, sae/sae.py:597-645 (encode :557-581, decode :583-595, loss
    :144-149; for topk l1_loss is None and loss == mse_loss, :617-626).  k = None: activation_fn_str = "relu"
    (get_activation_fn :813-830) with the L1 sparsity term l1_coefficient * mean_n ||f_n||_1 (:617-626, lp_norm = 1).

    batch_mean / n_global: the data-parallel form -- mean_n(x) over the GLOBAL
    batch and the global token count; default = this batch (single process, the reference).
    dead_mask [d_sae] bool (use_ghost_grads, training): adds _compute_ghost_residual_loss (sae/sae.py:151-179) -- also when
    no feature is dead (ghost_out is then zero and the term is a constant: the reference adds it all the same).
    ghost_global = (mean over the GLOBAL batch of the residual x - sae_out [d_in], the global batch's mse loss): the data-parallel form
    of the ghost term, whose two batch-wide quantities (:156, :172) are then the whole batch's, not this shard's; its mean runs over
    n_global tokens.
    target [N, d_in] (Transcoder.forward, sae/transcoder.py:66-116): the activation to reconstruct; P then holds the decoder's
    own bias ``b_dec_out`` (decode, :54-64; ``b_dec`` only centres the encoder input, :35-37) and optionally ``W_skip``
    [d_in, d_in] (``sae_out += x @ W_skip.mT`` on the RAW input, before LN-out, :73-76); loss and normaliser against it
    (:78; batch_mean is then the target's).
    layer_norm: True / "layer_norm" (:74-90), False / None / "none", or "constant_norm_rescale" (:60-72: x * sqrt(d_in) / ||x|| on the way in,
    / the same coefficient on the way out).  act (k = None): "relu" or "tanh-relu" = tanh(relu(.)) (get_activation_fn :823-830).
    lp_norm (k = None): the sparsity term is l1_coefficient * mean_n ||f_n||_p (:617, ``feature_acts.norm(p=self.lp_norm, dim=1)``).
    idx [N, k] (tests only: the fp32-vs-float64 noise floors): the selection to keep instead of hidden_pre's own top-k -- the same
    computation carried in another precision must not move to another set where two pre-activations tie within its noise."""
    dt = x.dtype.type
    N, d = x.shape
    mode = norm_mode(layer_norm)
    coeff = None
    if mode == "layer_norm":
        xh, mu, std = ln_in(x)
    elif mode == "constant_norm_rescale":                              # :60-72
        coeff = dt(d ** 0.5) / np.sqrt((x ** 2).sum(axis=-1, keepdims=True))
        xh, mu, std = x * coeff, np.zeros((N, 1), x.dtype), dt(1) / coeff
    else:
        xh, mu, std = x, np.zeros((N, 1), x.dtype), np.ones((N, 1), x.dtype)
    sae_in = xh - P["b_dec"]                                           # :563-565
    hidden_pre = sae_in @ P["W_enc"] + P["b_enc"]                      # :567-574
    if k is None:
        feats = np.maximum(hidden_pre, dt(0))                          # :576 with torch.nn.ReLU
        if act == "tanh-relu":
            feats = np.tanh(feats)                                     # :823-830
        else:
            assert act == "relu", act
        idx = vals = None
    else:
        if idx is None:
            idx, vals = topk_mask(hidden_pre, k)                       # :576
        else:
            vals = np.maximum(np.take_along_axis(hidden_pre, idx, axis=-1), dt(0))
        feats = np.zeros_like(hidden_pre)
        np.put_along_axis(feats, idx, vals, axis=-1)
    if target is None:
        pre_out = feats @ P["W_dec"] + P["b_dec"]                      # :584-591
        y = x
    else:
        pre_out = feats @ P["W_dec"] + P["b_dec_out"]                  # transcoder.py:54-64
        if P.get("W_skip") is not None:
            pre_out = pre_out + x @ P["W_skip"].T                      # transcoder.py:73-74
        y = target
    if mode == "layer_norm":
        sae_out = pre_out * std + mu                                   # :89-90 (no eps on the way out)
    elif mode == "constant_norm_rescale":
        sae_out = pre_out / coeff                                      # :68-70
    else:
        sae_out = pre_out
    bm = y.mean(axis=0, keepdims=True) if batch_mean is None else batch_mean.reshape(1, -1)
    nf = np.sqrt(((y - bm) ** 2).sum(axis=-1, keepdims=True))          # :145-147
    ng = N if n_global is None else n_global
    mse = ((sae_out - y) ** 2 / nf).sum() / dt(ng * y.shape[1])        # :146-148 (mean over N x the width of what is reconstructed:
    #                                                                     d_in, or a transcoder's d_out -- transcoder.py:12, 78)
    l0 = (feats > 0).sum(axis=-1).astype(np.float64).mean()            # train_sae.py:364
    l1 = None
    loss = mse
    lp_S = None
    if k is None:                                                      # :617-626: sparsity = ||f||_p per token, mean over the batch
        if lp_norm == 1:
            l1 = dt(l1_coefficient) * (np.abs(feats).sum(axis=-1).sum() / dt(ng))
        else:
            lp_S = (np.abs(feats) ** dt(lp_norm)).sum(axis=-1, keepdims=True)
            l1 = dt(l1_coefficient) * ((lp_S ** dt(1.0 / lp_norm)).sum() / dt(ng))
        loss = mse + l1
    gh = None
    if dead_mask is not None:                                          # sae/sae.py:151-179
        # (a Transcoder hands the ghost term its INPUT x, not the target, transcoder.py:82-86: the residual is x - sae_out and the rescaling
        # mse is _compute_mse_loss(x, sae_out), normalised by x's own centred norms -- as the reference computes it, d_out = d_in)
        ghost_mse = mse
        if target is not None and ghost_global is None:
            nfx = np.sqrt(((x - x.mean(axis=0, keepdims=True)) ** 2).sum(axis=-1, keepdims=True))
            ghost_mse = ((sae_out - x) ** 2 / nfx).sum() / dt(ng * d)
        res = x - sae_out
        rc = res - (res.mean(axis=0, keepdims=True) if ghost_global is None else np.asarray(ghost_global[0], x.dtype).reshape(1, -1))
        l2 = np.sqrt((res ** 2).sum(axis=-1))
        E = np.exp(hidden_pre[:, dead_mask])
        G0 = E @ P["W_dec"][dead_mask]
        s = l2 / (dt(1e-6) + np.sqrt((G0 ** 2).sum(axis=-1)) * dt(2))         # (detached)
        G = G0 * s[:, None]
        den = np.sqrt((rc ** 2).sum(axis=-1, keepdims=True))                   # (detached)
        mg = (G - res) ** 2 / den
        r = dt(ghost_mse if ghost_global is None else ghost_global[1]) / (mg + dt(1e-6))      # (detached)
        ghost = (r * mg).sum(dtype=np.float64) / (ng * x.shape[1])
        gh = dict(E=E, s=s, G=G, res=res, den=den, r=r, mask=dead_mask, loss=dt(ghost), ng=ng)
        loss = loss + dt(ghost)
    return dict(sae_in=sae_in, hidden_pre=hidden_pre, idx=idx, vals=vals, feature_acts=feats, sae_out=sae_out,
                mu=mu, std=std, norm_factor=nf, loss=dt(loss), mse_loss=dt(mse), l1_loss=None if l1 is None else dt(l1), l0=l0,
                ghost=gh, ghost_loss=None if gh is None else gh["loss"], target=target, act=act, lp_norm=lp_norm, lp_S=lp_S, norm_mode=mode)


def sae_backward(P: Dict[str, Array], x: Array, fw: Dict[str, Array], layer_norm: bool = True,
                 n_global: Optional[int] = None, l1_coefficient: float = 0.0, gate: Optional[Array] = None) -> Dict[str, Array]:
    """What ``loss.backward()`` (train_sae.py:392) deposits in the four ``.grad`` fields.  l1_coefficient (ReLU + L1
    forward only): d l1_loss / d f = l1_coefficient / N where f > 0 (the 1-norm's subgradient at 0 is 0, as torch's).
    gate (tests of the dense step): the ReLU gate [N, d_sae] to use instead of ``feature_acts > 0`` -- the gradient is
    discontinuous in the sign of hidden_pre, and an entry within fp32 summation noise of zero may fall on either side."""
    dt = x.dtype.type
    N, d = x.shape
    ng = N if n_global is None else n_global
    tc = fw.get("target") is not None
    d_out = dt(2.0) * (fw["sae_out"] - (fw["target"] if tc else x)) / fw["norm_factor"] / dt(ng * (fw["target"].shape[1] if tc else d))
    d_pre = d_out * fw["std"] if norm_mode(layer_norm) != "none" else d_out      # (constant_norm_rescale: std holds 1 / coefficient)
    feats = fw["feature_acts"]
    g = {}
    g["W_dec"] = feats.T @ d_pre
    d_feats = d_pre @ P["W_dec"].T
    if fw.get("l1_loss") is not None:
        if fw.get("lp_S") is None:
            d_feats = d_feats + dt(l1_coefficient) / dt(ng)
        else:                                                          # d ||f||_p / d f = ||f||_p^(1 - p) f^(p - 1)  (f >= 0; 0 where f = 0 as torch's)
            lp = dt(fw["lp_norm"])
            with np.errstate(divide="ignore", invalid="ignore"):
                t = np.where(fw["lp_S"] > 0, fw["lp_S"] ** dt(1.0 / fw["lp_norm"] - 1.0), dt(0))
                d_feats = d_feats + dt(l1_coefficient) / dt(ng) * t * np.where(feats > 0, feats ** (lp - dt(1)), dt(0))
    if fw.get("act") == "tanh-relu":
        d_feats = d_feats * (dt(1) - feats * feats)                    # d tanh(relu(h)) / d h = 1 - f^2 where h > 0
    d_hidden = np.where(feats > 0 if gate is None else gate, d_feats, dt(0))     # topk scatter + ReLU gates
    gh = fw.get("ghost")
    if gh is not None and gh["mask"].any():                           # gradient of the ghost residual loss: through ghost_out only
        dG0 = gh["r"] * dt(2) * (gh["G"] - gh["res"]) / gh["den"] / dt(gh.get("ng", N) * d) * gh["s"][:, None]
        g["W_dec"][gh["mask"]] += gh["E"].T @ dG0
        d_hidden[:, gh["mask"]] += (dG0 @ P["W_dec"][gh["mask"]].T) * gh["E"]
    g["W_enc"] = fw["sae_in"].T @ d_hidden
    g["b_enc"] = d_hidden.sum(axis=0)
    d_sae_in = d_hidden @ P["W_enc"].T
    if tc:                                                            # the two roles of b_dec are two parameters here
        g["b_dec"] = -d_sae_in.sum(axis=0)
        g["b_dec_out"] = d_pre.sum(axis=0)
        if P.get("W_skip") is not None:
            g["W_skip"] = d_pre.T @ x
    else:
        g["b_dec"] = d_pre.sum(axis=0) - d_sae_in.sum(axis=0)         # decode bias + "sae_in = x_hat - b_dec"
    return g


def renorm_decoder(P: Dict[str, Array]) -> None:
    """set_decoder_norm_to_unit_norm, sae/sae.py:275-277 (in place)."""
    P["W_dec"] /= np.linalg.norm(P["W_dec"], axis=1, keepdims=True)


def grad_total_norm(g: Dict[str, Array]) -> float:
    return float(np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g.values())))


def clip_and_project(P: Dict[str, Array], g: Dict[str, Array], max_grad_norm: Optional[float]) -> float:
    """clip_grad_norm_ over all four tensors (train_sae.py:394-397; coef = clamp(max/(norm+1e-6), max=1))
    then remove_gradient_parallel_to_decoder_directions (sae/sae.py:279-297).  In place; returns the
    pre-clip total norm."""
    total = grad_total_norm(g)
    if max_grad_norm:
        coef = min(max_grad_norm / (total + 1e-6), 1.0)
        for v in g.values():
            v *= v.dtype.type(coef)
    par = (g["W_dec"] * P["W_dec"]).sum(axis=1, keepdims=True)
    g["W_dec"] -= par * P["W_dec"]
    return total


def adam_step(P: Dict[str, Array], g: Dict[str, Array], m: Dict[str, Array], v: Dict[str, Array], lr: float, step: int,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam(params, lr) defaults (train_sae.py:229): no weight decay, no amsgrad."""
    bc1 = 1.0 - b1 ** step
    bc2s = np.sqrt(1.0 - b2 ** step)
    for key in P:
        dt = P[key].dtype.type
        m[key] += (g[key] - m[key]) * dt(1.0 - b1)
        v[key] *= dt(b2)
        v[key] += dt(1.0 - b2) * g[key] * g[key]
        denom = np.sqrt(v[key]) / dt(bc2s) + dt(eps)
        P[key] -= dt(lr / bc1) * (m[key] / denom)


def lr_lambda_cosine_warmup(step: int, warm_up_steps: int, training_steps: int, lr_end: float) -> float:
    """sae/training/get_scheduler.py:50-60 (``cosineannealingwarmup``; NB ``lr_end`` enters as a
    *multiplier*, train_sae.py:236 passes cfg.lr / 10)."""
    if step < warm_up_steps:
        return (step + 1) / warm_up_steps
    progress = (step - warm_up_steps) / (training_steps - warm_up_steps)
    return lr_end + 0.5 * (1 - lr_end) * (1 + np.cos(np.pi * progress))


def train_step(P: Dict[str, Array], opt: Dict[str, Dict[str, Array]], stats: Dict[str, Array], x: Array, k: Optional[int], lr: float,
               step: int, max_grad_norm: Optional[float] = 1.0, layer_norm: bool = True, l1_coefficient: float = 0.0,
               dead_feature_window: Optional[int] = None, target: Optional[Array] = None, gate: Optional[Array] = None,
               act: str = "relu", lp_norm: float = 1.0) -> Dict[str, float]:
    """VisionSAETrainer.train_step, sae/train_sae.py:278-411, in its order: renorm decoder -> forward ->
    firing statistics -> backward -> clip -> project -> Adam.  ``step`` is 1-based (Adam's step count).
    gate (tests of the ReLU steps, see sae_backward): the step as it continues when the ReLU gates [N, d_sae] of the entries within
    fp32 summation noise of zero fall as given -- the backward and the firing statistics under these gates."""
    renorm_decoder(P)                                                   # :306-307
    dead = None if dead_feature_window is None else stats["n_fwd_since_fired"] > dead_feature_window      # :330-332 (use_ghost_grads)
    fw = sae_forward(P, x, k, layer_norm, l1_coefficient=l1_coefficient, dead_mask=dead, target=target, act=act, lp_norm=lp_norm)
    fired = ((fw["feature_acts"] > 0) if gate is None else gate).sum(axis=0)      # :356-361
    stats["n_fwd_since_fired"] += 1
    stats["n_fwd_since_fired"][fired > 0] = 0
    stats["act_freq_scores"] += fired.astype(stats["act_freq_scores"].dtype)
    g = sae_backward(P, x, fw, layer_norm, l1_coefficient=l1_coefficient, gate=gate)
    total = clip_and_project(P, g, max_grad_norm)
    adam_step(P, g, opt["m"], opt["v"], lr, step)
    return dict(loss=float(fw["loss"]), mse_loss=float(fw["mse_loss"]), l0=float(fw["l0"]), grad_norm=total,
                l1_loss=None if fw["l1_loss"] is None else float(fw["l1_loss"]),
                ghost_loss=None if fw["ghost_loss"] is None else float(fw["ghost_loss"]))


# ---- Gated SAE (sae/sae.py:648-792), ReLU activation ----------------------------------------------------------------------
def gated_forward(P: Dict[str, Array], x: Array, layer_norm: bool = True, l1_coefficient: float = 0.0,
                  batch_mean: Optional[Array] = None, n_global: Optional[int] = None, k: Optional[int] = None,
                  active: Optional[Array] = None) -> Dict[str, Array]:
    """GatedSparseAutoencoder.forward (:730-771): gate path (sae_in @ W_enc + b_gate) > 0 (:703-706), magnitude path with shared weights
    sae_in @ (W_enc * exp(r_mag)) + b_mag (:708-712), auxiliary reconstruction of sae_in through the gate (:786-792).
    k = None: activation_fn_str = "relu" -- magnitudes relu(mag_pre), gate activations relu(gate_pre), L1 on them weighted by the
    decoder row norms (:780-784).  k given: activation_fn_str = "topk" -- BOTH go through TopK (:795-810; magnitudes :714, gate
    activations :773-778) and there is no L1 term (:741-745).
    active [N, d_sae] bool (tests of the dense gated step): the Heaviside gates to use instead of ``gate_pre > 0`` -- feature_acts is
    discontinuous in the sign of gate_pre, and an entry within fp32 summation noise of zero may fall on either side."""
    dt = x.dtype.type
    N, d = x.shape
    if layer_norm:
        xh, mu, std = ln_in(x)
    else:
        xh, mu, std = x, np.zeros((N, 1), x.dtype), np.ones((N, 1), x.dtype)
    S = xh - P["b_dec"]
    gate_pre = S @ P["W_enc"] + P["b_gate"]
    active = gate_pre > 0 if active is None else active
    mag_pre = S @ (P["W_enc"] * np.exp(P["r_mag"])) + P["b_mag"]
    if k is None:
        mags = np.maximum(mag_pre, dt(0))
        pg = np.maximum(gate_pre, dt(0))                               # _compute_gate_activation :773-778
    else:
        mags, pg = np.zeros_like(mag_pre), np.zeros_like(gate_pre)
        idx, vals = topk_mask(mag_pre, k)
        np.put_along_axis(mags, idx, vals, axis=-1)
        idx, vals = topk_mask(gate_pre, k)
        np.put_along_axis(pg, idx, vals, axis=-1)
    feats = np.where(active, mags, dt(0))
    pre_out = feats @ P["W_dec"] + P["b_dec"]
    sae_out = pre_out * std + mu if layer_norm else pre_out
    bm = x.mean(axis=0, keepdims=True) if batch_mean is None else batch_mean.reshape(1, -1)      # (data-parallel form: the GLOBAL batch's)
    ng = N if n_global is None else n_global
    nf = np.sqrt(((x - bm) ** 2).sum(axis=-1, keepdims=True))
    mse = ((sae_out - x) ** 2 / nf).sum() / dt(ng * d)
    wn = np.linalg.norm(P["W_dec"], axis=1)
    l1 = dt(l1_coefficient) * ((pg * wn).sum(axis=-1).sum() / dt(ng)) if k is None else dt(0)
    via = pg @ P["W_dec"] + P["b_dec"]
    aux = ((via - S) ** 2).sum(axis=-1).sum() / dt(ng)
    l0 = (feats > 0).sum(axis=-1).astype(np.float64).mean()
    return dict(sae_in=S, gate_pre=gate_pre, mag_pre=mag_pre, feature_acts=feats, pg=pg, via=via, sae_out=sae_out, mu=mu, std=std,
                norm_factor=nf, wn=wn, n_global=ng, loss=dt(mse + l1 + aux), mse_loss=dt(mse), l1_loss=dt(l1), aux_loss=dt(aux), l0=l0,
                topk=k is not None)


def gated_backward(P: Dict[str, Array], x: Array, fw: Dict[str, Array], layer_norm: bool = True, l1_coefficient: float = 0.0,
                   gates: Optional[Tuple[Array, Array]] = None) -> Dict[str, Array]:
    """loss.backward() of the gated forward.  b_enc takes no part in it (its .grad stays None in the reference: no entry here).
    gates = (feature_acts > 0, gate_pre > 0) to use instead of the oracle's own (tests: entries within summation noise of zero).
    Top-k form (fw["topk"]): the gradient reaches the kept entries only (TopK's scatter + its ReLU), and there is no L1 term."""
    dt = x.dtype.type
    N, d = x.shape
    S, feats, pg = fw["sae_in"], fw["feature_acts"], fw["pg"]
    topk = bool(fw.get("topk", False))
    if topk:
        l1_coefficient = 0.0
    on_f, on_g = (feats > 0, (pg > 0) if topk else (fw["gate_pre"] > 0)) if gates is None else gates
    N = fw["n_global"]                                                 # (every mean is over the global batch)
    d_out = dt(2.0) * (fw["sae_out"] - x) / fw["norm_factor"] / dt(N * d)
    dY = d_out * fw["std"] if layer_norm else d_out
    dVia = dt(2.0) * (fw["via"] - S) / dt(N)
    er = np.exp(P["r_mag"])
    dM = np.where(on_f, dY @ P["W_dec"].T, dt(0))                      # (no gradient through the Heaviside gate)
    dG = np.where(on_g, dVia @ P["W_dec"].T + dt(l1_coefficient) / dt(N) * fw["wn"], dt(0))
    dP = dM * er + dG
    g = {}
    g["W_dec"] = feats.T @ dY + pg.T @ dVia + (dt(l1_coefficient) / dt(N)) * pg.sum(axis=0)[:, None] * (P["W_dec"] / fw["wn"][:, None])
    g["W_enc"] = S.T @ dP
    g["b_gate"] = dG.sum(axis=0)
    g["b_mag"] = dM.sum(axis=0)
    g["r_mag"] = (dM * (fw["mag_pre"] - P["b_mag"])).sum(axis=0)       # d (p e^r) / d r = p e^r = mag_pre - b_mag
    dS = dP @ P["W_enc"].T - dVia                                      # through the two encoder paths and as the aux target
    g["b_dec"] = dY.sum(axis=0) + dVia.sum(axis=0) - dS.sum(axis=0)
    return g


def gated_train_step(P: Dict[str, Array], opt: Dict[str, Dict[str, Array]], stats: Dict[str, Array], x: Array, lr: float, step: int,
                     max_grad_norm: Optional[float] = 1.0, layer_norm: bool = True, l1_coefficient: float = 0.0,
                     k: Optional[int] = None, active: Optional[Array] = None) -> Dict[str, float]:
    """VisionSAETrainer.train_step (sae/train_sae.py:278-411) on a GatedSparseAutoencoder.  P, opt hold every parameter but b_enc
    (untouched by the optimizer: no gradient).  k: the top-k form (see gated_forward).  active: the step as it continues when the
    Heaviside gates within fp32 noise of zero fall as given (see gated_forward; the backward's gate ReLU follows them)."""
    renorm_decoder(P)
    fw = gated_forward(P, x, layer_norm, l1_coefficient, k=k, active=active)
    fired = (fw["feature_acts"] > 0).sum(axis=0)
    stats["n_fwd_since_fired"] += 1
    stats["n_fwd_since_fired"][fired > 0] = 0
    stats["act_freq_scores"] += fired.astype(stats["act_freq_scores"].dtype)
    g = gated_backward(P, x, fw, layer_norm, l1_coefficient, gates=None if active is None else (fw["feature_acts"] > 0, active))
    Pg = {k_: P[k_] for k_ in g}
    total = clip_and_project(Pg, g, max_grad_norm)
    adam_step(Pg, g, {k_: opt["m"][k_] for k_ in g}, {k_: opt["v"][k_] for k_ in g}, lr, step)
    return dict(loss=float(fw["loss"]), mse_loss=float(fw["mse_loss"]), l1_loss=float(fw["l1_loss"]), aux_loss=float(fw["aux_loss"]),
                l0=float(fw["l0"]), grad_norm=total)
