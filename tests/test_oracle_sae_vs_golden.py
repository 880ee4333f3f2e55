"""Pin oracle/sae_oracle.py against what the reference's real VisionSAETrainer.train_step produced
(tests/golden/gen_golden_sae.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import sae_oracle as O
from oracle.vit_oracle import fingerprint
from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state

from conftest import GOLDEN, rel_fro


def fresh(d_in, d_sae):
    P = {k: v.copy() for k, v in synth_sae_state(d_in, d_sae, 0).items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
    return P, opt, stats


def test_small_three_steps_full_tensors():
    g = np.load(os.path.join(GOLDEN, "sae_small_steps.npz"))
    d_in, d_sae, k, N = 64, 512, 8, 256
    P, opt, stats = fresh(d_in, d_sae)
    for t in range(3):
        x = synth_sae_batch(N, d_in, seed=t)
        # pieces of the step, checked individually before the fused step mutates P
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        fw = O.sae_forward(Pc, x, k)
        assert rel_fro(fw["hidden_pre"], g[f"s{t}_hidden_pre"]) < 1e-5
        grads = O.sae_backward(Pc, x, fw)
        for n in ("W_enc", "W_dec", "b_enc", "b_dec"):
            assert rel_fro(grads[n], g[f"s{t}_grad_{n}"]) < 2e-4, (t, n)
        out = O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1)
        loss, mse, l0, gn = g[f"s{t}_scalars"]
        assert abs(out["loss"] - loss) <= 1e-5 * abs(loss) and abs(out["mse_loss"] - mse) <= 1e-5 * abs(mse)
        assert out["l0"] == l0 == 8.0
        assert abs(out["grad_norm"] - gn) <= 1e-4 * gn
        for n in ("W_enc", "W_dec", "b_enc", "b_dec"):
            assert rel_fro(P[n], g[f"s{t}_param_{n}"]) < 1e-5, (t, n)
            assert rel_fro(opt["m"][n], g[f"s{t}_m_{n}"]) < 2e-4, (t, n)
            assert rel_fro(opt["v"][n], g[f"s{t}_v_{n}"]) < 4e-4, (t, n)
        assert np.array_equal(stats["act_freq_scores"], g[f"s{t}_act_freq"])
        assert np.array_equal(stats["n_fwd_since_fired"], g[f"s{t}_n_since"])


@pytest.mark.slow
def test_b32_config3_step_fingerprints():
    with open(os.path.join(GOLDEN, "sae_b32_steps.json")) as f:
        G = json.load(f)
    c = G["config"]
    P, opt, stats = fresh(c["d_in"], c["d_sae"])
    for t, want in enumerate(G["steps"][:2]):
        x = synth_sae_batch(c["n_tokens"], c["d_in"], seed=t)
        out = O.train_step(P, opt, stats, x, c["k"], lr=c["lr"], step=t + 1)
        assert abs(out["loss"] - want["loss"]) <= 1e-5 * want["loss"], t
        assert out["l0"] == want["l0"] == 32.0
        # torch's CPU vector_norm accumulates 18.9 M fp32 squares with ~1e-3 relative error (measured:
        # 9.5e-4 on a dense tensor of this size), so the reference's own clip norm is only good to ~1e-3;
        # the four gradient tensors themselves match to 1e-7 (fingerprints below)
        assert abs(out["grad_norm"] - want["grad_norm"]) <= 2e-3 * want["grad_norm"], t
        for n in ("W_enc", "W_dec", "b_enc", "b_dec"):
            fp = fingerprint(P[n])
            assert abs(fp["l2"] - want["params"][n]["l2"]) <= 1e-6 * want["params"][n]["l2"], (t, n)
            assert np.max(np.abs(np.array(fp["vals"]) - np.array(want["params"][n]["vals"]))) < 2e-5, (t, n)
        assert abs(fingerprint(stats["act_freq_scores"])["sum"] - want["act_freq"]["sum"]) < 0.5


def test_relu_l1_three_steps_vs_reference_fixture():
    """The ReLU + L1 form of the oracle (k = None) against what the reference's own StandardSparseAutoencoder
    (activation_fn_str = "relu", l1_coefficient = 2e-3) produced through its own train_step
    (tests/golden/gen_golden_sae_variants.py): scalars, statistics, parameters after step 3."""
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, d_sae, N, l1c = 64, 512, 256, 2e-3
    P = {n: g[f"relu_l1_init_{n}"].copy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": g["relu_l1_since0"].astype(np.float32).copy(), "act_freq_scores": np.zeros(d_sae, np.float32)}
    for t in range(3):
        out = O.train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=t), None, lr=1e-3, step=t + 1, l1_coefficient=l1c)
        loss, mse, l1, l0 = g[f"relu_l1_s{t}_scalars"][:4]
        assert abs(out["loss"] - loss) <= 1e-5 * abs(loss) and abs(out["mse_loss"] - mse) <= 1e-5 * abs(mse), (t, out, loss, mse)
        assert abs(out["l1_loss"] - l1) <= 1e-5 * abs(l1) and abs(out["l0"] - l0) <= 1e-6 * l0
        assert np.array_equal(stats["act_freq_scores"], g[f"relu_l1_s{t}_act_freq"])
        assert np.array_equal(stats["n_fwd_since_fired"], g[f"relu_l1_s{t}_n_since"])
    for n in P:
        assert rel_fro(P[n], g[f"relu_l1_s2_param_{n}"]) < 1e-5, n


def test_relu_ghost_three_steps_vs_reference_fixture():
    """Ghost gradients (use_ghost_grads, a third of the features dead from the start): the oracle's ReLU + L1 + ghost form
    against the reference's own run (relu_ghost of tests/golden/sae_variants_steps.npz)."""
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, d_sae, N, l1c = 64, 512, 256, 2e-3
    P = {n: g[f"relu_ghost_init_{n}"].copy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": g["relu_ghost_since0"].astype(np.float32).copy(), "act_freq_scores": np.zeros(d_sae, np.float32)}
    assert (stats["n_fwd_since_fired"] > 1).sum() > 100
    for t in range(3):
        out = O.train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=t), None, lr=1e-3, step=t + 1, l1_coefficient=l1c,
                           dead_feature_window=1)
        loss, mse, l1, l0, ghost = g[f"relu_ghost_s{t}_scalars"][:5]
        assert abs(out["loss"] - loss) <= 2e-5 * abs(loss) and abs(out["mse_loss"] - mse) <= 1e-5 * abs(mse), (t, out, loss, mse)
        assert abs(out["ghost_loss"] - ghost) <= 2e-5 * abs(ghost) and abs(out["l1_loss"] - l1) <= 1e-5 * abs(l1), (t, out, ghost)
        assert np.array_equal(stats["act_freq_scores"], g[f"relu_ghost_s{t}_act_freq"])
        assert np.array_equal(stats["n_fwd_since_fired"], g[f"relu_ghost_s{t}_n_since"])
    for n in P:
        assert rel_fro(P[n], g[f"relu_ghost_s2_param_{n}"]) < 2e-5, n


def test_topk_ghost_three_steps_vs_reference_fixture():
    """Ghost gradients on a top-k SAE (sae.py:151-179 behind TopK :795-810; the dead mask of train_sae.py:330-332): the oracle's
    top-k + ghost form against the reference's own run (topk_ghost of tests/golden/sae_variants_steps.npz)."""
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, d_sae, N, k = 64, 512, 256, 8
    P = {n: g[f"topk_ghost_init_{n}"].copy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": g["topk_ghost_since0"].astype(np.float32).copy(), "act_freq_scores": np.zeros(d_sae, np.float32)}
    assert (stats["n_fwd_since_fired"] > 1).sum() > 100
    for t in range(3):
        out = O.train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=t), k, lr=1e-3, step=t + 1, dead_feature_window=1)
        loss, mse, _, l0, ghost = g[f"topk_ghost_s{t}_scalars"][:5]
        assert abs(out["loss"] - loss) <= 2e-5 * abs(loss) and abs(out["mse_loss"] - mse) <= 1e-5 * abs(mse), (t, out, loss, mse)
        assert abs(out["ghost_loss"] - ghost) <= 2e-5 * abs(ghost) and out["l0"] == l0, (t, out, ghost)
        assert np.array_equal(stats["act_freq_scores"], g[f"topk_ghost_s{t}_act_freq"])
        assert np.array_equal(stats["n_fwd_since_fired"], g[f"topk_ghost_s{t}_n_since"])
    # (a dead feature's only gradient here is the ghost term: entries of ~1e-9, where fp32 summation-order noise -- numpy against torch --
    # is an ABSOLUTE error that Adam's g / (|g| + 1e-8) turns into lr-sized differences on those columns of W_enc: 3.2e-4 after three
    # steps with every loss of every step within 2e-5; the ReLU form above keeps its dead features' L1-free gradient larger)
    for n in P:
        assert rel_fro(P[n], g[f"topk_ghost_s2_param_{n}"]) < (1e-3 if n == "W_enc" else 2e-4), (n, rel_fro(P[n], g[f"topk_ghost_s2_param_{n}"]))   # (W_dec's dead rows: the same effect, 8e-5)


@pytest.mark.parametrize("variant,d_out", [("transcoder", 64), ("transcoder_dout", 40)])
def test_transcoder_three_steps_vs_reference_fixture(variant, d_out):
    """The Transcoder form of the oracle (target given, b_dec_out; top-k, k = 8) against the reference's own Transcoder through its own
    train_step: with the skip matrix at d_out = d_in (transcoder of tests/golden/sae_variants_steps.npz) and WITHOUT it at d_out = 40
    != d_in = 64 (transcoder_dout: the reference's skip term only type-checks at equal widths, transcoder.py:10, 73-76; the loss is the
    mean over N x d_out)."""
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, d_sae, N = 64, 512, 256
    names = [str(k) for k in g[f"{variant}_keys"]]
    assert sorted(names) == ["W_dec", "W_enc"] + (["W_skip"] if d_out == d_in else []) + ["b_dec", "b_dec_out", "b_enc"]
    P = {n: g[f"{variant}_init_{n}"].copy() for n in names}
    assert P["W_dec"].shape == (d_sae, d_out)
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": g[f"{variant}_since0"].astype(np.float32).copy(), "act_freq_scores": np.zeros(d_sae, np.float32)}
    for t in range(3):
        out = O.train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=t), 8, lr=1e-3, step=t + 1,
                           target=synth_sae_batch(N, d_in, seed=100 + t)[:, :d_out].copy())
        loss, mse, l1, l0 = g[f"{variant}_s{t}_scalars"][:4]
        assert np.isnan(l1) and out["l1_loss"] is None
        assert abs(out["loss"] - loss) <= 1e-5 * abs(loss) and abs(out["mse_loss"] - mse) <= 1e-5 * abs(mse), (t, out, loss, mse)
        assert abs(out["l0"] - l0) <= 1e-6 * l0
        assert np.array_equal(stats["act_freq_scores"], g[f"{variant}_s{t}_act_freq"])
        assert np.array_equal(stats["n_fwd_since_fired"], g[f"{variant}_s{t}_n_since"])
    for n in P:
        assert rel_fro(P[n], g[f"{variant}_s2_param_{n}"]) < 1e-5, n


@pytest.mark.parametrize("variant,k", [("gated", None), ("gated_topk", 8)])
def test_gated_three_steps_vs_reference_fixture(variant, k):
    """The gated form of the oracle (ReLU; top-k: TopK on the magnitudes AND on the gate activations, no L1 term, sae.py:741-745,
    773-778) against the reference's own GatedSparseAutoencoder through its own train_step (gated / gated_topk of
    tests/golden/sae_variants_steps.npz): loss / mse / l1 / l0 / auxiliary loss, statistics, parameters after step 3 (b_enc, which the
    gated forward never uses, must come out untouched)."""
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, d_sae, N, l1c = 64, 512, 256, 2e-3
    names = [str(k_) for k_ in g[f"{variant}_keys"]]
    assert names == ["W_enc", "b_gate", "r_mag", "b_mag", "W_dec", "b_enc", "b_dec"]
    P = {n: g[f"{variant}_init_{n}"].copy() for n in names if n != "b_enc"}
    opt = {"m": {k_: np.zeros_like(v) for k_, v in P.items()}, "v": {k_: np.zeros_like(v) for k_, v in P.items()}}
    stats = {"n_fwd_since_fired": g[f"{variant}_since0"].astype(np.float32).copy(), "act_freq_scores": np.zeros(d_sae, np.float32)}
    for t in range(3):
        out = O.gated_train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=t), lr=1e-3, step=t + 1, l1_coefficient=l1c, k=k)
        loss, mse, l1, l0, _, aux = g[f"{variant}_s{t}_scalars"]
        for got, want in ((out["loss"], loss), (out["mse_loss"], mse), (out["l1_loss"], l1), (out["aux_loss"], aux)):
            assert abs(got - want) <= 1e-5 * abs(want), (t, out, g[f"{variant}_s{t}_scalars"])
        assert abs(out["l0"] - l0) <= 1e-6 * l0
        assert np.array_equal(stats["act_freq_scores"], g[f"{variant}_s{t}_act_freq"])
        assert np.array_equal(stats["n_fwd_since_fired"], g[f"{variant}_s{t}_n_since"])
    for n in P:
        assert rel_fro(P[n], g[f"{variant}_s2_param_{n}"]) < 1e-5, n
    assert np.array_equal(g[f"{variant}_s2_param_b_enc"], g[f"{variant}_init_b_enc"])


TAIL = {"relu_constnorm": dict(k=None, layer_norm="constant_norm_rescale", l1_coefficient=2e-3),
        "topk_constnorm": dict(k=8, layer_norm="constant_norm_rescale"),
        "tanh_relu": dict(k=None, l1_coefficient=2e-3, act="tanh-relu"),
        "relu_lp2": dict(k=None, l1_coefficient=2e-3, lp_norm=2.0)}


@pytest.mark.parametrize("variant", sorted(TAIL))
def test_tail_variants_three_steps_vs_reference_fixture(variant):
    """Round 6, SURVEY.md 8(f) row 3's tail: normalize_activations = "constant_norm_rescale" (sae.py:60-72; ReLU + L1 and top-k), the
    "tanh-relu" activation (:823-830) and lp_norm = 2 (:617) in the oracle against what the reference's own StandardSparseAutoencoder
    produced through its own train_step (tests/golden/gen_golden_sae_tail.py): scalars, statistics, parameters after step 3."""
    g = np.load(os.path.join(GOLDEN, "sae_tail_steps.npz"))
    kw = dict(TAIL[variant])
    k = kw.pop("k")
    d_in, d_sae, N = 64, 512, 256
    P = {n: g[f"{variant}_init_{n}"].copy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": g[f"{variant}_since0"].astype(np.float32).copy(), "act_freq_scores": np.zeros(d_sae, np.float32)}
    for t in range(3):
        out = O.train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=t), k, lr=1e-3, step=t + 1, **kw)
        loss, mse, l1, l0 = g[f"{variant}_s{t}_scalars"][:4]
        assert abs(out["loss"] - loss) <= 1e-5 * abs(loss) and abs(out["mse_loss"] - mse) <= 1e-5 * abs(mse), (t, out, loss, mse)
        if k is None:
            assert abs(out["l1_loss"] - l1) <= 1e-5 * abs(l1), (t, out, l1)
        assert abs(out["l0"] - l0) <= 1e-6 * l0
        assert np.array_equal(stats["act_freq_scores"], g[f"{variant}_s{t}_act_freq"])
        assert np.array_equal(stats["n_fwd_since_fired"], g[f"{variant}_s{t}_n_since"])
    for n in P:
        assert rel_fro(P[n], g[f"{variant}_s2_param_{n}"]) < 1e-5, n


@pytest.mark.parametrize("variant,k", [("tc_topk_ghost", 8), ("tc_relu_ghost", None)])
def test_transcoder_ghost_three_steps_vs_reference_fixture(variant, k):
    """Round 6: ghost gradients on a Transcoder -- the ghost term sees the INPUT activation (residual x - sae_out, rescaled by
    _compute_mse_loss(x, sae_out): transcoder.py:82-86 + sae.py:151-179) while the loss reconstructs the target -- in the oracle against
    the reference's own Transcoder through its own train_step (tests/golden/sae_tail_steps.npz): top-k with the skip connection, ReLU + L1
    without; a third of the features dead from the start."""
    g = np.load(os.path.join(GOLDEN, "sae_tail_steps.npz"))
    d_in, d_sae, N = 64, 512, 256
    names = [str(n) for n in g[f"{variant}_keys"]]
    P = {n: g[f"{variant}_init_{n}"].copy() for n in names}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": g[f"{variant}_since0"].astype(np.float32).copy(), "act_freq_scores": np.zeros(d_sae, np.float32)}
    assert (stats["n_fwd_since_fired"] > 1).sum() > 100
    for t in range(3):
        out = O.train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=t), k, lr=1e-3, step=t + 1, dead_feature_window=1,
                           l1_coefficient=0.0 if k else 2e-3, target=synth_sae_batch(N, d_in, seed=100 + t))
        loss, mse, l1, l0, ghost = g[f"{variant}_s{t}_scalars"][:5]
        assert abs(out["loss"] - loss) <= 2e-5 * abs(loss) and abs(out["mse_loss"] - mse) <= 1e-5 * abs(mse), (t, out, loss, mse)
        assert abs(out["ghost_loss"] - ghost) <= 2e-5 * abs(ghost), (t, out, ghost)
        assert abs(out["l0"] - l0) <= 1e-6 * l0
        assert np.array_equal(stats["act_freq_scores"], g[f"{variant}_s{t}_act_freq"])
        assert np.array_equal(stats["n_fwd_since_fired"], g[f"{variant}_s{t}_n_since"])
    for n in P:
        # (a dead feature's only gradient is the ghost term: ~1e-9 entries Adam turns into lr-sized steps either way -- see the top-k ghost test)
        assert rel_fro(P[n], g[f"{variant}_s2_param_{n}"]) < (1e-3 if n == "W_enc" else 2e-4), (n, rel_fro(P[n], g[f"{variant}_s2_param_{n}"]))
