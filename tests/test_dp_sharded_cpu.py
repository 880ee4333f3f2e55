"""The data-parallel SAE step with a feature-sharded optimizer on CPU: the trainer's own
``_native_dp_step`` -- global batch mean all-reduce, reduce-scatter of the gradient rows, one small bucket (gb_dec | fire
counts | loss, mse, l0), scalar all-reduce of the ranks' clip-norm terms, clip / project / Adam on the rank's rows only,
asynchronous all-gather of the parameter rows -- runs under gloo with world 2 and world 4 around a CPU engine built from
the oracle (tests/_cpu_engine.py), and must land on the single-process oracle's parameters, losses and statistics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import sae_oracle as O
from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state

from conftest import rel_fro

D_IN, D_SAE, K, N, STEPS = 32, 256, 4, 128, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, b_dec_init):
    import torch.distributed as dist
    from vit_prisma_amd.sae import StandardSparseAutoencoder, VisionModelSAERunnerConfig, VisionSAETrainer
    from _cpu_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                                    # ranks START with different parameters on purpose
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN, activation_fn_str="topk",
        activation_fn_kwargs={"k": K}, normalize_activations="layer_norm", b_dec_init_method=b_dec_init, train_batch_size=N,
        lr=1e-3, max_grad_norm=1.0, _device="cpu", log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0, seed=7 + rank)
    sae = StandardSparseAutoencoder(cfg)
    if rank == 0:
        with torch.no_grad():
            for n, v in synth_sae_state(D_IN, D_SAE, 0).items():
                getattr(sae, n).copy_(torch.from_numpy(v))
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae).use_feature_parallel(False)     # the data-parallel mode
    assert tr.world == world and tr._shard(D_SAE) == (rank * D_SAE // world, (rank + 1) * D_SAE // world)
    # run the trainer's native branch on the CPU twin of the engine
    tr._native_ok = lambda *a, **k: True
    real_get = VisionSAETrainer._get_engine

    def get_engine(s, n_tokens):
        if tr._engine is None:
            for p in s.parameters():                                  # what _get_engine does first under DP
                dist.broadcast(p.data, src=0)
            tr._engine = OracleEngine(s, K, n_tokens)
        return tr._engine

    tr._get_engine = get_engine
    act, since, frac, opt, sched = tr.initialize_training_variables()
    losses = []
    for t in range(STEPS):
        x = torch.from_numpy(synth_sae_batch(N, D_IN, seed=t))
        xs = x[rank * (N // world):(rank + 1) * (N // world)][:, None, :].contiguous()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=xs, n_training_steps=t, n_training_tokens=t * N)
        losses.append((float(loss), float(l0)))
        assert tr.last_step_native
    assert len(tr._pending) == 3                                     # W_enc^T, W_dec, b_enc rows still in flight
    tr.sync_parameters()
    assert not tr._pending
    out = {n: getattr(sae, n).detach().numpy().copy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
    q.put((rank, out, losses, act.numpy().copy(), since.numpy().copy(), frac))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_step_equals_single_process_oracle(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "mean")) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    P = {kk: v.copy() for kk, v in synth_sae_state(D_IN, D_SAE, 0).items()}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(D_SAE, np.float32), "act_freq_scores": np.zeros(D_SAE, np.float32)}
    ref_losses = []
    for t in range(STEPS):
        ref = O.train_step(P, opt, stats, synth_sae_batch(N, D_IN, seed=t), K, lr=1e-3, step=t + 1)
        ref_losses.append((ref["loss"], ref["l0"]))
    for rank, params, losses, act, since, frac in got:
        for n in P:                                                   # every rank holds the complete, identical parameters
            assert rel_fro(params[n], P[n]) < 1e-5, (rank, n)
            assert np.array_equal(params[n], got[0][1][n]), (rank, n)
        for (l, l0), (rl, rl0) in zip(losses, ref_losses):
            assert abs(l - rl) <= 1e-5 * abs(rl) and abs(l0 - rl0) < 1e-6
        assert np.array_equal(act, stats["act_freq_scores"]) and np.array_equal(since, stats["n_fwd_since_fired"])
        assert frac == STEPS * N


# ---- ReLU + L1 on the dense step, data parallel (VisionSAETrainer._native_dense_step with a process group): tokens sharded, the
# global batch mean all-reduced, ONE all-reduce of the flat gradient buffer + one small bucket (fire counts | loss, mse, l0, -, l1),
# replicated optimizer
L1C = 2e-3


def _relu_worker(rank, world, port, q):
    import torch.distributed as dist
    from vit_prisma_amd.sae import StandardSparseAutoencoder, VisionModelSAERunnerConfig, VisionSAETrainer
    from _cpu_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN, activation_fn_str="relu",
        activation_fn_kwargs={}, l1_coefficient=L1C, normalize_activations="layer_norm", b_dec_init_method="mean", train_batch_size=N,
        lr=1e-3, max_grad_norm=1.0, _device="cpu", log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0, seed=7 + rank)
    sae = StandardSparseAutoencoder(cfg)
    if rank == 0:
        with torch.no_grad():
            for n, v in synth_sae_state(D_IN, D_SAE, 0).items():
                getattr(sae, n).copy_(torch.from_numpy(v))
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae)
    tr._native_kind = lambda *a, **k: "relu"                          # run the trainer's dense branch on the CPU twin

    def get_engine(s, n_tokens):
        if tr._engine is None:
            for p in s.parameters():
                dist.broadcast(p.data, src=0)
            tr._engine = OracleEngine(s, 1, n_tokens)
        return tr._engine

    tr._get_engine = get_engine
    act, since, frac, opt, sched = tr.initialize_training_variables()
    losses = []
    for t in range(STEPS):
        x = torch.from_numpy(synth_sae_batch(N, D_IN, seed=t))
        xs = x[rank * (N // world):(rank + 1) * (N // world)][:, None, :].contiguous()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=xs, n_training_steps=t, n_training_tokens=t * N)
        losses.append((float(loss), float(mse), float(l1), float(l0)))
        assert tr.last_step_native
    tr.sync_parameters()
    out = {n: getattr(sae, n).detach().numpy().copy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
    q.put((rank, out, losses, act.numpy().copy(), since.numpy().copy(), frac))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_relu_l1_data_parallel_step_equals_single_process_oracle(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_relu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = {k: v.copy() for k, v in synth_sae_state(D_IN, D_SAE, 0).items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(D_SAE, np.float32), "act_freq_scores": np.zeros(D_SAE, np.float32)}
    want = [O.train_step(P, opt, stats, synth_sae_batch(N, D_IN, seed=t), None, lr=1e-3, step=t + 1, l1_coefficient=L1C) for t in range(STEPS)]
    for rank, out, losses, act, since, frac in got:
        for t, (loss, mse, l1, l0) in enumerate(losses):
            assert abs(loss - want[t]["loss"]) <= 1e-5 * want[t]["loss"] and abs(mse - want[t]["mse_loss"]) <= 1e-5 * want[t]["mse_loss"]
            assert abs(l1 - want[t]["l1_loss"]) <= 1e-5 * want[t]["l1_loss"] and abs(l0 - want[t]["l0"]) <= 1e-6 * want[t]["l0"]
        for n in P:
            assert rel_fro(out[n], P[n]) < 2e-5, (rank, n)
        assert np.array_equal(act, stats["act_freq_scores"]) and np.array_equal(since, stats["n_fwd_since_fired"])
        assert frac == N * STEPS


# ---- Gated SAE, data parallel: the same orchestration around pv_sae_gated_step (batch_mean / n_global)
def _gated_worker(rank, world, port, q, init, topk=False):
    import torch.distributed as dist
    from vit_prisma_amd.sae import GatedSparseAutoencoder, VisionModelSAERunnerConfig, VisionSAETrainer
    from _cpu_engine import OracleGatedEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN,
        activation_fn_str="topk" if topk else "relu", activation_fn_kwargs={"k": K} if topk else {}, l1_coefficient=L1C,
        architecture="gated", normalize_activations="layer_norm", b_dec_init_method="mean",
        train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cpu", log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0,
        seed=7 + rank)
    tr = VisionSAETrainer(cfg, model=None, dataset=None)
    sae = tr.sparse_coder
    assert type(sae) is GatedSparseAutoencoder
    if rank == 0:
        with torch.no_grad():
            for n, v in init.items():
                getattr(sae, n).copy_(torch.from_numpy(v))
    tr._native_kind = lambda *a, **k: "gated"

    def get_engine(s, n_tokens):
        if tr._engine is None:
            for p in s.parameters():
                dist.broadcast(p.data, src=0)
            tr._engine = OracleGatedEngine(s, n_tokens)
            tr._engine.gated_topk, tr._engine.k = topk, K
        return tr._engine

    tr._get_engine = get_engine
    act, since, frac, opt, sched = tr.initialize_training_variables()
    losses = []
    for t in range(STEPS):
        x = torch.from_numpy(synth_sae_batch(N, D_IN, seed=t))
        xs = x[rank * (N // world):(rank + 1) * (N // world)][:, None, :].contiguous()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=xs, n_training_steps=t, n_training_tokens=t * N)
        losses.append((float(loss), float(mse), float(l1), float(l0), float(tr._engine.scalars[6])))
        assert tr.last_step_native
    out = {n: p.detach().numpy().copy() for n, p in sae.named_parameters()}
    q.put((rank, out, losses, act.numpy().copy(), since.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("topk", [False, True])
def test_gated_data_parallel_step_equals_single_process_oracle(topk):
    """(topk: the top-k form of the gated SAE -- TopK on the magnitudes and on the gate activations, no L1 term -- through the same
    token-sharded orchestration)"""
    world = 2
    rs = np.random.RandomState(9)
    init = dict(synth_sae_state(D_IN, D_SAE, 0))
    for name, scale in (("b_gate", 0.05), ("r_mag", 0.2), ("b_mag", 0.05)):
        init[name] = (rs.standard_normal(D_SAE) * scale).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gated_worker, args=(r, world, port, q, init, topk)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = {k: v.copy() for k, v in init.items() if k != "b_enc"}
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(D_SAE, np.float32), "act_freq_scores": np.zeros(D_SAE, np.float32)}
    want = [O.gated_train_step(P, opt, stats, synth_sae_batch(N, D_IN, seed=t), lr=1e-3, step=t + 1, l1_coefficient=L1C, k=K if topk else None)
            for t in range(STEPS)]
    for rank, out, losses, act, since in got:
        for t, (loss, mse, l1, l0, aux) in enumerate(losses):
            for gotv, key in ((loss, "loss"), (mse, "mse_loss"), (l1, "l1_loss"), (aux, "aux_loss")):
                assert abs(gotv - want[t][key]) <= 1e-5 * abs(want[t][key]), (t, key, gotv, want[t][key])
            assert abs(l0 - want[t]["l0"]) <= 1e-6 * want[t]["l0"]
        for n in P:
            assert rel_fro(out[n], P[n]) < 2e-5, (rank, n)
        assert np.array_equal(out["b_enc"], init["b_enc"])
        assert np.array_equal(act, stats["act_freq_scores"]) and np.array_equal(since, stats["n_fwd_since_fired"])


# ---- top-k Transcoder, data parallel: VisionSAETrainer._native_dp_step's transcoder branch (the TARGET's global mean, one all-reduce
# of the whole flat gradient buffer -- b_dec_out and W_skip ride in it -- replicated optimizer)
def _tc_init():
    init = dict(synth_sae_state(D_IN, D_SAE, 0))
    rs = np.random.RandomState(5)
    init["b_dec_out"] = (rs.standard_normal(D_IN) * 0.05).astype(np.float32)
    init["W_skip"] = (rs.standard_normal((D_IN, D_IN)) / np.sqrt(D_IN) * 0.3).astype(np.float32)
    return init


def _tc_worker(rank, world, port, q):
    import torch.distributed as dist
    from vit_prisma_amd.sae import Transcoder, VisionModelSAERunnerConfig, VisionSAETrainer
    from _cpu_engine import OracleTranscoderEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN, activation_fn_str="topk",
        activation_fn_kwargs={"k": K}, normalize_activations="layer_norm", b_dec_init_method="mean", train_batch_size=N, lr=1e-3,
        max_grad_norm=1.0, _device="cpu", log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0, seed=7 + rank,
        is_transcoder=True, transcoder_with_skip_connection=True, d_out=D_IN, out_hook_point_layer=1)
    tr = VisionSAETrainer(cfg, model=None, dataset=None)
    sae = tr.sparse_coder
    assert type(sae) is Transcoder
    if rank == 0:
        with torch.no_grad():
            for n, v in _tc_init().items():
                getattr(sae, n).copy_(torch.from_numpy(v))
    tr._native_kind = lambda *a, **k: "topk"

    def get_engine(s, n_tokens):
        if tr._engine is None:
            for p in s.parameters():
                dist.broadcast(p.data, src=0)
            tr._engine = OracleTranscoderEngine(s, K, n_tokens)
        return tr._engine

    tr._get_engine = get_engine
    act, since, frac, opt, sched = tr.initialize_training_variables()
    losses = []
    h = N // world
    for t in range(STEPS):
        pair = torch.stack([torch.from_numpy(synth_sae_batch(N, D_IN, seed=t)), torch.from_numpy(synth_sae_batch(N, D_IN, seed=100 + t))],
                           dim=1)[rank * h:(rank + 1) * h].contiguous()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=pair, n_training_steps=t, n_training_tokens=t * N)
        losses.append((float(loss), float(mse), float(l0)))
        assert tr.last_step_native and l1 is None
    out = {n: p.detach().numpy().copy() for n, p in sae.named_parameters()}
    q.put((rank, out, losses, act.numpy().copy(), since.numpy().copy(), frac))
    dist.barrier()
    dist.destroy_process_group()


def test_topk_transcoder_data_parallel_step_equals_single_process_oracle():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = {k: v.copy() for k, v in _tc_init().items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(D_SAE, np.float32), "act_freq_scores": np.zeros(D_SAE, np.float32)}
    want = [O.train_step(P, opt, stats, synth_sae_batch(N, D_IN, seed=t), K, lr=1e-3, step=t + 1, target=synth_sae_batch(N, D_IN, seed=100 + t))
            for t in range(STEPS)]
    for rank, out, losses, act, since, frac in got:
        for t, (loss, mse, l0) in enumerate(losses):
            assert abs(loss - want[t]["loss"]) <= 1e-5 * want[t]["loss"] and abs(mse - want[t]["mse_loss"]) <= 1e-5 * want[t]["mse_loss"]
            assert abs(l0 - want[t]["l0"]) <= 1e-6 * want[t]["l0"]
        for n in P:
            assert rel_fro(out[n], P[n]) < 2e-5, (rank, n)
            assert np.array_equal(out[n], got[0][1][n]), (rank, n)          # replicas stay identical
        assert np.array_equal(act, stats["act_freq_scores"]) and np.array_equal(since, stats["n_fwd_since_fired"])
        assert frac == N * STEPS


# ---- ghost gradients with a process group (use_ghost_grads; sae.py:151-179): the ghost term's two batch-wide quantities -- the residual's
# column mean (:156) and the mse loss (:172) -- are exchanged between the step and the ghost term (top-k) / after a pass without it
# (dense ReLU step): VisionSAETrainer._native_dp_step / _native_dense_step at world 2 against the single-process oracle
def _ghost_worker(rank, world, port, q, relu):
    import torch.distributed as dist
    from vit_prisma_amd.sae import StandardSparseAutoencoder, VisionModelSAERunnerConfig, VisionSAETrainer
    from _cpu_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN,
        activation_fn_str="relu" if relu else "topk", activation_fn_kwargs={} if relu else {"k": K}, l1_coefficient=L1C,
        normalize_activations="layer_norm", b_dec_init_method="mean", train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cpu",
        log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0, seed=7 + rank, use_ghost_grads=True, dead_feature_window=1)
    sae = StandardSparseAutoencoder(cfg)
    if rank == 0:
        with torch.no_grad():
            for n, v in synth_sae_state(D_IN, D_SAE, 0).items():
                getattr(sae, n).copy_(torch.from_numpy(v))
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae).use_feature_parallel(False)
    tr._native_kind = lambda *a, **k: "relu" if relu else "topk"

    def get_engine(s, n_tokens):
        if tr._engine is None:
            for p in s.parameters():
                dist.broadcast(p.data, src=0)
            tr._engine = OracleEngine(s, K, n_tokens)
        return tr._engine

    tr._get_engine = get_engine
    act, since, frac, opt, sched = tr.initialize_training_variables()
    since[::3] = 5.0                                              # a third of the features count as dead (window 1)
    losses = []
    for t in range(STEPS):
        x = torch.from_numpy(synth_sae_batch(N, D_IN, seed=t))
        xs = x[rank * (N // world):(rank + 1) * (N // world)][:, None, :].contiguous()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=xs, n_training_steps=t, n_training_tokens=t * N)
        losses.append((float(loss), float(mse), float(l0)))
        assert tr.last_step_native
    tr.sync_parameters()
    out = {n: getattr(sae, n).detach().numpy().copy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
    q.put((rank, out, losses, act.numpy().copy(), since.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("relu", [False, True])
def test_ghost_gradients_data_parallel_equal_single_process_oracle(relu):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ghost_worker, args=(r, world, port, q, relu)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = {k: v.copy() for k, v in synth_sae_state(D_IN, D_SAE, 0).items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(D_SAE, np.float32), "act_freq_scores": np.zeros(D_SAE, np.float32)}
    stats["n_fwd_since_fired"][::3] = 5.0
    want = [O.train_step(P, opt, stats, synth_sae_batch(N, D_IN, seed=t), None if relu else K, lr=1e-3, step=t + 1,
                         l1_coefficient=L1C if relu else 0.0, dead_feature_window=1) for t in range(STEPS)]
    assert all(w["ghost_loss"] is not None and w["ghost_loss"] > 0 for w in want)
    for rank, out, losses, act, since in got:
        for t, (loss, mse, l0) in enumerate(losses):
            assert abs(loss - want[t]["loss"]) <= 1e-5 * want[t]["loss"] and abs(mse - want[t]["mse_loss"]) <= 1e-5 * want[t]["mse_loss"], (t, loss, want[t])
            assert abs(l0 - want[t]["l0"]) <= 1e-6 * want[t]["l0"]
        for n in P:
            assert rel_fro(out[n], P[n]) < 1e-4, (rank, n, rel_fro(out[n], P[n]))
            assert np.array_equal(out[n], got[0][1][n]), (rank, n)          # replicas stay identical
        assert np.array_equal(act, stats["act_freq_scores"]) and np.array_equal(since, stats["n_fwd_since_fired"])
