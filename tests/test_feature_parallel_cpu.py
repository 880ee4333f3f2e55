"""The feature-parallel (tensor-parallel) top-k SAE step of vit_prisma_amd/sae/feature_parallel.py under gloo, world 2 and 4,
around the CPU twin of the shard engine (tests/_cpu_engine.py): candidates all-gathered, global top-k, partial
reconstructions all-reduced, shard-local backward / clip / project / Adam -- must land on the single-process oracle's
parameters, losses, l0 and firing counts (to fp32 summation order: the reconstruction is a sum of per-rank partial sums)."""
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


def _worker(rank, world, port, q, harvest_per_rank):
    import torch.distributed as dist
    from vit_prisma_amd.sae.feature_parallel import FeatureParallelSAE, shard_range
    from _cpu_engine import OracleShardEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = {n: torch.from_numpy(v.copy()) for n, v in synth_sae_state(D_IN, D_SAE, 0).items()}
    fp = FeatureParallelSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], K,
                            lambda We, Wd, be, bd: OracleShardEngine(We, Wd, be, bd, K, N), dist=dist, rank=rank, world=world)
    assert (fp.lo, fp.hi) == shard_range(D_SAE, rank, world) and fp.engine.d_sae == D_SAE // world
    losses, fires = [], []
    for t in range(STEPS):
        x = torch.from_numpy(synth_sae_batch(N, D_IN, seed=t))
        if harvest_per_rank:                                         # every rank brings its own tokens: all-gather first
            per = N // world
            x = fp.gather_tokens(x[rank * per:(rank + 1) * per].contiguous())
            assert tuple(x.shape) == (N, D_IN)
        loss, l0 = fp.step(x, lr=1e-3, max_grad_norm=1.0)
        losses.append((float(loss), float(l0)))
        fires.append(fp.fire_count.numpy().copy())
    P = fp.gather_parameters()
    q.put((rank, {n: v.numpy().copy() for n, v in P.items()}, losses, fires))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,harvest_per_rank", [(2, False), (4, True)])
def test_feature_parallel_step_equals_single_process_oracle(world, harvest_per_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, harvest_per_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    P = {kk: v.copy() for kk, v in synth_sae_state(D_IN, D_SAE, 0).items()}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(D_SAE, np.float32), "act_freq_scores": np.zeros(D_SAE, np.float32)}
    ref_losses, ref_fires = [], []
    for t in range(STEPS):
        before = stats["act_freq_scores"].copy()
        ref = O.train_step(P, opt, stats, synth_sae_batch(N, D_IN, seed=t), K, lr=1e-3, step=t + 1)
        ref_losses.append((ref["loss"], ref["l0"]))
        ref_fires.append(stats["act_freq_scores"] - before)
    for rank, params, losses, fires in got:
        for n in P:
            assert rel_fro(params[n], P[n]) < 1e-5, (rank, n)
            assert np.array_equal(params[n], got[0][1][n]), (rank, n)        # every rank gathers the same parameters
        for (l, l0), (rl, rl0) in zip(losses, ref_losses):
            assert abs(l - rl) <= 1e-5 * abs(rl) and abs(l0 - rl0) < 1e-6, (rank, l, rl, l0, rl0)
        for f, rf in zip(fires, ref_fires):
            assert np.array_equal(f, rf), rank


def test_global_topk_breaks_ties_by_feature_index():
    """Two pretend ranks, equal candidate values across shards -- the lower GLOBAL feature index wins, as torch.topk on the
    dense row (and the oracle) would have it.  (The host statement of pv_sae_tp_merge; the kernel itself is held to it in
    tests/test_native_sae_gpu.py.)"""
    from _cpu_engine import OracleShardEngine
    eng = OracleShardEngine.__new__(OracleShardEngine)
    eng.k, eng.d_sae = 2, 8
    vals = torch.tensor([[[5.0, 1.0]], [[5.0, 3.0]]])                            # [W, n, k]
    idxs = torch.tensor([[[7, 2]], [[1, 0]]], dtype=torch.int32)                 # local indices; rank 1's are global 9, 8
    gathered = torch.stack([vals.view(torch.int32), idxs], dim=1).contiguous()   # [W, 2, n, k]
    # candidates: (5.0, g7) (1.0, g2) | (5.0, g9) (3.0, g8): top-2 = the two 5.0s (g7 before g9)
    assert eng.tp_merge(gathered, 2, 0, 1).tolist() == [[5.0, 0.0]]
    assert eng.tp_merge(gathered, 2, 1, 1).tolist() == [[5.0, 0.0]]
    vals[1, 0, 1] = 5.0                                                          # now (5.0, g8) ties too: g7, g8 win, g9 loses
    gathered = torch.stack([vals.view(torch.int32), idxs], dim=1).contiguous()
    assert eng.tp_merge(gathered, 2, 0, 1).tolist() == [[5.0, 0.0]]
    assert eng.tp_merge(gathered, 2, 1, 1).tolist() == [[0.0, 5.0]]


def test_simulated_world_equals_the_process_group_step():
    """simulate_step (all ranks of a world in one process, exchanges by hand) lands on the oracle like the gloo run does."""
    from vit_prisma_amd.sae.feature_parallel import FeatureParallelSAE, gather_parameters_local, simulate_step
    from _cpu_engine import OracleShardEngine
    world = 4
    T = {n: torch.from_numpy(v.copy()) for n, v in synth_sae_state(D_IN, D_SAE, 0).items()}
    ranks = [FeatureParallelSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], K,
                                lambda We, Wd, be, bd: OracleShardEngine(We, Wd, be, bd, K, N), rank=r, world=world) for r in range(world)]
    P = {kk: v.copy() for kk, v in synth_sae_state(D_IN, D_SAE, 0).items()}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(D_SAE, np.float32), "act_freq_scores": np.zeros(D_SAE, np.float32)}
    for t in range(STEPS):
        x = synth_sae_batch(N, D_IN, seed=t)
        before = stats["act_freq_scores"].copy()
        ref = O.train_step(P, opt, stats, x, K, lr=1e-3, step=t + 1)
        loss, l0 = simulate_step(ranks, torch.from_numpy(x), lr=1e-3, max_grad_norm=1.0)
        assert abs(float(loss) - ref["loss"]) <= 1e-5 * abs(ref["loss"]) and abs(float(l0) - ref["l0"]) < 1e-6
        for fp in ranks:
            assert np.array_equal(fp.fire_count.numpy(), stats["act_freq_scores"] - before)
    got = gather_parameters_local(ranks)
    for n in P:
        assert rel_fro(got[n].numpy(), P[n]) < 1e-5, n


def _trainer_worker(rank, world, port, q, drop=False):
    """The same step through VisionSAETrainer.train_step (use_feature_parallel): tokens harvested per rank, statistics,
    sync_parameters() gathering the shards back into the module."""
    import torch.distributed as dist
    from vit_prisma_amd.sae import StandardSparseAutoencoder, VisionModelSAERunnerConfig, VisionSAETrainer
    from _cpu_engine import OracleShardEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN, activation_fn_str="topk",
        activation_fn_kwargs={"k": K}, normalize_activations="layer_norm", b_dec_init_method="mean", train_batch_size=N,
        lr=1e-3, max_grad_norm=1.0, _device="cpu", log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0, seed=7 + rank)
    sae = StandardSparseAutoencoder(cfg)
    if rank == 0:
        with torch.no_grad():
            for n, v in synth_sae_state(D_IN, D_SAE, 0).items():
                getattr(sae, n).copy_(torch.from_numpy(v))
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae).use_feature_parallel(True, drop_replicas=drop)
    tr._native_ok = lambda *a, **k: True
    tr._make_shard_engine = lambda s, max_tokens: (lambda We, Wd, be, bd: OracleShardEngine(We, Wd, be, bd, K, max_tokens))
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
        # drop_replicas: the module's matrices are released while the shards train ...
        assert (sae.W_enc.numel() == 0 and sae.W_dec.numel() == 0) if drop else sae.W_enc.shape == (D_IN, D_SAE)
        if drop and t == 0:
            tr.sync_parameters()                                # ... come back complete on request ...
            assert sae.W_enc.shape == (D_IN, D_SAE) and sae.W_dec.shape == (D_SAE, D_IN)
    tr.sync_parameters()                                        # ... (and are released again by the step after)
    out = {n: getattr(sae, n).detach().numpy().copy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
    q.put((rank, out, losses, act.numpy().copy(), since.numpy().copy(), frac))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("drop", [False, True])
def test_trainer_feature_parallel_world2_equals_single_process_oracle(drop):
    """drop: use_feature_parallel(True, drop_replicas=True) -- the full-size W_enc / W_dec of the module are released on every rank
    while the shards train and come back, complete and equal to the single-process run's, on sync_parameters()."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q, drop)) for r in range(world)]
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
        for n in P:
            assert rel_fro(params[n], P[n]) < 1e-5, (rank, n)
        for (l, l0), (rl, rl0) in zip(losses, ref_losses):
            assert abs(l - rl) <= 1e-5 * abs(rl) and abs(l0 - rl0) < 1e-6
        assert np.array_equal(act, stats["act_freq_scores"]) and np.array_equal(since, stats["n_fwd_since_fired"])
        assert frac == STEPS * N


def test_auto_mode_takes_feature_parallel_only_where_its_kernels_can_run():
    """use_feature_parallel(None): feature parallel needs world <= 8 (pv_sae_tp_merge), a shard of at least k features and a whole
    number of 4-feature groups per shard; everywhere else the token-sharded step (no such limits) must be chosen, and
    use_feature_parallel(True) keeps asking for the feature-parallel one."""
    from vit_prisma_amd.sae.config import VisionModelSAERunnerConfig
    from vit_prisma_amd.sae.trainer import VisionSAETrainer

    def trainer(d_in, exp, k, world):
        cfg = VisionModelSAERunnerConfig(hook_point_layer=1, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp,
                                         activation_fn_str="topk", activation_fn_kwargs={"k": k}, _device="cpu", log_to_wandb=False,
                                         n_checkpoints=0)
        tr = VisionSAETrainer(cfg, model=None, dataset=None)
        tr.world = world
        return tr

    tr = trainer(64, 8, 8, 2)
    assert tr._use_tp(tr.sparse_coder)                                       # 512 features over 2 ranks: fine
    assert trainer(64, 8, 8, 8)._use_tp(trainer(64, 8, 8, 8).sparse_coder)
    for d_in, exp, k, world in ((64, 8, 8, 16),        # two nodes: the merge kernel ranks the candidates of at most 8 ranks
                                (64, 1, 32, 4),         # a shard of 16 features cannot hold k = 32 candidates
                                (24, 1, 4, 2),          # a 12-feature shard ... is fine (multiple of 4)
                                (28, 1, 2, 2),          # a 14-feature shard is not a whole number of 4-feature groups
                                (64, 8, 8, 3)):         # 512 does not divide by 3
        tr = trainer(d_in, exp, k, world)
        want = (d_in, exp, k, world) == (24, 1, 4, 2)
        assert tr._use_tp(tr.sparse_coder) == want, (d_in, exp, k, world)
        tr.use_feature_parallel(True)
        assert tr._use_tp(tr.sparse_coder)                                   # forced: the hard error stays with the kernels
        tr.use_feature_parallel(False)
        assert not tr._use_tp(tr.sparse_coder)
    assert not trainer(64, 8, 8, 1)._use_tp(trainer(64, 8, 8, 1).sparse_coder)
