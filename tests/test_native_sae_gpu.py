"""GPU parity tests of the native SAE step (through the C ABI) against the numpy oracle and the
reference-generated golden fixtures.  Tolerance: 1e-4 relative on losses (north_star) and on every
gradient / parameter tensor (measured ~1e-6); top-k index SETS and firing statistics exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O
from oracle.vit_oracle import fingerprint
from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.sae import (StandardSparseAutoencoder, VisionActivationsStore, VisionModelSAERunnerConfig,
                                VisionSAETrainer)
from vit_prisma_amd.sae.native_sae import NativeSAE
from vit_prisma_amd.synth import ARCHS, synth_images, synth_sae_batch, synth_sae_state, synth_vit_state

from conftest import GOLDEN, rel_fro

pytestmark = pytest.mark.gpu
TOL = 1e-4
# Parameters after SEVERAL ghost steps against the reference's fixture: held to GHOST_RUN_X x the oracle's own fp32-vs-float64 distance over
# the same steps (run_param_floor).  Measured on the GPU: 32 x on the 64 -> 512 ReLU fixture (3.5e-4 against a floor of 1.1e-5 -- the two
# CPU computations share their BLAS summation orders and sit closer to each other than two independent fp32 computations do; every loss of
# those steps agrees to 1e-4), 1.1 x on the top-k one.  Round 5 had the constant 1e-3 here.
GHOST_RUN_X = 64.0


def fw_scalars(fw):
    """The scalars O.train_step returns, read off a forward on the renormed copy (the step itself is taken later, once the kernel's
    ReLU gates are known: see ``gate`` of O.train_step)."""
    opt = lambda key: None if fw.get(key) is None else float(fw[key])
    return dict(loss=float(fw["loss"]), mse_loss=float(fw["mse_loss"]), l0=float(fw["l0"]), l1_loss=opt("l1_loss"), ghost_loss=opt("ghost_loss"))


def truncated(reason):
    """A comparison that cannot go on entry for entry (a near-tie fell differently in two correct fp32 computations) ends HERE, visibly:
    the parametrization is reported as xfailed, not passed, and named in gpurun_out/truncated_tests.txt (profiles/r06_truncated_tests.txt
    is that file from the round's GPU run).  Round 5 ended such runs with a bare ``return``."""
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "truncated_tests.txt"), "a") as f:
        f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?')}: {reason}\n")
    pytest.xfail(reason)


def fresh(d_in, d_sae):
    sd = synth_sae_state(d_in, d_sae, 0)
    P = {k: v.copy() for k, v in sd.items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
    T = {k: torch.from_numpy(v.copy()).cuda() for k, v in sd.items()}
    return P, opt, stats, T


# (768, 49152): the x64 CLIP-B/32 SAEs of the reference's docs/sae_table.md (d_sae > 32768: two blocks of the CSR scan, 192
# candidate tiles, 3072 sampled values per token); (512, 65536): the plan's upper bound
@pytest.mark.parametrize("d_in,d_sae,k,n", [(64, 512, 8, 256), (96, 1024, 16, 300), (768, 24576, 32, 4096),
                                            (768, 49152, 32, 1024), (768, 49152, 64, 512), (512, 65536, 32, 300),
                                            (1280, 20480, 32, 1024), (1156, 4624, 16, 300),       # (d_in up to 1280: ViT-H/14's width; ragged beyond 1024)
                                            (768, 8192, 128, 512), (256, 4096, 256, 300), (64, 512, 100, 256)])      # (k > 64: the exact encoder + streaming top-k)
def test_native_step_vs_oracle(d_in, d_sae, k, n):
    P, opt, stats, T = fresh(d_in, d_sae)
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
    for t in range(2):
        x = synth_sae_batch(n, d_in, seed=t)
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        fw = O.sae_forward(Pc, x, k)
        # a token whose k-th and (k+1)-th largest pre-activations lie within fp32 summation noise of each other may keep either (more
        # likely the deeper k reaches into the bulk: k = 128 / 256): picked by the oracle alone, replaced by a safe token
        top = -np.partition(-fw["hidden_pre"], k, axis=1)[:, :k + 1]
        risky = (top[:, :k].min(axis=1) - top[:, k]) < 1e-5 * np.abs(fw["hidden_pre"]).max()
        if risky.any():
            x[risky] = x[np.flatnonzero(~risky)[0]]
            fw = O.sae_forward(Pc, x, k)
        gr = O.sae_backward(Pc, x, fw)
        ref = O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1)
        eng.renorm_decoder()
        eng.step(torch.from_numpy(x).cuda(), want_out=True)
        eng.grad_sqnorm()
        torch.cuda.synchronize()
        sc = eng.scalars.cpu().numpy()
        assert abs(sc[0] - ref["loss"]) <= TOL * ref["loss"] and abs(sc[1] - ref["mse_loss"]) <= TOL * ref["mse_loss"]
        assert sc[2] == ref["l0"] == float(k)
        assert abs(np.sqrt(sc[3]) - ref["grad_norm"]) <= TOL * ref["grad_norm"]
        idx = np.sort(eng.topk_idx[:n].cpu().numpy(), axis=1)
        assert np.array_equal(idx, np.sort(fw["idx"], axis=1))                       # exact index sets
        assert rel_fro(eng.sae_out[:n].cpu().numpy(), fw["sae_out"]) < TOL
        assert rel_fro(eng.grad_W_enc().cpu().numpy(), gr["W_enc"]) < TOL
        for name in ("W_dec", "b_enc", "b_dec"):
            assert rel_fro(eng.g[name].cpu().numpy(), gr[name]) < TOL, name
        eng.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for name in P:
            assert rel_fro(eng.params[name].cpu().numpy(), P[name]) < TOL, name
            assert rel_fro(eng.m[name].cpu().numpy(), opt["m"][name]) < TOL, name
            assert rel_fro(eng.v[name].cpu().numpy(), opt["v"][name]) < 2 * TOL, name
        assert np.array_equal(eng.act_freq_scores.cpu().numpy(), stats["act_freq_scores"])
        assert np.array_equal(eng.n_fwd_since_fired.cpu().numpy(), stats["n_fwd_since_fired"])


def test_native_config3_vs_reference_golden():
    """BASELINE config 3 (768 -> 24576, k = 32, N = 4096): losses / parameters of the reference's own
    VisionSAETrainer.train_step, 3 consecutive steps."""
    with open(os.path.join(GOLDEN, "sae_b32_steps.json")) as f:
        G = json.load(f)
    c = G["config"]
    _, _, _, T = fresh(c["d_in"], c["d_sae"])
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], c["k"], True, c["n_tokens"])
    for t, want in enumerate(G["steps"]):
        eng.train_step(torch.from_numpy(synth_sae_batch(c["n_tokens"], c["d_in"], seed=t)).cuda(), c["lr"], 1.0)
        torch.cuda.synchronize()
        sc = eng.scalars.cpu().numpy()
        assert abs(sc[0] - want["loss"]) <= TOL * want["loss"], (t, sc[0], want["loss"])
        assert sc[2] == want["l0"]
        # the reference's clip norm is itself only good to ~1e-3 (torch CPU vector_norm, see the oracle test)
        assert abs(np.sqrt(sc[3]) - want["grad_norm"]) <= 2e-3 * want["grad_norm"]
        for name in ("W_enc", "W_dec", "b_enc", "b_dec"):
            fp = fingerprint(eng.params[name].cpu().numpy())
            assert abs(fp["l2"] - want["params"][name]["l2"]) <= 1e-5 * want["params"][name]["l2"], (t, name)
            assert np.max(np.abs(np.array(fp["vals"]) - np.array(want["params"][name]["vals"]))) < 1e-4, (t, name)
        assert abs(float(eng.act_freq_scores.sum()) - want["act_freq"]["sum"]) < 0.5


def test_topk_edge_cases_ties_and_negatives():
    """Rows with massive ties (fallback radix path), all-negative rows (ReLU zeroes the kept values) and
    k larger than the positives."""
    d_in, d_sae, k, n = 64, 512, 8, 8
    _, _, _, T = fresh(d_in, d_sae)
    T["W_enc"].zero_()                                  # hidden_pre == b_enc for every token: 512-way ties per value
    T["b_enc"].copy_(torch.linspace(-1.0, 0.5, d_sae).cuda().round(decimals=1))
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
    idx, val, mu, sd = eng.encode_topk(torch.from_numpy(synth_sae_batch(n, d_in, 0)).cuda())
    torch.cuda.synchronize()
    b = T["b_enc"].cpu().numpy()
    kth = np.sort(b)[-k]
    for r in range(n):
        sel = idx[r].cpu().numpy()
        assert len(set(sel.tolist())) == k                                     # k distinct indices
        assert np.all(b[sel] >= kth) and np.allclose(val[r].cpu().numpy(), np.maximum(b[sel], 0))
    T["b_enc"].fill_(-1.0)                                                     # every pre-activation negative
    idx, val, _, _ = eng.encode_topk(torch.from_numpy(synth_sae_batch(n, d_in, 0)).cuda())
    assert float(val.abs().max()) == 0.0 and len(set(idx[0].cpu().tolist())) == k
    # > 1024 candidates (4096-way tie): the radix-select fallback (which of the tied columns are kept is
    # arbitrary, as in torch.topk, but deterministic)
    _, _, _, T2 = fresh(d_in, 4096)
    T2["W_enc"].zero_()
    T2["b_enc"].fill_(0.25)
    T2["b_enc"][100] = 0.75
    eng2 = NativeSAE(T2["W_enc"], T2["W_dec"], T2["b_enc"], T2["b_dec"], k, True, n)
    idx, val, _, _ = eng2.encode_topk(torch.from_numpy(synth_sae_batch(n, d_in, 0)).cuda())
    torch.cuda.synchronize()
    first = idx.clone()
    sel = idx[3].cpu().tolist()
    assert 100 in sel and len(set(sel)) == k and all(0 <= c < 4096 for c in sel)
    assert sorted(val[3].cpu().tolist()) == [0.25] * 7 + [0.75]
    idx_b, _, _, _ = eng2.encode_topk(torch.from_numpy(synth_sae_batch(n, d_in, 0)).cuda())
    assert torch.equal(idx_b, first)                                            # run-to-run deterministic


def test_trainer_native_path_matches_reference_and_store_harvests_natively():
    g = np.load(os.path.join(GOLDEN, "sae_small_steps.npz"))
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=64, expansion_factor=8, activation_fn_str="topk",
        activation_fn_kwargs={"k": 8}, normalize_activations="layer_norm", b_dec_init_method="mean",
        train_batch_size=256, lr=1e-3, max_grad_norm=1.0, _device="cuda", log_to_wandb=False,
        lr_scheduler_name="constant", n_checkpoints=0, context_size=17, store_batch_size=4, n_batches_in_buffer=4)
    sae = StandardSparseAutoencoder(cfg)
    with torch.no_grad():
        for n, v in synth_sae_state(64, 512, 0).items():
            getattr(sae, n).copy_(torch.from_numpy(v))
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae)
    act, since, frac, opt, sched = tr.initialize_training_variables()
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(256, 64, seed=t)).cuda()[:, None, :]
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act,
            n_forward_passes_since_fired=since, n_frac_active_tokens=frac, layer_acts=x, n_training_steps=t,
            n_training_tokens=t * 256)
        assert tr.last_step_native and l1 is None
        want = g[f"s{t}_scalars"]
        assert abs(float(loss) - want[0]) <= TOL * want[0] and float(l0) == want[2]
        for n in ("W_enc", "W_dec", "b_enc", "b_dec"):
            assert rel_fro(getattr(sae, n).detach().cpu().numpy(), g[f"s{t}_param_{n}"]) < TOL, (t, n)
        assert np.array_equal(act.cpu().numpy(), g[f"s{t}_act_freq"])
        assert np.array_equal(since.cpu().numpy(), g[f"s{t}_n_since"])

    # the store: harvest blocks.1.hook_resid_post of the tiny ViT through the native plan
    arch = ARCHS["tiny"]
    vit = HookedViT(HookedViTConfig(**arch, device="cuda"))
    vit.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()})
    vit = vit.cuda().eval()
    images = torch.from_numpy(synth_images(arch, 32, 3))
    ds = [(images[i], 0) for i in range(32)]
    store = VisionActivationsStore(cfg, vit, ds)
    assert vit.last_run_native
    assert store.storage_buffer.shape == ((16 + 8) * 17 // 2, 1, 64)          # kept half of the mixed buffer
    batch = store.next_batch()
    assert batch.shape[1:] == (1, 64) and batch.is_cuda
    acts = store.get_activations(images[:4].cuda())
    from oracle.vit_oracle import vit_forward
    _, c = vit_forward(synth_vit_state(arch, 0), arch, images[:4].numpy(), stop_at_layer=2,
                       names_filter=["blocks.1.hook_resid_post"])
    assert rel_fro(acts[:, :, 0].cpu().numpy(), c["blocks.1.hook_resid_post"]) < TOL


def _queue_get_or_fail(q, procs, timeout):
    """q.get that gives up as soon as a worker process has died (a crashed worker never puts anything on the queue: waiting out
    the full timeout cost 13 GPU-minutes per failing test on the metered box)."""
    import queue as _queue
    import time as _time
    t_end = _time.time() + timeout
    while True:
        try:
            return q.get(timeout=2.0)
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:
                raise AssertionError(f"worker process exited with {dead} before reporting")
            if _time.time() > t_end:
                raise AssertionError("worker processes did not report in time")


def _dp_gpu_worker(rank, world, port, q):
    """One of two processes that share cuda:0 over a gloo group (RCCL wants one device per rank; gloo moves the same
    tensors through the host) and run exactly what a rank of the 8-GPU job runs."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    d_in, d_sae, k, N = 64, 512, 8, 256
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=d_sae // d_in, activation_fn_str="topk",
        activation_fn_kwargs={"k": k}, normalize_activations="layer_norm", b_dec_init_method="mean", train_batch_size=N,
        lr=1e-3, max_grad_norm=1.0, _device="cuda", log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0)
    sae = StandardSparseAutoencoder(cfg)
    with torch.no_grad():
        for n, v in synth_sae_state(d_in, d_sae, 0).items():
            getattr(sae, n).copy_(torch.from_numpy(v))
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae).use_feature_parallel(False)     # the data-parallel mode
    assert tr.world == world
    act, since, frac, opt, sched = tr.initialize_training_variables()
    losses = []
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(N, d_in, seed=t)).to(dev)
        xs = x[rank * (N // world):(rank + 1) * (N // world)][:, None, :].contiguous()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=xs, n_training_steps=t, n_training_tokens=t * N)
        losses.append(float(loss))
    assert tr._engine is not None, "the native engine did not run"
    tr.sync_parameters()                                          # the last step's all-gather lands here
    # the bench legs a rank of the scaling run executes (tiny step counts: this is a does-it-complete check)
    from vit_prisma_amd.sae.bench_leg import sae_bench_leg, sae_end_to_end_leg
    leg = sae_bench_leg(dev, dist=dist, steps=2, warmup=1)
    e2e = sae_end_to_end_leg(dev, dist=dist, steps=3, warmup=1)
    if rank == 0:
        q.put(({n: getattr(sae, n).detach().cpu().numpy() for n in ("W_enc", "W_dec", "b_enc", "b_dec")}, losses,
               act.cpu().numpy(), leg["value"], e2e["value"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_native_data_parallel_world2_equals_single_process_oracle():
    """The multi-GPU path on real kernels: two ranks (sharing the one GPU of the test box, gloo collectives) each take
    half of every 256-token batch through VisionSAETrainer's native step -- global batch mean pre-reduction,
    reduce-scatter of the gradient rows, global clip norm from the ranks' terms, Adam on each rank's half of the
    features, asynchronous all-gather of the parameters -- and must land on the oracle's single-process parameters
    after three steps.  The bench's DP legs must complete too."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    params, losses, act, leg_tps, e2e_tps = _queue_get_or_fail(q, procs, 800)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    d_in, d_sae, k, N = 64, 512, 8, 256
    P = {kk: v.copy() for kk, v in synth_sae_state(d_in, d_sae, 0).items()}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
    for t in range(3):
        ref = O.train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=t), k, lr=1e-3, step=t + 1)
        assert abs(losses[t] - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (t, losses[t], ref["loss"])
    for n in P:
        assert rel_fro(params[n], P[n]) < 1e-4, n
    assert np.array_equal(act, stats["act_freq_scores"])
    assert leg_tps > 0 and e2e_tps > 0


def _tc_init(d_in, d_sae):
    init = dict(synth_sae_state(d_in, d_sae, 0))
    rs = np.random.RandomState(5)
    init["b_dec_out"] = (rs.standard_normal(d_in) * 0.05).astype(np.float32)
    init["W_skip"] = (rs.standard_normal((d_in, d_in)) / np.sqrt(d_in) * 0.3).astype(np.float32)
    return init


def _rccl_world1_worker(port, q, mode):
    """ONE rank in a process group on the nccl backend (= RCCL on ROCm): the trainer is told to take its multi-rank code paths
    (force_distributed_paths), so every collective of the 8-GPU job -- the in-place reduce_scatter_tensor of the gradient rows,
    the asynchronous all_gather_into_tensor of the updated parameter rows, the packed small buckets, the feature-parallel step's
    all-gathers / all-reduces, the dense step's flat-gradient all-reduce -- is issued on RCCL with the tensors it gets there."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    d_in, d_sae, k, N = 768, 6144, 32, 1024
    relu = mode in ("relu_dp", "relu_ghost_dp")
    tc = mode == "topk_tc_dp"
    ghost = mode.endswith("ghost_dp")
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=d_sae // d_in,
        activation_fn_str="relu" if relu else "topk", activation_fn_kwargs={} if relu else {"k": k}, l1_coefficient=3e-3,
        normalize_activations="layer_norm", b_dec_init_method="mean", train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cuda",
        log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0, use_ghost_grads=ghost, dead_feature_window=1 if ghost else 5000,
        **(dict(is_transcoder=True, transcoder_with_skip_connection=True, d_out=d_in, out_hook_point_layer=1) if tc else {}))
    if tc:
        from vit_prisma_amd.sae import Transcoder
        sae = Transcoder(cfg)
        init = _tc_init(d_in, d_sae)
    else:
        sae = StandardSparseAutoencoder(cfg)
        init = synth_sae_state(d_in, d_sae, 0)
    with torch.no_grad():
        for n, v in init.items():
            getattr(sae, n).copy_(torch.from_numpy(v))
    import copy
    single = None
    if mode == "topk_ghost_dp":
        # the same three steps on the single-process path first: the top-k ghost term's gradient is ill-conditioned in fp32 where the
        # ghost reconstruction meets the residual (r = mse / (mg + 1e-6) with mg -> 0), so after three Adam steps the PARAMETERS are
        # compared with the single-process engine's and the losses with the oracle's
        sae1 = copy.deepcopy(sae)
        tr1 = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae1).use_native(True)
        a1, s1, f1, o1, sc1 = tr1.initialize_training_variables()
        s1[::3] = 5.0
        for t in range(3):
            x = torch.from_numpy(synth_sae_batch(N, d_in, seed=10 + t)).to(dev)[:, None, :].contiguous()
            _, _, _, _, a1, s1, f1 = tr1.train_step(sparse_autoencoder=sae1, optimizer=o1, scheduler=sc1, act_freq_scores=a1,
                                                    n_forward_passes_since_fired=s1, n_frac_active_tokens=f1, layer_acts=x,
                                                    n_training_steps=t, n_training_tokens=t * N)
        tr1.sync_parameters()
        single = {n: getattr(sae1, n).detach().cpu().numpy() for n in init}
        single["act_freq"] = a1.cpu().numpy()
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae).use_native(True).force_distributed_paths(True)
    tr.use_feature_parallel(mode == "topk_tp")
    act, since, frac, opt, sched = tr.initialize_training_variables()
    if ghost:
        since[::3] = 5.0                          # a third of the features count as dead (window 1): the ghost term is live
    out, gates = [], []
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(N, d_in, seed=10 + t)).to(dev)[:, None, :].contiguous()
        if tc:
            x = torch.cat([x, torch.from_numpy(synth_sae_batch(N, d_in, seed=60 + t)).to(dev)[:, None, :]], dim=1).contiguous()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=x, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native
        out.append((float(loss), float(l0)))
        if mode == "relu_dp":
            # the ReLU gates as the kernels took them (dH != 0, left in the workspace by the dense GEMMs): entries within fp32 summation
            # noise of zero may fall on either side, and the oracle's step is continued under THESE gates (see the test)
            eng = tr._engine
            assert int(eng.relu_mode.item()) == 1, "from the synthetic init the step runs on the dense GEMMs"
            off = eng.lib.pv_debug_sae_ws_offset(eng._plan, b"hidden")
            dH = eng.workspace[off:off + N * d_sae * 4].view(torch.float32).view(N, d_sae)
            gates.append(np.packbits((dH != 0).cpu().numpy(), axis=1))
    took = {"topk_dp": tr._engine is not None and tr._fp is None and not tr._engine.lazy_w_enc,
            "topk_tp": tr._fp is not None, "relu_dp": tr._engine is not None,
            "topk_tc_dp": tr._engine is not None and tr._engine.transcoder and tr._fp is None,
            "topk_ghost_dp": tr._engine is not None and tr._fp is None, "relu_ghost_dp": tr._engine is not None}[mode]
    tr.sync_parameters()
    q.put((out, {n: getattr(sae, n).detach().cpu().numpy() for n in init}, act.cpu().numpy(), took, dist.get_backend(), single, gates))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["topk_dp", "topk_tp", "relu_dp", "topk_tc_dp", "topk_ghost_dp", "relu_ghost_dp"])
def test_multi_rank_steps_on_rccl_world1_equal_the_oracle(mode):
    """the trainer's data-parallel step (sharded optimizer), its feature-parallel step and the dense step's
    data-parallel form, each through torch.distributed on the NCCL backend (RCCL) with a world of one rank, against the
    single-process oracle after three steps (768 -> 6144, 1024 tokens).  What a one-GPU box can execute of the 8-GPU job: the same
    calls on the same backend; the arithmetic of more than one rank is covered by the gloo tests above and on CPU."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(port, q, mode))
    p.start()
    out, params, act, took, backend, single, gates = _queue_get_or_fail(q, [p], 600)
    p.join(timeout=120)
    assert p.exitcode == 0 and took and backend == "nccl"
    d_in, d_sae, k, N = 768, 6144, 32, 1024
    relu = mode in ("relu_dp", "relu_ghost_dp")
    ghost = mode.endswith("ghost_dp")            # (use_ghost_grads with a process group: the global residual mean / mse exchange)
    tc = mode == "topk_tc_dp"                    # (a top-k Transcoder with the skip connection: the token-sharded step's transcoder branch)
    P = {kk: v.copy() for kk, v in (_tc_init(d_in, d_sae) if tc else synth_sae_state(d_in, d_sae, 0)).items()}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
    if ghost:
        stats["n_fwd_since_fired"][::3] = 5.0
    # relu_dp runs from the synthetic init, where half of all pre-activations are positive: a few of the 6.3 M per step lie within fp32
    # summation noise of zero and take the other side of the ReLU -- the worker hands back the gates the kernels took and the oracle
    # continues under them (parameters then at 1e-4; round 5: a constant 5e-4).
    # The ghost forms: the same three steps carried in float64 beside the fp32 ones tell how far two correct computations of this run end
    # up apart (the top-k ghost term's gradient is ill-conditioned in fp32 where the ghost reconstruction meets the residual; Adam turns
    # such entries into lr-sized differences); the parameters are held to 6 x that distance where it exceeds 1e-4 (round 5: 1e-3).
    noisy = (relu and not gates) or single is not None
    if noisy:
        P64 = {kk: v.astype(np.float64) for kk, v in P.items()}
        opt64 = {w: {kk: v.astype(np.float64) for kk, v in opt[w].items()} for w in ("m", "v")}
        stats64 = {kk: v.copy() for kk, v in stats.items()}
    for t in range(3):
        x, y = synth_sae_batch(N, d_in, seed=10 + t), synth_sae_batch(N, d_in, seed=60 + t) if tc else None
        kw = dict(lr=1e-3, step=t + 1, l1_coefficient=3e-3 if relu else 0.0, dead_feature_window=1 if ghost else None)
        gate = None
        if gates:
            # relu_dp: the oracle's step under the kernels' own ReLU gates, which may differ from the oracle's only on entries within fp32
            # summation noise of zero (as test_relu_l1_dense_step_vs_oracle does; round 5 loosened the parameter bound to 5e-4 instead)
            Pc = {kk: v.copy() for kk, v in P.items()}
            O.renorm_decoder(Pc)
            fw = O.sae_forward(Pc, x, None, l1_coefficient=3e-3)
            got = np.unpackbits(gates[t], axis=1)[:, :d_sae].astype(bool)
            differs = got != (fw["feature_acts"] > 0)
            assert differs.sum() <= 1e-5 * differs.size and np.all(np.abs(fw["hidden_pre"][differs]) < 1e-5 * np.abs(fw["hidden_pre"]).max()), t
            gate = got if differs.any() else None
        ref = O.train_step(P, opt, stats, x, None if relu else k, target=y, gate=gate, **kw)
        if noisy:
            O.train_step(P64, opt64, stats64, x.astype(np.float64), None if relu else k, target=None if y is None else y.astype(np.float64), **kw)
        assert abs(out[t][0] - ref["loss"]) <= TOL * abs(ref["loss"]) and abs(out[t][1] - ref["l0"]) <= TOL * ref["l0"], (t, out[t], ref)
    for n in P:
        ptol = max(TOL, 6.0 * rel_fro(P[n], P64[n])) if noisy else TOL
        if single is not None:                   # (topk_ghost_dp: against the single-process engine, see the worker)
            assert rel_fro(params[n], single[n]) < ptol, n
        else:
            assert rel_fro(params[n], P[n]) < ptol, n
    want_act = single["act_freq"] if single is not None else stats["act_freq_scores"]
    assert np.abs(act - want_act).sum() <= TOL * want_act.sum()


def _rccl_world1_worker_b(port, q, mode, batch_file):
    """As _rccl_world1_worker, for the two forms round 5 left without RCCL coverage: the top-k GATED SAE
    (gated_topk_dp) and a top-k Transcoder between hook points of DIFFERENT width (topk_tc_dout_dp: 768 -> 1024, no skip connection).
    The batches come from the test (a file): it has replaced tokens whose top-k selection is a near-tie in the oracle's own numbers."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    B = np.load(batch_file)
    d_in, d_sae, k, N = 768, 6144, 32, 1024
    gated = mode == "gated_topk_dp"
    d_out = d_in if gated else 1024
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=d_sae // d_in, activation_fn_str="topk",
        activation_fn_kwargs={"k": k}, normalize_activations="layer_norm", b_dec_init_method="mean", train_batch_size=N, lr=1e-3,
        max_grad_norm=1.0, _device="cuda", log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0,
        **(dict(architecture="gated") if gated else dict(is_transcoder=True, transcoder_with_skip_connection=False, d_out=d_out,
                                                          out_hook_point_layer=1)))
    tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True).force_distributed_paths(True)
    sae = tr.sparse_coder
    names = [n for n, _ in sae.named_parameters()]
    with torch.no_grad():
        for n in names:
            getattr(sae, n).copy_(torch.from_numpy(B["init_" + n]))
    act, since, frac, opt, sched = tr.initialize_training_variables()
    out = []
    for t in range(3):
        x = torch.from_numpy(B[f"x{t}"]).to(dev)
        layer_acts = x[:, None, :].contiguous() if gated else _PairActs(x, torch.from_numpy(B[f"y{t}"]).to(dev))
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=layer_acts, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native, tr._native_why_not(sae)
        out.append((float(loss), float(l0)))
    took = tr._engine is not None and tr._fp is None and (tr._engine.gated_topk if gated else tr._engine.tc_widths == (d_in, d_out))
    tr.sync_parameters()
    q.put((out, {n: getattr(sae, n).detach().cpu().numpy() for n in names}, act.cpu().numpy(), bool(took), dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["gated_topk_dp", "topk_tc_dout_dp"])
def test_gated_topk_and_unequal_width_transcoder_on_rccl_world1_equal_the_oracle(mode, tmp_path):
    """the trainer's multi-rank step of the top-k gated SAE and of a top-k transcoder with d_out != d_in (768 -> 1024),
    through torch.distributed on the NCCL backend (RCCL) with a world of one rank, against the single-process oracle after three steps
    (768 -> 6144, 1024 tokens): losses, l0, firing statistics, every parameter at 1e-4."""
    import socket
    import torch.multiprocessing as mp
    d_in, d_sae, k, N = 768, 6144, 32, 1024
    gated = mode == "gated_topk_dp"
    d_out = d_in if gated else 1024
    rs = np.random.RandomState(9)
    P = {kk: v.copy() for kk, v in synth_sae_state(d_in, d_sae, 0).items()}
    if gated:
        for name, scale in (("b_gate", 0.05), ("r_mag", 0.2), ("b_mag", 0.05)):
            P[name] = (rs.standard_normal(d_sae) * scale).astype(np.float32)
    else:
        wd = rs.uniform(-1.0, 1.0, size=(d_sae, d_out)).astype(np.float32)
        P["W_dec"] = wd / np.linalg.norm(wd, axis=1, keepdims=True)
        P["b_dec_out"] = (rs.standard_normal(d_out) * 0.05).astype(np.float32)
    blob = {"init_" + n: v.copy() for n, v in P.items()}
    b_enc0 = P.pop("b_enc") if gated else None                   # (takes no part in a gated SAE's step: no gradient)
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
    refs = []
    for t in range(3):
        x = synth_sae_batch(N, d_in, seed=10 + t)
        y = None if gated else np.concatenate([synth_sae_batch(N, d_in, seed=60 + t), synth_sae_batch(N, d_in, seed=80 + t)], axis=1)[:, :d_out].copy()
        # tokens whose top-k selection is a near-tie in the oracle's own numbers may keep either entry: replaced by a safe token (picked by
        # the oracle alone, before any kernel runs -- as the single-process tests do)
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        pres = ([O.gated_forward(Pc, x, k=k)[key] for key in ("mag_pre", "gate_pre")] if gated else [O.sae_forward(Pc, x, k, target=y)["hidden_pre"]])
        risky = np.zeros(N, bool)
        for h in pres:
            top = -np.partition(-h, k, axis=1)[:, :k + 1]
            risky |= (top[:, :k].min(axis=1) - top[:, k]) < 1e-5 * np.abs(h).max()
        if risky.any():
            safe = np.flatnonzero(~risky)[0]
            x[risky] = x[safe]
            if y is not None:
                y[risky] = y[safe]
        blob[f"x{t}"] = x
        if y is not None:
            blob[f"y{t}"] = y
        refs.append(O.gated_train_step(P, opt, stats, x, lr=1e-3, step=t + 1, k=k) if gated
                    else O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1, target=y))
    batch_file = os.path.join(tmp_path, "batches.npz")
    np.savez(batch_file, **blob)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker_b, args=(port, q, mode, batch_file))
    p.start()
    out, params, act, took, backend = _queue_get_or_fail(q, [p], 600)
    p.join(timeout=120)
    assert p.exitcode == 0 and took and backend == "nccl"
    for t in range(3):
        assert abs(out[t][0] - refs[t]["loss"]) <= TOL * abs(refs[t]["loss"]) and abs(out[t][1] - refs[t]["l0"]) <= TOL * refs[t]["l0"], (t, out[t], refs[t])
    for n in P:
        assert rel_fro(params[n], P[n]) < TOL, n
    if gated:
        assert np.array_equal(params["b_enc"], b_enc0)
    assert np.abs(act - stats["act_freq_scores"]).sum() <= TOL * stats["act_freq_scores"].sum()


# ---------------------------------------------------------------------------------------------------
# the filtered encoder (sae_enc.hip): fp16 MFMA filter + exact fp32 re-scoring must be indistinguishable from the
# exact fp32 GEMM + streaming top-k it replaces
# ---------------------------------------------------------------------------------------------------
def _both_paths(eng, x, tuning, loop=-1):
    tuning("reset")
    tuning("gemm_loop", loop)                      # K loop of the filter GEMM: -1 software-pipelined (default), 0 barrier-then-fetch
    idx_f, val_f, mu_f, sd_f = (t.clone() for t in eng.encode_topk(x))
    n_fb = eng.fallback_rows()
    tuning("sae_exact", 1)
    idx_e, val_e, mu_e, sd_e = (t.clone() for t in eng.encode_topk(x))
    tuning("reset")
    torch.cuda.synchronize()
    return (idx_f, val_f), (idx_e, val_e), n_fb


@pytest.mark.parametrize("loop", [-1, 0])
@pytest.mark.parametrize("d_in,d_sae,k,n", [(128, 8192, 16, 600), (768, 24576, 32, 1100), (96, 4096, 64, 257), (104, 4096, 16, 300),
                                            (768, 49152, 32, 700), (768, 3072, 32, 900), (128, 2048, 8, 300), (1280, 20480, 32, 500)])
def test_filtered_encoder_equals_exact_path(d_in, d_sae, k, n, loop, tuning):
    _, _, _, T = fresh(d_in, d_sae)
    T["b_enc"].mul_(20.0)                                                  # biases that matter
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
    assert eng.filtered_encoder
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=5)).cuda()
    (idx_f, val_f), (idx_e, val_e), n_fb = _both_paths(eng, x, tuning, loop)
    assert n_fb == 0                                                       # ordinary data: the filter decides every token
    assert torch.equal(idx_f.sort(dim=1).values, idx_e.sort(dim=1).values)  # exact index sets
    # values: both are fp32 dot products of the same operands in different summation orders
    o_f, o_e = idx_f.sort(dim=1).indices, idx_e.sort(dim=1).indices
    assert float((val_f.gather(1, o_f) - val_e.gather(1, o_e)).abs().max()) <= 2e-6 * float(val_e.abs().max())
    # and against the oracle
    P = {kk: v.cpu().numpy() for kk, v in T.items()}
    fw = O.sae_forward(P, x.cpu().numpy(), k)
    assert np.array_equal(np.sort(idx_f.cpu().numpy(), axis=1), np.sort(fw["idx"], axis=1))


def test_filtered_encoder_exact_fallback_on_ties_range_and_outside_edits(tuning):
    d_in, d_sae, k, n = 64, 4096, 8, 40
    _, _, _, T = fresh(d_in, d_sae)
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, False, n)      # no input LayerNorm: raw magnitudes reach the GEMM
    assert eng.filtered_encoder
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=2)).cuda()
    x[3, 5] = 3.0e5                                                        # outside the fp16 range: that token must take the exact path
    x[7] *= 1e-7                                                           # fp16-subnormal inputs: covered by the bound's absolute term
    (idx_f, val_f), (idx_e, val_e), n_fb = _both_paths(eng, x, tuning)
    assert 1 <= n_fb <= 3
    assert torch.equal(idx_f.sort(dim=1).values, idx_e.sort(dim=1).values)
    # duplicated encoder columns: every value comes as an exactly tied pair -> the band around the k-th value holds both
    # members, the exact re-scoring ties, the index order decides (which member survives is arbitrary, values are not)
    with torch.no_grad():
        T["W_enc"][:, 1::2] = T["W_enc"][:, 0::2]
        T["b_enc"][1::2] = T["b_enc"][0::2]
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=3)).cuda()
    (idx_f, val_f), (idx_e, val_e), n_fb = _both_paths(eng, x, tuning)     # (the in-place edits above re-sync the shadows)
    close = lambda a, b: torch.allclose(a.sort(dim=1).values, b.sort(dim=1).values, rtol=2e-6, atol=1e-7)  # noqa: E731 (two fp32 summation orders)
    assert close(val_f, val_e)
    assert torch.equal(val_f.sort(dim=1).values[:, 0::2], val_f.sort(dim=1).values[:, 1::2])     # pairs tie EXACTLY within one path
    for r in range(n):
        assert len(set(idx_f[r].cpu().tolist())) == k
    # 2048 identical columns on top of every row: a 2048-way tie at the k-th value -> candidate-list overflow -> every
    # token is recomputed exactly (radix top-k)
    with torch.no_grad():
        T["W_enc"][:, :2048] = T["W_enc"][:, :1].clone()
        T["b_enc"][:2048] = 5.0
    (idx_f, val_f), (idx_e, val_e), n_fb = _both_paths(eng, x, tuning)
    assert n_fb == n
    assert close(val_f, val_e)
    assert bool((idx_f < 2048).all()) and all(len(set(idx_f[r].cpu().tolist())) == k for r in range(n))
    # a weight outside the fp16 range poisons the bound: every token takes the exact path, results stay right
    with torch.no_grad():
        T["W_enc"][3, 77] = 1.0e6
    (idx_f, val_f), (idx_e, val_e), n_fb = _both_paths(eng, x, tuning)
    assert n_fb == n and close(val_f, val_e)


# ---------------------------------------------------------------------------------------------------
# module-level inference on the HIP kernels (the SAE inside a ViT hook)
# ---------------------------------------------------------------------------------------------------
def _module(d_in, d_sae, k, return_out_only=False, normalize_activations="layer_norm"):
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=d_sae // d_in, activation_fn_str="topk",
        activation_fn_kwargs={"k": k}, normalize_activations=normalize_activations, _device="cuda", log_to_wandb=False)
    cfg.return_out_only = return_out_only
    sae = StandardSparseAutoencoder(cfg)
    with torch.no_grad():
        for n, v in synth_sae_state(d_in, d_sae, 0).items():
            getattr(sae, n).copy_(torch.from_numpy(v))
    return sae.eval()


@pytest.mark.parametrize("norm", ["layer_norm", "constant_norm_rescale", "none"])
def test_module_forward_and_encode_run_natively_and_equal_the_torch_path(norm):
    """(round 6: also under normalize_activations = "constant_norm_rescale" -- sae.py:60-72, the prep kernel's third mode -- and none)"""
    sae = _module(768, 24576, 32, normalize_activations=norm)
    x = torch.from_numpy(synth_sae_batch(300, 768, seed=4)).cuda().view(6, 50, 768)      # [batch, tokens, d_in] like a hook sees it
    with torch.no_grad():
        got = sae(x)
        assert sae.last_run_native, sae.native_fallback_reason
        sae_in_n, acts_n = sae.encode(x)
        assert sae.last_run_native
        sae.use_native(False)
        want = sae(x)
        sae_in_t, acts_t = sae.encode(x)
        assert not sae.last_run_native
        sae.use_native(None)
    assert len(got) == 7 and got[0].shape == x.shape and got[1].shape == (6, 50, 24576) and got[4] is None
    assert rel_fro(got[0].cpu().numpy(), want[0].cpu().numpy()) < 1e-5
    assert torch.equal(got[1] > 0, want[1] > 0)                                          # same active sets
    assert rel_fro(got[1].cpu().numpy(), want[1].cpu().numpy()) < 1e-5
    assert abs(float(got[2]) - float(want[2])) <= 1e-5 * float(want[2])                  # loss over dim 0 of the 3-D input, like the reference
    assert torch.equal(acts_n > 0, acts_t > 0) and rel_fro(sae_in_n.cpu().numpy(), sae_in_t.cpu().numpy()) < 1e-6
    # autograd recording -> PyTorch path (auto), error when forced
    out = sae(x)
    assert not sae.last_run_native and "autograd" in sae.native_fallback_reason and out[2].requires_grad
    with pytest.raises(Exception):
        sae.use_native(True)(x)
    sae.use_native(None)
    # a hook on the SAE's own hook points -> PyTorch path
    with torch.no_grad():
        seen = []
        sae.hook_hidden_post.add_hook(lambda t, hook: seen.append(t.shape))
        sae(x)
        assert not sae.last_run_native and seen
        sae.reset_hooks()


def test_sae_substitution_inside_a_vit_hook_is_native_end_to_end():
    """sae/evals/evals.py:321-392: the ViT forward with blocks.6.hook_resid_post replaced by the SAE's reconstruction.
    ViT split plan (HIP) -> Python hook -> SAE forward (HIP) -> rest of the ViT (HIP); against the all-PyTorch run."""
    from vit_prisma_amd.synth import ARCHS, synth_images, synth_vit_state
    arch = ARCHS["clip-vit-b32"]
    vit = HookedViT(HookedViTConfig(**arch, device="cuda"))
    vit.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()})
    vit = vit.cuda().eval()
    sae = _module(768, 24576, 32, return_out_only=True)
    x = torch.from_numpy(synth_images(arch, 4, 2)).cuda()
    name = "blocks.6.hook_resid_post"
    calls = []

    def substitute(t, hook):
        out = sae(t)
        calls.append(sae.last_run_native)
        return out

    with torch.no_grad():
        vit.use_native(True)
        got = vit.run_with_hooks(x, fwd_hooks=[(name, substitute)])
        assert vit.last_run_native and calls == [True]
        vit.use_native(False)
        sae.use_native(False)
        want = vit.run_with_hooks(x, fwd_hooks=[(name, substitute)])
        assert not vit.last_run_native and calls == [True, False]
    assert rel_fro(got.cpu().numpy(), want.cpu().numpy()) < 1e-4


def test_deferred_decoder_renorm_equals_the_explicit_pass():
    """step(renorm_decoder=True) + apply (the trainer's form: inverse row norms only, W_dec rewritten by the Adam kernel)
    lands on the same parameters as renorm_decoder() + step() + apply(), and W_dec holds un-normalised rows at the step
    boundary exactly like the reference's (train_sae.py:307 normalises at the START of the next step)."""
    d_in, d_sae, k, n = 128, 8192, 16, 512
    engs = []
    for _ in range(2):
        _, _, _, T = fresh(d_in, d_sae)
        T["W_dec"].mul_(torch.linspace(0.5, 2.0, d_sae, device="cuda")[:, None])       # rows far from unit norm
        engs.append(NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n))
    a, b = engs
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
        a.renorm_decoder(); a.step(x); a.grad_sqnorm(); a.apply(1e-3, 1.0)
        b.step(x, renorm_decoder=True); b.grad_sqnorm(); b.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        assert abs(float(a.scalars[0]) - float(b.scalars[0])) <= 1e-6 * float(a.scalars[0])
        for name in ("W_enc", "W_dec", "b_enc", "b_dec"):
            assert rel_fro(b.params[name].cpu().numpy(), a.params[name].cpu().numpy()) < 1e-6, (t, name)
    assert float((b.params["W_dec"].norm(dim=1) - 1).abs().max()) > 1e-5               # un-normalised after the optimizer step


@pytest.mark.parametrize("d_in,d_sae,k,n", [(768, 24576, 32, 1024), (128, 8192, 16, 512), (100, 4096, 8, 96)])
def test_every_gradient_row_is_written_and_the_clip_norm_comes_from_the_backward(d_in, d_sae, k, n):
    """The step has no zero_grad pass: rows of features no token kept are zeroed by their own kernel, every other row is
    stored once -- poison the buffer first.  The clip norm assembled from the backward kernels' per-feature terms
    (pv_sae_grad_sqnorm_step) equals the two-stage reduction over the whole buffer (pv_sae_grad_sqnorm)."""
    _, _, _, T = fresh(d_in, d_sae)
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
    for t in range(2):
        x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
        eng.flat_g.fill_(float("nan"))
        eng.step(x, renorm_decoder=True)
        assert bool(torch.isfinite(eng.flat_g[:eng.n_flat]).all())
        empty = eng.fire_count == 0
        assert int(empty.sum()) > 0 and float(eng.g["W_dec"][empty].abs().max()) == 0.0
        eng.grad_sqnorm(from_step=True)
        fused = float(eng.scalars[3])
        eng.grad_sqnorm()
        full = float(eng.scalars[3])
        ref = float((eng.flat_g[:eng.n_flat].double() ** 2).sum())
        assert abs(fused - ref) <= 1e-5 * ref and abs(full - ref) <= 1e-5 * ref, (fused, full, ref)
        eng.apply(1e-3, 1.0)
        eng.grad_sqnorm(from_step=True)                      # stale after apply: must fall back to the full pass
        assert abs(float(eng.scalars[3]) - ref) <= 1e-5 * ref


@pytest.mark.parametrize("d_in,d_sae,k,n", [(768, 24576, 32, 1024), (100, 4096, 8, 96)])
def test_sparse_gradient_step_lands_on_the_same_parameters_as_the_dense_one(d_in, d_sae, k, n):
    """PV_SAE_SPARSE_GRADS (what the single-process trainer passes): rows of features that kept no token are neither zeroed
    by the step nor read by apply.  Poisoned gradient buffers stay poisoned exactly there, the clip norm from the step's
    per-feature terms is unchanged, and parameters + Adam moments after apply equal the dense-gradient step's (to summation
    noise everywhere, to the bit on the rows in question: g = 0 either way).  Readers of the raw buffers refuse to run while such a step is pending."""
    engs = []
    for _ in range(2):
        _, _, _, T = fresh(d_in, d_sae)
        engs.append(NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n))
    dense, sparse = engs
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
        dense.step(x, renorm_decoder=True); dense.grad_sqnorm(from_step=True); dense.apply(1e-3, 1.0)
        sparse.flat_g.fill_(float("nan"))
        sparse.step(x, renorm_decoder=True, sparse_grads=True)
        empty = sparse.fire_count == 0
        assert int(empty.sum()) > 0
        assert bool(torch.isnan(sparse.g["W_dec"][empty]).all()) and bool(torch.isnan(sparse.g["W_enc"][empty]).all())
        assert bool(torch.isfinite(sparse.g["W_dec"][~empty]).all()) and bool(torch.isfinite(sparse.g["b_enc"]).all())
        with pytest.raises(RuntimeError):
            sparse.grad_sqnorm()
        with pytest.raises(RuntimeError):
            sparse.forward(x)
        sparse.grad_sqnorm(from_step=True)
        sparse.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for i in (0, 3):                                     # loss, clip norm
            assert abs(float(sparse.scalars[i]) - float(dense.scalars[i])) <= 1e-6 * abs(float(dense.scalars[i])), (t, i)
        # two runs of the step agree to summation-order noise (the order pairs enter a feature's list is scheduling-dependent,
        # DESIGN 3.1) -- except on the rows this flag is about, which see g = 0 in both and must be the same bits
        for name in ("W_enc", "W_dec", "b_enc", "b_dec"):
            assert rel_fro(sparse.params[name].cpu().numpy(), dense.params[name].cpu().numpy()) < 1e-6, (t, name)
        assert rel_fro(sparse.flat_m.cpu().numpy(), dense.flat_m.cpu().numpy()) < 1e-5, t
        assert rel_fro(sparse.flat_v.cpu().numpy(), dense.flat_v.cpu().numpy()) < 1e-5, t
        never = empty if t == 0 else (never & empty)        # features without a pair in every step so far: no noise to inherit
        assert int(never.sum()) > 0
        assert torch.equal(sparse.params["W_dec"][never], dense.params["W_dec"][never]), t
        assert torch.equal(sparse.params["W_enc"][:, never], dense.params["W_enc"][:, never]), t
        assert torch.equal(sparse._m["W_dec"][never], dense._m["W_dec"][never]) and torch.equal(sparse._v["W_encT"][never], dense._v["W_encT"][never]), t
        assert bool(torch.isfinite(sparse.flat_m).all()) and bool(torch.isfinite(sparse.flat_v).all())
    sparse.forward(x)                                        # nothing pending any more


def test_activation_cache_shards_written_from_the_native_harvest_match_the_references():
    """SURVEY.md 8f row 2 on the GPU: generate_cached_activations_from_dataset driven by the native run_with_cache writes the
    {idx}.pt fp16 shards the REFERENCE's writer produced for the same images (tests/golden/act_cache_tiny/)."""
    import tempfile
    from vit_prisma_amd.sae import CacheVisionActivationStore
    from vit_prisma_amd.synth import ARCHS, synth_vit_state
    gold = os.path.join(GOLDEN, "act_cache_tiny")
    ref = np.load(os.path.join(gold, "reference_reader.npz"))
    arch = ARCHS["tiny"]
    vit = HookedViT(HookedViTConfig(**arch, device="cuda"))
    vit.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    vit = vit.cuda().eval().use_native(True)
    imgs = torch.from_numpy(ref["images"])
    ds = torch.utils.data.TensorDataset(imgs, torch.zeros(len(imgs), dtype=torch.long))
    with tempfile.TemporaryDirectory() as tmp:
        cfg = VisionModelSAERunnerConfig(
            hook_point_layer=1, layer_subtype="hook_resid_post", d_in=64, expansion_factor=8, activation_fn_str="topk",
            activation_fn_kwargs={"k": 8}, context_size=17, store_batch_size=2, n_batches_in_buffer=4, train_batch_size=16,
            cached_activations_path=os.path.join(tmp, "cache"), use_cached_activations=True, _device="cuda", log_to_wandb=False)
        store = VisionActivationsStore(cfg, vit, ds, create_dataloader=False)
        assert store.generate_cached_activations_from_dataset(tokens_per_file=50) == 3 and vit.last_run_native
        for i, rows in enumerate((50, 50, 19)):
            ours, theirs = torch.load(os.path.join(tmp, "cache", f"{i}.pt")), torch.load(os.path.join(gold, f"{i}.pt"))
            assert ours.dtype == torch.float16 and tuple(ours.shape) == (rows, 1, 64) == tuple(theirs.shape)
            assert torch.allclose(ours.float(), theirs.float(), atol=4e-3, rtol=2e-3), i       # one fp16 ulp at |x| ~ 4
        # and the reader serves training batches from them on the GPU
        reader = CacheVisionActivationStore(cfg)
        b = reader.next_batch()
        assert b.is_cuda and b.shape == (16, 1, 64)


# ---------------------------------------------------------------------------------------------------
# feature-parallel step on the real kernels (pv_sae_tp_merge / pv_sae_tp_partial / pv_sae_tp_finish / pv_sae_tp_bucket_*), two
# ranks sharing the GPU over gloo: the small shape (exact encoder), a 4096-feature shard (filtered encoder) and the BENCH
# shape (768 -> 24576, 4096 tokens).  The choreography itself is also covered on CPU (tests/test_feature_parallel_cpu.py).
# Batches are synth_sae_batch(seed0 + t): the index-set comparison needs batches on which no token has its k-th and
# (k+1)-th pre-activation closer than fp32 summation noise -- at 768 -> 8192 seed 0 has one (token 381, relative gap
# 5.7e-8: the fp32 oracle and an fp64 evaluation disagree there; the kernels side with fp64, tools/tp_diag.py), seeds
# 10..12 have none (smallest gap 3.4e-6).
# ---------------------------------------------------------------------------------------------------


def _tp_gpu_worker(rank, world, port, q, d_in, d_sae, k, N, steps, seed0):
    import torch.distributed as dist
    from vit_prisma_amd.sae.feature_parallel import FeatureParallelSAE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    T = {n: torch.from_numpy(v.copy()).to(dev) for n, v in synth_sae_state(d_in, d_sae, 0).items()}
    fp = FeatureParallelSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k,
                            lambda We, Wd, be, bd: NativeSAE(We, Wd, be, bd, k, True, N), dist=dist, rank=rank, world=world)
    losses, fires = [], []
    for t in range(steps):
        x = torch.from_numpy(synth_sae_batch(N, d_in, seed=seed0 + t)).to(dev)
        loss, l0 = fp.step(x, lr=1e-3, max_grad_norm=1.0)
        losses.append((float(loss), float(l0)))
        fires.append(fp.fire_count.cpu().numpy().copy())
    P = fp.gather_parameters()
    if rank == 0:
        q.put(({n: v.cpu().numpy() for n, v in P.items()}, losses, fires))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("d_in,d_sae,k,N,seed0", [(64, 512, 8, 256, 0), (768, 8192, 32, 512, 10), (768, 24576, 32, 4096, 0)])
def test_feature_parallel_world2_equals_single_process_oracle(d_in, d_sae, k, N, seed0):
    """Two ranks, each with a NativeSAE over its half of the features: candidates all-gathered, global top-k, partial
    reconstructions all-reduced, shard-local backward / clip / project / Adam (vit_prisma_amd/sae/feature_parallel.py) --
    losses, l0, firing counts and the gathered parameters against the single-process oracle."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    steps = 3
    procs = [ctx.Process(target=_tp_gpu_worker, args=(r, 2, port, q, d_in, d_sae, k, N, steps, seed0)) for r in range(2)]
    for p in procs:
        p.start()
    params, losses, fires = _queue_get_or_fail(q, procs, 800)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    P = {kk: v.copy() for kk, v in synth_sae_state(d_in, d_sae, 0).items()}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
    for t in range(steps):
        before = stats["act_freq_scores"].copy()
        ref = O.train_step(P, opt, stats, synth_sae_batch(N, d_in, seed=seed0 + t), k, lr=1e-3, step=t + 1)
        assert abs(losses[t][0] - ref["loss"]) <= 1e-4 * abs(ref["loss"]) and abs(losses[t][1] - ref["l0"]) < 1e-4, (t, losses[t], ref)
        assert np.array_equal(fires[t], stats["act_freq_scores"] - before), t
    for n in P:
        assert rel_fro(params[n], P[n]) < 1e-4, n


def test_tp_merge_kernel_equals_the_host_statement():
    """pv_sae_tp_merge against the two-stable-sorts statement of the same rule (tests/_cpu_engine.py: value desc, global
    feature index asc), on candidates with many exact ties across ranks, zeros and a world of 8."""
    from _cpu_engine import OracleShardEngine
    from vit_prisma_amd import _native as N
    g = torch.Generator().manual_seed(3)
    for W, n, k, shard in ((2, 300, 32, 4096), (8, 257, 64, 3072), (4, 64, 5, 16), (1, 10, 8, 64)):
        vals = (torch.randint(0, 12, (W, n, k), generator=g).float() * 0.25)              # heavy ties, some zeros
        vals, _ = vals.sort(dim=2, descending=True)
        idx = torch.stack([torch.stack([torch.randperm(shard, generator=g)[:k] for _ in range(n)]) for _ in range(W)]).int()
        gathered = torch.stack([vals.view(torch.int32), idx], dim=1).contiguous()         # [W, 2, n, k]
        twin = OracleShardEngine.__new__(OracleShardEngine)
        twin.k, twin.d_sae = k, shard
        gd = gathered.cuda()
        for rank in range(W):
            out = torch.empty(n, k, dtype=torch.float32, device="cuda")
            N.check(N.lib().pv_sae_tp_merge(gd.data_ptr(), W, rank, n, k, shard, out.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "pv_sae_tp_merge")
            assert torch.equal(out.cpu(), twin.tp_merge(gathered, W, rank, n)), (W, rank)


@pytest.mark.parametrize("world,d_in,d_sae,k,N", [(4, 64, 512, 8, 256), (4, 768, 24576, 32, 4096), (8, 768, 24576, 32, 4096)])
def test_feature_parallel_simulated_world_equals_single_process_oracle(world, d_in, d_sae, k, N):
    """All ranks of a world of 4 / 8 on ONE GPU in lockstep (feature_parallel.simulate_step: the phases of the real step, the
    exchanges by hand) at the bench shape: losses, l0, firing counts and parameters against the single-process oracle.  At
    world 8 a shard is 3072 features: the filtered encoder's smallest plans."""
    from vit_prisma_amd.sae.feature_parallel import FeatureParallelSAE, gather_parameters_local, simulate_step
    P, opt, stats, T = fresh(d_in, d_sae)
    ranks = [FeatureParallelSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k,
                                lambda We, Wd, be, bd: NativeSAE(We, Wd, be, bd, k, True, N), rank=r, world=world) for r in range(world)]
    for t in range(2):
        x = synth_sae_batch(N, d_in, seed=t)
        before = stats["act_freq_scores"].copy()
        ref = O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1)
        loss, l0 = simulate_step(ranks, torch.from_numpy(x).cuda(), lr=1e-3, max_grad_norm=1.0)
        torch.cuda.synchronize()
        assert abs(float(loss) - ref["loss"]) <= TOL * abs(ref["loss"]) and abs(float(l0) - ref["l0"]) < 1e-4, (t, float(loss), ref["loss"])
        for fp in ranks:
            assert np.array_equal(fp.fire_count.cpu().numpy(), stats["act_freq_scores"] - before), (t, fp.rank)
    got = gather_parameters_local(ranks)
    for n in P:
        assert rel_fro(got[n].cpu().numpy(), P[n]) < TOL, n


# ---------------------------------------------------------------------------------------------------
# the dense fused step: ReLU + L1 SAEs (pv_sae_dense_step, csrc/sae_dense.hip)
# ---------------------------------------------------------------------------------------------------
# (the last four: the shape bench.py times this step at -- 768 -> 24576, 4096 tokens -- and the x64 SAEs every published CLIP-B/32 SAE of
# the reference is, docs/sae_table.md:12-36 -- 768 -> 49152)
def fp32_noise_floor(Pc, x, k, ln, dead, l1c=0.0, gate=None, opt=None, lr=1e-3, step=1):
    """How far is the fp32 ORACLE itself from the same computation carried in float64?  Per gradient tensor, the rel-Frobenius distance
    between the oracle's fp32 gradients and its float64 ones (same ReLU gates / the same top-k sets: the float64 run keeps the fp32
    run's selection).  The ghost term puts exp(hidden_pre) of the dead columns into the gradients, which turns the ABSOLUTE fp32
    summation noise of hidden_pre into a RELATIVE error of those entries: two correct fp32 implementations differ by this much,
    whatever their summation orders.  The ghost tests hold the kernels to a multiple of this floor per tensor (ghost_tolerances)
    instead of a constant argued in a comment (round-4 review).
    opt given: also "param:<name>" -- the same distance between the PARAMETERS after clip + projection + Adam on the two gradient
    sets (Adam's g / (sqrt(v) + eps) turns a noise-sized gradient entry into an lr-sized step in either direction): the bound the
    post-step parameters are held to (round 5 used a constant 1e-3 there)."""
    P64 = {kk: v.astype(np.float64) for kk, v in Pc.items()}
    x64 = x.astype(np.float64)
    fw32 = O.sae_forward(Pc, x, k, layer_norm=ln, l1_coefficient=l1c, dead_mask=dead)
    fw64 = O.sae_forward(P64, x64, k, layer_norm=ln, l1_coefficient=l1c, dead_mask=dead, idx=fw32["idx"])
    g = gate if gate is not None else (None if k is not None else fw32["feature_acts"] > 0)
    kw = {} if k is not None else dict(l1_coefficient=l1c, gate=g)
    gr32 = O.sae_backward(Pc, x, fw32, layer_norm=ln, **kw)
    gr64 = O.sae_backward(P64, x64, fw64, layer_norm=ln, **kw)
    floor = {name: rel_fro(gr32[name], gr64[name]) for name in gr32}
    if opt is not None:
        after = []
        for gr in (gr32, gr64):
            P2 = {kk: v.copy() for kk, v in Pc.items()}
            g2 = {kk: np.asarray(v, np.float32).copy() for kk, v in gr.items()}
            m2, v2 = ({kk: v.copy() for kk, v in opt[which].items()} for which in ("m", "v"))
            O.clip_and_project(P2, g2, 1.0)
            O.adam_step(P2, g2, m2, v2, lr, step)
            after.append(P2)
        for name in Pc:
            floor["param:" + name] = rel_fro(after[0][name], after[1][name])
    return floor


def run_param_floor(P0, since0, batches, k, targets=None, **kw):
    """name -> rel-Frobenius distance between the parameters the ORACLE ends at when it carries len(batches) train steps in fp32 and in
    float64 (from the same fp32 start): what two correct computations of this run are apart.  Parameter bounds of multi-step
    comparisons are 6 x this (as ghost_tolerances) where that exceeds 1e-4; round 5 had the constant 1e-3 there."""
    ends = []
    for dt in (np.float32, np.float64):
        P = {n: np.asarray(v, dt).copy() for n, v in P0.items()}
        opt = {w: {n: np.zeros_like(v) for n, v in P.items()} for w in ("m", "v")}
        stats = {"n_fwd_since_fired": np.asarray(since0, np.float32).copy(), "act_freq_scores": np.zeros(len(since0), np.float32)}
        for t, x in enumerate(batches):
            O.train_step(P, opt, stats, np.asarray(x, dt), k, step=t + 1, target=None if targets is None else np.asarray(targets[t], dt), **kw)
        ends.append(P)
    return {n: rel_fro(ends[0][n], ends[1][n]) for n in P0}


def ghost_tolerances(floor):
    """name -> max(TOL, 6 x floor[name]) (and "max": the largest over the gradient tensors).  6 x: numpy's matmul sums in blocks, the
    MFMA chain of the kernels sums the K = d_in (768) products of an entry in k order -- measured on the GPU, the kernels sit at
    1.2 ... 3.5 x the oracle's own floor (3.5 x: gW_dec at 768 -> 49152).  The bound so derived is 2e-4 ... 5e-4 depending on the shape
    (the loose end only without LayerNorm, where |hidden_pre| reaches ~50); no constant beside it (round 5 capped it at 5e-4 and kept
    5e-4 / 1e-3 as fallbacks)."""
    tol = {name: max(TOL, 6.0 * v) for name, v in floor.items()}
    tol["max"] = max(v for name, v in tol.items() if not name.startswith("param:"))
    return tol


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("d_in,d_sae,n,ln,ghost", [(64, 512, 256, True, False), (136, 1056, 300, False, False), (768, 8192, 1024, True, False),
                                                   (64, 512, 256, True, True), (136, 1056, 300, False, True), (768, 8192, 1024, True, True),
                                                   (768, 24576, 4096, True, False), (768, 24576, 4096, True, True),
                                                   (768, 49152, 1024, True, False), (768, 49152, 1024, True, True),
                                                   (1280, 10240, 512, True, True)])
def test_relu_l1_dense_step_vs_oracle(d_in, d_sae, n, ln, ghost):
    """pv_sae_dense_step + grad_sqnorm + apply against the ReLU + L1 form of the oracle (pinned to the reference fixtures by
    tests/test_oracle_sae_vs_golden.py): losses, l0, every gradient tensor, parameters, statistics; ragged shapes (partial
    tiles in M, N and K) and the no-LayerNorm form included.  ghost: use_ghost_grads with every fifth feature counted as
    dead (sae.py:151-179): the ghost residual loss and its gradient through the dead columns."""
    l1c, window = 3e-3, 3
    # ghost: exp(hidden_pre) turns the ABSOLUTE fp32 summation noise of hidden_pre into a RELATIVE error of the ghost
    # activations (d exp(h) / exp(h) = dh): without LayerNorm |hidden_pre| reaches ~50 here and the dead rows of gW_dec carry
    # ~1e-4 of it (the reference run on a GPU would differ from its CPU run by as much); after one such step the two
    # parameter sets are ~1e-4 apart and later steps are not comparable gate for gate: one step, every bound derived from the
    # oracle's own fp32-vs-float64 distance on this batch (fp32_noise_floor)
    P, opt, stats, T = fresh(d_in, d_sae)
    if ghost:
        stats["n_fwd_since_fired"][::5] = 10.0
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], 1, ln, n)
    eng.n_fwd_since_fired.copy_(torch.from_numpy(stats["n_fwd_since_fired"]))
    for t in range(1 if ghost else 2):
        x = synth_sae_batch(n, d_in, seed=t)
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        dead = (stats["n_fwd_since_fired"] > window) if ghost else None
        fw = O.sae_forward(Pc, x, None, layer_norm=ln, l1_coefficient=l1c, dead_mask=dead)
        gr = O.sae_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c)
        before = stats["act_freq_scores"].copy()
        ref = fw_scalars(fw)
        eng.dense_step(torch.from_numpy(x).cuda(), l1c, want_out=True,
                       dead_mask=(eng.n_fwd_since_fired > window) if ghost else None)
        eng.grad_sqnorm()
        torch.cuda.synchronize()
        sc = eng.scalars.cpu().numpy()
        assert abs(sc[0] - ref["loss"]) <= TOL * ref["loss"] and abs(sc[1] - ref["mse_loss"]) <= TOL * ref["mse_loss"], (sc, ref)
        # (an activation within fp32 summation noise of zero may fall on either side of the ReLU: 18 of 8.4 M at the largest shape)
        assert abs(sc[4] - ref["l1_loss"]) <= TOL * ref["l1_loss"] and abs(sc[2] - ref["l0"]) <= TOL * ref["l0"], (sc, ref)
        if ghost:
            assert abs(sc[5] - ref["ghost_loss"]) <= TOL * ref["ghost_loss"], (sc, ref)
        assert rel_fro(eng.sae_out[:n].cpu().numpy(), fw["sae_out"]) < TOL
        # The backward is discontinuous in the sign of hidden_pre: where |hidden_pre| is within fp32 summation noise of zero
        # the kernel's gate may differ from the oracle's.  The kernel's gates are read back (dH != 0, left in the workspace),
        # must differ from the oracle's only on such entries, and the gradients are compared under the kernel's gates.
        off = eng.lib.pv_debug_sae_ws_offset(eng._plan, b"hidden")
        dH = eng.workspace[off:off + n * d_sae * 4].view(torch.float32).view(n, d_sae).cpu().numpy()
        gate = dH != 0
        if ghost:
            gate[:, dead] = fw["feature_acts"][:, dead] > 0      # (the ghost term makes dH nonzero on dead columns whatever the gate)
        differs = gate != (fw["feature_acts"] > 0)
        assert differs.sum() <= 1e-5 * gate.size and np.all(np.abs(fw["hidden_pre"][differs]) < 1e-5 * np.abs(fw["hidden_pre"]).max())
        if differs.any():
            gr = O.sae_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c, gate=gate)
        # ghost: per-tensor bounds DERIVED from the oracle's own fp32-vs-float64 distance on this very batch (fp32_noise_floor)
        gt = ghost_tolerances(fp32_noise_floor(Pc, x, None, ln, dead, l1c, gate if differs.any() else None, opt=opt, step=t + 1)) if ghost else None
        tol_of = (lambda name: gt[name]) if gt else (lambda name: TOL)
        assert abs(np.sqrt(sc[3]) - grad_norm_of(gr)) <= (gt["max"] if gt else TOL) * grad_norm_of(gr)
        assert rel_fro(eng.grad_W_enc().cpu().numpy(), gr["W_enc"]) < tol_of("W_enc")
        for name in ("W_dec", "b_enc", "b_dec"):
            assert rel_fro(eng.g[name].cpu().numpy(), gr[name]) < tol_of(name), name
        # the oracle's step continues under the KERNEL's gates where they differ (round 5 stopped the comparison here)
        O.train_step(P, opt, stats, x, None, lr=1e-3, step=t + 1, layer_norm=ln, l1_coefficient=l1c,
                     dead_feature_window=window if ghost else None, gate=gate if differs.any() else None)
        fire_ref = stats["act_freq_scores"] - before
        assert np.abs(eng.fire_count.cpu().numpy() - fire_ref).sum() <= TOL * fire_ref.sum()
        eng.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for name in P:
            assert rel_fro(eng.params[name].cpu().numpy(), P[name]) < tol_of("param:" + name), name
        assert np.abs(eng.act_freq_scores.cpu().numpy() - stats["act_freq_scores"]).sum() <= TOL * stats["act_freq_scores"].sum()
        assert np.abs(eng.n_fwd_since_fired.cpu().numpy() - stats["n_fwd_since_fired"]).sum() <= 2


@pytest.mark.parametrize("kind", ["relu", "gated"])
def test_dense_steps_see_an_outside_edit_of_w_enc(kind):
    """dense_step / gated_step read the encoder through its transposed master W_encT: an in-place edit of the W_enc parameter between
    two steps (load_state_dict, an optimizer of the caller, resampling: the version counter moves) must reach the kernels, and the
    following apply must not write the stale copy back over it."""
    d_in, d_sae, n, l1c = 64, 512, 256, 3e-3
    P, opt, stats, T = fresh(d_in, d_sae)
    kw = {}
    if kind == "gated":
        rs = np.random.RandomState(9)
        kw["gated"] = {m: torch.from_numpy((rs.standard_normal(d_sae) * s_).astype(np.float32)).cuda()
                       for m, s_ in (("b_gate", 0.05), ("r_mag", 0.2), ("b_mag", 0.05))}
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=0)).cuda()

    def run(eng):
        (eng.gated_step if kind == "gated" else eng.dense_step)(x, l1c, want_out=True)
        torch.cuda.synchronize()
        return eng.sae_out[:n].clone(), eng.scalars.clone()

    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], 1, True, n, **kw)
    out0, _ = run(eng)
    with torch.no_grad():
        T["W_enc"].mul_(0.5)                                       # the outside edit (bumps the version counter)
    # a fresh engine over a copy of the edited parameters is the truth (copied BEFORE the next step: a step renormalises W_dec in place)
    T2 = {m: v.clone() for m, v in T.items()}
    kw2 = {"gated": {m: v.clone() for m, v in kw["gated"].items()}} if kind == "gated" else {}
    out1, sc1 = run(eng)
    assert rel_fro(out0.cpu().numpy(), out1.cpu().numpy()) > 1e-2   # the edit reached the kernels
    ref = NativeSAE(T2["W_enc"], T2["W_dec"], T2["b_enc"], T2["b_dec"], 1, True, n, **kw2)
    out_ref, sc_ref = run(ref)
    assert torch.equal(out1, out_ref) and torch.equal(sc1[:3], sc_ref[:3])
    for e in (eng, ref):
        e.grad_sqnorm()
        e.apply(1e-3, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(eng.W_encT, ref.W_encT) and torch.equal(T["W_enc"], T2["W_enc"])


@pytest.mark.parametrize("d_in,d_sae,k,n,ln", [(64, 512, 8, 256, True), (136, 1056, 16, 300, False), (768, 8192, 32, 1024, True)])
def test_topk_ghost_step_vs_oracle(d_in, d_sae, k, n, ln):
    """Ghost gradients on the top-k step (pv_sae_step + pv_sae_topk_ghost; sae.py:151-179 behind TopK :795-810) against the oracle's
    top-k + ghost form (pinned to the reference's own topk_ghost run by tests/test_oracle_sae_vs_golden.py): every fifth feature counts
    as dead; losses (mse, ghost, total), l0, the reconstruction, every gradient tensor, the clip norm.  One step, gradients and post-step
    parameters held to bounds derived from the oracle's own fp32-vs-float64 distance where the dead rows are concerned: exp(hidden_pre) turns the absolute fp32 summation noise of hidden_pre into a relative error
    of the ghost activations (see test_relu_l1_dense_step_vs_oracle)."""
    window = 3
    P, opt, stats, T = fresh(d_in, d_sae)
    stats["n_fwd_since_fired"][::5] = 10.0
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, ln, n)
    eng.n_fwd_since_fired.copy_(torch.from_numpy(stats["n_fwd_since_fired"]))
    x = synth_sae_batch(n, d_in, seed=10)
    Pc = {kk: v.copy() for kk, v in P.items()}
    O.renorm_decoder(Pc)
    dead = stats["n_fwd_since_fired"] > window
    fw = O.sae_forward(Pc, x, k, layer_norm=ln, dead_mask=dead)
    gr = O.sae_backward(Pc, x, fw, layer_norm=ln)
    before = stats["act_freq_scores"].copy()
    gt = ghost_tolerances(fp32_noise_floor(Pc, x, k, ln, dead, opt=opt))      # derived per-tensor bounds, gradients and post-step parameters
    ref = O.train_step(P, opt, stats, x, k, lr=1e-3, step=1, layer_norm=ln, dead_feature_window=window)
    xg = torch.from_numpy(x).cuda()
    dead_g = eng.n_fwd_since_fired > window                          # (before the step's statistics, train_sae.py:330-332)
    eng.renorm_decoder()
    eng.step(xg, want_out=True, renorm_decoder=False, sparse_grads=False)
    eng.topk_ghost(xg, dead_g)
    eng.grad_sqnorm()
    torch.cuda.synchronize()
    sc = eng.scalars.cpu().numpy()
    assert abs(sc[0] - ref["loss"]) <= TOL * ref["loss"] and abs(sc[1] - ref["mse_loss"]) <= TOL * ref["mse_loss"], (sc, ref)
    assert abs(sc[5] - ref["ghost_loss"]) <= TOL * ref["ghost_loss"] and abs(sc[2] - ref["l0"]) < 1e-4, (sc, ref)
    assert np.array_equal(np.sort(eng.topk_idx[:n].cpu().numpy(), axis=1), np.sort(fw["idx"], axis=1))
    assert rel_fro(eng.sae_out[:n].cpu().numpy(), fw["sae_out"]) < TOL
    tol_of = lambda name: gt[name]
    assert abs(np.sqrt(sc[3]) - grad_norm_of(gr)) <= gt["max"] * grad_norm_of(gr)
    assert rel_fro(eng.grad_W_enc().cpu().numpy(), gr["W_enc"]) < tol_of("W_enc")
    for name in ("W_dec", "b_enc", "b_dec"):
        assert rel_fro(eng.g[name].cpu().numpy(), gr[name]) < tol_of(name), name
    # the live features' rows are what the plain top-k step gives them (the ghost term reaches dead features only)
    live = ~dead
    assert rel_fro(eng.g["W_dec"].cpu().numpy()[live], gr["W_dec"][live]) < TOL
    assert np.array_equal(eng.fire_count.cpu().numpy(), stats["act_freq_scores"] - before)
    eng.apply(1e-3, 1.0)
    torch.cuda.synchronize()
    for name in P:
        # (Adam's g / (|g| + 1e-8) on the dead features' ~1e-9 gradient entries: the bound is the oracle's own fp32-vs-float64 distance
        # of the post-step parameters, see fp32_noise_floor)
        assert rel_fro(eng.params[name].cpu().numpy(), P[name]) < tol_of("param:" + name), name


def test_topk_ghost_trainer_runs_natively_and_matches_the_reference_fixture():
    """activation_fn_str = "topk" with use_ghost_grads through VisionSAETrainer.train_step on the HIP path, against what the REFERENCE's
    own classes produced through its own train_step (topk_ghost of tests/golden/sae_variants_steps.npz): three steps, losses and the
    ghost loss at 1e-4, statistics (a handful of entries may differ after a ghost step, as for the ReLU form), parameters at 6 x the
    oracle's own fp32-vs-float64 distance over the three steps times GHOST_RUN_X (run_param_floor)."""
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, exp, N = 64, 8, 256
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp, activation_fn_str="topk",
        activation_fn_kwargs={"k": 8}, normalize_activations="layer_norm", initialization_method="independent", b_dec_init_method="mean",
        train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cuda", _dtype="float32", log_to_wandb=False, use_ghost_grads=True,
        feature_sampling_window=1000, dead_feature_window=1, lr_scheduler_name="constant", n_checkpoints=0, verbose=False)
    tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True)
    model = tr.sparse_coder
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"topk_ghost_init_{n}"]).cuda())
    act, since, frac, opt, sched = tr.initialize_training_variables()
    since.copy_(torch.from_numpy(g["topk_ghost_since0"]).cuda())
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(N, d_in, seed=t)).cuda()[:, None, :]
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=x, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native and l1 is None
        want = g[f"topk_ghost_s{t}_scalars"]
        for got, w in ((loss, want[0]), (mse, want[1]), (l0, want[3]), (tr._engine.scalars[5], want[4])):
            assert abs(float(got) - w) <= TOL * abs(w), (t, float(got), w)
        af = g[f"topk_ghost_s{t}_act_freq"]
        assert np.abs(act.cpu().numpy() - af).sum() <= 1e-3 * af.sum()
        assert (since.cpu().numpy() != g[f"topk_ghost_s{t}_n_since"]).sum() <= 2
    floor = run_param_floor({n: g[f"topk_ghost_init_{n}"] for n, _ in model.named_parameters()}, g["topk_ghost_since0"],
                            [synth_sae_batch(N, d_in, seed=t) for t in range(3)], 8, lr=1e-3, dead_feature_window=1)
    for n, p in model.named_parameters():
        assert rel_fro(p.detach().cpu().numpy(), g[f"topk_ghost_s2_param_{n}"]) < max(TOL, GHOST_RUN_X * floor[n]), (n, floor[n])


def grad_norm_of(g):
    return float(np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g.values())))


@pytest.mark.parametrize("variant", ["relu_l1", "relu_ghost"])
def test_relu_variants_trainer_runs_natively_and_matches_the_reference_fixture(variant):
    """activation_fn_str = "relu" (with and without use_ghost_grads) through VisionSAETrainer.train_step on the dense HIP
    step, against what the REFERENCE's own classes produced through its own train_step
    (tests/golden/sae_variants_steps.npz): three steps, scalars at 1e-4, statistics exact, parameters after step 3 at 1e-4."""
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, exp, N = 64, 8, 256
    ghost = variant == "relu_ghost"
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp, activation_fn_str="relu",
        activation_fn_kwargs={}, normalize_activations="layer_norm", initialization_method="independent", b_dec_init_method="mean",
        train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cuda", _dtype="float32", log_to_wandb=False, use_ghost_grads=ghost,
        feature_sampling_window=1000, dead_feature_window=1 if ghost else 5000, lr_scheduler_name="constant", n_checkpoints=0,
        verbose=False, l1_coefficient=2e-3)
    tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True)
    model = tr.sparse_coder
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"{variant}_init_{n}"]).cuda())
    act, since, frac, opt, sched = tr.initialize_training_variables()
    since.copy_(torch.from_numpy(g[f"{variant}_since0"]).cuda())
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(N, d_in, seed=t)).cuda()[:, None, :]
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=x, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native
        want = g[f"{variant}_s{t}_scalars"]
        for got, w in ((loss, want[0]), (mse, want[1]), (l1, want[2]), (l0, want[3])):
            assert abs(float(got) - w) <= TOL * abs(w), (t, float(got), w)
        if ghost:
            assert abs(float(tr._engine.scalars[5]) - want[4]) <= TOL * abs(want[4])
        if not ghost:
            assert np.array_equal(act.cpu().numpy(), g[f"{variant}_s{t}_act_freq"]) and np.array_equal(since.cpu().numpy(), g[f"{variant}_s{t}_n_since"])
        else:
            # (after a step with ghost gradients the parameters agree to ~1e-4, not 1e-6: a handful of activations within that
            # distance of zero fall on the other side of the ReLU)
            af = g[f"{variant}_s{t}_act_freq"]
            assert np.abs(act.cpu().numpy() - af).sum() <= 1e-4 * af.sum()
            assert (since.cpu().numpy() != g[f"{variant}_s{t}_n_since"]).sum() <= 1
    # ghost: the dead features' only gradient is the ghost term -- entries of ~1e-9, where fp32 summation-order noise is an
    # ABSOLUTE error Adam's g / (|g| + 1e-8) turns into lr-sized differences on a few elements (measured 3.4e-4 on W_enc after
    # three steps; losses, the north-star quantity, stay within 1e-4 at every step above)
    floor = run_param_floor({n: g[f"{variant}_init_{n}"] for n, _ in model.named_parameters()}, g[f"{variant}_since0"],
                            [synth_sae_batch(N, d_in, seed=t) for t in range(3)], None, lr=1e-3, l1_coefficient=2e-3,
                            dead_feature_window=1 if ghost else None) if ghost else None
    for n, p in model.named_parameters():
        assert rel_fro(p.detach().cpu().numpy(), g[f"{variant}_s2_param_{n}"]) < (max(TOL, GHOST_RUN_X * floor[n]) if ghost else TOL), (n, floor and floor[n])


@pytest.mark.parametrize("variant", ["relu_constnorm", "topk_constnorm", "tanh_relu", "relu_lp2"])
def test_tail_variants_trainer_runs_natively_and_matches_the_reference_fixture(variant):
    """Round 6, the tail of SURVEY.md 8(f) row 3 on the HIP steps: normalize_activations = "constant_norm_rescale" (sae.py:60-72; a third
    mode of sae_prep_kernel) on the ReLU + L1 and the top-k step, activation_fn_str = "tanh-relu" (:823-830) and lp_norm = 2 (:617) in the
    dense step's epilogues -- each through VisionSAETrainer.train_step against what the REFERENCE's own classes produced through its own
    train_step (tests/golden/sae_tail_steps.npz, tests/golden/gen_golden_sae_tail.py): three steps, scalars at 1e-4, statistics exact,
    parameters after step 3 at 1e-4."""
    g = np.load(os.path.join(GOLDEN, "sae_tail_steps.npz"))
    d_in, exp, N = 64, 8, 256
    over = {"relu_constnorm": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, normalize_activations="constant_norm_rescale"),
            "topk_constnorm": dict(activation_fn_str="topk", activation_fn_kwargs={"k": 8}, normalize_activations="constant_norm_rescale"),
            "tanh_relu": dict(activation_fn_str="tanh-relu", activation_fn_kwargs={}, l1_coefficient=2e-3),
            "relu_lp2": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, lp_norm=2)}[variant]
    kw = dict(hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp, normalize_activations="layer_norm",
              initialization_method="independent", b_dec_init_method="mean", train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cuda",
              _dtype="float32", log_to_wandb=False, use_ghost_grads=False, feature_sampling_window=1000, dead_feature_window=5000,
              lr_scheduler_name="constant", n_checkpoints=0, verbose=False)
    kw.update(over)
    cfg = VisionModelSAERunnerConfig(**kw)
    tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True)
    model = tr.sparse_coder
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"{variant}_init_{n}"]).cuda())
    act, since, frac, opt, sched = tr.initialize_training_variables()
    since.copy_(torch.from_numpy(g[f"{variant}_since0"]).cuda())
    topk = cfg.activation_fn_str == "topk"
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(N, d_in, seed=t)).cuda()[:, None, :]
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=x, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native, tr._native_why_not(model)
        want = g[f"{variant}_s{t}_scalars"]
        for got, w in ((loss, want[0]), (mse, want[1]), (l0, want[3])) + (() if topk else ((l1, want[2]),)):
            assert abs(float(got) - w) <= TOL * abs(w), (t, float(got), w)
        assert np.array_equal(act.cpu().numpy(), g[f"{variant}_s{t}_act_freq"]) and np.array_equal(since.cpu().numpy(), g[f"{variant}_s{t}_n_since"])
    tr.sync_parameters()
    for n, p in model.named_parameters():
        assert rel_fro(p.detach().cpu().numpy(), g[f"{variant}_s2_param_{n}"]) < TOL, n


@pytest.mark.parametrize("variant", ["tc_topk_ghost", "tc_relu_ghost"])
def test_transcoder_ghost_trainer_runs_natively_and_matches_the_reference_fixture(variant):
    """Round 6: ghost gradients on a Transcoder on the HIP steps (pv_sae_topk_ghost / pv_sae_dense_step with a transcoder state: the ghost
    term sees the INPUT activation, transcoder.py:82-86 + sae.py:151-179) through VisionSAETrainer.train_step, against what the REFERENCE's
    own Transcoder produced through its own train_step (tests/golden/sae_tail_steps.npz): top-k with the skip connection, ReLU + L1
    without; three steps, losses and the ghost loss at 1e-4, statistics (a handful of entries may differ after a ghost step), parameters at
    6 x the oracle's own fp32-vs-float64 distance over the three steps."""
    g = np.load(os.path.join(GOLDEN, "sae_tail_steps.npz"))
    d_in, exp, N = 64, 8, 256
    topk = variant == "tc_topk_ghost"
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp, is_transcoder=True, d_out=d_in,
        out_hook_point_layer=6, transcoder_with_skip_connection=topk, activation_fn_str="topk" if topk else "relu",
        activation_fn_kwargs={"k": 8} if topk else {}, l1_coefficient=2e-3, normalize_activations="layer_norm",
        initialization_method="independent", b_dec_init_method="mean", train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cuda",
        _dtype="float32", log_to_wandb=False, use_ghost_grads=True, feature_sampling_window=1000, dead_feature_window=1,
        lr_scheduler_name="constant", n_checkpoints=0, verbose=False)
    tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True)
    model = tr.sparse_coder
    assert [n for n, _ in model.named_parameters()] == [str(n) for n in g[f"{variant}_keys"]]
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"{variant}_init_{n}"]).cuda())
    act, since, frac, opt, sched = tr.initialize_training_variables()
    since.copy_(torch.from_numpy(g[f"{variant}_since0"]).cuda())
    xs = [synth_sae_batch(N, d_in, seed=t) for t in range(3)]
    ys = [synth_sae_batch(N, d_in, seed=100 + t) for t in range(3)]
    for t in range(3):
        layer_acts = torch.stack([torch.from_numpy(xs[t]), torch.from_numpy(ys[t])], dim=1).cuda()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=layer_acts, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native, tr._native_why_not(model)
        want = g[f"{variant}_s{t}_scalars"]
        for got, w in ((loss, want[0]), (mse, want[1]), (l0, want[3]), (tr._engine.scalars[5], want[4])) + (() if topk else ((l1, want[2]),)):
            assert abs(float(got) - w) <= TOL * abs(w), (t, float(got), w)
        af = g[f"{variant}_s{t}_act_freq"]
        assert np.abs(act.cpu().numpy() - af).sum() <= 1e-3 * af.sum()
        assert (since.cpu().numpy() != g[f"{variant}_s{t}_n_since"]).sum() <= 2
    tr.sync_parameters()
    names = [n for n, _ in model.named_parameters()]
    floor = run_param_floor({n: g[f"{variant}_init_{n}"] for n in names}, g[f"{variant}_since0"], xs, 8 if topk else None, targets=ys,
                            lr=1e-3, dead_feature_window=1, l1_coefficient=0.0 if topk else 2e-3)
    # the top-k form's ghost gradient is ill-conditioned in fp32 where the ghost reconstruction meets the residual (r = mse / (mg + 1e-6)
    # with mg -> 0; the reason test_multi_rank_steps_on_rccl_world1 compares topk_ghost_dp with the single-process engine): measured
    # 2.4e-3 on b_enc (a third of its entries belong to dead features whose only gradient is that term) with every loss, the ghost loss
    # included, within 1e-4 at each of the three steps
    ill = 5e-3 if topk else 0.0
    for n, p in model.named_parameters():
        assert rel_fro(p.detach().cpu().numpy(), g[f"{variant}_s2_param_{n}"]) < max(TOL, GHOST_RUN_X * floor[n], ill), (n, floor[n])


def test_step_is_bit_reproducible_from_run_to_run():
    """The position of a pair inside its feature's list is drawn by an integer atomic in the selection kernel; the lists are
    put into token order before the backward (csr_sort_short_kernel, sae_long_sort_kernel), so nothing the step computes depends
    on that draw: two engines from the same state on the same batch agree BIT FOR BIT -- reconstruction, losses, every gradient
    tensor, parameters and moments after the optimizer step -- at the bench shape, three steps in a row."""
    d_in, d_sae, k, n = 768, 24576, 32, 4096
    engines = []
    for _ in range(2):
        _, _, _, T = fresh(d_in, d_sae)
        engines.append(NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n))
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
        for e in engines:
            e.step(x, want_out=True, renorm_decoder=True)
            e.grad_sqnorm(from_step=True)
        torch.cuda.synchronize()
        a, b = engines
        assert torch.equal(a.sae_out, b.sae_out) and torch.equal(a.scalars, b.scalars)
        assert torch.equal(torch.sort(a.topk_idx[:n], dim=1).values, torch.sort(b.topk_idx[:n], dim=1).values)
        assert torch.equal(a.flat_g, b.flat_g), float((a.flat_g - b.flat_g).abs().max())
        for e in engines:
            e.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for name in ("W_dec", "b_enc", "b_dec"):
            assert torch.equal(a.params[name], b.params[name]), name
        assert torch.equal(a.W_encT, b.W_encT) and torch.equal(a.flat_m, b.flat_m) and torch.equal(a.flat_v, b.flat_v)


@pytest.mark.parametrize("d_in,d_sae,k,n,sparse", [(768, 24576, 32, 4096, True), (768, 24576, 32, 4096, False), (256, 4096, 16, 1000, True),
                                                   (1024, 8192, 64, 777, True)])
def test_folded_launches_equal_the_single_launches_bitwise(d_in, d_sae, k, n, sparse):
    """Round 6 folded the step's launch-bound kernels into their neighbours -- batch-mean partials and the weight bound into the prep
    launch, the mean's second stage into the threshold launch, the loss normaliser into an idle wave of the select kernel, the CSR
    scan into the decode launch, the loss into the post + fill launch, the two list sorts into one launch, the clip norm into the last
    column sum (PV_SAE_FUSED_SQNORM), the bias vectors' Adam into the encoder's -- each in the arithmetic ORDER of the launch it
    replaces.  So the folded step (tuning key sae_fold = 1, the default) and the step of single launches (sae_fold = 0; the clip
    norm's block-wise sum from a launch of its own) agree BIT FOR BIT: reconstruction, scalars (loss, mse, l0, clip norm), the selected sets, every
    gradient, parameters and moments after the optimizer step, firing statistics -- three steps in a row, ragged token counts too."""
    from vit_prisma_amd import _native as NV
    engines = []
    for _ in range(2):
        _, _, _, T = fresh(d_in, d_sae)
        engines.append(NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n))
    try:
        for t in range(3):
            x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
            for fold, e in enumerate(engines):
                NV.set_tuning("sae_fold", fold)
                e.step(x, want_out=True, renorm_decoder=True, sparse_grads=sparse, fused_sqnorm=True)
                e.grad_sqnorm(from_step=True)                   # (nothing to do: the step left the clip norm)
            torch.cuda.synchronize()
            a, b = engines
            assert a._sq_fused and b._sq_fused
            # the block-wise clip norm against pv_sae_grad_sqnorm_step's (the same terms in another order: fp32 summation noise apart)
            fused_sq = float(b.scalars[3])
            b._sq_fused = False
            b.grad_sqnorm(from_step=True)
            torch.cuda.synchronize()
            assert abs(float(b.scalars[3]) - fused_sq) <= 2e-6 * fused_sq, (float(b.scalars[3]), fused_sq)
            b.scalars[3] = fused_sq
            assert torch.equal(a.sae_out, b.sae_out)
            assert torch.equal(a.scalars[:4], b.scalars[:4]), (a.scalars.tolist(), b.scalars.tolist())
            assert torch.equal(torch.sort(a.topk_idx[:n], dim=1).values, torch.sort(b.topk_idx[:n], dim=1).values)
            assert torch.equal(a.fire_count, b.fire_count) and torch.equal(a.act_freq_scores, b.act_freq_scores)
            if not sparse:
                assert torch.equal(a.flat_g, b.flat_g), float((a.flat_g - b.flat_g).abs().max())
            for fold, e in enumerate(engines):
                NV.set_tuning("sae_fold", fold)
                e.apply(1e-3, 1.0)
            torch.cuda.synchronize()
            for name in ("W_dec", "b_enc", "b_dec"):
                assert torch.equal(a.params[name], b.params[name]), name
            assert torch.equal(a.W_encT, b.W_encT) and torch.equal(a.flat_m, b.flat_m) and torch.equal(a.flat_v, b.flat_v)
    finally:
        NV.set_tuning("reset")


@pytest.mark.parametrize("case", ["range", "pair_ties", "massive_ties", "weight_range"])
def test_folded_step_recomputes_undecided_tokens_inline_and_equals_the_listed_form(case):
    """The inline exact path of the select kernel (tuning key sae_inline_fb; off by default: one undecided token costs the step 0.3 ms of
    latency, MEASURED.md): a token the filter cannot decide (outside the fp16 range, candidate-list overflow from massive ties, a
    poisoned weight bound) is recomputed exactly inside the select kernel by its own workgroup, in the arithmetic of the two fallback
    kernels.  On the adversarial inputs of test_filtered_encoder_exact_fallback_* the folded step with the inline path and the step of
    single launches that lists such tokens for sae_fb_hidden_kernel + sae_topk_kernel count the same undecided tokens and agree bit
    for bit -- reconstruction, scalars, gradients, parameters after the optimizer step.  (The folded step's DEFAULT form, which lists
    them too, is covered by the same comparison in test_folded_launches_* on ordinary data and by case "listed" here.)"""
    from vit_prisma_amd import _native as NV
    d_in, d_sae, k, n = 64, 4096, 8, 40
    engines, states = [], []
    for _ in range(2):
        _, _, _, T = fresh(d_in, d_sae)
        with torch.no_grad():
            if case == "pair_ties":
                T["W_enc"][:, 1::2] = T["W_enc"][:, 0::2]
                T["b_enc"][1::2] = T["b_enc"][0::2]
            if case in ("massive_ties", "weight_range"):
                T["W_enc"][:, :2048] = T["W_enc"][:, :1].clone()
                T["b_enc"][:2048] = 5.0
            if case == "weight_range":
                T["W_enc"][3, 77] = 1.0e6
        states.append(T)
        engines.append(NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, False, n))     # (no input LayerNorm: raw magnitudes reach the GEMM)
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=2)).cuda()
    if case == "range":
        x[3, 5] = 3.0e5
        x[7] *= 1e-7
    try:
        for t in range(2):
            counts = []
            for fold, e in enumerate(engines):
                NV.set_tuning("sae_fold", fold)
                NV.set_tuning("sae_inline_fb", fold)
                e.step(x, want_out=True, renorm_decoder=True, fused_sqnorm=True)
                counts.append(e.fallback_rows())
            torch.cuda.synchronize()
            a, b = engines
            assert counts[0] == counts[1] and (counts[0] == n if case in ("massive_ties", "weight_range") else counts[0] >= (1 if case == "range" else 0)), counts
            assert torch.equal(a.sae_out, b.sae_out) and torch.equal(a.scalars[:4], b.scalars[:4])
            assert torch.equal(a.topk_val[:n].sort(dim=1).values, b.topk_val[:n].sort(dim=1).values)
            assert torch.equal(a.flat_g, b.flat_g)
            for fold, e in enumerate(engines):
                NV.set_tuning("sae_fold", fold)
                e.apply(1e-3, 1.0)
            torch.cuda.synchronize()
            assert torch.equal(a.W_encT, b.W_encT) and torch.equal(a.params["W_dec"], b.params["W_dec"]) and torch.equal(a.params["b_enc"], b.params["b_enc"])
    finally:
        NV.set_tuning("reset")


def test_sample_pass_on_128_row_tiles_equals_256_row_tiles_bitwise():
    """The sample GEMM of the filtered encoder (every 16th feature: 4096 x 1536 outputs at the bench shape) runs on 128 x 256 tiles where
    256 x 256 ones would occupy 96 of 256 CUs.  An output element's K order does not depend on the tile it sits in, so the thresholds,
    the candidate lists and everything behind them are those of the 256-row form (tuning key enc_tm256 = 1) to the bit."""
    from vit_prisma_amd import _native as NV
    d_in, d_sae, k, n = 768, 24576, 32, 4000                    # (ragged last M tile)
    engines = []
    for _ in range(2):
        _, _, _, T = fresh(d_in, d_sae)
        engines.append(NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n))
    try:
        for t in range(2):
            x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
            for big, e in enumerate(engines):
                NV.set_tuning("enc_tm256", big)
                e.step(x, want_out=True, renorm_decoder=True, fused_sqnorm=True)
                e.apply(1e-3, 1.0)
            torch.cuda.synchronize()
            a, b = engines
            ws = NV  # noqa: F841
            assert torch.equal(a.sae_out, b.sae_out) and torch.equal(a.scalars[:4], b.scalars[:4])
            assert torch.equal(torch.sort(a.topk_idx[:n], dim=1).values, torch.sort(b.topk_idx[:n], dim=1).values)
            assert torch.equal(a.W_encT, b.W_encT) and torch.equal(a.params["W_dec"], b.params["W_dec"])
    finally:
        NV.set_tuning("reset")


# ---------------------------------------------------------------------------------------------------
# Transcoder (sae/transcoder.py; pv_sae_transcoder) on the two fused steps
# ---------------------------------------------------------------------------------------------------
def fresh_transcoder(d_in, d_sae, skip):
    P, opt, stats, T = fresh(d_in, d_sae)
    rs = np.random.RandomState(5)
    P["b_dec_out"] = (rs.standard_normal(d_in) * 0.05).astype(np.float32)
    if skip:
        P["W_skip"] = (rs.standard_normal((d_in, d_in)) / np.sqrt(d_in) * 0.3).astype(np.float32)
    for n in ("b_dec_out", "W_skip"):
        if n in P:
            opt["m"][n], opt["v"][n] = np.zeros_like(P[n]), np.zeros_like(P[n])
            T[n] = torch.from_numpy(P[n].copy()).cuda()
    return P, opt, stats, T


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("d_in,d_sae,k,n,ln,skip", [(64, 512, 8, 256, True, True), (136, 1056, 16, 300, False, True),
                                                    (768, 8192, 32, 1024, True, True), (768, 8192, 32, 1024, True, False),
                                                    (64, 512, None, 256, True, True), (136, 1056, None, 300, False, False),
                                                    (768, 8192, None, 1024, True, True),
                                                    # the benchmarked shape (768 -> 24576, 4096 tokens) and the published x64 width, both steps
                                                    (768, 24576, 32, 4096, True, True), (768, 24576, None, 4096, True, True),
                                                    (768, 49152, 32, 1024, True, True), (768, 49152, None, 1024, True, False),
                                                    (1280, 10240, 32, 512, True, True)])
def test_transcoder_steps_vs_oracle(d_in, d_sae, k, n, ln, skip):
    """A Transcoder (target activation, b_dec_out, optional W_skip) on the top-k step (k given) and on the dense ReLU + L1 step
    (k = None) against the oracle's transcoder form (pinned to the reference's own Transcoder run by
    tests/test_oracle_sae_vs_golden.py): losses, l0, every gradient tensor incl. the two extra ones, the clip norm,
    parameters and statistics after the optimizer step; ragged shapes and the no-LayerNorm form included."""
    l1c = 3e-3
    P, opt, stats, T = fresh_transcoder(d_in, d_sae, skip)
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k or 1, ln, n, b_dec_out=T["b_dec_out"], W_skip=T.get("W_skip"))
    assert eng.transcoder
    for t in range(2):
        # (seeds 10..: token 381 of seed 0 has its 32nd and 33rd pre-activation at 768 -> 8192 within fp32 summation noise of each
        # other -- profiles/r03_tp_near_tie_diag.txt -- and the top-k SET is what this test compares)
        x, y = synth_sae_batch(n, d_in, seed=10 + t), synth_sae_batch(n, d_in, seed=50 + t)
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        fw = O.sae_forward(Pc, x, k, layer_norm=ln, l1_coefficient=l1c, target=y)
        if k is not None:
            # a token whose k-th and (k+1)-th pre-activations lie within fp32 summation noise of each other may keep either one
            # (measured: 1 token of 1024 at 768 -> 8192 in step 2, tools/tc_diag.py): picked by the oracle alone, replaced by a
            # safe token (as test_native_step_vs_oracle does), so that the comparison runs to the end of the step
            top = -np.partition(-fw["hidden_pre"], k, axis=1)[:, :k + 1]
            risky = (top[:, :k].min(axis=1) - top[:, k]) < 1e-5 * np.abs(fw["hidden_pre"]).max()
            if risky.any():
                safe = np.flatnonzero(~risky)[0]
                x[risky], y[risky] = x[safe], y[safe]
                fw = O.sae_forward(Pc, x, k, layer_norm=ln, l1_coefficient=l1c, target=y)
        gr = O.sae_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c)
        before = stats["act_freq_scores"].copy()
        ref = fw_scalars(fw)
        xg, yg = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        if k is None:
            eng.dense_step(xg, l1c, want_out=True, target=yg)
            eng.grad_sqnorm()
        else:
            # the second step in the sparse-rows form the trainer uses (rows of features that kept no token neither written nor
            # read: the gradient buffers are then not comparable row for row, the clip norm and the parameters are)
            sparse = t == 1
            eng.step(xg, want_out=True, renorm_decoder=True, target=yg, sparse_grads=sparse)
            eng.grad_sqnorm(from_step=True)
        torch.cuda.synchronize()
        sc = eng.scalars.cpu().numpy()
        assert abs(sc[0] - ref["loss"]) <= TOL * ref["loss"] and abs(sc[1] - ref["mse_loss"]) <= TOL * ref["mse_loss"], (sc, ref)
        assert abs(sc[2] - ref["l0"]) <= TOL * ref["l0"], (sc, ref)
        if k is not None:
            # the top-k SET is discontinuous in hidden_pre: after an optimizer step the two parameter sets are ~1e-7 apart and a
            # token whose k-th and (k+1)-th pre-activations lie within fp32 summation noise of each other may keep either one
            # (measured: 1 token of 1024 at 768 -> 8192 in step 2, tools/tc_diag.py).  Such tokens must be near-ties in the
            # oracle's own numbers; the comparison of everything behind the selection ends there.
            same = (np.sort(eng.topk_idx[:n].cpu().numpy(), axis=1) == np.sort(fw["idx"], axis=1)).all(axis=1)
            assert same.all(), ("top-k sets differ on tokens the oracle does not call near-ties", np.flatnonzero(~same))
        assert rel_fro(eng.sae_out[:n].cpu().numpy(), fw["sae_out"]) < TOL
        if k is None:
            assert abs(sc[4] - ref["l1_loss"]) <= TOL * ref["l1_loss"]
            off = eng.lib.pv_debug_sae_ws_offset(eng._plan, b"hidden")
            dH = eng.workspace[off:off + n * d_sae * 4].view(torch.float32).view(n, d_sae).cpu().numpy()
            gate = dH != 0
            differs = gate != (fw["feature_acts"] > 0)            # (ReLU gates within fp32 summation noise of zero: see the dense test)
            assert differs.sum() <= 1e-5 * gate.size and np.all(np.abs(fw["hidden_pre"][differs]) < 1e-5 * np.abs(fw["hidden_pre"]).max())
            if differs.any():
                gr = O.sae_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c, gate=gate)
        else:
            differs = np.zeros(1, bool)
        assert sorted(gr) == sorted(P)
        assert abs(np.sqrt(sc[3]) - grad_norm_of(gr)) <= TOL * grad_norm_of(gr), (np.sqrt(sc[3]), grad_norm_of(gr))
        if not (k is not None and t == 1):
            assert rel_fro(eng.grad_W_enc().cpu().numpy(), gr["W_enc"]) < TOL
            for name in [m for m in P if m != "W_enc"]:
                assert rel_fro(eng.g[name].cpu().numpy(), gr[name]) < TOL, name
        else:
            for name in ("b_dec", "b_dec_out") + (("W_skip",) if skip else ()):
                assert rel_fro(eng.g[name].cpu().numpy(), gr[name]) < TOL, name
        # (the oracle's step continues under the KERNEL's ReLU gates where they differ; round 5 stopped the comparison here)
        O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1, layer_norm=ln, l1_coefficient=l1c, target=y,
                     gate=gate if differs.any() else None)
        fire_ref = stats["act_freq_scores"] - before
        assert np.abs(eng.fire_count.cpu().numpy() - fire_ref).sum() <= TOL * fire_ref.sum()
        eng.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for name in P:
            assert rel_fro(eng.params[name].cpu().numpy(), P[name]) < TOL, name
        assert np.abs(eng.act_freq_scores.cpu().numpy() - stats["act_freq_scores"]).sum() <= TOL * stats["act_freq_scores"].sum()


class _PairActs:
    """layer_acts[:, 0, :] -> x, layer_acts[:, 1, :] -> target for tensors of different widths: what train_step indexes
    (train_sae.py:299-301; the reference's own store concatenates the two buffers and therefore only serves equal widths)."""

    def __init__(self, x, y):
        self.x, self.y, self.shape = x, y, x.shape

    def __getitem__(self, idx):
        return (self.x, self.y)[idx[1]]


@pytest.mark.parametrize("d_in,d_out,d_sae,k,n,ln", [(64, 40, 512, 8, 256, True), (768, 1024, 8192, 32, 1024, True),
                                                      (1024, 768, 8192, 32, 1024, True), (136, 72, 1056, None, 300, False),
                                                      (768, 1024, 8192, None, 512, True), (1024, 768, 24576, 32, 2048, True)])
def test_transcoder_of_unequal_widths_vs_oracle(d_in, d_out, d_sae, k, n, ln):
    """A skip-less Transcoder between hook points of DIFFERENT width (transcoder.py:12: W_dec [d_sae, d_out]; the loss is the mean over
    N x d_out) on the top-k step (k) and on the ReLU + L1 step (k = None): every row padded to D = max(d_in, d_out)
    (pv_sae_transcoder.d_in_true / d_out_true) against the oracle on the real widths (pinned to the reference's own run at 64 -> 40 by
    tests/test_oracle_sae_vs_golden.py) -- losses, l0, every gradient and parameter on the real entries, and the padding exactly zero
    before and after the optimizer step."""
    l1c = 3e-3
    D = max(d_in, d_out)
    rs = np.random.RandomState(5)
    sd = synth_sae_state(d_in, d_sae, 0)
    P = {"W_enc": sd["W_enc"].copy(), "b_enc": sd["b_enc"].copy(), "b_dec": sd["b_dec"].copy(),
         "b_dec_out": (rs.standard_normal(d_out) * 0.05).astype(np.float32)}
    wd = rs.uniform(-1.0, 1.0, size=(d_sae, d_out)).astype(np.float32)
    P["W_dec"] = wd / np.linalg.norm(wd, axis=1, keepdims=True)
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}

    def pad(a, shape, sl):
        buf = torch.zeros(shape, dtype=torch.float32, device="cuda")
        buf[sl].copy_(torch.from_numpy(a))
        return buf

    real = {"W_enc": (slice(0, d_in),), "b_dec": (slice(0, d_in),), "W_dec": (slice(None), slice(0, d_out)), "b_dec_out": (slice(0, d_out),),
            "b_enc": (slice(None),)}
    T = {"W_enc": pad(P["W_enc"], (D, d_sae), real["W_enc"]), "b_dec": pad(P["b_dec"], (D,), real["b_dec"]),
         "W_dec": pad(P["W_dec"], (d_sae, D), real["W_dec"]), "b_dec_out": pad(P["b_dec_out"], (D,), real["b_dec_out"]),
         "b_enc": torch.from_numpy(P["b_enc"].copy()).cuda()}
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k or 1, ln, n, b_dec_out=T["b_dec_out"], tc_widths=(d_in, d_out))

    def padding_is_zero(get):
        for name, sl in real.items():
            full = get(name)
            mask = torch.ones_like(full, dtype=torch.bool)
            mask[sl] = False
            assert float(full[mask].abs().sum()) == 0.0 if mask.any() else True, name

    for t in range(2):
        x = synth_sae_batch(n, d_in, seed=10 + t)
        y = synth_sae_batch(n, max(d_out, d_in), seed=50 + t)[:, :d_out].copy() if d_out <= d_in else \
            np.concatenate([synth_sae_batch(n, d_in, seed=50 + t), synth_sae_batch(n, d_in, seed=70 + t)], axis=1)[:, :d_out].copy()
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        fw = O.sae_forward(Pc, x, k, layer_norm=ln, l1_coefficient=l1c, target=y)
        if k is not None:
            top = -np.partition(-fw["hidden_pre"], k, axis=1)[:, :k + 1]
            risky = (top[:, :k].min(axis=1) - top[:, k]) < 1e-5 * np.abs(fw["hidden_pre"]).max()
            if risky.any():
                safe = np.flatnonzero(~risky)[0]
                x[risky], y[risky] = x[safe], y[safe]
                fw = O.sae_forward(Pc, x, k, layer_norm=ln, l1_coefficient=l1c, target=y)
        gr = O.sae_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c)
        ref = fw_scalars(fw)
        xg, yg = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        if k is None:
            eng.dense_step(xg, l1c, want_out=True, target=yg)
        else:
            eng.step(xg, want_out=True, renorm_decoder=True, target=yg)
        eng.grad_sqnorm()
        torch.cuda.synchronize()
        sc = eng.scalars.cpu().numpy()
        assert abs(sc[0] - ref["loss"]) <= TOL * ref["loss"] and abs(sc[1] - ref["mse_loss"]) <= TOL * ref["mse_loss"], (sc, ref)
        assert abs(sc[2] - ref["l0"]) <= TOL * ref["l0"]
        assert rel_fro(eng.sae_out[:n, :d_out].cpu().numpy(), fw["sae_out"]) < TOL
        grads = {"W_enc": eng.grad_W_enc(), **{m: eng.g[m] for m in ("W_dec", "b_enc", "b_dec", "b_dec_out")}}
        gate = None
        if k is None:
            # ReLU gates within fp32 summation noise of zero may fall on either side: the kernel's gates are read back, must differ from the
            # oracle's only on such entries, and gradients + the rest of the step are compared under them (round 5: a constant 5e-4)
            off = eng.lib.pv_debug_sae_ws_offset(eng._plan, b"hidden")
            dH = eng.workspace[off:off + n * d_sae * 4].view(torch.float32).view(n, d_sae).cpu().numpy()
            differs = (dH != 0) != (fw["feature_acts"] > 0)
            assert differs.sum() <= 1e-5 * differs.size and np.all(np.abs(fw["hidden_pre"][differs]) < 1e-5 * np.abs(fw["hidden_pre"]).max())
            if differs.any():
                gate = dH != 0
                gr = O.sae_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c, gate=gate)
        O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1, layer_norm=ln, l1_coefficient=l1c, target=y, gate=gate)
        gtol = TOL
        for name, sl in real.items():
            assert rel_fro(grads[name][sl].cpu().numpy(), gr[name]) < gtol, name
        padding_is_zero(lambda name: grads[name])
        assert abs(np.sqrt(sc[3]) - grad_norm_of(gr)) <= gtol * grad_norm_of(gr)
        eng.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for name, sl in real.items():
            assert rel_fro(eng.params[name][sl].cpu().numpy(), P[name]) < gtol, name
        padding_is_zero(lambda name: eng.params[name])


@pytest.mark.parametrize("k", [8, None])
def test_unequal_width_transcoder_steps_on_token_shards_sum_to_the_whole_batch(k):
    """The token-sharded form (batch_mean = the TARGET's global mean over its d_out real columns, n_global) of the padded transcoder
    steps: two half batches add up to the whole batch's gradients and losses (top-k step and ReLU + L1 step)."""
    d_in, d_out, d_sae, n, l1c = 72, 136, 2048, 512, 3e-3
    D = max(d_in, d_out)
    rs = np.random.RandomState(3)
    sd = synth_sae_state(d_in, d_sae, 0)

    def pad(a, shape, sl):
        buf = torch.zeros(shape, dtype=torch.float32, device="cuda")
        buf[sl].copy_(torch.from_numpy(np.ascontiguousarray(a)))
        return buf

    wd = rs.uniform(-1.0, 1.0, size=(d_sae, d_out)).astype(np.float32)
    eng = NativeSAE(pad(sd["W_enc"], (D, d_sae), (slice(0, d_in),)), pad(wd / np.linalg.norm(wd, axis=1, keepdims=True), (d_sae, D), (slice(None), slice(0, d_out))),
                    torch.from_numpy(sd["b_enc"].copy()).cuda(), pad(sd["b_dec"], (D,), (slice(0, d_in),)), k or 1, True, n,
                    b_dec_out=pad((rs.standard_normal(d_out) * 0.05).astype(np.float32), (D,), (slice(0, d_out),)), tc_widths=(d_in, d_out))
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=0)).cuda()
    y = torch.from_numpy(synth_sae_batch(n, d_out, seed=50)).cuda()

    def run(xs, ys, **kk):
        if k is None:
            eng.dense_step(xs, l1c, update_stats=False, target=ys, **kk)
        else:
            eng.step(xs, update_stats=False, renorm_decoder=True, target=ys, **kk)
        torch.cuda.synchronize()
        return eng.flat_g.clone(), eng.scalars.clone(), eng.fire_count.clone()

    g_all, sc_all, fire_all = run(x, y)
    bm = y.mean(dim=0)                                            # [d_out]: padded by the engine
    h = n // 2
    g0, sc0, f0 = run(x[:h].contiguous(), y[:h].contiguous(), batch_mean=bm, n_global=n)
    g1, sc1, f1 = run(x[h:].contiguous(), y[h:].contiguous(), batch_mean=bm, n_global=n)
    assert rel_fro((g0 + g1).cpu().numpy(), g_all.cpu().numpy()) < TOL
    for slot in (0, 1) + ((4,) if k is None else ()):
        assert abs(float(sc0[slot] + sc1[slot]) - float(sc_all[slot])) <= TOL * abs(float(sc_all[slot])), slot
    assert torch.equal(f0 + f1, fire_all)


def test_transcoder_of_unequal_widths_through_the_trainer_matches_the_reference_fixture():
    """is_transcoder with d_out = 40 != d_in = 64 (top-k, k = 8, no skip connection) through VisionSAETrainer.train_step on the fused
    HIP step -- the module's parameters become views of the engine's padded storage -- against what the REFERENCE's own Transcoder
    produced through its own train_step (transcoder_dout of tests/golden/sae_variants_steps.npz)."""
    from vit_prisma_amd.sae import Transcoder
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, d_out, exp, N = 64, 40, 8, 256
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp, activation_fn_str="topk",
        activation_fn_kwargs={"k": 8}, normalize_activations="layer_norm", initialization_method="independent", b_dec_init_method="mean",
        train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cuda", _dtype="float32", log_to_wandb=False, use_ghost_grads=False,
        feature_sampling_window=1000, dead_feature_window=5000, lr_scheduler_name="constant", n_checkpoints=0, verbose=False,
        is_transcoder=True, transcoder_with_skip_connection=False, d_out=d_out, out_hook_point_layer=6)
    tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True)
    model = tr.sparse_coder
    assert type(model) is Transcoder and tuple(model.W_dec.shape) == (d_in * exp, d_out)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"transcoder_dout_init_{n}"]).cuda())
    act, since, frac, opt, sched = tr.initialize_training_variables()
    for t in range(3):
        pair = _PairActs(torch.from_numpy(synth_sae_batch(N, d_in, seed=t)).cuda(),
                         torch.from_numpy(synth_sae_batch(N, d_in, seed=100 + t)[:, :d_out].copy()).cuda())
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=pair, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native and l1 is None
        want = g[f"transcoder_dout_s{t}_scalars"]
        for got, w in ((loss, want[0]), (mse, want[1]), (l0, want[3])):
            assert abs(float(got) - w) <= TOL * abs(w), (t, float(got), w)
        assert np.array_equal(act.cpu().numpy(), g[f"transcoder_dout_s{t}_act_freq"])
    for n, p in model.named_parameters():
        assert tuple(p.shape) == g[f"transcoder_dout_s2_param_{n}"].shape
        assert rel_fro(p.detach().cpu().numpy(), g[f"transcoder_dout_s2_param_{n}"]) < TOL, n


def test_transcoder_trainer_runs_natively_and_matches_the_reference_fixture():
    """is_transcoder (top-k, k = 8, with the skip connection) through VisionSAETrainer.train_step on the fused HIP step, against
    what the REFERENCE's own Transcoder produced through its own train_step (tests/golden/sae_variants_steps.npz)."""
    from vit_prisma_amd.sae import Transcoder
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, exp, N = 64, 8, 256
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp, activation_fn_str="topk",
        activation_fn_kwargs={"k": 8}, normalize_activations="layer_norm", initialization_method="independent", b_dec_init_method="mean",
        train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cuda", _dtype="float32", log_to_wandb=False, use_ghost_grads=False,
        feature_sampling_window=1000, dead_feature_window=5000, lr_scheduler_name="constant", n_checkpoints=0, verbose=False,
        is_transcoder=True, transcoder_with_skip_connection=True, d_out=64, out_hook_point_layer=6)
    tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True)
    model = tr.sparse_coder
    assert type(model) is Transcoder
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"transcoder_init_{n}"]).cuda())
    act, since, frac, opt, sched = tr.initialize_training_variables()
    for t in range(3):
        pair = torch.stack([torch.from_numpy(synth_sae_batch(N, d_in, seed=t)), torch.from_numpy(synth_sae_batch(N, d_in, seed=100 + t))],
                           dim=1).cuda()
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=pair, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native and l1 is None
        want = g[f"transcoder_s{t}_scalars"]
        for got, w in ((loss, want[0]), (mse, want[1]), (l0, want[3])):
            assert abs(float(got) - w) <= TOL * abs(w), (t, float(got), w)
        assert np.array_equal(act.cpu().numpy(), g[f"transcoder_s{t}_act_freq"]) and np.array_equal(since.cpu().numpy(), g[f"transcoder_s{t}_n_since"])
    for n, p in model.named_parameters():
        assert rel_fro(p.detach().cpu().numpy(), g[f"transcoder_s2_param_{n}"]) < TOL, n


# ---------------------------------------------------------------------------------------------------
# Gated SAE (sae.py:648-792; pv_sae_gated_step)
# ---------------------------------------------------------------------------------------------------
def _shut_gates(P, x, ln, open_max):
    """b_gate shifted so that no token of batch x has more than open_max gates open (a trained gated SAE: L0 of tens): the regime the
    sparse form of the step is for."""
    pre = O.gated_forward({**P, "b_enc": None}, x, layer_norm=ln)["gate_pre"]
    kth = -np.partition(-pre, open_max, axis=1)[:, open_max]
    P["b_gate"] = (P["b_gate"] - kth.max()).astype(np.float32)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("d_in,d_sae,n,ln,form", [
    (64, 512, 256, True, "dense"), (136, 1056, 300, False, "dense"), (768, 8192, 1024, True, "dense"),
    (768, 24576, 4096, True, "dense"), (768, 49152, 1024, True, "dense"),      # (the benchmarked and the published shapes)
    # a batch whose gates are mostly shut runs SPARSE (pv_sae_gated_step_sparse; "fallback": the same batch with a capacity it cannot
    # be held in -- the mode word sends it to the dense GEMMs; "forced": sparse=False)
    (768, 8192, 1024, True, "sparse"), (256, 2048, 300, False, "sparse"), (768, 24576, 4096, True, "sparse"),
    (1024, 16384, 512, True, "sparse"), (1280, 10240, 512, True, "sparse"), (768, 8192, 1024, True, "fallback"),
    (768, 8192, 1024, True, "forced")])
def test_gated_step_vs_oracle(d_in, d_sae, n, ln, form):
    """pv_sae_gated_step(_sparse) + grad_sqnorm + apply against the oracle's gated form (pinned to the reference's own
    GatedSparseAutoencoder run by tests/test_oracle_sae_vs_golden.py): the four losses, l0, every gradient tensor, the clip norm,
    parameters and statistics after the optimizer step; ragged shapes and the no-LayerNorm form included."""
    l1c = 3e-3
    # (Adam's first step moves every weight by lr: at 1e-3 it opens hundreds of gates per token on the second batch)
    lr = 1e-3 if form == "dense" else 2e-5
    P, opt, stats, T = fresh(d_in, d_sae)
    rs = np.random.RandomState(9)
    for name, scale in (("b_gate", 0.05), ("r_mag", 0.2), ("b_mag", 0.05)):
        P[name] = (rs.standard_normal(d_sae) * scale).astype(np.float32)
        opt["m"][name], opt["v"][name] = np.zeros_like(P[name]), np.zeros_like(P[name])
    if form != "dense":
        _shut_gates(P, synth_sae_batch(n, d_in, seed=0), ln, 128)
    for name in ("b_gate", "r_mag", "b_mag"):
        T[name] = torch.from_numpy(P[name].copy()).cuda()
    b_enc0 = P.pop("b_enc")
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], 1, ln, n, gated={m: T[m] for m in ("b_gate", "r_mag", "b_mag")})
    kw = {"sparse": False} if form == "forced" else ({"cap": 8} if form == "fallback" else {})
    drift = None                                                  # |kernel's - oracle's| parameters after the previous step (see below)
    for t in range(2):
        x = synth_sae_batch(n, d_in, seed=t)
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        fw = O.gated_forward(Pc, x, layer_norm=ln, l1_coefficient=l1c)
        if form != "dense":
            # with tens of open gates per token ONE gate within fp32 summation noise of zero moves the token's reconstruction by percents
            # (see below): such tokens -- picked by the oracle alone, before the kernel runs -- are replaced by a copy of a safe one, so
            # that every tensor of the sparse form is compared entry for entry
            risky = np.abs(fw["gate_pre"]).min(axis=1) < 1e-5 * np.abs(fw["gate_pre"]).max()
            if risky.any():
                x[risky] = x[np.flatnonzero(~risky)[0]]
                fw = O.gated_forward(Pc, x, layer_norm=ln, l1_coefficient=l1c)
        gr = O.gated_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c)
        before = stats["act_freq_scores"].copy()
        eng.gated_step(torch.from_numpy(x).cuda(), l1c, want_out=True, **kw)
        eng.grad_sqnorm()
        torch.cuda.synchronize()
        if form != "forced":
            # which form ran: sparse exactly when the filter applies to the shape and every token's open gates fit the capacity
            open_max = int((fw["gate_pre"] > 0).sum(axis=1).max())
            assert eng.gated_mode == (0 if form == "sparse" else 1), (form, eng.gated_mode, open_max, float(fw["l0"]))
            assert (open_max <= 256) == (form != "dense") and (form != "fallback" or open_max > 8)
        # feature_acts is DISCONTINUOUS in the gate pre-activation (a Heaviside step times a magnitude, sae.py:705-716): where it lies
        # within noise of zero the kernel and numpy may open different gates, and the token's reconstruction moves by a whole decoder
        # row.  The dense GEMMs leave the gates they took behind (dG = (dVia W_dec^T + l1 / N)[gate > 0] in the second half of their
        # scratch: nonzero exactly where the gate is open): they may differ from the oracle's only on entries that close to zero, and
        # the oracle's step is then CONTINUED UNDER THE KERNEL'S GATES -- every tensor compared entry for entry, both steps (round 5 ended
        # such a run with a silent return; thousands of gates per token make replacing the tokens at risk impossible here: 81 % of them).
        active = None
        if form in ("dense", "fallback", "forced"):
            hs = eng._gt_scratch[:2 * n * d_sae * 4].view(torch.float32).view(2 * n, d_sae)
            gate_k = (hs[n:] != 0).cpu().numpy()
            differs = gate_k != (fw["gate_pre"] > 0)
            if differs.any():
                # how far may the two gate pre-activations be apart?  fp32 summation noise (2e-5 of the largest entry covers the 5 sigma
                # tail of a K = 768 accumulation among 1e8 gates), and -- from the second step on -- what the PARAMETERS differ by: Adam's
                # first steps move a weight whose gradient is noise by +-lr whichever sign the noise has, so single entries of W_enc are
                # 2 lr apart after a step although the tensors agree to 1e-6 (measured: gates up to 3.6e-5 fall differently in step 2);
                # the bound is taken entry by entry from the two parameter sets as they stood after the previous step:
                # |sae_in| |dW_enc| + |db_gate| + |db_dec| |W_enc|
                bound = 2e-5 * np.abs(fw["gate_pre"]).max()
                if drift is not None:
                    bound = bound + np.abs(fw["sae_in"]) @ drift["W_enc"] + drift["b_gate"] + drift["b_dec"] @ np.abs(Pc["W_enc"])
                    bound = bound[differs]
                assert differs.sum() <= max(8, int(1e-5 * n * d_sae)) and np.all(np.abs(fw["gate_pre"])[differs] <= bound), \
                    (t, differs.sum(), np.abs(fw["gate_pre"][differs]).max())
                active = gate_k
                fw = O.gated_forward(Pc, x, layer_norm=ln, l1_coefficient=l1c, active=active)
                gr = O.gated_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c, gates=(fw["feature_acts"] > 0, active))
        sc = eng.scalars.cpu().numpy()
        for slot, key in ((0, "loss"), (1, "mse_loss"), (4, "l1_loss"), (6, "aux_loss"), (2, "l0")):
            assert abs(sc[slot] - float(fw[key])) <= TOL * abs(float(fw[key])), (key, sc, float(fw[key]))
        got_out = eng.sae_out[:n].cpu().numpy()
        tok_err = np.linalg.norm(got_out - fw["sae_out"], axis=1) / np.linalg.norm(fw["sae_out"], axis=1)
        off = tok_err > TOL
        if off.any():
            # (the sparse form, whose tokens at risk were replaced above: a gate that still fell differently ends the comparison visibly)
            assert off.sum() <= max(4, int(4e-7 * n * d_sae)) and np.all(np.abs(fw["gate_pre"][off]).min(axis=1) < 2e-5 * np.abs(fw["gate_pre"]).max()), (off.sum(), tok_err.max())
            assert rel_fro(got_out, fw["sae_out"]) < 1e-3
            truncated(f"gated step {t}: {int(off.sum())} token(s) opened another gate within fp32 noise of zero; losses compared, tensors to 1e-3")
        assert rel_fro(got_out, fw["sae_out"]) < TOL
        # the magnitude path's ReLU gate of the backward as the kernel took it (dP = dM e^r + dG is what the scratch holds at the end, so
        # it is read off the gradients it shapes): entries within summation noise of zero may fall on either side -- compare under
        # the oracle's gates first and fall back to a norm-level statement when one differs
        bad = [name for name in gr if rel_fro((eng.grad_W_enc() if name == "W_enc" else eng.g[name]).cpu().numpy(), gr[name]) >= TOL]
        if bad:
            # at most a handful of gate flips: every tensor still agrees to 1e-3 and the losses above to 1e-4
            for name in gr:
                assert rel_fro((eng.grad_W_enc() if name == "W_enc" else eng.g[name]).cpu().numpy(), gr[name]) < 1e-3, name
            small = np.minimum(np.abs(fw["gate_pre"]), np.where(fw["gate_pre"] > 0, np.abs(fw["mag_pre"]), np.inf)).min()
            assert small < 1e-5 * np.abs(fw["gate_pre"]).max(), (bad, small)
            truncated(f"gated step {t}: a backward ReLU gate within fp32 noise of zero fell differently ({bad}); every gradient compared to 1e-3")
        O.gated_train_step(P, opt, stats, x, lr=lr, step=t + 1, layer_norm=ln, l1_coefficient=l1c, active=active)
        assert float(eng.g["b_enc"].abs().max()) == 0.0
        assert abs(np.sqrt(sc[3]) - grad_norm_of(gr)) <= TOL * grad_norm_of(gr)
        fire_ref = stats["act_freq_scores"] - before
        assert np.abs(eng.fire_count.cpu().numpy() - fire_ref).sum() <= TOL * fire_ref.sum()
        eng.apply(lr, 1.0)
        torch.cuda.synchronize()
        for name in P:
            assert rel_fro(eng.params[name].cpu().numpy(), P[name]) < TOL, name
        drift = {name: np.abs(eng.params[name].cpu().numpy() - P[name]) for name in ("W_enc", "b_gate", "b_dec")}
        assert np.array_equal(eng.params["b_enc"].cpu().numpy(), b_enc0)
        assert np.abs(eng.act_freq_scores.cpu().numpy() - stats["act_freq_scores"]).sum() <= TOL * stats["act_freq_scores"].sum()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("d_in,d_sae,k,n,ln", [(64, 512, 8, 256, True), (136, 1056, 16, 300, False), (768, 8192, 32, 1024, True),
                                               (768, 24576, 32, 4096, True), (1024, 16384, 64, 512, True), (1280, 10240, 32, 512, True)])
def test_gated_topk_step_vs_oracle(d_in, d_sae, k, n, ln):
    """pv_sae_gated_topk_step + grad_sqnorm + apply against the oracle's top-k gated form (pinned to the reference's own run,
    tests/test_oracle_sae_vs_golden.py): losses, l0, the two k-sparse lists, every gradient tensor, the clip norm, parameters and
    statistics after the optimizer step; the exact-encoder shapes (d_sae < 2048, ragged) and the filtered ones."""
    P, opt, stats, T = fresh(d_in, d_sae)
    rs = np.random.RandomState(9)
    for name, scale in (("b_gate", 0.05), ("r_mag", 0.2), ("b_mag", 0.05)):
        P[name] = (rs.standard_normal(d_sae) * scale).astype(np.float32)
        opt["m"][name], opt["v"][name] = np.zeros_like(P[name]), np.zeros_like(P[name])
        T[name] = torch.from_numpy(P[name].copy()).cuda()
    b_enc0 = P.pop("b_enc")
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, ln, n, gated={m: T[m] for m in ("b_gate", "r_mag", "b_mag")},
                    gated_topk=True)
    for t in range(2):
        x = synth_sae_batch(n, d_in, seed=t)
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        fw = O.gated_forward(Pc, x, layer_norm=ln, k=k)
        # a kept magnitude whose gate pre-activation lies within fp32 summation noise of zero may fall on either side of the
        # Heaviside step (a whole decoder row of difference): such tokens -- picked by the oracle alone -- are replaced by a safe one
        # (likewise a k-th / (k+1)-th entry of either path that close to each other: the two top-k selections may differ by that pair)
        idx_m = np.argsort(-fw["mag_pre"], axis=1, kind="stable")[:, :k]
        risky = np.abs(np.take_along_axis(fw["gate_pre"], idx_m, axis=1)).min(axis=1) < 1e-5 * np.abs(fw["gate_pre"]).max()
        for pre in (fw["mag_pre"], fw["gate_pre"]):
            top = -np.partition(-pre, k, axis=1)[:, :k + 1]
            risky |= (top[:, :k].min(axis=1) - top[:, k]) < 1e-5 * np.abs(pre).max()
        if risky.any():
            x[risky] = x[np.flatnonzero(~risky)[0]]
            fw = O.gated_forward(Pc, x, layer_norm=ln, k=k)
        gr = O.gated_backward(Pc, x, fw, layer_norm=ln)
        before = stats["act_freq_scores"].copy()
        ref = O.gated_train_step(P, opt, stats, x, lr=1e-3, step=t + 1, layer_norm=ln, k=k)
        eng.gated_topk_step(torch.from_numpy(x).cuda(), want_out=True)
        eng.grad_sqnorm()
        torch.cuda.synchronize()
        sc = eng.scalars.cpu().numpy()
        for slot, key in ((0, "loss"), (1, "mse_loss"), (6, "aux_loss"), (2, "l0")):
            assert abs(sc[slot] - ref[key]) <= TOL * abs(ref[key]), (key, sc, ref)
        assert sc[4] == 0.0 and ref["l1_loss"] == 0.0
        # the two lists: rows [0, n) = feature_acts at the magnitude path's top-k, rows [n, 2n) = the gate activations
        got_idx, got_val = eng.topk_idx.cpu().numpy(), eng.topk_val.cpu().numpy()
        dense_f, dense_g = np.zeros((n, d_sae), np.float32), np.zeros((n, d_sae), np.float32)
        np.put_along_axis(dense_f, got_idx[:n], got_val[:n], axis=1)
        np.put_along_axis(dense_g, got_idx[n:2 * n], got_val[n:2 * n], axis=1)
        assert rel_fro(dense_f, fw["feature_acts"]) < TOL and rel_fro(dense_g, fw["pg"]) < TOL
        assert rel_fro(eng.sae_out[:n].cpu().numpy(), fw["sae_out"]) < TOL
        for name in gr:
            assert rel_fro((eng.grad_W_enc() if name == "W_enc" else eng.g[name]).cpu().numpy(), gr[name]) < TOL, name
        assert float(eng.g["b_enc"].abs().max()) == 0.0
        assert abs(np.sqrt(sc[3]) - grad_norm_of(gr)) <= TOL * grad_norm_of(gr)
        fire_ref = stats["act_freq_scores"] - before
        assert np.array_equal(eng.fire_count.cpu().numpy(), fire_ref)
        eng.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for name in P:
            assert rel_fro(eng.params[name].cpu().numpy(), P[name]) < TOL, name
        assert np.array_equal(eng.params["b_enc"].cpu().numpy(), b_enc0)
        assert np.array_equal(eng.act_freq_scores.cpu().numpy(), stats["act_freq_scores"])


@pytest.mark.parametrize("kind", ["gated", "relu_transcoder", "topk_transcoder"])
def test_variant_steps_on_token_shards_sum_to_the_whole_batch(kind):
    """The data-parallel form of the gated step and of the ReLU transcoder step (batch_mean / n_global: tokens sharded over ranks,
    gradients summed by the caller): two half batches with the GLOBAL mean and token count give gradients and losses that add up
    to the whole batch's (one engine, the halves one after the other; statistics off)."""
    d_in, d_sae, n, l1c = 136, 1056, 512, 3e-3
    rs = np.random.RandomState(4)
    P, opt, stats, T = fresh(d_in, d_sae)
    kw = {}
    if kind == "gated":
        kw["gated"] = {m: torch.from_numpy((rs.standard_normal(d_sae) * s_).astype(np.float32)).cuda()
                       for m, s_ in (("b_gate", 0.05), ("r_mag", 0.2), ("b_mag", 0.05))}
    else:
        kw["b_dec_out"] = torch.from_numpy((rs.standard_normal(d_in) * 0.05).astype(np.float32)).cuda()
        kw["W_skip"] = torch.from_numpy((rs.standard_normal((d_in, d_in)) / np.sqrt(d_in) * 0.3).astype(np.float32)).cuda()
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], 16 if kind == "topk_transcoder" else 1, True, n, **kw)
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=0)).cuda()
    y = torch.from_numpy(synth_sae_batch(n, d_in, seed=50)).cuda()
    ref = y if kind != "gated" else x

    def run(xs, ys, **kk):
        if kind == "gated":
            eng.gated_step(xs, l1c, update_stats=False, **kk)
        elif kind == "topk_transcoder":
            eng.step(xs, update_stats=False, renorm_decoder=True, target=ys, **kk)
        else:
            eng.dense_step(xs, l1c, update_stats=False, target=ys, **kk)
        torch.cuda.synchronize()
        return eng.flat_g.clone(), eng.scalars.clone(), eng.fire_count.clone()

    g_all, sc_all, fire_all = run(x, y)
    bm = ref.mean(dim=0)
    h = n // 2
    g0, sc0, f0 = run(x[:h].contiguous(), y[:h].contiguous(), batch_mean=bm, n_global=n)
    g1, sc1, f1 = run(x[h:].contiguous(), y[h:].contiguous(), batch_mean=bm, n_global=n)
    assert rel_fro((g0 + g1).cpu().numpy(), g_all.cpu().numpy()) < TOL
    for slot in (0, 1, 4) + ((6,) if kind == "gated" else ()):
        assert abs(float(sc0[slot] + sc1[slot]) - float(sc_all[slot])) <= TOL * abs(float(sc_all[slot])), slot
    assert abs(float(sc0[2] + sc1[2]) / 2 - float(sc_all[2])) <= TOL * float(sc_all[2])
    assert torch.equal(f0 + f1, fire_all)


@pytest.mark.parametrize("kind", ["relu", "topk"])
def test_ghost_gradient_steps_on_token_shards_sum_to_the_whole_batch(kind):
    """Ghost gradients with the tokens sharded over ranks (pv_sae_ghost.err_colmean / mse_global / n_global): the ghost term normalises
    by the residual's column mean and rescales by the mse loss of the WHOLE batch (sae.py:156, :172), so two half batches that are
    given both give gradients and losses that add up to the whole batch's -- on the dense ReLU + L1 step and on the top-k step +
    pv_sae_topk_ghost (one engine, the halves one after the other; statistics off; a third of the features dead)."""
    d_in, d_sae, n, l1c, k = 136, 1056, 512, 3e-3, 16
    P, opt, stats, T = fresh(d_in, d_sae)
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=0)).cuda()
    dead = torch.zeros(d_sae, dtype=torch.bool, device="cuda")
    dead[::3] = True

    def run(xs, ghost_global=None, **kk):
        if kind == "relu":
            eng.dense_step(xs, l1c, update_stats=False, want_out=True, dead_mask=dead, ghost_global=ghost_global, **kk)
        else:
            eng.renorm_decoder()
            eng.step(xs, update_stats=False, renorm_decoder=False, want_out=True, **kk)
            eng.topk_ghost(xs, dead, ghost_global=ghost_global)
        torch.cuda.synchronize()
        return eng.flat_g.clone(), eng.scalars.clone(), eng.sae_out[:xs.shape[0]].clone()

    g_all, sc_all, out_all = run(x)
    assert float(sc_all[5]) > 0
    glob = ((out_all - x).mean(dim=0), sc_all[1:2].clone(), n)
    bm = x.mean(dim=0)
    h = n // 2
    g0, sc0, _ = run(x[:h].contiguous(), ghost_global=glob, batch_mean=bm, n_global=n)
    g1, sc1, _ = run(x[h:].contiguous(), ghost_global=glob, batch_mean=bm, n_global=n)
    assert rel_fro((g0 + g1).cpu().numpy(), g_all.cpu().numpy()) < TOL
    for slot in (0, 1, 5) + ((4,) if kind == "relu" else ()):
        assert abs(float(sc0[slot] + sc1[slot]) - float(sc_all[slot])) <= TOL * abs(float(sc_all[slot])), slot
    # and without the global quantities a sharded call is refused rather than silently wrong
    from vit_prisma_amd._native import NativeError
    with pytest.raises(NativeError):
        eng.dense_step(x[:h].contiguous(), l1c, update_stats=False, dead_mask=dead, batch_mean=bm, n_global=n)


@pytest.mark.parametrize("variant", ["gated", "gated_topk"])
def test_gated_trainer_runs_natively_and_matches_the_reference_fixture(variant):
    """architecture = "gated" (ReLU, and the top-k form: TopK on the magnitudes and on the gate activations, k = 8) through
    VisionSAETrainer.train_step on the HIP step, against what the REFERENCE's own GatedSparseAutoencoder produced through its own
    train_step (tests/golden/sae_variants_steps.npz)."""
    from vit_prisma_amd.sae import GatedSparseAutoencoder
    g = np.load(os.path.join(GOLDEN, "sae_variants_steps.npz"))
    d_in, exp, N = 64, 8, 256
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp,
        activation_fn_str="relu" if variant == "gated" else "topk", activation_fn_kwargs={} if variant == "gated" else {"k": 8},
        normalize_activations="layer_norm", initialization_method="independent", b_dec_init_method="mean",
        train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cuda", _dtype="float32", log_to_wandb=False, use_ghost_grads=False,
        feature_sampling_window=1000, dead_feature_window=5000, lr_scheduler_name="constant", n_checkpoints=0, verbose=False,
        l1_coefficient=2e-3, architecture="gated")
    tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True)
    model = tr.sparse_coder
    assert type(model) is GatedSparseAutoencoder
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"{variant}_init_{n}"]).cuda())
    act, since, frac, opt, sched = tr.initialize_training_variables()
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(N, d_in, seed=t)).cuda()[:, None, :]
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=x, n_training_steps=t, n_training_tokens=t * N)
        assert tr.last_step_native
        want = g[f"{variant}_s{t}_scalars"]
        for got, w in ((loss, want[0]), (mse, want[1]), (l1, want[2]), (l0, want[3]), (tr._engine.scalars[6], want[5])):
            assert abs(float(got) - w) <= TOL * abs(w), (t, float(got), w)
        assert np.array_equal(act.cpu().numpy(), g[f"{variant}_s{t}_act_freq"]) and np.array_equal(since.cpu().numpy(), g[f"{variant}_s{t}_n_since"])
    for n, p in model.named_parameters():
        assert rel_fro(p.detach().cpu().numpy(), g[f"{variant}_s2_param_{n}"]) < TOL, n


def test_store_harvest_prefetch_on_a_side_stream_serves_the_same_batches():
    """The store issues the NEXT refill's ViT forwards on a side stream while the current half buffer is served
    (sae/store.py: overlap_harvest): same images, same order, the same permutations drawn at the same points -- every batch
    it serves must be bit-identical to the synchronous store's, across several refills."""
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=64, expansion_factor=8, activation_fn_str="topk",
        activation_fn_kwargs={"k": 8}, normalize_activations="layer_norm", b_dec_init_method="mean",
        train_batch_size=64, lr=1e-3, max_grad_norm=1.0, _device="cuda", log_to_wandb=False,
        lr_scheduler_name="constant", n_checkpoints=0, context_size=17, store_batch_size=4, n_batches_in_buffer=4)
    arch = ARCHS["tiny"]
    vit = HookedViT(HookedViTConfig(**arch, device="cuda"))
    vit.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()})
    vit = vit.cuda().eval()
    images = torch.from_numpy(synth_images(arch, 32, 3))
    ds = [(images[i], 0) for i in range(32)]

    def serve(overlap: bool, legacy: bool = False):
        torch.manual_seed(1234)
        store = VisionActivationsStore(cfg, vit, ds, create_dataloader=False)
        store.overlap_harvest = overlap                         # (set before the first refill is scheduled)
        if legacy:                                              # one forward per store batch + the reference's buf[...] = acts copy
            store.HARVEST_IMAGES, store.direct_tap = 0, False
        store.storage_buffer = store.get_buffer(cfg.n_batches_in_buffer)
        store.dataloader = store.get_data_loader()
        out = []
        for _ in range(14):                                     # 16 * 17 / 64 = 4.25 batches per refill: three refills
            b = store.next_batch()
            out.append(b.clone())
            if overlap:
                torch.mm(torch.ones(512, 512, device="cuda"), torch.ones(512, 512, device="cuda"))   # main-stream work in between
        torch.cuda.synchronize()
        return out, store

    sync_batches, _ = serve(False)
    over_batches, st = serve(True)
    assert st._side_stream is not None and vit.last_run_native
    assert len(sync_batches) == len(over_batches)
    for a, b in zip(sync_batches, over_batches):
        assert a.shape == b.shape and torch.equal(a, b)
    # store batches coalesced into one forward (4-image store batches -> one 16-image forward per refill here) and the harvest
    # kernel storing straight into the buffer slice: the same bits as one forward per store batch + the copy
    n0 = vit._native.n_forward
    legacy_batches, _ = serve(False, legacy=True)
    n_legacy = vit._native.n_forward - n0
    n0 = vit._native.n_forward
    again, _ = serve(False)
    assert n_legacy >= 2 * (vit._native.n_forward - n0) > 0
    for a, b in zip(sync_batches, legacy_batches):
        assert a.dtype == b.dtype and torch.equal(a, b)


@pytest.mark.parametrize("vit_dtype", [torch.float32, torch.bfloat16])
def test_store_taps_straight_into_its_buffer(vit_dtype):
    """VisionActivationsStore._harvest_raw: the kernel that produces blocks.L.hook_resid_post writes it into the store's own
    buffer slice (pv_tap.dst = the slice: "caching costs one extra HBM store"), several store batches per forward -- and the rows
    are those of a plain run_with_cache on the same images, widened to cfg.dtype exactly as the reference's assignment
    (activations_store.py:326-355) would."""
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=1, layer_subtype="hook_resid_post", d_in=64, expansion_factor=8, activation_fn_str="topk",
        activation_fn_kwargs={"k": 8}, normalize_activations="layer_norm", b_dec_init_method="mean",
        train_batch_size=64, lr=1e-3, max_grad_norm=1.0, _device="cuda", log_to_wandb=False,
        lr_scheduler_name="constant", n_checkpoints=0, context_size=17, store_batch_size=4, n_batches_in_buffer=4)
    arch = ARCHS["tiny"]
    vit = HookedViT(HookedViTConfig(**arch, dtype=vit_dtype, device="cuda"))
    vit.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()})
    vit = vit.to(vit_dtype).cuda().eval()
    images = torch.from_numpy(synth_images(arch, 32, 3)).to(vit_dtype)
    ds = [(images[i], 0) for i in range(32)]
    store = VisionActivationsStore(cfg, vit, ds, create_dataloader=False)
    store.image_dataloader_iter = iter([images[i:i + 4].cuda() for i in range(0, 32, 4)])        # (fixed order)
    buf, _ = store._harvest_raw(4)
    torch.cuda.synchronize()
    assert vit.last_run_native and buf.dtype == vit_dtype and tuple(buf.shape) == (16, 17, 1, 64)
    assert store.n_buffer_copies == 0          # (round-4 advisor: the `buf[...] = acts` pass ran on a stacked copy although the kernel had written the rows)
    with torch.no_grad():
        _, cache = vit.run_with_cache(images[:16].cuda(), names_filter=[cfg.hook_point], stop_at_layer=2)
    assert torch.equal(buf[:, :, 0, :], cache[cfg.hook_point])
    # the arena was not involved: the forward's only tap went to the buffer
    assert buf.data_ptr() != cache[cfg.hook_point].data_ptr()
    torch.manual_seed(0)
    store._prefetched = None
    store.image_dataloader_iter = iter([images[i:i + 4].cuda() for i in range(0, 32, 4)])
    rows = store.get_buffer(4)
    torch.manual_seed(0)
    perm = torch.randperm(16 * 17, device="cuda")
    assert rows.dtype == cfg.dtype and torch.equal(rows, cache[cfg.hook_point].reshape(-1, 1, 64)[perm].to(cfg.dtype))


def test_checkpoint_mid_run_and_lazy_w_enc_match_the_torch_path(tmp_path):
    """(a) trainer.checkpoint() between steps renormalises W_dec on the live parameters (sae.py:275-277): the engine must not
    reuse the inverse row norms of its last apply afterwards -- the native run must keep tracking the PyTorch
    path of the same trainer through a checkpoint.  (b) single process: the engine trains W_enc in its transposed master and
    leaves the parameter's own layout stale until somebody reads it -- sae.W_enc, state_dict() and the saved checkpoint
    must nevertheless hold the trained values."""
    def make(native: bool, folder):
        cfg = VisionModelSAERunnerConfig(
            hook_point_layer=1, layer_subtype="hook_resid_post", d_in=64, expansion_factor=8, activation_fn_str="topk",
            activation_fn_kwargs={"k": 8}, normalize_activations="layer_norm", b_dec_init_method="mean", train_batch_size=256,
            lr=1e-3, max_grad_norm=1.0, _device="cuda", log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0,
            context_size=17, checkpoint_path=str(folder))
        sae = StandardSparseAutoencoder(cfg)
        with torch.no_grad():
            for n, v in synth_sae_state(64, 512, 0).items():
                getattr(sae, n).copy_(torch.from_numpy(v))
        tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae).use_native(native)
        return cfg, sae, tr, list(tr.initialize_training_variables())

    runs = {}
    for native in (True, False):
        cfg, sae, tr, st = make(native, tmp_path / ("native" if native else "torch"))
        for t in range(5):
            x = torch.from_numpy(synth_sae_batch(256, 64, seed=t)).cuda()[:, None, :]
            _, _, _, _, st[0], st[1], st[2] = tr.train_step(
                sparse_autoencoder=sae, optimizer=st[3], scheduler=st[4], act_freq_scores=st[0], n_forward_passes_since_fired=st[1],
                n_frac_active_tokens=st[2], layer_acts=x, n_training_steps=t, n_training_tokens=t * 256)
            assert tr.last_step_native == native
            if native and t == 1:
                eng = tr._engine
                assert eng.lazy_w_enc and eng._w_enc_stale
                stale = sae._parameters["W_enc"].detach().clone()               # the registry does not trigger the sync ...
                fresh = sae.W_enc.detach().clone()                              # ... attribute access does
                assert not eng._w_enc_stale and torch.equal(fresh, eng.W_encT.t()) and not torch.equal(stale, fresh)
            if t == 2:
                tr.checkpoint(sae, (t + 1) * 256, st[0], st[2])
        runs[native] = {n: p.detach().cpu().numpy().copy() for n, p in sae.named_parameters()}
        if native:
            path = [f for f in os.listdir(tmp_path / "native") if f.endswith(".pt") and "sparsity" not in f][0]
            blob = torch.load(tmp_path / "native" / path, weights_only=False)
            ck = blob["state_dict"]["W_enc"].cpu().numpy()
            # the checkpoint (after step 3) holds TRAINED values: two Adam steps of lr 1e-3 away from the final ones (each moves a
            # 0.036-sized weight by up to 1e-3), three away from the initial ones
            assert np.isfinite(ck).all() and 1e-3 < rel_fro(ck, runs[True]["W_enc"]) < 0.08
            assert rel_fro(ck, synth_sae_state(64, 512, 0)["W_enc"]) > 2e-2
    for n in runs[True]:
        assert rel_fro(runs[True][n], runs[False][n]) < TOL, n

# ---------------------------------------------------------------------------------------------------
# the ReLU + L1 step, sparse where the batch allows it (pv_sae_relu_step: fp16 filter with the threshold -B_n, exact fp32
# re-scoring, per-token lists of the positive activations on the k-sparse kernels; the dense GEMMs when a token cannot be held)
# ---------------------------------------------------------------------------------------------------
def _shift_b_enc_for_l0(P, x, ln, want_l0):
    """Move b_enc down so that a token keeps about want_l0 features on this batch: the regime a trained ReLU SAE with a strong L1
    term lives in (and the one bench.py's ReLU leg settles into after ~6 steps from the synthetic init)."""
    fw = O.sae_forward(P, x, None, layer_norm=ln)
    q = np.quantile(fw["hidden_pre"].ravel()[::7], 1.0 - want_l0 / P["W_enc"].shape[1])
    P["b_enc"] -= np.float32(q)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("d_in,d_sae,n,ln,l0,tc", [(64, 2048, 256, True, 12, False), (136, 2304, 300, False, 20, False),
                                                    (768, 8192, 1024, True, 24, False), (768, 8192, 1024, True, 24, True),
                                                    (768, 24576, 4096, True, 32, False), (768, 49152, 1024, True, 60, False),
                                                    (1280, 20480, 512, True, 40, False)])
def test_relu_step_sparse_vs_oracle(d_in, d_sae, n, ln, l0, tc):
    """relu_step in its sparse form against the ReLU + L1 oracle: the step must report mode 0, keep exactly the oracle's positive
    entries (sets equal up to entries within fp32 summation noise of zero) with the exact fp32 values, and give the losses,
    l0, reconstruction, every gradient tensor, the clip norm, parameters and statistics of the reference's train step -- ragged
    shapes, no LayerNorm, a Transcoder with the skip connection, the bench shape (768 -> 24576 x 4096) and the x64 width included."""
    l1c = 3e-3
    P, opt, stats, T = (fresh_transcoder(d_in, d_sae, True) if tc else fresh(d_in, d_sae))
    _shift_b_enc_for_l0(P, synth_sae_batch(n, d_in, seed=10), ln, l0)
    T["b_enc"].copy_(torch.from_numpy(P["b_enc"]))
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], 1, ln, n, **({"b_dec_out": T["b_dec_out"], "W_skip": T["W_skip"]} if tc else {}))
    for t in range(2):
        x = synth_sae_batch(n, d_in, seed=10 + t)
        y = synth_sae_batch(n, d_in, seed=50 + t) if tc else None
        Pc = {kk: v.copy() for kk, v in P.items()}
        O.renorm_decoder(Pc)
        fw = O.sae_forward(Pc, x, None, layer_norm=ln, l1_coefficient=l1c, target=y)
        before = stats["act_freq_scores"].copy()
        ref = fw_scalars(fw)
        eng.relu_step(torch.from_numpy(x).cuda(), l1c, want_out=True, target=torch.from_numpy(y).cuda() if tc else None)
        eng.grad_sqnorm()
        torch.cuda.synchronize()
        assert int(eng.relu_mode.item()) == 0, "the sparse form should hold this batch"
        idx, val, cnt = (v.cpu().numpy() for v in eng.relu_pairs())
        gate = np.zeros((n, d_sae), bool)
        rows = np.repeat(np.arange(n), cnt)
        cols = np.concatenate([idx[i, :cnt[i]] for i in range(n)]) if cnt.sum() else np.zeros(0, np.int64)
        gate[rows, cols] = True
        assert gate.sum() == cnt.sum()                                   # no feature twice in a token's list
        differs = gate != (fw["feature_acts"] > 0)
        assert differs.sum() <= 1e-5 * gate.size and np.all(np.abs(fw["hidden_pre"][differs]) < 1e-5 * np.abs(fw["hidden_pre"]).max())
        # values: the exact fp32 pre-activations (descending within a token)
        vals = np.concatenate([val[i, :cnt[i]] for i in range(n)])
        assert np.all(vals > 0) and all(np.all(np.diff(val[i, :cnt[i]]) <= 0) for i in range(0, n, max(1, n // 64)))
        assert np.abs(vals - fw["hidden_pre"][rows, cols]).max() <= 1e-5 * np.abs(fw["hidden_pre"]).max()
        sc = eng.scalars.cpu().numpy()
        assert abs(sc[0] - ref["loss"]) <= TOL * ref["loss"] and abs(sc[1] - ref["mse_loss"]) <= TOL * ref["mse_loss"], (sc, ref)
        assert abs(sc[4] - ref["l1_loss"]) <= TOL * ref["l1_loss"] and abs(sc[2] - ref["l0"]) <= TOL * ref["l0"] + 1e-6, (sc, ref)
        assert rel_fro(eng.sae_out[:n].cpu().numpy(), fw["sae_out"]) < TOL
        gr = O.sae_backward(Pc, x, fw, layer_norm=ln, l1_coefficient=l1c, gate=gate if differs.any() else None)
        assert abs(np.sqrt(sc[3]) - grad_norm_of(gr)) <= TOL * grad_norm_of(gr)
        assert rel_fro(eng.grad_W_enc().cpu().numpy(), gr["W_enc"]) < TOL
        for name in [m for m in gr if m != "W_enc"]:
            assert rel_fro(eng.g[name].cpu().numpy(), gr[name]) < TOL, name
        # (the oracle's step continues under the KERNEL's lists where an entry within noise of zero fell differently)
        O.train_step(P, opt, stats, x, None, lr=1e-3, step=t + 1, layer_norm=ln, l1_coefficient=l1c, target=y,
                     gate=gate if differs.any() else None)
        fire_ref = stats["act_freq_scores"] - before
        assert np.array_equal(eng.fire_count.cpu().numpy(), fire_ref)
        eng.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for name in P:
            assert rel_fro(eng.params[name].cpu().numpy(), P[name]) < TOL, name
        assert np.array_equal(eng.act_freq_scores.cpu().numpy(), stats["act_freq_scores"])
        assert np.array_equal(eng.n_fwd_since_fired.cpu().numpy(), stats["n_fwd_since_fired"])


@pytest.mark.parametrize("d_in,d_sae,n", [(64, 2048, 256), (768, 8192, 1024)])
def test_relu_step_goes_dense_exactly_when_a_token_cannot_be_held(d_in, d_sae, n):
    """The device-side switch: with a capacity of exactly the largest token's count the step runs sparse, with one group of four
    less it must raise its mode word and run the dense GEMMs -- bit for bit what dense_step computes -- and the two forms agree to
    summation-order noise.  At the synthetic init (half of all features positive) the step is dense whatever the capacity."""
    l1c = 3e-3
    P, opt, stats, T = fresh(d_in, d_sae)
    x = synth_sae_batch(n, d_in, seed=3)
    _shift_b_enc_for_l0(P, x, True, 20)
    Pc = {kk: v.copy() for kk, v in P.items()}
    O.renorm_decoder(Pc)
    cnt = (O.sae_forward(Pc, x, None, layer_norm=True)["feature_acts"] > 0).sum(axis=1)
    cap_ok = int((cnt.max() + 3) // 4 * 4)
    xg = torch.from_numpy(x).cuda()

    def run(step, **kw):
        # (the decoder renormalised by the one in-place kernel beforehand: relu_step defers its renorm -- 1 / |row| now, the rows
        # rewritten later -- which rounds differently from dense_step's in-place division; the bit-for-bit claim is about the step)
        Tl = {m: torch.from_numpy(v.copy()).cuda() for m, v in P.items()}
        eng = NativeSAE(Tl["W_enc"], Tl["W_dec"], Tl["b_enc"], Tl["b_dec"], 1, True, n)
        eng.renorm_decoder()
        getattr(eng, step)(xg, l1c, want_out=True, renorm_decoder=False, **kw)
        eng.grad_sqnorm()
        torch.cuda.synchronize()
        return eng

    sparse = run("relu_step", cap=cap_ok)
    dense_forced = run("relu_step", cap=cap_ok - 4)
    dense = run("dense_step")
    assert int(sparse.relu_mode.item()) == 0 and int(dense_forced.relu_mode.item()) == 1
    assert int(sparse.relu_pairs()[2].max().item()) == int(cnt.max())
    for a in ("flat_g", "scalars", "sae_out", "fire_count"):
        assert torch.equal(getattr(dense_forced, a), getattr(dense, a)), a
    assert rel_fro(sparse.flat_g.cpu().numpy(), dense.flat_g.cpu().numpy()) < 1e-5
    assert rel_fro(sparse.sae_out.cpu().numpy(), dense.sae_out.cpu().numpy()) < 1e-5
    assert torch.allclose(sparse.scalars[:5], dense.scalars[:5], rtol=1e-5, atol=0) and torch.equal(sparse.fire_count, dense.fire_count)
    # the init state: 50 % of the features fire
    P0, _, _, T0 = fresh(d_in, d_sae)
    T1 = {m: v.clone() for m, v in T0.items()}                  # (before any step: a step renormalises W_dec in place)
    eng = NativeSAE(T0["W_enc"], T0["W_dec"], T0["b_enc"], T0["b_dec"], 1, True, n)
    eng.renorm_decoder()
    eng.relu_step(xg, l1c, renorm_decoder=False)
    ref = NativeSAE(T1["W_enc"], T1["W_dec"], T1["b_enc"], T1["b_dec"], 1, True, n)
    ref.renorm_decoder()
    ref.dense_step(xg, l1c, renorm_decoder=False)
    torch.cuda.synchronize()
    assert int(eng.relu_mode.item()) == 1 and torch.equal(eng.flat_g, ref.flat_g) and torch.equal(eng.scalars, ref.scalars)
    assert torch.equal(eng.act_freq_scores, ref.act_freq_scores) and torch.equal(eng.n_fwd_since_fired, ref.n_fwd_since_fired)
    # ... and with the renorm left to the steps (deferred in relu_step: a dense step rewrites the rows itself before its GEMMs;
    # in place, first, in dense_step): the same step to rounding
    _, _, _, T2 = fresh(d_in, d_sae)
    T3 = {m: v.clone() for m, v in T2.items()}
    a = NativeSAE(T2["W_enc"], T2["W_dec"], T2["b_enc"], T2["b_dec"], 1, True, n)
    b = NativeSAE(T3["W_enc"], T3["W_dec"], T3["b_enc"], T3["b_dec"], 1, True, n)
    for t in range(2):
        a.relu_step(xg, l1c); a.grad_sqnorm(); a.apply(1e-3, 1.0)
        b.dense_step(xg, l1c); b.grad_sqnorm(); b.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        assert int(a.relu_mode.item()) == 1
        assert torch.allclose(a.scalars[:5], b.scalars[:5], rtol=1e-5, atol=0), t
        for name in ("W_enc", "W_dec", "b_enc", "b_dec"):
            assert rel_fro(a.params[name].cpu().numpy(), b.params[name].cpu().numpy()) < 1e-6, (t, name)


@pytest.mark.parametrize("sg", [False, True])
def test_relu_step_is_bit_reproducible_and_follows_a_collapsing_run(sg):
    """(a) two engines from the same state agree bit for bit over steps that run dense, then sparse; (b) bench.py's ReLU leg from
    the synthetic init (768 -> 8192 here): L0 falls from half of the features to a few within ~8 steps -- the step must change
    form on its own (mode 1, later 0) and track the oracle through the transition.  sg: as the single-process trainer runs it
    (PV_SAE_SPARSE_GRADS, the clip norm from the step's per-feature terms, the decoder renorm deferred across both forms)."""
    d_in, d_sae, n, l1c = 768, 8192, 1024, 8e-5
    P, opt, stats, _ = fresh(d_in, d_sae)
    engines = []
    for _ in range(2):
        _, _, _, T = fresh(d_in, d_sae)
        engines.append(NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], 1, True, n))
    modes = []
    for t in range(12):
        x = synth_sae_batch(n, d_in, seed=t % 4)
        ref = O.train_step(P, opt, stats, x, None, lr=1e-3, step=t + 1, l1_coefficient=l1c)
        xg = torch.from_numpy(x).cuda()
        for e in engines:
            e.relu_step(xg, l1c, want_out=True, sparse_grads=sg)
            e.grad_sqnorm(from_step=sg)
        torch.cuda.synchronize()
        a, b = engines
        assert torch.equal(a.flat_g, b.flat_g) and torch.equal(a.scalars, b.scalars) and torch.equal(a.sae_out, b.sae_out), t
        modes.append(int(a.relu_mode.item()))
        sc = a.scalars.cpu().numpy()
        # (a ReLU gate within fp32 noise of zero may fall either way: losses at 1e-4, l0 to a handful of entries)
        assert abs(sc[0] - ref["loss"]) <= 2e-4 * abs(ref["loss"]) and abs(sc[2] - ref["l0"]) <= 1e-3 * ref["l0"] + 0.01, (t, sc, ref)
        for e in engines:
            e.apply(1e-3, 1.0)
    torch.cuda.synchronize()
    assert modes[0] == 1 and modes[-1] == 0 and modes == sorted(modes, reverse=True), modes
    assert torch.equal(engines[0].W_encT, engines[1].W_encT) and torch.equal(engines[0].params["W_dec"], engines[1].params["W_dec"])


@pytest.mark.parametrize("d_in,d_sae,n,l0", [(768, 8192, 1024, 24), (64, 2048, 256, 0)])
def test_relu_sparse_gradient_step_lands_on_the_same_parameters_as_the_complete_one(d_in, d_sae, n, l0):
    """PV_SAE_SPARSE_GRADS on relu_step (what the single-process trainer passes).  A step that ran SPARSE leaves the gradient rows of
    features no token kept untouched (poison stays exactly there) and apply takes them as zero; a step that ran DENSE (l0 = 0: the
    synthetic init, half of all features positive) writes complete buffers and marks every feature live.  Either way the clip norm
    from the step's per-feature terms equals the full pass's and parameters + Adam moments land where the complete-gradient run's do."""
    l1c = 3e-3
    P, _, _, _ = fresh(d_in, d_sae)
    if l0:
        _shift_b_enc_for_l0(P, synth_sae_batch(n, d_in, seed=0), True, l0)
    engs = []
    for _ in range(2):
        T = {m: torch.from_numpy(v.copy()).cuda() for m, v in P.items()}
        engs.append(NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], 1, True, n))
    full, sparse = engs
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
        full.relu_step(x, l1c); full.grad_sqnorm(); full.apply(1e-3, 1.0)
        sparse.flat_g.fill_(float("nan"))
        sparse.relu_step(x, l1c, sparse_grads=True)
        torch.cuda.synchronize()
        mode = int(sparse.relu_mode.item())
        assert mode == int(full.relu_mode.item()) == (0 if l0 else 1)
        empty = sparse.fire_count == 0
        if mode == 0:
            assert int(empty.sum()) > 0
            assert bool(torch.isnan(sparse.g["W_dec"][empty]).all()) and bool(torch.isnan(sparse.g["W_enc"][empty]).all())
            assert bool(torch.isfinite(sparse.g["W_dec"][~empty]).all()) and bool(torch.isfinite(sparse.g["b_enc"]).all())
        else:
            assert bool(torch.isfinite(sparse.flat_g[:sparse.n_flat]).all())
        with pytest.raises(RuntimeError):
            sparse.grad_sqnorm()
        sparse.grad_sqnorm(from_step=True)
        sparse.apply(1e-3, 1.0)
        torch.cuda.synchronize()
        for i in (0, 1, 2, 3, 4):                            # loss, mse, l0, clip norm, l1
            assert abs(float(sparse.scalars[i]) - float(full.scalars[i])) <= 2e-6 * abs(float(full.scalars[i])), (t, i)
        for name in ("W_enc", "W_dec", "b_enc", "b_dec"):
            assert rel_fro(sparse.params[name].cpu().numpy(), full.params[name].cpu().numpy()) < 1e-6, (t, name)
        assert rel_fro(sparse.flat_m.cpu().numpy(), full.flat_m.cpu().numpy()) < 1e-5, t
        assert rel_fro(sparse.flat_v.cpu().numpy(), full.flat_v.cpu().numpy()) < 1e-5, t
        assert bool(torch.isfinite(sparse.flat_m).all()) and bool(torch.isfinite(sparse.flat_v).all())
