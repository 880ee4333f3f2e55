"""CPU tests of the SAE host side: module + trainer (PyTorch path) against what the reference's real
train_step produced, config round trip, and the data-parallel algebra over a 2-process gloo group."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sae_oracle as O
from vit_prisma_amd.sae import (StandardSparseAutoencoder, VisionModelSAERunnerConfig, VisionSAETrainer)
from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state

from conftest import GOLDEN, rel_fro


def make_cfg(d_in=64, expansion=8, k=8, n=256, **kw):
    base = dict(hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=expansion,
                activation_fn_str="topk", activation_fn_kwargs={"k": k}, normalize_activations="layer_norm",
                initialization_method="independent", b_dec_init_method="mean", train_batch_size=n, lr=1e-3,
                max_grad_norm=1.0, _device="cpu", _dtype="float32", log_to_wandb=False, lr_scheduler_name="constant",
                n_checkpoints=0)
    base.update(kw)
    return VisionModelSAERunnerConfig(**base)


def test_config_json_round_trip_and_derived():
    cfg = make_cfg(768, 32, 32, 4096)
    assert cfg.hook_point == "blocks.6.hook_resid_post" and cfg.d_sae == 24576
    assert cfg.total_training_tokens == 1_300_000 * 50 and cfg.total_training_steps == 65_000_000 // 4096
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "config.json")
        cfg.save_config(p)
        assert VisionModelSAERunnerConfig.load_config(p) == cfg
    with pytest.raises(ValueError):
        make_cfg(b_dec_init_method="bogus")


def test_module_and_trainer_torch_path_match_reference_three_steps():
    g = np.load(os.path.join(GOLDEN, "sae_small_steps.npz"))
    cfg = make_cfg()
    sae = StandardSparseAutoencoder(cfg)
    assert list(sae.state_dict().keys()) == ["W_dec", "W_enc", "b_enc", "b_dec"]
    assert list(sae.hook_dict.keys()) == ["hook_sae_in", "hook_hidden_pre", "hook_hidden_post", "hook_sae_out"]
    with torch.no_grad():
        for n, v in synth_sae_state(64, 512, 0).items():
            getattr(sae, n).copy_(torch.from_numpy(v))
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae)
    act, since, frac, opt, sched = tr.initialize_training_variables()
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(256, 64, seed=t))[:, None, :]
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act,
            n_forward_passes_since_fired=since, n_frac_active_tokens=frac, layer_acts=x, n_training_steps=t,
            n_training_tokens=t * 256)
        want = g[f"s{t}_scalars"]
        assert l1 is None and not tr.last_step_native
        assert abs(float(loss) - want[0]) <= 1e-5 * want[0] and float(l0) == want[2]
        for n in ("W_enc", "W_dec", "b_enc", "b_dec"):
            assert rel_fro(getattr(sae, n).detach().numpy(), g[f"s{t}_param_{n}"]) < 1e-5, (t, n)
        assert np.array_equal(act.numpy(), g[f"s{t}_act_freq"]) and np.array_equal(since.numpy(), g[f"s{t}_n_since"])
    assert frac == 3 * 256
    # 7-tuple contract of forward (sae.py:637-645)
    out = sae(torch.from_numpy(synth_sae_batch(8, 64, 9)))
    assert len(out) == 7 and out[0].shape == (8, 64) and out[1].shape == (8, 512) and out[4] is None
    assert int((out[1] > 0).sum(-1).max()) <= 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d_in, d_sae, k, N = 64, 512, 8, 256
    P = {kk: v.copy() for kk, v in synth_sae_state(d_in, d_sae, 0).items()}
    O.renorm_decoder(P)
    x = synth_sae_batch(N, d_in, seed=0)
    xs = x[rank * (N // world):(rank + 1) * (N // world)]
    # the two pre-reductions of SURVEY.md 8e, exactly as VisionSAETrainer._native_step issues them
    bm = torch.from_numpy(xs.sum(axis=0))
    dist.all_reduce(bm)
    fw = O.sae_forward(P, xs, k, batch_mean=(bm / N).numpy(), n_global=N)
    g = O.sae_backward(P, xs, fw, n_global=N)
    flat = torch.from_numpy(np.concatenate([g[n].ravel() for n in ("W_enc", "W_dec", "b_enc", "b_dec")]))
    dist.all_reduce(flat)                                    # ONE collective for all four gradients
    loss = torch.tensor([float(fw["loss"])], dtype=torch.float64)
    dist.all_reduce(loss)
    if rank == 0:
        q.put((flat.numpy(), float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_algebra_equals_single_process_gloo_world2():
    """Sum over ranks of shard gradients computed with the GLOBAL batch mean and 1/N_global scaling ==
    the single-process gradient at the global batch (the oracle for the 8-GPU path)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat, loss = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d_in, d_sae, k, N = 64, 512, 8, 256
    P = {kk: v.copy() for kk, v in synth_sae_state(d_in, d_sae, 0).items()}
    O.renorm_decoder(P)
    x = synth_sae_batch(N, d_in, seed=0)
    fw = O.sae_forward(P, x, k)
    g = O.sae_backward(P, x, fw)
    ref = np.concatenate([g[n].ravel() for n in ("W_enc", "W_dec", "b_enc", "b_dec")])
    assert rel_fro(flat, ref) < 1e-5
    assert abs(loss - float(fw["loss"])) <= 1e-5 * float(fw["loss"])


def test_on_disk_activation_cache_format_matches_the_reference(tmp_path):
    """SURVEY.md 8f row 2.  tests/golden/act_cache_tiny/ holds the {idx}.pt shards the REFERENCE's writer produced for
    the tiny model (7 images, blocks.1.hook_resid_post, 50 tokens per file) and what the reference's
    CacheVisionActivationStore read back (gen_golden_act_cache.py).  Our writer must produce the same files, our
    reader the same buffer."""
    from vit_prisma_amd import HookedViT, HookedViTConfig
    from vit_prisma_amd.sae import CacheVisionActivationStore, VisionActivationsStore
    from vit_prisma_amd.synth import ARCHS, synth_vit_state
    gold = os.path.join(GOLDEN, "act_cache_tiny")
    ref = np.load(os.path.join(gold, "reference_reader.npz"))
    arch = ARCHS["tiny"]
    model = HookedViT(HookedViTConfig(**arch, device="cpu"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model.eval()
    imgs = torch.from_numpy(ref["images"])
    ds = torch.utils.data.TensorDataset(imgs, torch.zeros(len(imgs), dtype=torch.long))
    cfg = make_cfg(hook_point_layer=1, context_size=17, store_batch_size=2, n_batches_in_buffer=4, train_batch_size=16,
                   cached_activations_path=str(tmp_path / "cache"), use_cached_activations=True)
    store = VisionActivationsStore(cfg, model, ds, create_dataloader=False)
    assert store.generate_cached_activations_from_dataset(tokens_per_file=50) == 3
    for i, rows in enumerate((50, 50, 19)):
        ours, theirs = torch.load(tmp_path / "cache" / f"{i}.pt"), torch.load(os.path.join(gold, f"{i}.pt"))
        assert ours.dtype == torch.float16 and tuple(ours.shape) == (rows, 1, 64) == tuple(theirs.shape)
        assert torch.allclose(ours.float(), theirs.float(), atol=4e-3, rtol=2e-3), i       # one fp16 ulp at |x| ~ 4
    # reader: on the reference's own shards, bit-identical to the reference's reader; then the served batches
    rcfg = make_cfg(hook_point_layer=1, context_size=17, store_batch_size=2, n_batches_in_buffer=4, train_batch_size=16,
                    cached_activations_path=gold, use_cached_activations=True)
    reader = CacheVisionActivationStore(rcfg)
    assert np.array_equal(reader.get_buffer(2).numpy(), ref["buffer_2_batches"])
    assert reader.storage_buffer.shape[1:] == (1, 64) and reader.next_batch().shape == (16, 1, 64)
    seen = sum(reader.next_batch().shape[0] for _ in range(12))                              # crosses several refills
    assert seen > 0
    with pytest.raises(ValueError):
        CacheVisionActivationStore(make_cfg(use_cached_activations=False))


class _LoggedImages(torch.utils.data.Dataset):
    def __init__(self, n):
        self.x = torch.randn(n, 3, 32, 32, generator=torch.Generator().manual_seed(3))
        self.seen = []

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        self.seen.append(int(i))
        return self.x[i], 0


def _harvest_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vit_prisma_amd import HookedViT, HookedViTConfig
    from vit_prisma_amd.sae import VisionActivationsStore
    from vit_prisma_amd.synth import ARCHS, synth_vit_state
    arch = ARCHS["tiny"]
    model = HookedViT(HookedViTConfig(**arch, device="cpu"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    data = _LoggedImages(16)
    cfg = make_cfg(hook_point_layer=1, context_size=17, store_batch_size=2, n_batches_in_buffer=4, train_batch_size=32)
    store = VisionActivationsStore(cfg, model.eval(), data)         # fills one buffer: 4 store batches = 8 images = one epoch share
    first_epoch = list(data.seen[:8])
    batch = store.next_batch()
    q.put((rank, first_epoch, tuple(batch.shape), tuple(store.storage_buffer.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_harvest_shards_images_across_ranks_gloo_world2():
    """with torch.distributed initialised every rank's VisionActivationsStore draws a disjoint shard of
    each epoch's images (no collective in harvesting) and serves train_batch_size / world tokens per step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_harvest_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, seen0, b0, buf0), (r1, seen1, b1, buf1) = got
    assert (r0, r1) == (0, 1)
    assert len(seen0) == len(seen1) == 8 and not set(seen0) & set(seen1) and set(seen0) | set(seen1) == set(range(16))
    assert b0 == b1 == (16, 1, 64)                      # 32 tokens per global step -> 16 per rank
    assert buf0[1:] == buf1[1:] == (1, 64)


def test_readers_of_w_enc_trigger_the_native_sync_hook():
    """A native training engine may keep W_enc's own layout stale between steps (NativeSAE.lazy_w_enc); every way of reaching
    the parameter through the module -- attribute access, state_dict(), parameters() / named_parameters() -- must call the
    registered materialisation first; the raw registry (what the trainer's hot loop uses) must not."""
    from vit_prisma_amd.sae import StandardSparseAutoencoder, VisionModelSAERunnerConfig
    cfg = VisionModelSAERunnerConfig(hook_point_layer=1, layer_subtype="hook_resid_post", d_in=16, expansion_factor=2,
                                     activation_fn_str="topk", activation_fn_kwargs={"k": 2}, _device="cpu", log_to_wandb=False)
    sae = StandardSparseAutoencoder(cfg)
    calls = []
    object.__setattr__(sae, "_native_sync_fn", lambda: calls.append(1))
    _ = sae._parameters["W_enc"]
    _ = sae.W_dec
    assert calls == []
    _ = sae.W_enc
    assert len(calls) == 1
    sae.state_dict()
    assert len(calls) >= 2
    n = len(calls)
    list(sae.parameters())
    assert len(calls) > n
    n = len(calls)
    x = torch.randn(4, 16)
    sae(x)                                               # the PyTorch forward reads self.W_enc
    assert len(calls) > n
