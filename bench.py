"""bench.py -- headline measurement of the MI355X-native hot path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one ``HookedViT.run_with_cache`` pass (ALL hooks, 214 cache entries) of CLIP ViT-B/32
@224 over one batch of 512 synthetic images per GPU, bf16 storage (BASELINE.json configs[1]),
inputs already resident in HBM.  With N > 1 (one process per GPU, launched by torch.distributed.run)
the image batches are sharded across ranks with no data-path collective (weak scaling).
Rank 0 prints ONE JSON line; ``value`` is the whole-job images/s.

Extra objects on the same line:
  roofline      dominant kernel (the MFMA GEMM family): algorithmic FLOPs per launch / average launch
                duration measured with HIP events on the launch stream inside the timed region
  kernels       the same live measurement for the other kernel families (attention, layernorm)
  cpu_baseline  the reference's CPU path (PyTorch fp32 hook path, every host core, warm-up 3 / time 5 / median;
                rank 0, N == 1 only); the numpy oracle's figure rides along as cpu_baseline.numpy_oracle
  sae           SAE train-step tokens/s (BASELINE.json configs[2]; second half of the metric)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0

# algorithmic work per image, CLIP ViT-B/32 all hooks (SURVEY.md 8d / DESIGN.md section 4)
FLOP_PER_IMAGE_B32 = 8.818e9
TAP_BYTES_PER_IMAGE_B32_BF16 = 20.34e6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="images per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sae", action="store_true")
    ap.add_argument("--no-l14", action="store_true", help="skip the L/14@336 pattern-only leg")
    ap.add_argument("--l14-batch", type=int, default=128, help="images per GPU of the L/14@336 leg (128 = BASELINE config 5; a rehearsal of "
                                                                 "many ranks on ONE GPU passes less: 32.7 GB of pattern taps per rank at 128)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--allow-overrides", action="store_true", help="A/B runs only: measure with PV_* env / tuning overrides (recorded)")
    ap.add_argument("--sae-parallel", default="feature", choices=["data", "feature"],
                    help="under torchrun: the SAE legs shard the features (feature, default: token-sized collectives only) or "
                         "tokens + optimizer (data: parameter-sized reduce-scatter / all-gather)")
    ap.add_argument("--leg-timeout", type=float, default=420.0,
                    help="under torchrun: seconds the secondary legs (SAE, L/14) may take before the main line is printed without them")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B runs only (needs --allow-overrides): pv_debug_set_tuning(KEY, VALUE) before measuring")
    return ap.parse_args()


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def _usable_cpus() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def _best_thread_count(run_once) -> tuple:
    """PyTorch's CPU path does not scale to every hardware thread of a big host (on the 256-thread GPU box
    torch.set_num_threads(256) made one B/32 forward take 106 s; the eager path is hundreds of small ops, each an OpenMP
    fork/join).  The baseline is therefore the BEST thread count of an upward sweep (8, 16, 32, ... usable CPUs), which
    is the most favourable honest figure for the CPU.  Returns (threads, seconds per call, {threads: seconds})."""
    ncpu = _usable_cpus()
    cands = sorted({c for c in (8, 16, 32, 64, 128, ncpu) if c <= ncpu} or {ncpu})
    seen, best = {}, None
    for nt in cands:
        torch.set_num_threads(nt)
        run_once()                                   # warm-up at this thread count
        t0 = time.perf_counter()
        run_once()
        dt = time.perf_counter() - t0
        seen[nt] = round(dt, 4)
        if best is None or dt < best[1]:
            best = (nt, dt)
        elif dt > 1.5 * best[1]:
            break                                    # past the knee: larger counts only get worse
    torch.set_num_threads(best[0])
    return best[0], best[1], seen


def cpu_baseline_torch(seconds: float, bs: int = 32) -> dict:
    """The reference's CPU path as SURVEY.md 8(d) defines it: the PyTorch hook path (this repo's faithful
    re-implementation of HookedViT.run_with_cache, pinned to reference-generated goldens by
    tests/test_vit_host_vs_golden.py; /root/reference does not exist on the GPU box) in fp32 on every host core
    (torch.set_num_threads(os.cpu_count())), all 214 hooks, warm-up 3 / time 5 / median."""
    from vit_prisma_amd import HookedViT, HookedViTConfig
    from vit_prisma_amd.synth import ARCHS, synth_images, synth_vit_state
    ncpu = os.cpu_count() or 1
    arch = ARCHS["clip-vit-b32"]
    model = HookedViT(HookedViTConfig(**arch, dtype=torch.float32, device="cpu"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model = model.eval().use_native(False)
    x = torch.from_numpy(synth_images(arch, bs, 1))

    def once():
        with torch.no_grad():
            _, cache = model.run_with_cache(x)
        assert len(cache) == 214 and not model.last_run_native

    nt, _, sweep = _best_thread_count(once)          # (its runs are the warm-up)
    times = []
    t_all = time.perf_counter()
    for i in range(5):
        t0 = time.perf_counter()
        once()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > seconds:
            break
    med = _median(times)
    return {"value": round(bs / med, 1), "unit": "images/s", "cores": nt, "kind": "port",
            "sample": f"median of {len(times)} x run_with_cache(all 214 hooks) of CLIP ViT-B/32 at bs={bs}, fp32, PyTorch CPU hook path "
                      f"(the reference's algorithm op for op) at the best thread count of a sweep {sweep} (s per forward) on a "
                      f"{ncpu}-thread host ({_usable_cpus()} usable), {med * 1e3:.0f} ms per forward" + _port_vs_reference("vit")}


def _port_vs_reference(which: str) -> str:
    """What the build-container measurement of the REFERENCE'S OWN classes beside this port says (tools/cpu_reference_vs_port.py ->
    profiles/r04_cpu_reference_vs_port.json; /root/reference does not exist on the GPU box, so 'kind' stays 'port')."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_cpu_reference_vs_port.json")) as f:
            d = json.load(f)[which]
        ref = d.get("reference_images_per_s", d.get("reference_tokens_per_s"))
        port = d.get("port_images_per_s", d.get("port_tokens_per_s"))
        return (f"; the reference's own classes timed beside this port on the {json.load(open(os.path.join(ROOT, 'profiles', 'r04_cpu_reference_vs_port.json')))['host']['cores']} "
                f"cores of the build container: reference {ref}, port {port} (profiles/r04_cpu_reference_vs_port.json)")
    except Exception:
        return ""


def sae_cpu_baseline_torch(seconds: float) -> dict:
    """The reference trainer's step on CPU: VisionSAETrainer.train_step on the PyTorch-autograd path (the reference
    algorithm verbatim, train_sae.py:278-411), fp32, all host cores; warm-up 1 / time up to 5 / median."""
    from vit_prisma_amd.sae import StandardSparseAutoencoder, VisionModelSAERunnerConfig, VisionSAETrainer
    from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state
    d_in, d_sae, k, n_tok = 768, 24576, 32, 4096
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=d_sae // d_in,
        activation_fn_str="topk", activation_fn_kwargs={"k": k}, normalize_activations="layer_norm",
        b_dec_init_method="mean", train_batch_size=n_tok, lr=1e-3, max_grad_norm=1.0, _device="cpu", log_to_wandb=False,
        lr_scheduler_name="constant", n_checkpoints=0)
    sae = StandardSparseAutoencoder(cfg)
    with torch.no_grad():
        for n, v in synth_sae_state(d_in, d_sae, 0).items():
            getattr(sae, n).copy_(torch.from_numpy(v))
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae)
    act, since, frac, opt, sched = tr.initialize_training_variables()
    x = torch.from_numpy(synth_sae_batch(n_tok, d_in, seed=0))[:, None, :]
    state = [act, since, frac, 0]

    def once():
        i = state[3]
        _, _, _, _, state[0], state[1], state[2] = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=state[0], n_forward_passes_since_fired=state[1],
            n_frac_active_tokens=state[2], layer_acts=x, n_training_steps=i, n_training_tokens=i * n_tok)
        state[3] += 1
        assert not tr.last_step_native

    nt, _, sweep = _best_thread_count(once)
    times = []
    t_all = time.perf_counter()
    for i in range(5):
        t0 = time.perf_counter()
        once()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > seconds:
            break
    med = _median(times)
    return {"value": round(n_tok / med, 1), "unit": "tokens/s", "cores": nt, "kind": "port",
            "sample": f"median of {len(times)} full train steps of {n_tok} tokens (768 -> 24576, k=32), fp32, PyTorch CPU autograd path "
                      f"(the reference trainer's algorithm) at the best thread count of a sweep {sweep} (s per step), {med * 1e3:.0f} ms per step"
                      + _port_vs_reference("sae")}


def cpu_baseline(seconds: float) -> dict:
    """Oracle (port of the reference algorithm) on the host cores, same workload at bs=16 per worker
    (images/s is flat in batch size on CPU).  numpy's elementwise kernels are
    single-threaded, so the host is filled with several concurrent forwards (numpy releases the GIL)
    each with a bounded BLAS thread pool; `cores` = worker threads x BLAS threads actually used."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.vit_oracle import vit_forward
    from vit_prisma_amd.synth import ARCHS, synth_images, synth_vit_state
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    arch = ARCHS["clip-vit-b32"]
    sd = synth_vit_state(arch, 0)
    imgs = synth_images(arch, 16, 1)
    ncpu = os.cpu_count() or 1
    workers = max(1, min(16, ncpu // 4))
    blas = max(1, min(8, ncpu // workers))

    def one(_):
        _, cache = vit_forward(sd, arch, imgs)
        return len(cache)

    def run():
        vit_forward(sd, arch, imgs)          # warm-up
        n, t0 = 0, time.perf_counter()
        with ThreadPoolExecutor(workers) as ex:
            while True:
                assert all(k == 214 for k in ex.map(one, range(workers)))
                n += workers
                dt = time.perf_counter() - t0
                if dt >= seconds or n >= 64 * workers:
                    break
        return n, dt

    if threadpool_limits is not None:
        with threadpool_limits(limits=blas):
            n, dt = run()
    else:
        n, dt = run()
    return {"value": 16 * n / dt, "unit": "images/s", "cores": workers * blas, "kind": "port",
            "sample": f"{n} x run_with_cache(all 214 hooks) of CLIP ViT-B/32 at bs=16, fp32 numpy oracle, {workers} concurrent "
                      f"forwards x {blas} BLAS threads, {dt:.1f} s on a {ncpu}-thread host"}


def sae_cpu_baseline(seconds: float) -> dict:
    """The oracle's full SAE train step (numpy port of the reference algorithm) on the host cores."""
    from oracle import sae_oracle as O
    from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state
    d_in, d_sae, k, n_tok = 768, 24576, 32, 4096
    P = {kk: v.copy() for kk, v in synth_sae_state(d_in, d_sae, 0).items()}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
    x = synth_sae_batch(n_tok, d_in, seed=0)
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(P, opt, stats, x, k, lr=1e-3, step=n + 1)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 8:
            break
    return {"value": n_tok * n / dt, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} full train steps of {n_tok} tokens (768 -> 24576, k=32), fp32 numpy oracle (BLAS-threaded), {dt:.1f} s"}


def pmc_traffic(kernel_prefix: str):
    """HBM bytes per launch of a kernel family from the newest committed rocprofv3 PMC summary of the B/32 bs=512 forward
    (profiles/r*_pmc_traffic_vit.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 2x FETCH correction), as the
    launch-weighted mean over every kernel of the family (the GEMM family = all its template instances: what the
    HIP events of the timed region bracket).  None when no PMC summary is committed."""
    import glob
    # the B/32 bs=512 passes only (r02_pmc_traffic_vit.json, round 1: r01_pmc_traffic_v7.json), newest round first by NAME:
    # modification times mean nothing in a fresh copy of the tree, and the SAE / L/14 summaries hold other workloads
    paths = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json"))
                   if os.path.basename(p).endswith(("_vit.json", "_v7.json")))
    for path in reversed(paths):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        tot = n = 0
        names = []
        for k, v in d.get("kernels", {}).items():
            if k.startswith(kernel_prefix) and "hbm_bytes_per_launch_corrected" in v:
                tot += v["hbm_bytes_per_launch_corrected"] * v.get("launches", 1)
                n += v.get("launches", 1)
                names.append(k)
        if n:
            return {"bytes_per_launch": int(tot / n), "source": os.path.basename(path), "kernel": ", ".join(names)}
    return None


def l14_pattern_leg(dev, dist, batch: int = 128, steps: int = 10, warmup: int = 2) -> dict:
    """BASELINE.json configs[4] / SURVEY.md 8d config 5: CLIP ViT-L/14 @336, bs = 128, bf16, only the 24
    ``attn.hook_pattern`` tensors tapped ([128, 16, 577, 577] each = 32.7 GB of taps per step)."""
    import time as _t
    import torch
    from vit_prisma_amd import HookedViT, HookedViTConfig
    from vit_prisma_amd.synth import ARCHS, synth_vit_state
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    arch = ARCHS["clip-vit-l14-336"]
    model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    images = torch.randn(batch, 3, 336, 336, device=dev, generator=g).to(torch.bfloat16)
    keep = lambda n: n.endswith("attn.hook_pattern")  # noqa: E731
    n_keys = 0
    with torch.no_grad():
        for _ in range(warmup):
            out, cache = model.run_with_cache(images, names_filter=keep)
            n_keys = len(cache)
            del out, cache
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = _t.perf_counter()
        for _ in range(steps):
            out, cache = model.run_with_cache(images, names_filter=keep)
            del out, cache
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = _t.perf_counter() - t0
    assert model.last_run_native and n_keys == arch["n_layers"], (model.native_fallback_reason, n_keys)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ips = batch * steps * world / elapsed
    flop_img, bytes_img = 381.92e9, 255.7e6                      # SURVEY.md 8d, algorithmic work per image
    return {"metric": "images/sec run_with_cache (attn.hook_pattern only) CLIP ViT-L/14 @336", "value": round(ips, 1),
            "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 2),
            "dtype": "bf16", "config": {"workload": f"CLIP ViT-L/14@336, bs={batch}/GPU, 24 pattern taps"},
            "roofline": {"bound": "mfma", "achieved": round(ips / world * flop_img / 1e12, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ips / world * flop_img / 1e12 / PEAK_BF16_TFLOPS, 4),
                         "hbm_GBps_algorithmic": round(ips / world * bytes_img / 1e9, 1)}}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # BENCH_BACKEND=gloo: rehearsal of the multi-rank code path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives go through the host) -- never a measurement, the line carries the backend
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        # launched by torch.distributed.run (also with a single rank, so the RCCL path is exercised on 1 GPU)
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from vit_prisma_amd import HookedViT, HookedViTConfig, _native as N
    from vit_prisma_amd.synth import ARCHS, synth_vit_state

    # a measurement is only valid on the kernels the library picks by itself: no PV_* environment variable, no
    # pv_debug_set_tuning override (the launch path reads neither the environment nor anything else that could skip work)
    pv_env = sorted(k for k in os.environ if k.startswith("PV_"))
    for kv in a.tune:
        assert a.allow_overrides, "--tune is for A/B runs: pass --allow-overrides"
        key, val = kv.split("=")
        N.set_tuning(key, int(val))
    if (pv_env or N.get_tuning("any")) and not a.allow_overrides:
        raise SystemExit(f"bench.py refuses to measure with kernel overrides active: env {pv_env}, tuning {N.get_tuning('any')}")

    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    arch = ARCHS["clip-vit-b32"]
    model = HookedViT(HookedViTConfig(**arch, dtype=dtype, device="cuda"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model = model.to(dtype).to(dev).eval().use_native(True)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(a.batch, 3, 224, 224, device=dev, generator=g).to(dtype)

    def step():
        out, cache = model.run_with_cache(images)
        return out, cache

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(a.warmup):
            out, cache = step()
            n_keys = len(cache)
            del out, cache
        N.prof_reset()
        N.prof_enable(True, kinds=("gemm",))      # the timed region carries HIP events around the dominant kernel only
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out, cache = step()
            del out, cache
        barrier()
        elapsed = time.perf_counter() - t0
        N.prof_enable(False)
        # the secondary kernels' figures come from two extra, UNTIMED steps (every event pair costs stream time)
        N.prof_enable(True, kinds=("attention", "layernorm"))
        for _ in range(2):
            out, cache = step()
            del out, cache
        torch.cuda.synchronize(dev)
        N.prof_enable(False)
    assert n_keys == 214 and model.last_run_native
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    images_total = a.batch * a.steps * world
    value = images_total / elapsed
    ms_per_step = elapsed / a.steps * 1e3

    gemm = N.prof_read("gemm")
    attn = N.prof_read("attention")
    ln = N.prof_read("layernorm")
    peak_tf = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
    gemm_tf = gemm["flops"] / max(gemm["ms"], 1e-9) / 1e9
    roofline = {
        "kernel": "gemm_kernel_v8 / v7 family (bf16 MFMA GEMM, 320x256 tile, fused bias / activation / residual / tap-store epilogues; v8 = the persistent form, one workgroup per CU walking its tiles, where a launch has more tiles than CUs; + the patch-embed and head GEMMs)",
        "bound": "mfma", "achieved": round(gemm_tf, 2), "peak": peak_tf, "unit": "TFLOP/s",
        "frac": round(gemm_tf / peak_tf, 4),
        "launches": gemm["launches"], "avg_launch_us": round(gemm["ms"] * 1e3 / max(gemm["launches"], 1), 2),
        "flops_per_launch": gemm["flops"] / max(gemm["launches"], 1),
        "hbm_GBps_algorithmic": round(gemm["bytes"] / max(gemm["ms"], 1e-9) / 1e6, 1),
        "share_of_step": round(gemm["ms"] / (ms_per_step * a.steps), 4),
        "traffic": None,
    }
    # per GEMM instance: both roofs (the bf16 MFMA peak and the HBM floor of its own algorithmic bytes), max of the two fractions --
    # the O-projection and MLP-1 move the most bytes per flop of the four
    inst = {}
    for nm, tag in N.GEMM_TAGS.items():
        g_ = N.prof_read_tag("gemm", tag)
        if g_["launches"]:
            tf_ = g_["flops"] / max(g_["ms"], 1e-9) / 1e9
            gb_ = g_["bytes"] / max(g_["ms"], 1e-9) / 1e6
            inst[nm] = {"launches": g_["launches"], "avg_launch_us": round(g_["ms"] * 1e3 / g_["launches"], 2),
                        "TFLOPs": round(tf_, 1), "frac_mfma": round(tf_ / peak_tf, 4), "GBps_algorithmic": round(gb_, 1),
                        "frac_hbm": round(gb_ / PEAK_HBM_GBS, 4), "frac": round(max(tf_ / peak_tf, gb_ / PEAK_HBM_GBS), 4)}
    roofline["instances"] = inst
    tr = pmc_traffic("gemm_kernel") if a.dtype == "bf16" and a.batch == 512 else None
    if tr is not None:
        roofline["traffic"] = tr["bytes_per_launch"]
        roofline["traffic_source"] = f"{tr['source']} ({tr['kernel']}); algorithmic bytes per launch = {gemm['bytes'] / max(gemm['launches'], 1):.4g}"
    kernels = {}
    for nm, k in (("attention", attn), ("layernorm", ln)):
        kernels[nm] = {
            "bound": "hbm", "achieved": round(k["bytes"] / max(k["ms"], 1e-9) / 1e6, 1), "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": round(k["bytes"] / max(k["ms"], 1e-9) / 1e6 / PEAK_HBM_GBS, 4),
            "launches": k["launches"], "avg_launch_us": round(k["ms"] * 1e3 / max(k["launches"], 1), 2),
            "share_of_step": round(k["ms"] / (ms_per_step * 2), 4), "sampled": "2 untimed steps after the timed region"}
    per_gpu = value / world
    whole = {
        "flop_frac_of_mfma_peak": round(per_gpu * FLOP_PER_IMAGE_B32 / (peak_tf * 1e12), 4),
        "tap_store_GBps": round(per_gpu * TAP_BYTES_PER_IMAGE_B32_BF16 * (1.0 if a.dtype == "bf16" else 36.68 / 20.34) / 1e9, 1),
    }

    line = {
        "metric": "images/sec run_with_cache (all hooks) CLIP ViT-B/32 @224",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic (randn images, seeded random weights in the converted CLIP layout)",
        "config": {"workload": f"CLIP ViT-B/32 run_with_cache, all 214 hooks tapped to the HBM arena, "
                               f"bs={a.batch}/GPU, {a.dtype}", "images_per_gpu_per_step": a.batch,
                   "cache_keys": n_keys, "parallelism": f"image-batch sharding x{world}, no data-path collective"},
        "roofline": roofline, "kernels": kernels, "whole_forward": whole,
        "overrides": {"env": pv_env, "tuning": N.get_tuning("any"), "tune": list(a.tune)},
    }
    if backend != "nccl":
        line["rehearsal_backend"] = backend

    # The secondary legs come after the main line is complete.  Under torchrun (world > 1) they contain collectives; a rank
    # that fails or stalls there must not take the main line with it: a watchdog prints what is there and leaves.
    import threading
    done = threading.Event()

    def emit():
        if rank == 0:
            print(json.dumps(line), flush=True)

    def emergency():
        if not done.is_set():
            line["secondary_legs"] = f"abandoned after {a.leg_timeout:.0f} s (see stderr)"
            line["ok"] = False
            sys.stderr.write(f"[bench rank {rank}] secondary legs did not finish in {a.leg_timeout:.0f} s: printing the main line only\n")
            emit()
            sys.stdout.flush()
            os._exit(0)

    wd = None
    if world > 1:
        wd = threading.Timer(a.leg_timeout, emergency)
        wd.daemon = True
        wd.start()

    def leg(name, fn):
        """Single process: errors propagate (a broken leg must be seen).  Under torchrun: recorded in the line."""
        if world == 1:
            return fn()
        try:
            return fn()
        except Exception as e:                                   # noqa: BLE001
            sys.stderr.write(f"[bench rank {rank}] leg {name} failed: {type(e).__name__}: {e}\n")
            return {"error": f"{type(e).__name__}: {e}"}

    if not a.no_sae:
        from vit_prisma_amd.sae.bench_leg import sae_bench_leg, sae_end_to_end_leg
        del model, images
        torch.cuda.empty_cache()
        fp = a.sae_parallel == "feature"
        sae = leg("sae", lambda: sae_bench_leg(dev, dist=dist, feature_parallel=fp))
        torch.cuda.empty_cache()
        e2e = leg("sae_end_to_end", lambda: sae_end_to_end_leg(dev, dist=dist, feature_parallel=fp))
        torch.cuda.empty_cache()
        if world == 1:
            # the same loop at the reference config's own store shape (store_batch_size 32 x n_batches_in_buffer 20, config.py:351-352):
            # 16 000 tokens per refill = 3.9 train steps, the ViT harvests 32 images at a time
            e2e_ref = leg("sae_end_to_end_ref_store", lambda: sae_end_to_end_leg(dev, dist=None, steps=40, warmup=24, store_bs=32, n_buf=20))
            if isinstance(e2e, dict) and isinstance(e2e_ref, dict):
                e2e["reference_store_shape"] = {k: e2e_ref.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "harvested_tokens_per_trained_token", "error") if k in e2e_ref}
            torch.cuda.empty_cache()
        if world > 1:
            # SURVEY.md 8(d) config 4 asks for both: strong (global batch 4096, above) and weak (4096 tokens per GPU) scaling
            weak = leg("sae_weak", lambda: sae_bench_leg(dev, dist=dist, feature_parallel=False, weak=True))
            # ... and the OTHER partitioning of the strong-scaling step beside the one `sae` ran (--sae-parallel feature, the default: features
            # sharded for good; "data": north_star's -- tokens sharded by image batch, the gradient rows reduce-scattered, the optimizer
            # sharded): no hardware curve exists for either (profiles/README.md), the first 8-GPU run decides between them
            torch.cuda.empty_cache()
            other = leg("sae_other_partitioning", lambda: sae_bench_leg(dev, dist=dist, feature_parallel=not fp))
        else:
            # the ReLU + L1 SAE (every published CLIP SAE of the reference) on the dense fused step
            weak = None
            # ReLU + L1: the run from the synthetic init once L0 has collapsed (10 warm-up steps; the first ~5 are dense), the same run
            # from step 1 (dense, then sparse), and the L0 of the published x64 SAEs (3.5 % of the features: dense)
            relu = leg("sae_relu_l1", lambda: sae_bench_leg(dev, dist=None, steps=20, warmup=10, activation="relu"))
            relu_init = leg("sae_relu_l1_from_init", lambda: sae_bench_leg(dev, dist=None, steps=10, warmup=0, activation="relu"))
            relu_pub = leg("sae_relu_l1_published_l0", lambda: sae_bench_leg(dev, dist=None, steps=5, warmup=2, activation="relu",
                                                                              relu_target_l0=0.035 * 24576))
            relu_l64 = leg("sae_relu_l1_l0_64", lambda: sae_bench_leg(dev, dist=None, steps=10, warmup=2, activation="relu",
                                                                       relu_target_l0=64.0))
            from vit_prisma_amd.sae.bench_leg import sae_variants_leg
            torch.cuda.empty_cache()
            variants = leg("sae_variants", lambda: sae_variants_leg(dev))
        if rank == 0:
            line["sae"] = sae
            sae["end_to_end"] = e2e
            if weak is not None:
                sae["weak_scaling_data_parallel"] = weak
                sae["strong_scaling_data_parallel" if fp else "strong_scaling_feature_parallel"] = other
            else:
                sae["relu_l1"] = relu
                if isinstance(relu, dict):
                    pick = ("value", "unit", "ms_per_step", "steps", "warmup", "l0", "sparse_steps", "dense_steps", "config", "roofline", "error")
                    relu["from_init"] = {k: relu_init.get(k) for k in pick if isinstance(relu_init, dict) and k in relu_init}
                    relu["published_l0"] = {k: relu_pub.get(k) for k in pick if isinstance(relu_pub, dict) and k in relu_pub}
                    relu["l0_64"] = {k: relu_l64.get(k) for k in pick if isinstance(relu_l64, dict) and k in relu_l64}
                sae["variants"] = variants
            if world == 1 and not a.no_cpu_baseline:
                sae["cpu_baseline"] = sae_cpu_baseline_torch(10.0)
                sae["cpu_baseline"]["numpy_oracle"] = sae_cpu_baseline(6.0)
    if not a.no_l14:
        torch.cuda.empty_cache()
        l14 = leg("l14_336_pattern", lambda: l14_pattern_leg(dev, dist, batch=a.l14_batch))
        if rank == 0:
            line["l14_336_pattern"] = l14
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_torch(a.cpu_seconds)
        line["cpu_baseline"]["numpy_oracle"] = cpu_baseline(min(a.cpu_seconds, 8.0))
        line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
    done.set()
    if wd is not None:
        wd.cancel()

    def leg_errors(obj, path=""):
        out = []
        if isinstance(obj, dict):
            if "error" in obj and isinstance(obj["error"], str):
                out.append(f"{path or 'line'}: {obj['error']}")
            for k, v in obj.items():
                out.extend(leg_errors(v, f"{path}.{k}" if path else k))
        return out

    # the headline numbers of the secondary legs once more, short and LAST (a consumer that keeps only the head or only the tail of this
    # long line still sees them: round 5's driver record lost the step-only SAE number inside the end-to-end leg's text)
    def pick(obj, *path):
        for k in path:
            obj = obj.get(k) if isinstance(obj, dict) else None
        return obj

    sae_obj = line.get("sae")
    if isinstance(sae_obj, dict):
        summary = {
            "sae_step_only_tokens_per_s": pick(sae_obj, "value"), "sae_step_only_ms": pick(sae_obj, "ms_per_step"),
            "sae_step_roofline_frac": pick(sae_obj, "roofline", "frac"), "sae_step_hbm_GB": pick(sae_obj, "roofline", "traffic"),
            "sae_end_to_end_tokens_per_s": pick(sae_obj, "end_to_end", "value"),
            "sae_relu_l1_published_l0_ms": pick(sae_obj, "relu_l1", "published_l0", "ms_per_step"),
            "sae_relu_l1_sparse_ms": pick(sae_obj, "relu_l1", "ms_per_step"),
            "sae_gated_relu_ms": pick(sae_obj, "variants", "gated_relu", "ms_per_step"),
            "sae_weak_scaling_tokens_per_s": pick(sae_obj, "weak_scaling_data_parallel", "value"),
            "l14_336_images_per_s": pick(line, "l14_336_pattern", "value"),
        }
        line["summary"] = {k: v for k, v in summary.items() if v is not None}
    errs = leg_errors(line)
    line["ok"] = not errs                          # a failed secondary leg is recorded and flagged here (the main line stays valid)
    if errs:
        line["leg_errors"] = errs
    emit()
    if dist is not None:
        # (the line is out: a rank that died in a secondary leg must not hold the others in a barrier)
        try:
            dist.destroy_process_group()
        except Exception:                                        # noqa: BLE001
            pass
    # (exit code stays 0 under torchrun: a non-zero rank exit would make the launcher tear the job down and lose the line;
    #  consumers read "ok" / "leg_errors")


if __name__ == "__main__":
    main()
