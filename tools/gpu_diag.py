"""GPU bring-up diagnostics (not a test, not shipped): runs every native kernel path against the
oracle / torch and prints per-key errors so that many bugs can be fixed per GPU call.

    python tools/gpu_diag.py [section ...]      sections: unit tiny b32 bf16 perf l14
Writes gpurun_out/diag.json.
"""
from __future__ import annotations

import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.vit_oracle import vit_forward  # noqa: E402
from vit_prisma_amd import HookedViT, HookedViTConfig, _native as N  # noqa: E402
from vit_prisma_amd.synth import ARCHS, synth_images, synth_vit_state  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
RES = {}
dev = torch.device("cuda:0")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))


def stream():
    return torch.cuda.current_stream(dev).cuda_stream


def unit():
    L = N.lib()
    r = {}
    for eb, dt in ((2, torch.int16), (4, torch.int32)):
        x = torch.randint(-30000, 30000, (5, 77, 45), dtype=dt, device=dev)
        y = torch.empty(5, 45, 77, dtype=dt, device=dev)
        N.check(L.pv_transpose_batched(eb, x.data_ptr(), y.data_ptr(), 5, 77, 45, stream()), "transpose")
        torch.cuda.synchronize()
        r[f"transpose{eb}"] = bool(torch.equal(y, x.transpose(1, 2).contiguous()))
    for name, dtype, code, tol in (("f32", torch.float32, 0, 1e-5), ("bf16", torch.bfloat16, 1, 1e-2)):
        for (M, Nn, K) in ((128, 128, 128), (200, 96, 64), (257, 10, 432), (64, 160, 96), (1000, 768, 3072), (33, 24, 588)):
            g = torch.Generator(device="cpu").manual_seed(M + Nn + K)
            A = torch.randn(M, K, generator=g).to(dev).to(dtype)
            Bt = torch.randn(Nn, K, generator=g).to(dev).to(dtype)
            bias = torch.randn(Nn, generator=g).to(dev).to(dtype)
            Cc = torch.full((M, Nn), float("nan"), device=dev, dtype=dtype)
            N.check(L.pv_gemm_bias(code, A.data_ptr(), K, Bt.data_ptr(), K, bias.data_ptr(), Cc.data_ptr(), Nn, M, Nn, K,
                                   stream()), "gemm")
            torch.cuda.synchronize()
            ref = A.double() @ Bt.double().T + bias.double()
            e = rel(Cc.double().cpu().numpy(), ref.cpu().numpy())
            r[f"gemm_{name}_{M}x{Nn}x{K}"] = e
            print(f"gemm {name} {M}x{Nn}x{K}: rel {e:.3e} {'OK' if e < tol else 'FAIL'}", flush=True)
            if not (e < tol):
                d = (Cc.double() - ref).abs()
                bad = (d > 10 * tol * ref.abs().max()).nonzero()
                print("   first bad idx:", bad[:8].tolist(), "n_bad", len(bad), "nan", int(torch.isnan(Cc.double()).sum()))
    RES["unit"] = r


def build(arch_name, dtype):
    arch = ARCHS[arch_name]
    cfg = HookedViTConfig(**arch, dtype=dtype, device="cuda")
    model = HookedViT(cfg)
    sd = {k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}
    model.load_state_dict(sd, strict=True)
    model = model.to(dtype).to(dev).eval()
    model.use_native(True)
    return model, arch


def compare(arch_name, bs, dtype, label, tol, stop=None, names_filter=None, budget=None):
    model, arch = build(arch_name, dtype)
    imgs = synth_images(arch, bs, 1)
    t0 = time.time()
    o_ref, c_ref = vit_forward(synth_vit_state(arch, 0), arch, imgs, stop_at_layer=stop, names_filter=names_filter)
    t_or = time.time() - t0
    with torch.no_grad():
        out, cache = model.run_with_cache(torch.from_numpy(imgs).to(dev).to(dtype), stop_at_layer=stop,
                                          names_filter=names_filter)
    torch.cuda.synchronize()
    assert model.last_run_native
    r = {"keys_equal": list(cache.keys()) == list(c_ref.keys()), "oracle_s": t_or, "errs": {}}
    worst = (0.0, None)
    nbad = 0
    for k in c_ref:
        if k not in cache.cache_dict:
            print(f"[{label}] MISSING {k}")
            continue
        got = cache[k].float().cpu().numpy()
        if got.shape != c_ref[k].shape:
            print(f"[{label}] SHAPE {k}: {got.shape} vs {c_ref[k].shape}")
            nbad += 1
            continue
        e = rel(got, c_ref[k])
        r["errs"][k] = e
        lim = tol if budget is None else max(2.0 * budget.get(k, {}).get("rel_fro", tol), 1e-6)
        if not (e <= lim):
            nbad += 1
            if nbad <= 40:
                print(f"[{label}] BAD {k}: rel {e:.3e} (limit {lim:.3e}) nan={int(np.isnan(got).sum())}")
        if e > worst[0] or e != e:
            worst = (e, k)
    eo = rel(out.float().cpu().numpy(), o_ref)
    r["out_err"] = eo
    r["n_bad"] = nbad
    print(f"[{label}] keys_equal={r['keys_equal']} n_keys={len(c_ref)} n_bad={nbad} worst={worst} out_err={eo:.3e}", flush=True)
    RES[label] = r
    return model


def tiny():
    for arch in ("tiny", "tiny-ragged"):
        for dtype, tol, nm in ((torch.float32, 2e-5, "f32"), (torch.bfloat16, 3e-2, "bf16")):
            try:
                compare(arch, 3, dtype, f"{arch}_{nm}", tol)
                compare(arch, 2, dtype, f"{arch}_{nm}_stop1", tol, stop=1)
            except Exception:
                traceback.print_exc()
                RES[f"{arch}_{nm}"] = {"exception": traceback.format_exc()}


def b32():
    compare("clip-vit-b32", 16, torch.float32, "b32_f32_bs16", 1e-4)
    compare("clip-vit-b32", 16, torch.float32, "b32_f32_stop7", 1e-4, stop=7, names_filter=["blocks.6.hook_resid_post"])


def bf16():
    with open(os.path.join(ROOT, "tests", "golden", "vit_b32_bf16_budget.json")) as f:
        budget = json.load(f)["budget"]
    compare("clip-vit-b32", 4, torch.bfloat16, "b32_bf16_bs4", 3e-2, budget=budget)


def perf():
    r = {}
    for dtype, nm in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        model, arch = build("clip-vit-b32", dtype)
        for bs in (16, 128, 512):
            x = torch.randn(bs, 3, 224, 224, device=dev).to(dtype)
            with torch.no_grad():
                for _ in range(3):
                    out, cache = model.run_with_cache(x)
                    del out, cache
                torch.cuda.synchronize()
                t0 = time.time()
                n = 10
                for _ in range(n):
                    out, cache = model.run_with_cache(x)
                    del out, cache
                torch.cuda.synchronize()
                dt = (time.time() - t0) / n
            r[f"{nm}_bs{bs}_all"] = bs / dt
            print(f"perf {nm} bs={bs} all hooks: {dt * 1e3:.2f} ms  {bs / dt:.0f} img/s  arena alloc/reuse {model._native.arena.n_alloc}/{model._native.arena.n_reuse}", flush=True)
            with torch.no_grad():
                for _ in range(2):
                    model.run_with_cache(x, names_filter=["blocks.6.hook_resid_post"], stop_at_layer=7)
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(n):
                    model.run_with_cache(x, names_filter=["blocks.6.hook_resid_post"], stop_at_layer=7)
                torch.cuda.synchronize()
                dt = (time.time() - t0) / n
            r[f"{nm}_bs{bs}_harvest7"] = bs / dt
            print(f"perf {nm} bs={bs} harvest(stop 7): {dt * 1e3:.2f} ms  {bs / dt:.0f} img/s", flush=True)
        del model
    RES["perf"] = r


def l14():
    arch = ARCHS["clip-vit-l14-336"]
    want = [f"blocks.{l}.attn.{h}" for l in (0, 23) for h in ("hook_attn_scores", "hook_pattern")]
    compare("clip-vit-l14-336", 1, torch.float32, "l14_f32_bs1", 1e-4, names_filter=want)
    model, _ = build("clip-vit-l14-336", torch.bfloat16)
    x = torch.randn(32, 3, 336, 336, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        flt = lambda n: n.endswith("attn.hook_pattern")  # noqa: E731
        for _ in range(2):
            model.run_with_cache(x, names_filter=flt)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            model.run_with_cache(x, names_filter=flt)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
    print(f"perf l14 bf16 bs=32 pattern hooks: {dt * 1e3:.1f} ms {32 / dt:.0f} img/s", flush=True)
    RES["l14_perf"] = 32 / dt


def sae():
    from oracle import sae_oracle as O
    from vit_prisma_amd.sae.native_sae import NativeSAE
    from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state
    r = {}
    for (d_in, d_sae, k, Ntok, nsteps) in ((64, 512, 8, 256, 3), (96, 1024, 16, 300, 2), (768, 24576, 32, 4096, 2)):
        tag = f"{d_in}x{d_sae}k{k}N{Ntok}"
        sd = synth_sae_state(d_in, d_sae, 0)
        P = {kk: v.copy() for kk, v in sd.items()}
        opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
        stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
        T = {kk: torch.from_numpy(v.copy()).to(dev) for kk, v in sd.items()}
        eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, Ntok)
        for t in range(nsteps):
            x = synth_sae_batch(Ntok, d_in, seed=t)
            # oracle pieces
            Pc = {kk: v.copy() for kk, v in P.items()}
            O.renorm_decoder(Pc)
            fw = O.sae_forward(Pc, x, k)
            gr = O.sae_backward(Pc, x, fw)
            out = O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1)
            xg = torch.from_numpy(x).to(dev)
            eng.renorm_decoder()
            eng.step(xg, want_out=True)
            eng.grad_sqnorm()
            torch.cuda.synchronize()
            sc = eng.scalars.cpu().numpy()
            idx = eng.topk_idx[:Ntok].cpu().numpy()
            same_set = float(np.mean([set(a.tolist()) == set(b.tolist()) for a, b in zip(idx, fw["idx"])]))
            e = {
                "loss_rel": abs(sc[0] - out["loss"]) / out["loss"], "l0": float(sc[2]), "l0_ref": out["l0"],
                "gnorm_rel": abs(np.sqrt(sc[3]) - out["grad_norm"]) / out["grad_norm"], "topk_set_match": same_set,
                "sae_out": rel(eng.sae_out[:Ntok].cpu().numpy(), fw["sae_out"]),
                "gW_enc": rel(eng.grad_W_enc().cpu().numpy(), gr["W_enc"]), "gW_dec": rel(eng.g["W_dec"].cpu().numpy(), gr["W_dec"]),
                "gb_enc": rel(eng.g["b_enc"].cpu().numpy(), gr["b_enc"]), "gb_dec": rel(eng.g["b_dec"].cpu().numpy(), gr["b_dec"]),
            }
            eng.apply(1e-3, 1.0)
            torch.cuda.synchronize()
            for kk in P:
                e["p_" + kk] = rel(eng.params[kk].cpu().numpy(), P[kk])
                e["m_" + kk] = rel(eng.m[kk].cpu().numpy(), opt["m"][kk])
            e["act_freq_eq"] = bool(np.array_equal(eng.act_freq_scores.cpu().numpy(), stats["act_freq_scores"]))
            e["n_since_eq"] = bool(np.array_equal(eng.n_fwd_since_fired.cpu().numpy(), stats["n_fwd_since_fired"]))
            r[f"{tag}_s{t}"] = e
            print(f"[sae {tag} step {t}]", {kk: (f"{v:.2e}" if isinstance(v, float) else v) for kk, v in e.items()}, flush=True)
        if d_in == 768:
            xs = [torch.from_numpy(synth_sae_batch(Ntok, d_in, seed=100 + i)).to(dev) for i in range(4)]
            for i in range(3):
                eng.train_step(xs[i % 4], 1e-3)
            torch.cuda.synchronize()
            N.prof_reset(); N.prof_enable(True)
            t0 = time.time()
            n = 20
            for i in range(n):
                eng.train_step(xs[i % 4], 1e-3)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n
            N.prof_enable(False)
            r["perf_tokens_per_s"] = Ntok / dt
            print(f"sae perf: {dt * 1e3:.3f} ms/step  {Ntok / dt:.0f} tokens/s", flush=True)
            for kind in ("sae_encode_topk", "sae_backward", "sae_apply", "gemm"):
                pr = N.prof_read(kind)
                print("   ", kind, {kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in pr.items()},
                      "avg_us", round(pr["ms"] * 1e3 / max(pr["launches"], 1), 1), flush=True)
    RES["sae"] = r


if __name__ == "__main__":
    sections = sys.argv[1:] or ["unit", "tiny", "b32", "bf16", "perf"]
    print(torch.cuda.get_device_name(0), flush=True)
    for s in sections:
        print(f"===== {s} =====", flush=True)
        try:
            globals()[s]()
        except Exception:
            traceback.print_exc()
            RES[s + "_exception"] = traceback.format_exc()
        with open(os.path.join(OUT, "diag.json"), "w") as f:
            json.dump(RES, f, indent=1, default=str)
