"""Per-rank GPU time of the feature-parallel SAE step at world 1 / 2 / 4 / 8, measured on ONE GPU: all ranks of a world run in
lockstep (feature_parallel.simulate_step), each rank's four phases are bracketed by events on the stream, the collectives are
done by hand and NOT timed.  Output: JSON {world: {phase: mean us of the slowest rank, ..., "step_compute_us": sum}} -- the
compute column of DESIGN.md section 5's table (the collectives' column is a model until the driver's SCALE run)."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vit_prisma_amd.sae.feature_parallel import FeatureParallelSAE, simulate_step     # noqa: E402
from vit_prisma_amd.sae.native_sae import NativeSAE                                     # noqa: E402
from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state                       # noqa: E402

d_in, d_sae, k, n = 768, 24576, 32, 4096
steps, warm = 12, 4
dev = torch.device("cuda:0")
sd = synth_sae_state(d_in, d_sae, 0)
xs = [torch.from_numpy(synth_sae_batch(n, d_in, seed=i)).to(dev) for i in range(4)]
out = {"shape": f"{d_in} -> {d_sae}, k = {k}, {n} tokens per step (global)", "steps": steps}

# the single-process fused step, for reference
T = {kk: torch.from_numpy(v.copy()).to(dev) for kk, v in sd.items()}
eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
for i in range(warm):
    eng.train_step(xs[i % 4], 1e-3, 1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps):
    eng.train_step(xs[i % 4], 1e-3, 1.0)
e1.record()
torch.cuda.synchronize()
out["single_process_step_us"] = round(e0.elapsed_time(e1) * 1e3 / steps, 1)
del eng, T

WORLDS = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 4, 8)
for W in WORLDS:
    T = {kk: torch.from_numpy(v.copy()).to(dev) for kk, v in sd.items()}
    ranks = [FeatureParallelSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k,
                                lambda We, Wd, be, bd: NativeSAE(We, Wd, be, bd, k, True, n), rank=r, world=W) for r in range(W)]
    ev = {}

    def on_phase(tag, r):
        name, edge = tag.split(":")
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.setdefault((name, r), []).append(e)

    for i in range(warm):
        simulate_step(ranks, xs[i % 4], 1e-3, 1.0)
    torch.cuda.synchronize()
    for i in range(steps):
        simulate_step(ranks, xs[i % 4], 1e-3, 1.0, on_phase=on_phase)
    torch.cuda.synchronize()
    res = {}
    for name in ("encode", "partial", "finish", "apply"):
        per_rank = []
        for r in range(W):
            es = ev[(name, r)]
            per_rank.append(sum(es[2 * j].elapsed_time(es[2 * j + 1]) for j in range(steps)) * 1e3 / steps)
        res[name] = round(max(per_rank), 1)
    res["step_compute_us"] = round(sum(res[p] for p in ("encode", "partial", "finish", "apply")), 1)
    res["filtered_encoder"] = bool(ranks[0].engine.filtered_encoder)
    res["loss"] = float(ranks[0].loss)
    out[f"world_{W}"] = res
    del ranks, T
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
