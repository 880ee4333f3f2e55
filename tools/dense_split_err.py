"""How far are the dense ReLU + L1 step's tensors from the same step carried in float64 -- on the exact fp32 matrix path (tuning key
dense_fp32 = 1) and on the split-fp16 one (round 6)?  768 -> 8192, 1024 tokens: sae_out, dH (left in the workspace), every gradient
tensor; max-abs error relative to the tensor's largest entry and rel-Frobenius.  The oracle's own fp32 run stands beside both."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import sae_oracle as O
from vit_prisma_amd import _native as N
from vit_prisma_amd.sae.native_sae import NativeSAE
from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state

d_in, d_sae, n, l1c = 768, int(os.environ.get("D_SAE", "8192")), 1024, 3e-3
sd = synth_sae_state(d_in, d_sae, 0)
x = synth_sae_batch(n, d_in, seed=0)
P64 = {k: v.astype(np.float64) for k, v in sd.items()}
O.renorm_decoder(P64)
fw64 = O.sae_forward(P64, x.astype(np.float64), None, l1_coefficient=l1c)
gr64 = O.sae_backward(P64, x.astype(np.float64), fw64, l1_coefficient=l1c)
P32 = {k: v.copy() for k, v in sd.items()}
O.renorm_decoder(P32)
fw32 = O.sae_forward(P32, x, None, l1_coefficient=l1c)
gr32 = O.sae_backward(P32, x, fw32, l1_coefficient=l1c, gate=fw64["feature_acts"] > 0)
ref = {"sae_out": fw64["sae_out"], "hidden_pre(f>0)": np.where(fw64["feature_acts"] > 0, fw64["hidden_pre"], 0.0), **{"g" + k: v for k, v in gr64.items()}}


def err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return {"max_abs_over_max": float(np.abs(a - b).max() / np.abs(b).max()), "rel_fro": float(np.linalg.norm(a - b) / np.linalg.norm(b))}


out = {"oracle_fp32": {"sae_out": err(fw32["sae_out"], ref["sae_out"]), **{"g" + k: err(v, gr64[k]) for k, v in gr32.items()}}}
for fp32 in (1, 0):
    N.set_tuning("reset")
    N.set_tuning("dense_fp32", fp32)
    T = {k: torch.from_numpy(v.copy()).cuda() for k, v in sd.items()}
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], 1, True, n)
    eng.dense_step(torch.from_numpy(x).cuda(), l1c, want_out=True)
    torch.cuda.synchronize()
    got = {"sae_out": eng.sae_out[:n].cpu().numpy(), "gW_enc": eng.grad_W_enc().cpu().numpy(),
           **{"g" + k: eng.g[k].cpu().numpy() for k in ("W_dec", "b_enc", "b_dec")}}
    off = eng.lib.pv_debug_sae_ws_offset(eng._plan, b"hidden")
    dH = eng.workspace[off:off + n * d_sae * 4].view(torch.float32).view(n, d_sae).cpu().numpy()
    flips = int(((dH != 0) != (fw64["feature_acts"] > 0)).sum())
    worst = float(np.abs(fw64["hidden_pre"][(dH != 0) != (fw64["feature_acts"] > 0)]).max() / np.abs(fw64["hidden_pre"]).max()) if flips else 0.0
    tag = "kernel_fp32" if fp32 else "kernel_split_fp16"
    out[tag] = {k: err(v, ref[k]) for k, v in got.items()}
    out[tag]["relu_gates_other_than_float64"] = {"count": flips, "largest_abs_hidden_pre_over_max": worst}
N.set_tuning("reset")
print(json.dumps(out, indent=1))
