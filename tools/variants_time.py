"""Step time of the SAE variants on their fused HIP steps at the bench shape (768 -> 24576, 4096 tokens): top-k transcoder with the
skip connection, ReLU transcoder, gated SAE -- next to the plain top-k and ReLU + L1 steps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vit_prisma_amd.sae.native_sae import NativeSAE
from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state

d_in, d_sae, k, n = 768, 24576, 32, 4096
dev = torch.device("cuda:0")
sd = synth_sae_state(d_in, d_sae, 0)
xs = [torch.from_numpy(synth_sae_batch(n, d_in, seed=i)).to(dev) for i in range(4)]
ys = [torch.from_numpy(synth_sae_batch(n, d_in, seed=50 + i)).to(dev) for i in range(4)]
rs = np.random.RandomState(3)

def T():
    return {kk: torch.from_numpy(v.copy()).to(dev) for kk, v in sd.items()}

def vec(nn, s=0.05):
    return torch.from_numpy((rs.standard_normal(nn) * s).astype(np.float32)).to(dev)

def timed(step, reps=10, warm=3):
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        step(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

out = {"shape": f"{d_in} -> {d_sae}, {n} tokens"}
t = T(); e = NativeSAE(t["W_enc"], t["W_dec"], t["b_enc"], t["b_dec"], k, True, n); e.lazy_w_enc = True
out["topk_ms"] = timed(lambda i: e.train_step(xs[i % 4], 1e-3, 1.0)); del e, t
t = T(); W_skip = torch.from_numpy((rs.standard_normal((d_in, d_in)) / np.sqrt(d_in) * 0.3).astype(np.float32)).to(dev)
e = NativeSAE(t["W_enc"], t["W_dec"], t["b_enc"], t["b_dec"], k, True, n, b_dec_out=vec(d_in), W_skip=W_skip)
def tc_topk(i):
    e.step(xs[i % 4], renorm_decoder=True, target=ys[i % 4]); e.grad_sqnorm(from_step=True); e.apply(1e-3, 1.0)
out["transcoder_topk_skip_ms"] = timed(tc_topk)
def tc_relu(i):
    e.dense_step(xs[i % 4], 3e-3, target=ys[i % 4]); e.grad_sqnorm(); e.apply(1e-3, 1.0)
out["transcoder_relu_skip_ms"] = timed(tc_relu); del e, t
t = T(); e = NativeSAE(t["W_enc"], t["W_dec"], t["b_enc"], t["b_dec"], 1, True, n)
def relu(i):
    e.dense_step(xs[i % 4], 3e-3); e.grad_sqnorm(); e.apply(1e-3, 1.0)
out["relu_l1_ms"] = timed(relu); del e, t
t = T(); e = NativeSAE(t["W_enc"], t["W_dec"], t["b_enc"], t["b_dec"], 1, True, n, gated=dict(b_gate=vec(d_sae), r_mag=vec(d_sae, 0.2), b_mag=vec(d_sae)))
def gated(i):
    e.gated_step(xs[i % 4], 3e-3); e.grad_sqnorm(); e.apply(1e-3, 1.0)
out["gated_ms"] = timed(gated)
out["gated_loss"] = float(e.scalars[0])
for kk in list(out):
    if kk.endswith("_ms"):
        out[kk] = round(out[kk], 3)
        out[kk.replace("_ms", "_tokens_per_s")] = round(n / out[kk] * 1e3)
print(json.dumps(out, indent=1))
