import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import sae_oracle as O
from vit_prisma_amd.sae.native_sae import NativeSAE
from vit_prisma_amd.synth import synth_sae_batch
from test_native_sae_gpu import fresh_transcoder, rel_fro
d_in, d_sae, k, n = 768, 8192, 32, 1024
P, opt, stats, T = fresh_transcoder(d_in, d_sae, True)
eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n, b_dec_out=T["b_dec_out"], W_skip=T["W_skip"])
for t in range(2):
    x, y = synth_sae_batch(n, d_in, seed=10 + t), synth_sae_batch(n, d_in, seed=50 + t)
    Pc = {kk: v.copy() for kk, v in P.items()}
    O.renorm_decoder(Pc)
    fw = O.sae_forward(Pc, x, k, target=y)
    gr = O.sae_backward(Pc, x, fw)
    ref = O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1, target=y)
    xg, yg = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    eng.step(xg, want_out=True, renorm_decoder=True, target=yg)
    eng.grad_sqnorm(from_step=True)
    torch.cuda.synchronize()
    out = eng.sae_out[:n].cpu().numpy()
    per_tok = np.linalg.norm(out - fw["sae_out"], axis=1) / np.linalg.norm(fw["sae_out"], axis=1)
    print(t, "sae_out rel_fro", rel_fro(out, fw["sae_out"]), "per-token max", per_tok.max(), "median", np.median(per_tok), "n>1e-4", (per_tok > 1e-4).sum())
    skip_ref = x @ Pc["W_skip"].T
    skip_ref64 = x.astype(np.float64) @ Pc["W_skip"].T.astype(np.float64)
    sk = eng._tc_scratch[: n * d_in * 4].view(torch.float32).view(n, d_in).cpu().numpy()
    print("   skip term: kernel vs numpy fp32", rel_fro(sk, skip_ref), " kernel vs fp64", rel_fro(sk, skip_ref64), " numpy fp32 vs fp64", rel_fro(skip_ref, skip_ref64),
          " |x| max", np.abs(x).max(), "std(x)", x.std())
    idx_same = np.array_equal(np.sort(eng.topk_idx[:n].cpu().numpy(), axis=1), np.sort(fw["idx"], axis=1))
    print("   idx sets equal", idx_same, "loss", float(eng.scalars[0]), ref["loss"])
    for name in P:
        gk = eng.grad_W_enc().cpu().numpy() if name == "W_enc" else eng.g[name].cpu().numpy()
        print("   grad", name, rel_fro(gk, gr[name]))
    eng.apply(1e-3, 1.0)
    torch.cuda.synchronize()
    for name in P:
        print("   param", name, rel_fro(eng.params[name].cpu().numpy(), P[name]))
