"""One tiny native SAE train step on the GPU, checked against the oracle (called by
__graft_entry__.smoke())."""
from __future__ import annotations

import numpy as np
import torch


def sae_smoke(dev: torch.device) -> None:
    from oracle import sae_oracle as O
    from ..synth import synth_sae_batch, synth_sae_state
    from .native_sae import NativeSAE
    d_in, d_sae, k, n = 64, 512, 8, 256
    sd = synth_sae_state(d_in, d_sae, 0)
    P = {kk: v.copy() for kk, v in sd.items()}
    opt = {"m": {kk: np.zeros_like(v) for kk, v in P.items()}, "v": {kk: np.zeros_like(v) for kk, v in P.items()}}
    stats = {"n_fwd_since_fired": np.zeros(d_sae, np.float32), "act_freq_scores": np.zeros(d_sae, np.float32)}
    T = {kk: torch.from_numpy(v.copy()).to(dev) for kk, v in sd.items()}
    eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
    for t in range(2):
        x = synth_sae_batch(n, d_in, seed=t)
        ref = O.train_step(P, opt, stats, x, k, lr=1e-3, step=t + 1)
        eng.train_step(torch.from_numpy(x).to(dev), 1e-3, 1.0)
        torch.cuda.synchronize()
        loss = float(eng.scalars[0])
        assert abs(loss - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (loss, ref["loss"])
        for kk in P:
            got = eng.params[kk].cpu().numpy()
            err = np.linalg.norm(got - P[kk]) / np.linalg.norm(P[kk])
            assert err < 1e-4, (kk, err)
