"""Per-workgroup timeline of the filter GEMM (sae_enc_gemm_kernel<1>: 1536 tiles of 256 x 256, K = 768, one 512-thread workgroup per CU at a
time) from a debug build (-DPV_ENC_TRACE: PV_NATIVE_LIB=tools/variants/libpvnative_enctrace.so): start, K loop done, end of the hit-list
epilogue on the 100 MHz wall clock.  How long are the three phases of a tile, how long are the gaps between a CU's tiles?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vit_prisma_amd.synth import synth_sae_state, synth_sae_batch
from vit_prisma_amd.sae.native_sae import NativeSAE

d_in, d_sae, k, n = 768, 24576, 32, 4096
T = {kk: torch.from_numpy(v.copy()).cuda() for kk, v in synth_sae_state(d_in, d_sae, 0).items()}
eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
off = eng.lib.pv_debug_sae_ws_offset(eng._plan, b"hidden") + 4 * 8192 * 8
nt = 16 * 96
for t in range(3):
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
    eng.workspace[off:off + nt * 32].zero_()
    eng.step(x, renorm_decoder=True, sparse_grads=True, fused_sqnorm=True)
    eng.apply(1e-3, 1.0)
    torch.cuda.synchronize()
tr = eng.workspace[off:off + nt * 32].view(torch.int64).view(nt, 4).cpu().numpy()
t0 = tr[:, 0].min()
start, end, kdone, hw = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0, tr[:, 3]
print(f"kernel span {end.max():.1f} us over {nt} tiles; tile lifetime mean {np.mean(end - start):.1f} median {np.median(end - start):.1f} p99 {np.percentile(end - start, 99):.1f} us")
print(f"  start -> K loop done: mean {np.mean(kdone - start):.1f} median {np.median(kdone - start):.1f} p99 {np.percentile(kdone - start, 99):.1f} us")
print(f"  epilogue (K loop done -> end): mean {np.mean(end - kdone):.1f} median {np.median(end - kdone):.1f} p99 {np.percentile(end - kdone, 99):.1f} us")
# per CU: group by hardware id, order by start, gaps between one tile's end and the next one's start
cu = hw
gaps, per_cu = [], []
for c in np.unique(cu):
    m = np.where(cu == c)[0]
    o = m[np.argsort(start[m])]
    per_cu.append(len(o))
    for a, b in zip(o[:-1], o[1:]):
        gaps.append(start[b] - end[a])
gaps = np.array(gaps)
print(f"  distinct hw ids {len(per_cu)}, tiles per id: min {min(per_cu)} max {max(per_cu)}; gap between a CU's tiles: mean {gaps.mean():.2f} median {np.median(gaps):.2f} p99 {np.percentile(gaps, 99):.2f} us")
for t in range(0, int(end.max()) + 1, 20):
    print(f"  t = {t:4d} us: {int(((start <= t) & (end > t)).sum()):4d} tiles alive, {int(((kdone <= t) & (end > t)).sum()):4d} in their epilogue")
