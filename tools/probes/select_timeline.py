"""Per-workgroup timeline of sae_select_kernel (debug build -DPV_SEL_TRACE: PV_NATIVE_LIB=tools/variants/libpvnative_seltrace.so): start,
candidate-pass-done and end stamps of every token's workgroup on the 100 MHz wall clock + its hardware id, one top-k step at the bench
shape.  Prints the workgroup lifetime distribution, how many are alive over time, and the spread over XCDs / CUs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vit_prisma_amd.synth import synth_sae_state, synth_sae_batch
from vit_prisma_amd.sae.native_sae import NativeSAE

d_in, d_sae, k, n = 768, 24576, 32, 4096
T = {kk: torch.from_numpy(v.copy()).cuda() for kk, v in synth_sae_state(d_in, d_sae, 0).items()}
eng = NativeSAE(T["W_enc"], T["W_dec"], T["b_enc"], T["b_dec"], k, True, n)
off = eng.lib.pv_debug_sae_ws_offset(eng._plan, b"hidden")
for t in range(3):
    x = torch.from_numpy(synth_sae_batch(n, d_in, seed=t)).cuda()
    eng.workspace[off:off + n * 32].zero_()
    eng.step(x, renorm_decoder=True, sparse_grads=True, fused_sqnorm=True)
    eng.apply(1e-3, 1.0)
    torch.cuda.synchronize()
tr = eng.workspace[off:off + n * 32].view(torch.int64).view(n, 4).cpu().numpy()
t0 = tr[:, 0].min()
start, end, mid, hw = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0, tr[:, 3]      # us
life = end - start
print(f"kernel span {end.max():.1f} us; workgroup lifetime: mean {life.mean():.1f} median {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f} us")
print(f"candidate pass (start -> stamp 2): mean {(mid - start).mean():.1f} median {np.median(mid - start):.1f} p99 {np.percentile(mid - start, 99):.1f} us; rest: mean {(end - mid).mean():.1f} p99 {np.percentile(end - mid, 99):.1f}")
for t in range(0, int(end.max()) + 1, 10):
    alive = int(((start <= t) & (end > t)).sum())
    started = int((start <= t).sum())
    print(f"  t = {t:4d} us: {alive:5d} workgroups alive, {started:5d} started")
xcc = (hw >> 32) & 0xf
print("workgroups per XCC (XCC_ID):", np.bincount(xcc.astype(np.int64), minlength=8).tolist())
order = np.argsort(start)
print("start time of workgroup (by block id) 0, 1023, 2047, 3071, 4095:", [round(float(start[i]), 1) for i in (0, 1023, 2047, 3071, 4095)])
print("sum of lifetimes / (span x 2048 slots) =", round(float(life.sum() / (end.max() * 2048)), 3))
