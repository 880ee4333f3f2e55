PV_VIT_SPLIT=2 python -m pytest tests/test_native_vit_gpu.py -m gpu -x -q -k "full_size or fp32_b32_bs16 or cache_lifetime" 2>&1 | tail -3
python - <<'PY'
import os, torch, numpy as np, subprocess, sys, json
# value parity: split vs unsplit at bs=64 (two processes because the switch is read once)
code = '''
import torch, sys, os
sys.path.insert(0, os.getcwd())
from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.synth import ARCHS, synth_vit_state, synth_images
arch = ARCHS["clip-vit-b32"]
m = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
m = m.to(torch.bfloat16).cuda().eval().use_native(True)
x = torch.from_numpy(synth_images(arch, 64, 3)).cuda().bfloat16()
with torch.no_grad():
    out, cache = m.run_with_cache(x)
torch.cuda.synchronize()
import hashlib
h = hashlib.sha256()
for k in cache.keys():
    h.update(cache[k].float().cpu().numpy().tobytes())
h.update(out.float().cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
'''
d = []
for env in ({}, {"PV_VIT_SPLIT": "2"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    d.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")])
    if r.returncode: print(r.stderr[-800:])
print("bit-identical split vs unsplit at bs=64:", d[0] == d[1], d)
PY
for r in 1 2; do
for v in split one; do
  e=PV_VIT_SPLIT=2; [ $v = one ] && e=A=1
  env $e python bench.py --no-sae --no-l14 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['avg_launch_us'])"
done; done
