# round 4, fifth GPU pass: top-k + ghost gradients natively, the sparse attempt's hysteresis, regression of what the pass touched
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 500 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 \
  -k "topk_ghost or relu_step or relu_variants or relu_l1_dense_step_vs_oracle" > $O/t_sae.log 2>&1; echo "rc=$? $(( $(date +%s) - T0 ))s" >> $O/t_sae.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/t_sae.log | head -12
grep -E "^E " $O/t_sae.log | head -20
timeout 300 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -p no:cacheprovider --timeout=200 -k "long_sequence or l14_bs128" > $O/t_attn.log 2>&1; echo "rc=$?" >> $O/t_attn.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/t_attn.log | head
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
dev = torch.device("cuda:0")
for name, kw in (("published_l0", dict(steps=16, warmup=2, relu_target_l0=0.035 * 24576)), ("steady", dict(steps=20, warmup=10)),
                 ("from_init", dict(steps=10, warmup=0)), ("l0_64", dict(steps=10, warmup=2, relu_target_l0=64.0))):
    r = sae_bench_leg(dev, activation="relu", **kw)
    print("relu", name, r["ms_per_step"], "l0", round(r["l0"], 1), "sparse/dense", r["sparse_steps"], r["dense_steps"], flush=True)
PY
echo "total $(( $(date +%s) - T0 ))s"
