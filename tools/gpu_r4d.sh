# round 4, fourth GPU pass: the strip-in-LDS long-sequence attention kernel -- parity, then time against the two-pass kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 400 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -p no:cacheprovider --timeout=200 \
  -k "long_sequence or l14_bs128 or hooks_inside_the_attention" > $O/t_attn.log 2>&1; echo "rc=$? $(( $(date +%s) - T0 ))s" >> $O/t_attn.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/t_attn.log | head -20
grep -E "^E " $O/t_attn.log | head -20
for lean in 0 1; do
  PV_ATTN_LEAN=$lean timeout 200 python tools/attn_l14_time.py > $O/attn_time_lean$lean.json 2> $O/attn_time_lean$lean.err; echo "lean=$lean: $(cat $O/attn_time_lean$lean.json)"; tail -2 $O/attn_time_lean$lean.err
done
timeout 300 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=200 -k "goes_dense or relu_step_sparse" > $O/t_relu.log 2>&1; echo "rc=$?" >> $O/t_relu.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/t_relu.log | head
python - <<'PY'
import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
dev = torch.device("cuda:0")
for name, kw in (("published_l0", dict(steps=5, warmup=2, relu_target_l0=0.035 * 24576)), ("steady", dict(steps=20, warmup=10)),
                 ("from_init", dict(steps=10, warmup=0))):
    r = sae_bench_leg(dev, activation="relu", **kw)
    print("relu", name, r["ms_per_step"], "l0", round(r["l0"], 1), "sparse/dense", r["sparse_steps"], r["dense_steps"], flush=True)
PY
echo "total $(( $(date +%s) - T0 ))s"
