R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ad; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -x -k "reproducible or native_step_vs_oracle or feature_parallel or simulated or config3" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -15 $O/t.log | cut -c1-300
timeout 300 python - <<'PY'
import json, torch
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
for _ in range(2):
    r = sae_bench_leg(torch.device("cuda:0"), dist=None)
    print(json.dumps({"tokens_per_s": r["value"], "ms_per_step": r["ms_per_step"], "kernels": r["kernels"]}), flush=True)
PY
