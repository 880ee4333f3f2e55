R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -k "trainer or checkpoint or relu_variants or substitution" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -6 $O/t.log | cut -c1-300
timeout 600 python - > $O/sae_leg.json 2> $O/sae_leg.err <<'PY'
import json, torch
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
for _ in range(2):
    r = sae_bench_leg(torch.device("cuda:0"), dist=None)
    print(json.dumps({"tokens_per_s": r["value"], "ms_per_step": r["ms_per_step"], "kernels": r["kernels"]}), flush=True)
PY
cat $O/sae_leg.json; tail -3 $O/sae_leg.err
