R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3af; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -x -k "transcoder" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -12 $O/t.log | cut -c1-300
timeout 300 python - <<'PY'
import json, torch
from vit_prisma_amd.sae.bench_leg import sae_variants_leg
print(json.dumps(sae_variants_leg(torch.device("cuda:0"))))
PY
