# round 3, seventh GPU pass: the store's harvest prefetch (determinism test, end-to-end leg with and without it)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -k "prefetch or trainer_native_path" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log
timeout 900 python - > $O/e2e.json 2> $O/e2e.err <<'PY'
import json, torch
from vit_prisma_amd.sae.bench_leg import sae_end_to_end_leg
dev = torch.device("cuda:0")
for ov in (False, True, False, True):
    r = sae_end_to_end_leg(dev, overlap_harvest=ov)
    print(json.dumps({"overlap": ov, "tokens_per_s": r["value"], "ms_per_step": r["ms_per_step"], "ratio": r["harvested_tokens_per_trained_token"]}), flush=True)
    torch.cuda.empty_cache()
PY
cat $O/e2e.json; tail -3 $O/e2e.err
