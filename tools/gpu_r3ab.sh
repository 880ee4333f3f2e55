R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python - > $O/e2e.jsonl 2> $O/err.log <<'PY'
import json, torch
from vit_prisma_amd.sae.bench_leg import sae_end_to_end_leg
for bs, nb in ((256, 8), (512, 4), (256, 8), (512, 4), (1024, 2)):
    r = sae_end_to_end_leg(torch.device("cuda:0"), dist=None, store_bs=bs, n_buf=nb)
    print(json.dumps({"store_bs": bs, "n_buf": nb, "tokens_per_s": r["value"], "ms_per_step": r["ms_per_step"], "ratio": r.get("harvested_tokens_per_trained_token")}), flush=True)
    torch.cuda.empty_cache()
PY
cat $O/e2e.jsonl; tail -3 $O/err.log
