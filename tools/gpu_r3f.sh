# round 3, sixth GPU pass: dense step after the scratch fix (tests + the ReLU bench leg), TP phase times after the csr_post fix
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -k "relu_l1" > $O/dense_tests.log 2>&1; echo "rc=$?" >> $O/dense_tests.log
tail -6 $O/dense_tests.log
timeout 600 python - > $O/relu_leg.json 2> $O/relu_leg.err <<'PY'
import json, torch
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
print(json.dumps(sae_bench_leg(torch.device("cuda:0"), dist=None, steps=8, warmup=2, activation="relu")))
PY
cat $O/relu_leg.json; tail -3 $O/relu_leg.err
timeout 600 python tools/tp_shard_times.py > $O/tp_shard_times.json 2> $O/tp_shard_times.err; cat $O/tp_shard_times.json | tr -d '\n' | cut -c1-1500; echo
