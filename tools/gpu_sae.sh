# round-2 GPU call 3: the rebuilt SAE step -- parity tests, step-only bench, kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_native_sae_gpu.py -m gpu -x -q > $O/tests_sae.log 2>&1; echo "tests rc=$?" >> $O/tests_sae.log
timeout 300 python -c "
import torch, json
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
print(json.dumps(sae_bench_leg(torch.device('cuda', 0))))" > $O/bench_sae.json 2> $O/bench_sae.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sae -o sae -- python $R/tools/prof_sae.py > $O/prof_sae.out 2> $O/prof_sae.err
cp $O/prof_sae/sae_kernel_stats.csv $O/sae_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_sae
cd $R; tail -25 $O/tests_sae.log; cat $O/bench_sae.json; tail -3 $O/bench_sae.err; head -30 $O/sae_kernel_stats.csv | cut -c1-150
