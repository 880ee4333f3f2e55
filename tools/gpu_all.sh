# full GPU regression: every -m gpu test, SAE step bench + kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c9; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -m gpu -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?" >> $O/tests_gpu.log
timeout 300 python -c "
import torch, json
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
print(json.dumps(sae_bench_leg(torch.device('cuda', 0))))" > $O/bench_sae.json 2> $O/bench_sae.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sae -o sae -- python $R/tools/prof_sae.py > $O/prof_sae.out 2> $O/prof_sae.err
cp $O/prof_sae/sae_kernel_stats.csv $O/sae_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_sae
cd $R; tail -30 $O/tests_gpu.log; cat $O/bench_sae.json; tail -3 $O/bench_sae.err
