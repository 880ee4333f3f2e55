# round-2 third pass: GPU tests, the torchrun rehearsal of bench.py, SAE leg numbers
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/ -m gpu -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?" >> $O/tests_gpu.log
tail -6 $O/tests_gpu.log
timeout 400 python bench.py --no-cpu-baseline --no-l14 > $O/bench_sae.json 2> $O/bench_sae.err; tail -c 300 $O/bench_sae.err
python - <<PY
import json
d=json.load(open('$O/bench_sae.json'))
print('b32', d['value'], d['ms_per_step'], 'gemm', d['roofline']['achieved'])
print('sae', d['sae']['value'], d['sae']['ms_per_step'], d['sae']['kernels'], 'e2e', d['sae']['end_to_end']['value'])
PY
bash tools/gpu_dist_rehearsal.sh > $O/rehearsal.log 2>&1; tail -c 1500 $O/rehearsal.log
