# first GPU pass of round 3: what round 2 wrote but could not run on hardware any more
#   1. the feature-parallel SAE step at the bench-sized shard (filtered encoder, d_in = 768)
#   2. bench.py under torchrun with two ranks sharing the GPU over gloo, SAE step-only leg data-parallel vs feature-parallel
#      (completion + loss agreement; two ranks on one GPU say nothing about speed)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; rm -rf $O; mkdir -p $O
cd $R
PV_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -k feature_parallel > $O/tp_tests.log 2>&1; echo "rc=$?" >> $O/tp_tests.log
tail -5 $O/tp_tests.log
for mode in data feature; do
  BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-l14 --sae-parallel $mode > $O/gloo2_$mode.json 2> $O/gloo2_$mode.err; echo "rc=$?" >> $O/gloo2_$mode.err
  python - <<PY
import json
try:
    d = json.load(open('$O/gloo2_$mode.json'))
    s = d.get('sae', {})
    print('$mode', s.get('ms_per_step'), s.get('final_loss'), s.get('config', {}).get('parallelism'), s.get('error'))
except Exception as e:
    print('$mode', 'no line:', e)
PY
  tail -2 $O/gloo2_$mode.err
done
