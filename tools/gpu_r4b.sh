# round 4, second GPU pass: the new parity cases only (every stage under its own timeout), then A/B of the four-wave GEMM
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 600 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 --durations=15 \
  -k "relu_step or rccl or store_taps or store_harvest or outside_edit" > $O/t_sae_new.log 2>&1; echo "rc=$? $(( $(date +%s) - T0 ))s" >> $O/t_sae_new.log
tail -25 $O/t_sae_new.log | grep -E "passed|failed|FAILED|ERROR|rc=|Error" | head -20
timeout 600 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 --durations=10 -s \
  -k "mutating_hooks_on_b32 or hooks_inside_the_attention or ragged_shapes or do_not_depend" > $O/t_vit_new.log 2>&1; echo "rc=$? $(( $(date +%s) - T0 ))s" >> $O/t_vit_new.log
grep -E "passed|failed|FAILED|ERROR|rc=|worst error" $O/t_vit_new.log | head -20
timeout 900 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=400 --durations=15 \
  -k "relu_l1_dense_step or transcoder_steps or gated_step" > $O/t_sae_big.log 2>&1; echo "rc=$? $(( $(date +%s) - T0 ))s" >> $O/t_sae_big.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/t_sae_big.log | head -20
for tile in -1 9; do
  timeout 240 python bench.py --no-cpu-baseline --no-sae --no-l14 --allow-overrides --tune gemm_tile=$tile > $O/bench_tile$tile.json 2> $O/bench_tile$tile.err
  python - <<PY
import json
try:
    d = json.load(open('$O/bench_tile$tile.json'))
    print('tile $tile: img/s', d.get('value'), 'ms', d.get('ms_per_step'), 'gemm frac', d.get('roofline', {}).get('frac'), 'avg us', d.get('roofline', {}).get('avg_launch_us'))
except Exception as e:
    print('tile $tile: no line', e)
PY
done
for tile in 5 9; do PV_TILE=$tile timeout 120 python tools/gemm_ab_tile.py > $O/gemm_tile$tile.txt 2>&1; cat $O/gemm_tile$tile.txt | tail -6; done
echo "total $(( $(date +%s) - T0 ))s"
