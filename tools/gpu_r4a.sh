# round 4, first GPU pass: the whole -m gpu suite with the new parity cases (bench-size dense steps, B/32 intra-block hook
# fixtures, RCCL world-1 steps, store coalescing), then a bench line without the CPU legs (store shapes, baseline numbers)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
grep -E "passed|failed|error|worst error" $O/gpu_tests.log | tail -15
grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head -30
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open('$O/bench.json'))
    s = d.get('sae', {})
    print('vit', d.get('value'), d.get('ms_per_step'), d.get('roofline', {}).get('frac'))
    print('sae', s.get('ms_per_step'), 'e2e', s.get('end_to_end', {}).get('value'), 'refshape', s.get('end_to_end', {}).get('reference_store_shape', {}).get('value'))
    print('relu', s.get('relu_l1', {}).get('ms_per_step'), 'l14', d.get('l14_336_pattern', {}).get('value'))
except Exception as e:
    print('no bench line:', e)
PY
tail -3 $O/bench.err
