# round 4, third GPU pass: the fixed parity cases + the embedding / final stage hooks, then the whole bench line (ReLU legs by regime)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 500 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 \
  -k "rccl or outside_edit or goes_dense or (gated_step and 24576)" > $O/t_sae_fix.log 2>&1; echo "rc=$? $(( $(date +%s) - T0 ))s" >> $O/t_sae_fix.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/t_sae_fix.log | head -12
timeout 500 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 \
  -k "embedding_and_final or boundary_hooks" > $O/t_vit_edge.log 2>&1; echo "rc=$? $(( $(date +%s) - T0 ))s" >> $O/t_vit_edge.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/t_vit_edge.log | head -12
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - T0 ))s"
python - <<PY
import json
try:
    d = json.load(open('$O/bench.json'))
    s = d.get('sae', {})
    print('vit', d.get('value'), d.get('ms_per_step'), d.get('roofline', {}).get('frac'))
    print('sae', s.get('ms_per_step'), 'e2e', s.get('end_to_end', {}).get('value'), 'refshape', s.get('end_to_end', {}).get('reference_store_shape', {}).get('value'))
    r = s.get('relu_l1', {})
    print('relu steady', r.get('ms_per_step'), 'l0', r.get('l0'), 'sparse/dense', r.get('sparse_steps'), r.get('dense_steps'), r.get('kernels'))
    for k in ('from_init', 'published_l0', 'l0_64'):
        q = r.get(k, {})
        print('relu', k, q.get('ms_per_step'), 'l0', q.get('l0'), 'sparse/dense', q.get('sparse_steps'), q.get('dense_steps'), q.get('error'))
    print('l14', d.get('l14_336_pattern', {}).get('value'), 'ok', d.get('ok'))
except Exception as e:
    print('no bench line:', e)
PY
tail -3 $O/bench.err
echo "total $(( $(date +%s) - T0 ))s"
