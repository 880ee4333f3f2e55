# radix-select threshold in the select kernel: SAE GPU tests, kernel stats, bench (SAE legs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 700 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 > $O/t_sae.log 2>&1; echo "sae tests rc=$? $(( $(date +%s) - T0 ))s"; tail -4 $O/t_sae.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python $R/tools/prof_sae.py > $O/prof_sae.out 2> $O/prof_sae.err
cp $O/p/s_kernel_stats.csv $O/sae_kernel_stats.csv; rm -rf $O/p
head -8 $O/sae_kernel_stats.csv | awk -F'",' '{print substr($1,1,70), $2}' | cut -d, -f1-3
cd $R
timeout 500 python bench.py --no-l14 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - T0 ))s"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{"metric"')][0])
print('b32', d['value'], d['ms_per_step'])
s=d['sae']; print('sae', s['value'], s['ms_per_step'], s['roofline']['frac'], s.get('kernels'), 'e2e', s['end_to_end']['value'], 'ref-store', s['end_to_end'].get('reference_store_shape',{}).get('value'))
r=s['relu_l1']; print('relu', r['value'], r['ms_per_step'], r.get('sparse_steps'), r.get('dense_steps'))
print('variants', {k: (v.get('value'), v.get('ms_per_step')) for k, v in s.get('variants', {}).items()})
PY
echo "total $(( $(date +%s) - T0 ))s"
