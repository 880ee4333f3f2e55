# sparse-gradient / deferred-renorm ReLU step + single-round filter epilogue: every SAE GPU test, then the bench line without the ViT-only legs,
# kernel stats of the top-k and ReLU steps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 600 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 -x > $O/t_sae.log 2>&1; echo "sae tests rc=$? $(( $(date +%s) - T0 ))s"; tail -4 $O/t_sae.log
timeout 500 python bench.py --no-l14 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - T0 ))s"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sae -o sae -- python $R/tools/prof_sae.py > $O/prof_sae.out 2> $O/prof_sae.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_relu -o relu -- python $R/tools/prof_relu.py > $O/prof_relu.out 2> $O/prof_relu.err
for n in sae relu; do cp $O/prof_$n/${n}_kernel_stats.csv $O/${n}_kernel_stats.csv 2>/dev/null; done
rm -rf $O/prof_sae $O/prof_relu
python - <<PY
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{"metric"')][0])
print('b32', d['value'], d['ms_per_step'])
s=d['sae']; print('sae', s['value'], s['ms_per_step'], s['roofline']['frac'], s.get('kernels'), 'e2e', s['end_to_end']['value'], 'ref-store', s['end_to_end'].get('reference_store_shape',{}).get('value'))
r=s['relu_l1']; print('relu', r['value'], r['ms_per_step'], r.get('sparse_steps'), r.get('dense_steps'), r.get('kernels'), {k: (r[k].get('ms_per_step'), r[k].get('sparse_steps'), r[k].get('dense_steps')) for k in ('from_init','published_l0','l0_64') if k in r})
print('variants', {k: (v.get('value'), v.get('ms_per_step')) for k, v in s.get('variants', {}).items()})
PY
head -8 $O/sae_kernel_stats.csv | cut -c1-120
echo "total $(( $(date +%s) - T0 ))s"
