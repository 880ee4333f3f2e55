# every SAE GPU test on the one-round filter epilogue + sparse-gradient ReLU step, then A/B of the epilogue (kernel stats) and the bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 700 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 > $O/t_sae.log 2>&1; echo "sae tests rc=$? $(( $(date +%s) - T0 ))s"; tail -6 $O/t_sae.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  PV_TUNE=enc_rounds=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sae$v -o sae -- python $R/tools/prof_sae.py > $O/prof_sae$v.out 2> $O/prof_sae$v.err
  PV_TUNE=enc_rounds=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_relu$v -o relu -- python $R/tools/prof_relu.py > $O/prof_relu$v.out 2> $O/prof_relu$v.err
  cp $O/prof_sae$v/sae_kernel_stats.csv $O/sae_rounds${v}_kernel_stats.csv; cp $O/prof_relu$v/relu_kernel_stats.csv $O/relu_rounds${v}_kernel_stats.csv
  rm -rf $O/prof_sae$v $O/prof_relu$v
  echo "enc_rounds=$v"; grep -h 'sae_enc_gemm_kernel<1' $O/sae_rounds${v}_kernel_stats.csv $O/relu_rounds${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
  grep -o "'ms_per_step': [0-9.]*" $O/prof_sae$v.out $O/prof_relu$v.out | head -4
done
cd $R
timeout 500 python bench.py --no-l14 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - T0 ))s"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{"metric"')][0])
print('b32', d['value'], d['ms_per_step'])
s=d['sae']; print('sae', s['value'], s['ms_per_step'], s['roofline']['frac'], s.get('kernels'), 'e2e', s['end_to_end']['value'], 'ref-store', s['end_to_end'].get('reference_store_shape',{}).get('value'))
r=s['relu_l1']; print('relu', r['value'], r['ms_per_step'], r.get('sparse_steps'), r.get('dense_steps'), r.get('kernels'), {k: (r[k].get('ms_per_step'), r[k].get('sparse_steps'), r[k].get('dense_steps')) for k in ('from_init','published_l0','l0_64') if k in r})
PY
echo "total $(( $(date +%s) - T0 ))s"
