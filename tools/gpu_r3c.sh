# round 3, third GPU pass: the feature-parallel tests that had not run (merge kernel, simulated worlds, world-2 gloo at the bench
# shape), a kernel trace of ONE rank's work at world 8 (where does the per-rank floor sit?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -k "feature_parallel or tp_merge or filtered_encoder_equals" > $O/tp_tests.log 2>&1; echo "rc=$?" >> $O/tp_tests.log
tail -8 $O/tp_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tp8 -o tp8 -- python $R/tools/tp_shard_times.py 8 > $O/tp8.json 2> $O/tp8.err
python - <<'PY'
import csv, glob, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3c'
f = glob.glob(O + '/prof_tp8/**/*kernel_stats.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:45]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  {float(r['Percentage']):5.1f}%  {r['Name'][:110]}")
PY
