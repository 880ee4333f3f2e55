# round 3, fourth GPU pass: the dense ReLU + L1 step (first time on hardware), kernel trace of one rank's work at world 8
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -x -k "relu_l1" > $O/dense_tests.log 2>&1; echo "rc=$?" >> $O/dense_tests.log
tail -25 $O/dense_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tp8 -o tp8 -- python $R/tools/tp_shard_times.py 8 > $O/tp8.json 2> $O/tp8.err
tail -3 $O/tp8.err
python - <<'PY'
import csv, glob, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3d'
f = glob.glob(O + '/prof_tp8/**/*kernel_stats.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:45]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  {float(r['Percentage']):5.1f}%  {r['Name'][:110]}")
PY
