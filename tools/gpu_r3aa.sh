R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3aa; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o e2e -- python $R/tools/prof_e2e.py > $O/e2e.out 2> $O/e2e.err
cat $O/e2e.out | tail -1
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/prof/e2e_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:28]:
    print(f"{float(r['Percentage']):6.2f}% calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
cp $O/prof/e2e_kernel_stats.csv $O/e2e_kernel_stats.csv; rm -rf $O/prof
cd $R; timeout 300 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -k "fallback_dispatch" 2>&1 | tail -2
