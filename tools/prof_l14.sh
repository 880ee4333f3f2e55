cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_l14
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_l14 -o l14 -- python $R/tools/l14_run.py > $R/gpurun_out/prof_l14.json 2> $R/gpurun_out/prof_l14.err
python - <<'PY'
import csv,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/prof_l14/l14_kernel_stats.csv')))
for r in rows[:12]:
    print(f"{float(r['Percentage']):6.2f}% calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
PY
cat $R/gpurun_out/prof_l14.json | tail -1 | cut -c1-300
