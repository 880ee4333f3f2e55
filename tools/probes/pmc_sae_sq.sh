# SQ counters per kernel of the top-k SAE step (one pass): where do the launch-bound kernels wait?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p -o p -- python $R/tools/prof_sae.py > $O/log.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:26]:
    c = max(n[k], 1)
    print(f"{k:60s} n={c:3d} gui={v['GRBM_GUI_ACTIVE']/c:9.0f} waves={v['SQ_WAVES']/c:8.0f} wave_cyc={v['SQ_WAVE_CYCLES']/c:11.0f} wait_any={v['SQ_WAIT_ANY']/max(v['SQ_WAVE_CYCLES'],1):.2f} wait_inst={v['SQ_WAIT_INST_ANY']/max(v['SQ_WAVE_CYCLES'],1):.2f} active={v['SQ_ACTIVE_INST_ANY']/max(v['SQ_WAVE_CYCLES'],1):.2f}")
PY
rm -rf $O/p
