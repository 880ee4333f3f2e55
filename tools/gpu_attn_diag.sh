# where does the T = 577 attention kernel's time go?  kernel time with / without the pattern tap + PMC passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_diag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in pattern none; do
  MODE=$M timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python $R/tools/attn_diag.py > $O/run_$M.log 2>&1
  grep attn_ $O/p/t_kernel_stats.csv | cut -d, -f1-4 | sed "s/^/$M: /" >> $O/summary.txt
  rm -rf $O/p
done
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  MODE=pattern ITERS=2 timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc$i -o c -- python $R/tools/attn_diag.py > $O/pmc$i.log 2>&1
  python - "$O/pmc$i" >> $O/summary.txt 2>&1 <<'P'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counters in", sys.argv[1]); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "attn_" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("pmc", k, "per launch", sum(v) / len(v), "launches", len(v))
P
  rm -rf $O/pmc$i
done
cat $O/summary.txt
