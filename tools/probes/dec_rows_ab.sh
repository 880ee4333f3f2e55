# decode kernel with 4 (shipped) / 8 rows of W_dec in flight per trip: kernel time, step time, loss (must agree)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in base dec8 base dec8; do
  lib=$R/vit_prisma_amd/libpvnative.so; [ $v != base ] && lib=$R/tools/variants/libpvnative_$v.so
  PV_ALLOW_STALE_LIB=1 PV_NATIVE_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec_$v -o p -- python $R/tools/prof_sae.py > $O/prof_dec_$v.log 2>&1
  f=$(find $O/prof_dec_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sae_decode_kernel" in r["Name"]:
        print("  decode", r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us  min", round(float(r["MinNs"]) / 1e3, 1))
PY
  grep -o "ms_per_step.: [0-9.]*\|final_loss.: [0-9.e-]*" $O/prof_dec_$v.log | head -2 | tr '\n' ' '; echo
  rm -rf $O/prof_dec_$v
done
