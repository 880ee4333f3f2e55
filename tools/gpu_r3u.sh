R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/variants_time.py > $O/variants.json 2> $O/err.log; echo "rc=$?"
cat $O/variants.json; tail -5 $O/err.log
