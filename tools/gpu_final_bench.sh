# per-key parity ratios at the bench configurations + the final bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fb; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python tools/parity_ratios.py > $O/parity.log 2>&1; cp gpurun_out/parity_ratios.json $O/ 2>/dev/null
python bench.py > $O/bench_full.json 2> $O/bench_full.err
tail -12 $O/parity.log; tail -c 600 $O/bench_full.json; tail -2 $O/bench_full.err
