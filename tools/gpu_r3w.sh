R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O
cd $R
timeout 120 ./tools/probes/write_bw_probe.bin > $O/write_bw.log 2>&1; echo "rc=$?" >> $O/write_bw.log
cat $O/write_bw.log
