R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/two_stream_probe.py > $O/two_stream.txt 2>&1; cat $O/two_stream.txt | grep -v amdgpu.ids
