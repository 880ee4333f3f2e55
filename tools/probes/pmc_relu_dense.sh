# HBM traffic per kernel of the ReLU + L1 step at the published L0 (dense form): separate FETCH_SIZE / WRITE_SIZE passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_rd; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEPS=3 timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- python $R/tools/prof_relu_dense.py > $O/f.log 2>&1
STEPS=3 timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o p -- python $R/tools/prof_relu_dense.py > $O/w.log 2>&1
python $R/tools/pmc_traffic.py $O/f $O/w $O/pmc_traffic_relu_dense.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/prof_relu_dense.py (ReLU + L1 steps at the published L0, dense split-fp16 form); KB per launch; hbm_bytes = (2*FETCH + WRITE)*1024" > $O/pmc_traffic_relu_dense.txt
head -30 $O/pmc_traffic_relu_dense.txt
rm -rf $O/f $O/w
