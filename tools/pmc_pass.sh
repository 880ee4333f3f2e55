# one rocprofv3 PMC pass per counter group over a short ViT bench run -> gpurun_out/pmc_<tag>/*counter_collection.csv,
# summarised per kernel (mean per launch) by tools/pmc_summary.py.  usage: bash tools/pmc_pass.sh TAG "CTR1 CTR2 ..."
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; ctrs=$2
rm -rf $R/gpurun_out/pmc_$tag
timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_$tag.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_$tag
