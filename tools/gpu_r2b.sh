# round-2 second pass: GPU tests, full bench line, PMC diagnosis of the two K-loop forms
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/ -m gpu -q -x > $O/tests_gpu.log 2>&1; echo "tests rc=$?" >> $O/tests_gpu.log
tail -4 $O/tests_gpu.log
timeout 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 300 $O/bench_full.err
python - <<PY
import json
d=json.load(open('$O/bench_full.json'))
print('b32', d['value'], d['ms_per_step'], 'gemm', d['roofline']['achieved'], d['roofline']['avg_launch_us'])
print('sae', d['sae']['value'], d['sae']['ms_per_step'], d['sae']['kernels'], 'e2e', d['sae']['end_to_end']['value'])
print('l14', d['l14_336_pattern']['value'], d['l14_336_pattern']['ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE TA_TA_BUSY_sum SQ_INSTS_LDS"
i=1
for ctrs in "$P1" "$P2"; do
  REPS=4 timeout 240 rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc$i -o p -- python $R/tools/gemm_ab.py 0,1 > $O/pmc$i.log 2>&1
  python $R/tools/pmc_by_grid.py $O/pmc$i gemm_kernel_v7 > $O/pmc_loop_$i.json 2>> $O/pmc$i.log
  rm -rf $O/pmc$i
  i=$((i+1))
done
head -c 1500 $O/pmc_loop_1.json
