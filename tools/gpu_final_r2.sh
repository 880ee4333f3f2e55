# last pass of the round: every -m gpu test, smoke(), the bench line, a PMC look at the non-GEMM kernels of the forward
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fin2; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -m gpu -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?" >> $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/tests_gpu.log; tail -2 $O/smoke.log
python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 200 $O/bench_full.err
python - <<PY
import json
d=json.load(open('$O/bench_full.json'))
print('b32', d['value'], d['ms_per_step'], 'gemm', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'])
print('sae', d['sae']['value'], d['sae']['ms_per_step'], d['sae']['roofline']['frac'], 'e2e', d['sae']['end_to_end']['value'])
print('l14', d['l14_336_pattern']['value'], d['l14_336_pattern']['ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
P2="GRBM_GUI_ACTIVE TA_TA_BUSY_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
i=1
for ctrs in "$P1" "$P2"; do
  timeout 240 rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc$i -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc$i.log 2>&1
  python $R/tools/pmc_by_grid.py $O/pmc$i attn ln_kernel gemm_kernel_v7 > $O/pmc_forward_$i.json 2>> $O/pmc$i.log
  rm -rf $O/pmc$i
  i=$((i+1))
done
head -c 600 $O/pmc_forward_2.json
