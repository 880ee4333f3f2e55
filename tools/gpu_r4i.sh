# where the select kernel's 135 us go: timing ablations (ranking pass twice / no exact re-scoring loads) + SQ counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T0=$(date +%s)
for v in base rank2 norescore; do
  L=$R/vit_prisma_amd/libpvnative.so; [ $v != base ] && L=$R/tools/variants/libpvnative_$v.so
  PV_NATIVE_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o s -- python $R/tools/prof_sae.py > $O/$v.out 2> $O/$v.err
  cp $O/p_$v/s_kernel_stats.csv $O/${v}_kernel_stats.csv; rm -rf $O/p_$v
  echo $v; grep -h 'sae_select_kernel\|sae_decode_kernel\|sae_backward_kernel' $O/${v}_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}' | cut -d, -f1-3
done
echo "ablations $(( $(date +%s) - T0 ))s"
timeout 120 rocprofv3 -L > $O/counters.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -o p -- python $R/tools/prof_sae.py > $O/pmc1.log 2>&1
python $R/tools/pmc_mfma.py $O/pmc1 $O/pmc_sq_waits.json "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -- python tools/prof_sae.py" select decode backward adam enc_gemm > $O/pmc_sq_waits.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2 -o p -- python $R/tools/prof_sae.py > $O/pmc2.log 2>&1
python $R/tools/pmc_mfma.py $O/pmc2 $O/pmc_sq_insts.json "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -- python tools/prof_sae.py" select decode backward adam enc_gemm > $O/pmc_sq_insts.txt 2>&1
rm -rf $O/pmc1 $O/pmc2
tail -3 $O/pmc1.log; tail -3 $O/pmc2.log
python - <<PY
import json
for f in ('pmc_sq_waits','pmc_sq_insts'):
    try:
        d=json.load(open('$O/'+f+'.json'))
        for k,v in d['kernels'].items():
            if 'select' in k or 'decode' in k or 'backward_kernel' in k: print(f, k[:40], v)
    except Exception as e: print(f, 'failed', e)
PY
echo "total $(( $(date +%s) - T0 ))s"
