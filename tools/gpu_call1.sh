# round-2 GPU call 1: new parity tests at the bench configs, per-key ratios, MFMA-utilisation PMC pass, CPU baseline on the box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_vit_gpu.py -m gpu -x -q -k "bs512 or l14_bs128 or do_not_depend or within_reference_bf16_budget or ragged_shapes or library_is_loaded" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 600 python tools/parity_ratios.py > $O/parity.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
python $R/tools/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 (B/32 bs=512 bf16 all hooks)" > $O/pmc_mfma.txt 2>&1
rm -rf $O/pmc_mfma
cd $R && timeout 600 python bench.py --no-sae --no-l14 > $O/bench_vit.json 2> $O/bench_vit.err
tail -5 $O/tests.log; tail -30 $O/parity.log; cat $O/pmc_mfma.txt | head; tail -c 1500 $O/bench_vit.json
