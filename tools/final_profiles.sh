# round evidence run: kernel-trace stats (ViT leg, SAE leg, L/14 leg), PMC traffic passes (ViT + SAE), MFMA-utilisation pass, full bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vit -o vit -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 10 --warmup 3 > $O/prof_vit_bench.json 2> $O/prof_vit.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sae -o sae -- python $R/tools/prof_sae.py > $O/prof_sae.out 2> $O/prof_sae.err
STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l14 -o l14 -- python $R/tools/l14_run.py > $O/prof_l14.json 2> $O/prof_l14.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sae --no-l14; counter unit KB per launch; hbm_bytes = (2*FETCH + WRITE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)" > $O/pmc_traffic.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_sae_fetch -o p -- python $R/tools/prof_sae.py > $O/pmc_sae_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_sae_write -o p -- python $R/tools/prof_sae.py > $O/pmc_sae_write.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_sae_fetch $O/pmc_sae_write $O/pmc_traffic_sae.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/prof_sae.py (7 SAE train steps 768 -> 24576, k = 32, N = 4096); KB per launch; hbm_bytes = (2*FETCH + WRITE)*1024" > $O/pmc_traffic_sae.txt
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
python $R/tools/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1" > $O/pmc_mfma.txt 2>&1
cd $R && python bench.py > $O/bench_full.json 2> $O/bench_full.err
cp $O/prof_vit/vit_kernel_stats.csv $O/vit_kernel_stats.csv; cp $O/prof_sae/sae_kernel_stats.csv $O/sae_kernel_stats.csv; cp $O/prof_l14/l14_kernel_stats.csv $O/l14_kernel_stats.csv
tail -c 1500 $O/bench_full.json; tail -3 $O/bench_full.err; cat $O/pmc_traffic_sae.txt | head -30
