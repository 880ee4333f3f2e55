# round-6 evidence run: every -m gpu test, smoke(), the full bench line, rocprofv3 kernel stats (ViT leg, SAE top-k leg, ReLU sparse leg,
# L/14 leg), PMC traffic passes (ViT + SAE), MFMA-utilisation pass, per-rank times of the feature-parallel step.  Every stage under its
# own timeout; summaries are copied into profiles/r06_* by hand.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6final; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
rm -f $R/gpurun_out/truncated_tests.txt
timeout 2400 python -m pytest tests/ -m gpu -q -p no:cacheprovider --timeout=1500 --durations=8 > $O/tests_gpu.log 2>&1; echo "tests rc=$? $(( $(date +%s) - T0 ))s" >> $O/tests_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
grep -E "passed|failed|rc=" $O/tests_gpu.log | tail -4; grep -E "^FAILED|^ERROR" $O/tests_gpu.log | head; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? $(( $(date +%s) - T0 ))s"; tail -c 300 $O/bench_full.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vit -o vit -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 10 --warmup 3 > $O/prof_vit_bench.json 2> $O/prof_vit.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sae -o sae -- python $R/tools/prof_sae.py > $O/prof_sae.out 2> $O/prof_sae.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_relu -o relu -- python $R/tools/prof_relu.py > $O/prof_relu.out 2> $O/prof_relu.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_reludense -o reludense -- python $R/tools/prof_relu_dense.py > $O/prof_reludense.out 2> $O/prof_reludense.err
VARIANT=gated_relu timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gateddense -o gateddense -- python $R/tools/sae_variant_time.py > $O/prof_gateddense.out 2> $O/prof_gateddense.err
VARIANT=gated_relu_l0_64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gated -o gated -- python $R/tools/sae_variant_time.py > $O/prof_gated.out 2> $O/prof_gated.err
STEPS=6 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l14 -o l14 -- python $R/tools/l14_run.py > $O/prof_l14.json 2> $O/prof_l14.err
echo "kernel traces done $(( $(date +%s) - T0 ))s"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_vit.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sae --no-l14; counter unit KB per launch; hbm_bytes = (2*FETCH + WRITE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)" > $O/pmc_traffic_vit.txt
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_sae_fetch -o p -- python $R/tools/prof_sae.py > $O/pmc_sae_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_sae_write -o p -- python $R/tools/prof_sae.py > $O/pmc_sae_write.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_sae_fetch $O/pmc_sae_write $O/pmc_traffic_sae.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/prof_sae.py (7 SAE train steps 768 -> 24576, k = 32, N = 4096); KB per launch; hbm_bytes = (2*FETCH + WRITE)*1024" > $O/pmc_traffic_sae.txt
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_relu_fetch -o p -- python $R/tools/prof_relu.py > $O/pmc_relu_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_relu_write -o p -- python $R/tools/prof_relu.py > $O/pmc_relu_write.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_relu_fetch $O/pmc_relu_write $O/pmc_traffic_relu.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/prof_relu.py (7 ReLU + L1 train steps in the sparse form, 768 -> 24576, L0 ~ 16, N = 4096); KB per launch; hbm_bytes = (2*FETCH + WRITE)*1024" > $O/pmc_traffic_relu.txt
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
python $R/tools/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python bench.py --no-sae --no-l14 --no-cpu-baseline --steps 2 --warmup 1" > $O/pmc_mfma.txt 2>&1
echo "pmc done $(( $(date +%s) - T0 ))s"
STEPS=40 REPS=3 timeout 300 python $R/tools/sae_fold_ab.py > $O/sae_fold_ab.txt 2>&1
timeout 200 python $R/tools/dense_split_err.py > $O/dense_split_err.json 2> $O/dense_split_err.err
timeout 300 python $R/tools/tp_shard_times.py > $O/tp_shard_times.json 2> $O/tp_shard_times.err
for n in vit sae relu reludense gateddense gated l14; do cp $O/prof_$n/${n}_kernel_stats.csv $O/${n}_kernel_stats.csv 2>/dev/null; done
rm -rf $O/prof_vit $O/prof_sae $O/prof_relu $O/prof_reludense $O/prof_gateddense $O/prof_gated $O/prof_l14 $O/pmc_fetch $O/pmc_write $O/pmc_sae_fetch $O/pmc_sae_write $O/pmc_relu_fetch $O/pmc_relu_write $O/pmc_mfma
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_full.json') if l.startswith('{"metric"')][0])
print('b32', d['value'], d['ms_per_step'], 'gemm', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'], 'ok', d.get('ok'))
print('instances', {k: (v['avg_launch_us'], v['frac_mfma'], v['frac_hbm']) for k, v in d['roofline'].get('instances', {}).items()})
s=d['sae']; print('sae', s['value'], s['ms_per_step'], s['roofline']['frac'], 'e2e', s['end_to_end']['value'], 'ref-store', s['end_to_end'].get('reference_store_shape',{}).get('value'))
r=s['relu_l1']; print('relu', r['value'], r['ms_per_step'], r.get('sparse_steps'), r.get('dense_steps'), {k: (r[k].get('ms_per_step'), r[k].get('sparse_steps'), r[k].get('dense_steps')) for k in ('from_init','published_l0','l0_64') if k in r})
print('variants', {k: (v.get('value'), v.get('ms_per_step')) for k, v in s.get('variants', {}).items()})
print('l14', d['l14_336_pattern']['value'], d['l14_336_pattern']['ms_per_step'])
print('summary', d.get('summary'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], s.get('cpu_baseline', {}).get('value'))
PY
head -14 $O/relu_kernel_stats.csv | cut -c1-150
echo "total $(( $(date +%s) - T0 ))s"
