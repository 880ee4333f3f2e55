# round 3: token-sorted / XCD-ranged long-list backward -- parity (whole SAE suite), step time, kernel stats, PMC traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -x -k "native_step_vs_oracle or every_gradient_row or sparse_gradient or feature_parallel_simulated or config3 or data_parallel" > $O/sae_tests.log 2>&1; echo "rc=$?" >> $O/sae_tests.log
tail -8 $O/sae_tests.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sae -o sae -- python $R/tools/prof_sae.py > $O/prof_sae.out 2> $O/prof_sae.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_sae_fetch -o p -- python $R/tools/prof_sae.py > $O/pmc_sae_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_sae_write -o p -- python $R/tools/prof_sae.py > $O/pmc_sae_write.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_sae_fetch $O/pmc_sae_write $O/pmc_traffic_sae.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/prof_sae.py (7 SAE train steps 768 -> 24576, k = 32, N = 4096); KB per launch; hbm_bytes = (2*FETCH + WRITE)*1024" > $O/pmc_traffic_sae.txt
cp $O/prof_sae/sae_kernel_stats.csv $O/sae_kernel_stats.csv; rm -rf $O/prof_sae $O/pmc_sae_fetch $O/pmc_sae_write
grep -o "'ms_per_step': [0-9.]*" $O/prof_sae.out | head -2
python - <<'PY'
import csv, json, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3l'
rows = list(csv.DictReader(open(O + '/sae_kernel_stats.csv')))
print('sum', sum(int(r['TotalDurationNs']) for r in rows) / 7 / 1e3)
for r in rows[:14]:
    print(f"{int(r['TotalDurationNs'])/7/1e3:8.1f}  {r['Name'][:90]}")
d = json.load(open(O + '/pmc_traffic_sae.json'))['kernels']
steps = max(v['launches'] for k, v in d.items() if 'sae_decode_kernel' in k)
print('traffic GB/step', sum(v['hbm_bytes_per_launch_corrected'] * v['launches'] for v in d.values()) / steps / 1e9)
for k, v in sorted(d.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch_corrected'] * kv[1]['launches'])[:8]:
    print(f"{v['hbm_bytes_per_launch_corrected']*v['launches']/steps/1e6:8.1f} MB  {k[:60]}")
PY
