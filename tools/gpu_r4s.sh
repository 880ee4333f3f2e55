# last look at the final binary: smoke(), a cross-section of the GPU suite, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; rm -rf $O; mkdir -p $O
cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 200 python -m pytest tests/ -m gpu -q -p no:cacheprovider --timeout=150 -k "library_is_loaded or config3 or bit_reproducible_from_run or relu_sparse_gradient or b32_within_reference_bf16_budget or topk_ghost_step" > $O/t.log 2>&1; echo "rc=$?"; tail -2 $O/t.log
timeout 200 python bench.py --no-l14 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{"metric"')][0])
s=d['sae']; r=s['relu_l1']
print('b32', d['value'], d['ms_per_step'], d['roofline']['frac'], 'sae', s['ms_per_step'], 'relu', r['ms_per_step'], 'e2e', s['end_to_end']['value'], s['end_to_end']['reference_store_shape']['value'], {k:(v['value']) for k,v in s['variants'].items()}, d.get('ok'))
PY
