# round 3, fifth GPU pass: whole SAE suite (dense step, x64, feature parallel), the bench line with the new legs, torchrun rehearsal
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_native_sae_gpu.py -m gpu -q > $O/sae_tests.log 2>&1; echo "rc=$?" >> $O/sae_tests.log
tail -12 $O/sae_tests.log
timeout 900 python bench.py --no-l14 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3e'
try:
    d = json.loads([l for l in open(O + '/bench.json') if l.startswith('{"metric"')][0])
    print('vit', d['value'], d['ms_per_step'], d['roofline']['frac'], 'ok', d.get('ok'))
    s = d['sae']; print('sae', s['value'], s['ms_per_step'], s['roofline']['frac'])
    print('e2e', s['end_to_end']['value'])
    r = s['relu_l1']; print('relu', json.dumps(r)[:900])
except Exception as e:
    print('no line', e)
PY
tail -5 $O/bench.err
BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-l14 > $O/gloo2.json 2> $O/gloo2.err; echo "gloo2 rc=$?"
python - <<'PY'
import json, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3e'
try:
    d = json.loads([l for l in open(O + '/gloo2.json') if l.startswith('{"metric"')][0])
    s = d['sae']
    print('gloo2 ok', d.get('ok'), 'sae', s.get('ms_per_step'), s.get('final_loss'), s.get('config', {}).get('parallelism'))
    print('e2e', s['end_to_end'].get('value'), s['end_to_end'].get('config', {}).get('parallelism'), s['end_to_end'].get('error'))
    w = s.get('weak_scaling_data_parallel', {}); print('weak', w.get('value'), w.get('ms_per_step'), w.get('final_loss'), w.get('error'))
except Exception as e:
    print('no line', e)
PY
tail -4 $O/gloo2.err
