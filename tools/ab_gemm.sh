mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*
python -m pytest tests/test_native_vit_gpu.py -m gpu -x -q 2>&1 | tail -8
PV_ATTN_WG=1 python -m pytest tests/test_native_vit_gpu.py -m gpu -x -q -k attention_core 2>&1 | tail -3
run() { name=$1; shift; env "$@" python bench.py --no-sae --no-cpu-baseline --steps 30 --warmup 8 > gpurun_out/ab/b_$name.json 2>/dev/null; }
run wave A=1
run wg PV_ATTN_WG=1
run wave2 A=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/b_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('avg_launch_us'), j['kernels']['attention']['avg_launch_us'], j['kernels']['layernorm']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
