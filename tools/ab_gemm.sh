set -x
mkdir -p gpurun_out/ab
python -m pytest tests/test_native_vit_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/gemm_bench.py > gpurun_out/ab/mb_v4.txt 2>&1
PV_GEMM_V5=1 python tools/gemm_bench.py > gpurun_out/ab/mb_v5.txt 2>&1
PV_GEMM_DBG=8 python tools/gemm_bench.py > gpurun_out/ab/mb_v4_nopre.txt 2>&1
for rep in 1 2; do
python bench.py --no-sae --no-cpu-baseline --steps 30 --warmup 8 > gpurun_out/ab/b_v4_$rep.json 2>/dev/null
PV_GEMM_DBG=8 python bench.py --no-sae --no-cpu-baseline --steps 30 --warmup 8 > gpurun_out/ab/b_v4nopre_$rep.json 2>/dev/null
PV_GEMM_V5=1 python bench.py --no-sae --no-cpu-baseline --steps 30 --warmup 8 > gpurun_out/ab/b_v5_$rep.json 2>/dev/null
done
for d in 1 2 3; do PV_GEMM_DBG=$d python bench.py --no-sae --no-cpu-baseline --steps 30 --warmup 8 > gpurun_out/ab/b_v4_dbg$d.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/b_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('avg_launch_us'), {k:v.get('avg_us') for k,v in j.get('kernels',{}).items()})
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/ab/mb_*.txt
