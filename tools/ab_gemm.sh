for r in 1 2; do
for w in 8 12 16; do
  PV_V7_WBLK=$w python bench.py --no-sae --no-l14 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wblk $w bench', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['avg_launch_us'])"
done; done
