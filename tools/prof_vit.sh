# rocprofv3 kernel trace of the bench command (ViT leg only) -> gpurun_out/prof_vit/*_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_vit
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vit -o vit -- python $R/bench.py --no-sae --no-l14 --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/prof_vit_bench.json 2> $R/gpurun_out/prof_vit.err
ls -la $R/gpurun_out/prof_vit/
head -30 $R/gpurun_out/prof_vit/*kernel_stats.csv
python - <<'PY'
import csv,glob,os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/prof_vit/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# gaps inside the last forward: find last 'patch' gemm kernel (a_mode PATCH -> gemm_kernel<...>) occurrence
names=[r['Kernel_Name'] for r in rows]
last=max(i for i,n in enumerate(names) if 'l2norm' in n)
first=max(i for i,n in enumerate(names[:last]) if 'gemm_kernel<' in n and 'PATCH' not in n and i<last-50) if False else None
# forward = from previous l2norm+1 to last l2norm
prev=max(i for i,n in enumerate(names[:last]) if 'l2norm' in n)
seg=rows[prev+1:last+1]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
span=int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp'])
print('one forward: kernels',len(seg),'span us',span/1e3,'busy us',busy/1e3,'gaps us',(span-busy)/1e3)
PY
