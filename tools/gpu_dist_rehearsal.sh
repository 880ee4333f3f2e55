# the torchrun code path of bench.py: (1) one rank over RCCL, (2) two ranks sharing the GPU over gloo (rehearsal only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dist; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-l14 > $O/nccl1.json 2> $O/nccl1.err; echo "rc=$?" >> $O/nccl1.err
BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-l14 > $O/gloo2.json 2> $O/gloo2.err; echo "rc=$?" >> $O/gloo2.err
tail -c 1200 $O/nccl1.json; tail -3 $O/nccl1.err; tail -c 1500 $O/gloo2.json; tail -5 $O/gloo2.err
