#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp8.log; : > $L
echo "== 2 ranks on one GPU, gloo rehearsal" >> $L
NVX_BENCH_BACKEND=gloo NVX_BENCH_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --streams 1024 --frames 4 >> $L 2>&1
echo "rc=$?" >> $L
echo "== torchrun world 1 (driver style)" >> $L
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu >> $L 2>&1
echo "rc=$?" >> $L
grep -v "amdgpu.ids\|^W1\|^\*\*\*\|OMP_NUM" $L | cut -c1-600
