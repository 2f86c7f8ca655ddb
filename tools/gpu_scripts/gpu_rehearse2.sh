#!/bin/bash
# Rehearsal of the multi-process bench path on ONE GPU: two gloo ranks, both on device 0, half the streams each
# (RCCL refuses two ranks on one device, so the collectives run on gloo; the data path is the real one).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
export NVX_BENCH_BACKEND=gloo NVX_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --streams 2048 --steps 10 --warmup 2 > gpurun_out/rehearse2.log 2> gpurun_out/rehearse2.err
echo "rehearse rc=$?"; tail -1 gpurun_out/rehearse2.log | cut -c1-1500
timeout -k 10 300 python bench.py --streams 2048 --steps 10 --warmup 2 --no-cpu --verify 32 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('single 2048:', j['ms_per_step'], j['parity'])"
