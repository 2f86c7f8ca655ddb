#!/bin/bash
# Round 4 rehearsals of the multi-GPU forms on ONE GPU: 4 gloo ranks sharing device 0 (bench.py starts them itself), and ONE
# process with an nvx_group of 8 members, all on device 0 (4096 streams in all either way).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04r; rm -rf $O; mkdir -p $O
cd $R
NVX_BENCH_BACKEND=gloo NVX_BENCH_DEVICE=0 timeout -k 10 600 python3 bench.py --gpus 4 --streams 1024 --steps 10 --warmup 2 --no-cpu > $O/ranks4.json 2> $O/ranks4.err; echo "4 ranks rc=$?"; cut -c1-300 $O/ranks4.json
NVX_BENCH_GROUP_DEVICES=0,0,0,0,0,0,0,0 timeout -k 10 600 python3 bench.py --gpus 8 --group --streams 512 --steps 10 --warmup 2 > $O/group8.json 2> $O/group8.err; echo "group of 8 rc=$?"; cut -c1-300 $O/group8.json
