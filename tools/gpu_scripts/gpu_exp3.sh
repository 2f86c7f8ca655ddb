#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp3.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 300 python bench.py --no-cpu --steps 6 --warmup 2 $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], r['demod_avg_launch_ms'], d['parity'])" >> $L 2>&1; }
ARGS="" run NVX_NT=0
ARGS="" run NVX_NT=1
ARGS="--streams 2816" run NVX_NT=1
ARGS="--streams 5632" run NVX_NT=1
ARGS="--streams 2048" run NVX_NT=1
ARGS="--streams 1024" run NVX_NT=1
ARGS="--streams 3072" run NVX_NT=1
cat $L
