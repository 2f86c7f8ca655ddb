#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp4.log; : > $L
timeout -k 10 600 python -m pytest tests -x -q -m gpu >> $L 2>&1 || { tail -30 $L; exit 1; }
run() { echo "== $*" >> $L; env "$@" timeout -k 10 300 python bench.py --no-cpu --steps 6 --warmup 2 $ARGS 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], r['demod_avg_launch_ms'], d['parity'])" >> $L 2>&1; }
ARGS="" run NVX_PREFETCH=1
ARGS="" run NVX_PREFETCH=2
ARGS="" run NVX_WAVES_PER_CU=8
ARGS="" run NVX_WAVES_PER_CU=10
ARGS="--streams 2816" run NVX_PREFETCH=1
cat $L
