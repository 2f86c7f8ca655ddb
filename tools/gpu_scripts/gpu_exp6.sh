#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp6.log; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu >> $L 2>&1 || { tail -30 $L; exit 1; }
timeout -k 10 300 python bench.py --steps 10 --warmup 2 2>/dev/null >> $L
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null >> $L
grep -o '"demod_avg_launch_ms": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"parity": [a-z]*\|"value": [0-9.]*\|passed.*' $L
