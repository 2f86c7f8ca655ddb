#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp9.log; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu >> $L 2>&1 || { tail -30 $L; exit 1; }
timeout -k 10 600 python bench.py --wideband 512 --frames 12 --steps 6 --warmup 1 --no-cpu 2>/dev/null >> $L
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null >> $L
grep -o 'passed.*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"parity": [a-z]*\|"frac": [0-9.]*' $L | paste -sd' '
