#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp10.log; : > $L
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu >> $L 2>&1 || { tail -30 $L; exit 1; }
for v in 0 1 0 1; do
echo "== early_reload $v" >> $L
NVX_EARLY_RELOAD=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null >> $L
done
grep -o '^== .*\|passed.*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"frac": [0-9.]*' $L | paste -sd' ' | sed 's/== /\n== /g'
