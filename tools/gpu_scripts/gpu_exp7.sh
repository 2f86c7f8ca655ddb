#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp7.log; : > $L
echo "== variant y2run80 tests" >> $L
NAVTEX_AMD_LIB=$R/tools/_bin/libnavtex_amd_y2run80.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu >> $L 2>&1 || { tail -30 $L; exit 1; }
for i in 1 2; do
echo "== default" >> $L
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null >> $L
echo "== y2run80" >> $L
NAVTEX_AMD_LIB=$R/tools/_bin/libnavtex_amd_y2run80.so timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null >> $L
done
grep -o '^== .*\|"demod_avg_launch_ms": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"parity": [a-z]*\|passed.*' $L | paste -sd' ' | sed 's/== /\n== /g'
