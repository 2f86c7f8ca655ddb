#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp5.log; : > $L
timeout -k 10 600 python -m pytest tests -x -q -m gpu >> $L 2>&1 || { tail -30 $L; exit 1; }
timeout -k 10 300 python bench.py --steps 8 --warmup 2 2>/dev/null >> $L
tail -3 $L | cut -c1-1800
