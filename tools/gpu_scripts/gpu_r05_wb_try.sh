#!/bin/bash
# The restructured fused wideband kernel: its parity tests first (bounded), then the A/B against the round-4 form.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; mkdir -p gpurun_out; L=gpurun_out/wb_try.log; : > $L
timeout -k 10 500 python3 -m pytest tests/test_wideband.py -x -q -m gpu >> $L 2>&1; rc=$?
tail -5 $L
[ $rc -ne 0 ] && exit $rc
ROUNDS=${ROUNDS:-3} bash tools/gpu_scripts/gpu_r05_wb_ab.sh "$@"
