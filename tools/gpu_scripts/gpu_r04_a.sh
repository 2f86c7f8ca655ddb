#!/bin/bash
# Round 4: the seal tests, the hand-over tests around them, then the headline A/B against the round-3 kernels and the
# seal-less build of today's (ROUNDS interleaved rounds, medians).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04a; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_integrity.py -x -q -m gpu > $O/integrity.log 2>&1; rc=$?; tail -25 $O/integrity.log; echo "integrity rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "handoff or dependent or randomized or carry" > $O/parity.log 2>&1; rc=$?; tail -5 $O/parity.log; echo "parity rc=$rc"
[ $rc -ne 0 ] && exit $rc
ROUNDS=${ROUNDS:-8} bash tools/gpu_scripts/gpu_abn.sh tools/_bin/libnavtex_amd_r3base.so - tests/_variants/libnavtex_amd_sealoff.so -- --no-legs > $O/ab.log 2>&1; rc=$?; tail -4 $O/ab.log; echo "ab rc=$rc"
exit $rc
