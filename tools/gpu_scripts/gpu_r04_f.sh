#!/bin/bash
# Round 4 campaigns on the shipped code: the seal tests, then full-size soak runs and ragged push-mode cases with the seals'
# counters (expected: 0 stale, 0 failed), all against the oracle / run to run.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04f; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_integrity.py -x -q -m gpu > $O/integrity.log 2>&1; rc=$?; tail -4 $O/integrity.log; echo "integrity rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python tools/gpu_scripts/soak_units.py ${SOAK:-90} > $O/soak_units.log 2>&1; rc=$?; tail -3 $O/soak_units.log; echo "soak rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/gpu_scripts/sweep_ragged.py 300 ${RAGGED:-80} > $O/sweep_ragged.log 2>&1; rc=$?; tail -2 $O/sweep_ragged.log; echo "ragged rc=$rc"
grep -o "'stale_repaired': [0-9]*" $O/sweep_ragged.log | sort | uniq -c
exit $rc
