#!/bin/bash
# GPU parity suite only; the log lands in gpurun_out/suite.log
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/suite.log; : > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu >> $L 2>&1; rc=$?
tail -15 $L; exit $rc
