#!/bin/bash
# round 3: the -m gpu suite, then the headline kernel A/B against the round-3 base build (tools/_bin/libnavtex_amd_r3base.so)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -15 $O/suite.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_scripts/gpu_abn.sh tools/_bin/libnavtex_amd_r3base.so - -- --no-legs > $O/ab_headline.log 2>&1; echo "ab rc=$?"; cat $O/ab_headline.log
