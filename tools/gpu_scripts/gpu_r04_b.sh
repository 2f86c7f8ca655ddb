#!/bin/bash
# Round 4: bench.py's self-verification at test size (the GPU tests around it), then the driver's command at full size.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04b; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_bench_checks.py -x -q -m gpu > $O/checks.log 2>&1; rc=$?; tail -30 $O/checks.log; echo "checks rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"; tail -5 $O/bench.err; cut -c1-3000 $O/bench.json
exit $rc
