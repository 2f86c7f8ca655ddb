#!/bin/bash
# parity tests, bench, rocprofv3 kernel trace
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $R/gpurun_out/test2.log 2>&1; echo "pytest rc=$?" >> $R/gpurun_out/test2.log
tail -5 $R/gpurun_out/test2.log
timeout -k 10 600 python bench.py > $R/gpurun_out/bench2.log 2>&1; echo "bench rc=$?" >> $R/gpurun_out/bench2.log
tail -3 $R/gpurun_out/bench2.log
export TMPDIR=/tmp
cd $R && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/prof2.log 2>&1; echo "prof rc=$?" >> $R/gpurun_out/prof2.log
tail -3 $R/gpurun_out/prof2.log
find $R/gpurun_out/prof2 -name "*stats*" | head
