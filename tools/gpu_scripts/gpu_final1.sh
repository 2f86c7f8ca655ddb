#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/final1; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py > $O/trace.log 2>&1; echo "trace rc=$?"
cat $O/trace/*/*kernel_stats.csv
cut -c1-2500 $O/bench.json
