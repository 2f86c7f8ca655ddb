#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp1.log; : > $L
for pad in 0 64 1024 1088 16448; do
  echo "== pad $pad" >> $L
  timeout -k 10 300 python bench.py --no-cpu --steps 6 --warmup 2 --pitch-pad $pad 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['demod_avg_launch_ms'], d['parity'])" >> $L 2>&1
done
cat $L
