#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/exp2.log; : > $L
timeout -k 10 600 python -m pytest tests -x -q -m gpu >> $L 2>&1 || { tail -30 $L; exit 1; }
for pfd in 1 2 1 2; do
  echo "== prefetch $pfd" >> $L
  NVX_PREFETCH=$pfd timeout -k 10 300 python bench.py --no-cpu --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['demod_avg_launch_ms'], d['parity'])" >> $L 2>&1
done
cat $L
