#!/bin/bash
# thirds throughout for small independent launches: -m gpu suite, then the small BASELINE configs
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; rc=$?
echo "suite rc=$rc: $(tail -1 $O/suite.log)"; [ $rc -ne 0 ] && { tail -30 $O/suite.log; exit $rc; }
for split in 0 auto; do
  if [ $split = auto ]; then unset NVX_TAIL_SPLIT; else export NVX_TAIL_SPLIT=$split; fi
  for S in 1 3 16 64; do
    timeout -k 10 200 python3 bench.py --streams $S --frames 62 --steps 40 --warmup 5 --no-cpu --no-legs --no-stage0-extra 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read().splitlines()[-1]); print('NVX_TAIL_SPLIT=$split streams $S frames 62: step', r['ms_per_step'], 'ms  cascade', r['roofline']['avg_launch_ms'], ' demod span', r['roofline']['demod_span_ms'], ' value', r['value'], 'parity', r['parity'])"
  done
done | tee $O/small_configs.txt
