#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
for S in 1 3 16 64 256; do for F in 12 62; do
    timeout -k 10 200 python3 bench.py --streams $S --frames $F --steps 40 --warmup 5 --no-cpu --no-legs --no-stage0-extra 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read().splitlines()[-1]); print('streams $S frames $F: step', r['ms_per_step'], 'ms  cascade', r['roofline']['avg_launch_ms'], ' demod span', r['roofline']['demod_span_ms'], ' value', r['value'], 'parity', r['parity'])"
done; done | tee $O/small_configs.txt
