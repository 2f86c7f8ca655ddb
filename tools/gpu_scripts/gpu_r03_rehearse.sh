#!/bin/bash
# Rehearsal of the multi-rank path on ONE GPU through bench.py's own launcher (no torchrun on the command line):
# N gloo ranks on device 0, 4096 / N streams each (RCCL refuses several ranks on one device).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03reh; mkdir -p $O
cd $R
export NVX_BENCH_BACKEND=gloo NVX_BENCH_DEVICE=0
for n in 2 4; do
  timeout -k 10 400 python3 bench.py --gpus $n --streams $((4096 / n)) --steps 10 --warmup 3 --no-cpu > $O/rehearse_${n}ranks.json 2> $O/rehearse_${n}ranks.err; echo "n=$n rc=$?"
  python3 -c "
import json; r=json.load(open('$O/rehearse_${n}ranks.json'))
print('n', r['n_gpus'], 'step', r['ms_per_step'], 'value', r['value'], 'parity', r['parity'], r['parity_streams_checked'], r['ranks'])"
done
unset NVX_BENCH_BACKEND NVX_BENCH_DEVICE
timeout -k 10 300 python3 bench.py --gpus 1 --streams 2048 --steps 10 --warmup 3 --no-cpu --no-legs --no-stage0-extra --verify 32 > $O/one_rank_2048.json 2>/dev/null; python3 -c "
import json; r=json.load(open('$O/one_rank_2048.json')); print('1 rank x 2048: step', r['ms_per_step'])"
