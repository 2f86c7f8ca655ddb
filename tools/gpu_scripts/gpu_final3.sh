#!/bin/bash
# round artefacts: gpu_final2 (default bench + kernel stats + PMC) plus the variant-A and wideband bench lines
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
bash $R/tools/gpu_scripts/gpu_final2.sh > $R/gpurun_out/final2.out 2>&1 || { tail -20 $R/gpurun_out/final2.out; exit 1; }
O=$R/gpurun_out/final2
timeout -k 10 400 python bench.py --variant-a --frames 96 --steps 6 --warmup 1 > $O/variant_a.json 2> $O/variant_a.err; echo "variant-a rc=$?"
timeout -k 10 400 python bench.py --wideband 512 --frames 12 --steps 10 --warmup 2 > $O/wideband.json 2> $O/wideband.err; echo "wideband rc=$?"
timeout -k 10 400 python bench.py --stage0 cic3 > $O/cic3.json 2> $O/cic3.err; echo "cic3 rc=$?"
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"parity": [a-z]*\|"frac": [0-9.]*\|"value": [0-9.]*' $O/bench.json $O/variant_a.json $O/wideband.json $O/cic3.json | paste -sd' '
tail -32 $R/gpurun_out/final2.out | head -30
