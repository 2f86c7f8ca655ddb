#!/bin/bash
# A/B of kernel builds on one box: gpu_ab.sh LIB_A LIB_B [bench args...]; "-" = the product library.
# Runs A B A B (interleaved against clock/box drift), prints ms_per_step and cascade avg per run.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
A=$1; B=$2; shift 2
mkdir -p $R/gpurun_out; L=$R/gpurun_out/ab.log; : > $L
for lib in $A $B $A $B; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    echo "== $lib" >> $L
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu "$@" 2>/dev/null >> $L || { echo FAILED >> $L; tail -5 $L; exit 1; }
done
python - <<PY
import json
tag=None
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        print(f"{tag:45s} step {j['ms_per_step']:.3f} ms  cascade {j['roofline'].get('avg_launch_ms')} ms  frac {j['roofline']['frac']:.4f} parity {j.get('parity')}")
PY
