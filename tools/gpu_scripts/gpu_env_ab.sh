#!/bin/bash
# A/B of an environment switch on one box: gpu_env_ab.sh VAR VAL_A VAL_B [bench args...]; two interleaved rounds.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
V=$1; A=$2; B=$3; shift 3
mkdir -p $R/gpurun_out; L=$R/gpurun_out/env_ab.log; : > $L
for val in $A $B $A $B; do
    echo "== $V=$val" >> $L
    env $V=$val timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-stage0-extra "$@" 2>/dev/null >> $L || { echo FAILED >> $L; tail -5 $L; exit 1; }
done
python - <<PY
import json
tag=None
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        print(f"{tag:30s} step {j['ms_per_step']:.3f} ms  cascade {j['roofline'].get('avg_launch_ms')} ms  frac {j['roofline']['frac']:.4f} parity {j.get('parity')}")
PY
