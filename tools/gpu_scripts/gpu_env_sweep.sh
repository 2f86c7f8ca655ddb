#!/bin/bash
# sweep of an environment switch on one box: gpu_env_sweep.sh VAR "v1 v2 ..." [bench args...]; ROUNDS (default 2) interleaved rounds
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
V=$1; VALS=$2; shift 2
mkdir -p $R/gpurun_out; L=$R/gpurun_out/env_sweep.log; : > $L
for round in $(seq 1 ${ROUNDS:-2}); do for val in $VALS; do
    echo "== $V=$val" >> $L
    env $V=$val timeout -k 10 300 python bench.py --no-cpu --no-stage0-extra --no-legs --verify 32 "$@" 2>/dev/null >> $L || { echo FAILED >> $L; tail -5 $L; exit 1; }
done; done
python - <<PY
import json, collections, statistics
tag=None; by=collections.OrderedDict()
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        by.setdefault(tag, []).append((j['ms_per_step'], j['roofline'].get('avg_launch_ms'), j.get('parity')))
for tag, v in by.items():
    print(f"{tag:30s} step {[x[0] for x in v]}  cascade {[x[1] for x in v]}  parity {all(x[2] for x in v)}")
PY
