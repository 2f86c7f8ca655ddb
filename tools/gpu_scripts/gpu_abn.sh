#!/bin/bash
# A/B/.../N of kernel builds on one box: gpu_abn.sh LIB... [-- bench args]; "-" = the product library.
# ROUNDS (default 2) interleaved rounds (against clock/box drift); prints ms_per_step, cascade average and parity per run.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" = "--" ] && shift
mkdir -p $R/gpurun_out; L=$R/gpurun_out/abn.log; : > $L
for round in $(seq 1 ${ROUNDS:-2}); do for lib in "${libs[@]}"; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    echo "== $lib" >> $L
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-stage0-extra --verify 32 "$@" 2>/dev/null >> $L || { echo FAILED >> $L; tail -5 $L; exit 1; }
done; done
python - <<PY
import json
import statistics, collections
tag=None; by=collections.OrderedDict()
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        by.setdefault(tag, []).append(j['roofline'].get('avg_launch_ms'))
        print(f"{tag:45s} step {j['ms_per_step']:.3f} ms  cascade {j['roofline'].get('avg_launch_ms')} ms  frac {j['roofline']['frac']:.4f} parity {j.get('parity')}")
for tag, v in by.items():
    print(f"{tag:45s} cascade median {statistics.median(v):.3f}  min {min(v):.3f}  mean {statistics.mean(v):.3f}  n {len(v)}")
PY
