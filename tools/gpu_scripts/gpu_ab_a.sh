#!/bin/bash
# A/B of kernel builds in ONE bench mode: gpu_ab_a.sh "<bench args>" LIB...   ("-" = the product library; parity is
# expected to fail for timing probes)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mode=$1; shift
mkdir -p $R/gpurun_out; L=$R/gpurun_out/ab_a.log; : > $L
for round in 1 2; do for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    echo "== $lib" >> $L
    timeout -k 10 300 python bench.py --warmup 2 --no-cpu --verify 32 $mode 2>>$R/gpurun_out/ab_a.err >> $L || { echo FAILED >> $L; tail -5 $L; tail -5 $R/gpurun_out/ab_a.err; }
done; done
python - <<PY
import json
tag=None
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        print(f"{tag:50s} step {j['ms_per_step']:8.3f} ms  cascade {j['roofline'].get('avg_launch_ms')} ms  parity {j.get('parity')}")
PY
