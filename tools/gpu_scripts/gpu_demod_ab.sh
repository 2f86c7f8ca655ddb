#!/bin/bash
# demodulator timing of kernel builds: gpu_demod_ab.sh LIB...; prints demod_span_ms (front + FSM) per build
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/demod_ab.log; : > $L
for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    echo "== $lib" >> $L
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null >> $L || { echo FAILED >> $L; tail -5 $L; exit 1; }
done
python - <<PY
import json
tag=None
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        print(f"{tag:45s} step {j['ms_per_step']:.3f} ms  demod {j['roofline'].get('demod_span_ms')} ms  parity {j.get('parity')}")
PY
