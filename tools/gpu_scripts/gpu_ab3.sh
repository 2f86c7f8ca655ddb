#!/bin/bash
# A/B of kernel builds in the three bench modes (headline RAW, Variant A, wideband), two interleaved rounds.
#   gpu_ab3.sh LIB... ("-" = the product library)   -> gpurun_out/ab3.log
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/ab3.log; : > $L
for round in 1 2; do for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    for mode in "" "--variant-a --frames 96 --steps 5" "--wideband 512 --frames 12"; do
        echo "== $lib | ${mode:-headline}" >> $L
        timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --verify 32 $mode 2>>$R/gpurun_out/ab3.err >> $L || { echo FAILED >> $L; tail -5 $L; tail -5 $R/gpurun_out/ab3.err; exit 1; }
    done
done; done
python - <<PY
import json
tag=None
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        ch=j.get("channeliser") or {}
        print(f"{tag:60s} step {j['ms_per_step']:8.3f} ms  cascade {j['roofline'].get('avg_launch_ms')} ms  chan {ch.get('avg_launch_ms')}  parity {j.get('parity')}")
PY
