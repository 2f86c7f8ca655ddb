#!/bin/bash
# Variant A (4096 streams x 96 frames at 252 kS/s) and the wideband kernel, A/B of library builds on one box:
# gpu_r05_va_ab.sh LIB... ; ROUNDS interleaved rounds; timing-only probes give wrong bits (exit status 3 is accepted).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/va_ab.log; : > $L
cd $R
for round in $(seq 1 ${ROUNDS:-3}); do for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    echo "== $lib variant_a" >> $L
    timeout -k 10 300 python3 bench.py --variant-a --frames 96 --steps 5 --warmup 1 --no-cpu --verify 16 2>/dev/null >> $L; rc=$?
    [ $rc -ne 0 ] && [ $rc -ne 3 ] && { echo "FAILED rc=$rc" >> $L; tail -5 $L; exit 1; }
    echo "== $lib wideband" >> $L
    timeout -k 10 200 python3 bench.py --wideband 512 --frames 12 --steps 10 --warmup 2 --no-cpu --verify 16 2>/dev/null >> $L; rc=$?
    [ $rc -ne 0 ] && [ $rc -ne 3 ] && { echo "FAILED rc=$rc" >> $L; tail -5 $L; exit 1; }
done; done
python3 - <<PY
import json, statistics, collections
tag=None; by=collections.OrderedDict()
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        by.setdefault(tag, []).append(j['roofline'].get('avg_launch_ms'))
for tag, v in by.items():
    print(f"{tag:60s} kernel ms {v}  median {statistics.median(v):.3f}")
PY
