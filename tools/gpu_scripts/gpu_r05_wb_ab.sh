#!/bin/bash
# The fused wideband kernel, A/B/.../N of library builds on one box: gpu_r05_wb_ab.sh LIB... ("-" = the product);
# ROUNDS (default 4) interleaved rounds of `bench.py --wideband 512 --frames 12`; kernel average, step, parity per run; medians.
# (Elimination probes produce wrong bits on purpose: parity false, exit status 3 -- the line is still taken.)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; L=$R/gpurun_out/wb_ab.log; : > $L
cd $R
for round in $(seq 1 ${ROUNDS:-4}); do for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    echo "== $lib" >> $L
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
        k=j['roofline'].get('avg_launch_ms')
        by.setdefault(tag, []).append(k)
        print(f"{tag:48s} step {j['ms_per_step']:.3f} ms  kernel {k} ms  fir3 {j['roofline'].get('fir3_avg_launch_ms')}  frac {j['roofline']['frac']}  parity {j.get('parity')}")
for tag, v in by.items():
    print(f"{tag:48s} kernel median {statistics.median(v):.3f}  min {min(v):.3f}  mean {statistics.mean(v):.3f}  n {len(v)}")
PY
