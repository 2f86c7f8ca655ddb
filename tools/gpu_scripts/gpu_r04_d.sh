#!/bin/bash
# Round 4: same-box A/B of FIR3 inside the wave (the previous commit's library) against FIR3 as its own kernel, both
# 252 kS/s kernel families; ROUNDS interleaved rounds, medians of the cascade kernel and of the step.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O
cd $R
ab() {  # tag, bench args...
    local tag=$1; shift
    for round in $(seq 1 ${ROUNDS:-4}); do for lib in tools/_bin/libnavtex_amd_f3in.so -; do
        if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
        echo "== $lib" >> $O/$tag.log
        timeout -k 10 300 python bench.py --no-cpu --verify 32 --no-legs --no-stage0-extra "$@" 2>/dev/null >> $O/$tag.log || { echo FAILED >> $O/$tag.log; tail -5 $O/$tag.log; exit 1; }
    done; done
    unset NAVTEX_AMD_LIB
    python - $O/$tag.log $tag <<'PY'
import json, sys, statistics, collections
tag=None; by=collections.OrderedDict()
for line in open(sys.argv[1]):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        by.setdefault(tag, []).append((j['roofline'].get('avg_launch_ms'), j['ms_per_step'], j.get('parity')))
for tag, v in by.items():
    print(f"{sys.argv[2]:10s} {tag:40s} cascade median {statistics.median(x[0] for x in v):.3f} (all {[x[0] for x in v]})  step median {statistics.median(x[1] for x in v):.3f} (all {[x[1] for x in v]}) parity {all(x[2] for x in v)}")
PY
}
ab variant_a --variant-a --frames 96 --steps 8 --warmup 2
ab wideband --wideband 512 --frames 12 --steps 12 --warmup 2
