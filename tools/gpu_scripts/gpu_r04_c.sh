#!/bin/bash
# Round 4, FIR3 as its own kernel behind the fused wideband kernel: the tests that touch the wideband paths and the seals,
# then same-box A/B against the previous commit's library (FIR3 inside the wave), ROUNDS interleaved rounds.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_integrity.py tests/test_wideband.py tests/test_gpu_group.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
for round in $(seq 1 ${ROUNDS:-6}); do for lib in tools/_bin/libnavtex_amd_f3in.so -; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    echo "== $lib" >> $O/wideband.log
    timeout -k 10 300 python bench.py --no-cpu --verify 32 --wideband 512 --frames 12 --steps 20 --warmup 3 2>/dev/null >> $O/wideband.log || { echo FAILED; exit 1; }
done; done
unset NAVTEX_AMD_LIB
python - $O/wideband.log <<'PY'
import json, sys, statistics, collections
tag=None; by=collections.OrderedDict()
for line in open(sys.argv[1]):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line)
        by.setdefault(tag, []).append((j['roofline'].get('avg_launch_ms'), j['ms_per_step'], j.get('parity'), j['roofline'].get('fir3_avg_launch_ms')))
for tag, v in by.items():
    print(f"wideband 512 x 12  {tag[-36:]:36s} fused kernel median {statistics.median(x[0] for x in v):.3f} (all {[x[0] for x in v]})  step median {statistics.median(x[1] for x in v):.3f} min {min(x[1] for x in v):.3f} (all {[x[1] for x in v]})  nvx_fir3 {[x[3] for x in v]}  parity {all(x[2] for x in v)}")
PY
