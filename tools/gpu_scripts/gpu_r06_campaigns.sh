#!/bin/bash
# Round-6 campaigns on the shipped code (the host-input path changed this round: StreamClose, ended streams skipped; per-entry
# sample counts, one demodulator stream): full-size soak runs with the seals' counters, ragged push-mode cases with and
# without ragged ends, random configurations in the three unit modes, the wideband soak.  Everything against the oracle or
# run to run; expected: 0 differences, 0 stale, 0 failed launches.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06camp; rm -rf $O; mkdir -p $O
cd $R
run() { local name=$1; shift; timeout -k 10 ${T:-500} "$@" > $O/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc: $(tail -1 $O/$name.log | cut -c1-300)"; [ $rc -ne 0 ] && { tail -20 $O/$name.log; exit $rc; }; }
run soak_units python3 tools/gpu_scripts/soak_units.py ${SOAK:-3000}
run soak_wideband python3 tools/gpu_scripts/soak_wideband.py ${WSOAK:-600}
run sweep_ragged python3 tools/gpu_scripts/sweep_ragged.py 5000 ${RAGGED:-400}
run sweep_ragged_tails python3 tools/gpu_scripts/sweep_ragged.py 9000 ${RAGGED:-400} --tails
run sweep_random python3 tools/gpu_scripts/sweep_random_configs.py 3000 ${RANDOM_N:-200}
NVX_INDEPENDENT=0 run sweep_random_handover python3 tools/gpu_scripts/sweep_random_configs.py 4000 ${RANDOM_N:-200}
NVX_INDEPENDENT=1 run sweep_random_independent python3 tools/gpu_scripts/sweep_random_configs.py 5000 ${RANDOM_N:-200}
run soak_long_run python3 tools/gpu_scripts/soak_long_run.py ${LONG:-150}
grep -ho "'stale_repaired': [0-9]*" $O/sweep_ragged*.log | sort | uniq -c
