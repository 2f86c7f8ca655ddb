#!/bin/bash
# Round 4: the one-off campaigns of the earlier rounds again, on the shipped code (sealed state blocks, 272-entry blocks,
# FIR3 outside the fused wideband kernel): random configurations, the third-order stage 0, the long carry without reset.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04g; rm -rf $O; mkdir -p $O
cd $R
run() { local name=$1; shift; timeout -k 10 900 "$@" > $O/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -2 $O/$name.log; return $rc; }
run random_configs python tools/gpu_scripts/sweep_random_configs.py 3000 300 || exit 1
run random_configs_handover env NVX_INDEPENDENT=0 python tools/gpu_scripts/sweep_random_configs.py 4000 150 || exit 1
run random_configs_independent env NVX_INDEPENDENT=1 python tools/gpu_scripts/sweep_random_configs.py 5000 150 || exit 1
run cic3 python tools/gpu_scripts/sweep_cic3.py 500 60 || exit 1
run long_run python tools/gpu_scripts/soak_long_run.py 600 || exit 1
