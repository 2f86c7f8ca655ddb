#!/bin/bash
# per-kernel average durations (rocprofv3 --stats) of kernel builds on one box: gpu_kstats_ab.sh LIB...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
cd $R
for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    O=$R/gpurun_out/kstats_$(basename $lib .so); rm -rf $O; mkdir -p $O
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-cpu --steps 8 --warmup 2 > $O/log.txt 2>&1 || { echo FAILED; tail -5 $O/log.txt; exit 1; }
    echo "== $lib"; cut -d, -f1,2,4 $O/*/*kernel_stats.csv | grep "nvx_"
done
