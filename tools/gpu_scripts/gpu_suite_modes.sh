#!/bin/bash
# the -m gpu suite three times: automatic unit mode, dependent units forced, independent units forced
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
for mode in auto 0 1; do
    L=$R/gpurun_out/suite_$mode.log
    if [ $mode = auto ]; then unset NVX_INDEPENDENT; else export NVX_INDEPENDENT=$mode; fi
    timeout -k 10 900 python -m pytest tests -x -q -m gpu > $L 2>&1; rc=$?
    echo "== NVX_INDEPENDENT=$mode rc=$rc: $(tail -1 $L)"
    [ $rc -ne 0 ] && { tail -40 $L; exit $rc; }
done
exit 0
