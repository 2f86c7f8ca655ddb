#!/bin/bash
# Which kernels does the GPU suite launch?  `pytest -m gpu` under rocprofv3 --kernel-trace --stats (the test processes
# and the children they start all write their own stats file); the union of kernel names with their call counts goes to
# gpurun_out/r05suite/suite_kernels.txt -> profiles/r05/suite_kernels.txt, which tests/test_isa.py holds against the list
# of kernels the library contains.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05suite; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 -m pytest tests -q -m gpu -x > $O/suite.log 2>&1; rc=$?
echo "suite under rocprofv3 rc=$rc"; tail -3 $O/suite.log
python3 - $O > $O/suite_kernels.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
calls = collections.Counter(); files = 0
for f in glob.glob(f"{O}/trace/**/*kernel_stats.csv", recursive=True):
    files += 1
    for r in csv.DictReader(open(f)):
        calls[r["Name"]] += int(r["Calls"])
print(f"# kernels launched while `python3 -m pytest tests -q -m gpu -x` ran under rocprofv3 --kernel-trace --stats ({files} processes wrote a stats file); name, calls")
for k in sorted(calls):
    if "nvx_" in k:
        print(f"{k}    # {calls[k]}")
PY
cat $O/suite_kernels.txt
rm -rf $O/trace
exit $rc
