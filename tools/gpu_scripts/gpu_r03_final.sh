#!/bin/bash
# Round-3 artefacts, one box: the driver's bench command (line with all side legs), rocprofv3 kernel stats of the same
# command, the PMC passes of the headline kernel (traffic, issue mix, LDS).  Progress goes to stdout step by step.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03final; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/trace.log 2>&1; echo "trace rc=$?"
grep -h '^{' $O/trace.log | tail -1 > $O/bench_under_rocprof.json
B="python3 bench.py --no-cpu --no-stage0-extra --no-legs --verify 32 --steps 4 --warmup 1"
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1; echo "p2 rc=$?"
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1; echo "p3 rc=$?"
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p4 -- $B > $O/p4.log 2>&1; echo "p4 rc=$?"
cat $O/trace/*/*kernel_stats.csv > $O/kernel_stats.csv; cat $O/kernel_stats.csv
python3 - $O > $O/pmc.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "nvx_fir_cascade" in r["Kernel_Name"]]
        last = max(int(r["Dispatch_Id"]) for r in rows)
        acc = collections.OrderedDict()
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for k, v in acc.items():
            print(f"{p} {k:24s} {v:.6g}")
PY
cat $O/pmc.txt
cut -c1-1500 $O/bench.json
