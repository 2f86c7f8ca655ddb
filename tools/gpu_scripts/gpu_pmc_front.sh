#!/bin/bash
# PMC counters of nvx_demod_front (last dispatch) from two separate passes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/pmc_front; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python3 bench.py --no-cpu --steps 3 --warmup 1"
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1; echo "p2 rc=$?"
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for p in ("p1", "p2"):
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "nvx_demod_front" in r["Kernel_Name"]]
        last = max(int(r["Dispatch_Id"]) for r in rows)
        acc = collections.OrderedDict()
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for k, v in acc.items():
            print(f"{p} {k:24s} {v:.6g}")
PY
