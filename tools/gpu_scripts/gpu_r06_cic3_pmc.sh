#!/bin/bash
# Round 6, measurement only: the issue / wait / LDS counters of the third-order stage 0's kernel (nvx_fir_cascade_cic3_1)
# beside the headline kernel's on the same box (gpu_r06_final.sh collects only its traffic).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06cic3; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python3 bench.py --no-cpu --no-stage0-extra --no-legs --verify 32 --steps 4 --warmup 1"
pmc() { local d=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
    timeout -k 10 280 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $O/$d -- $B "$@" > $O/$d.log 2>&1; echo "$d rc=$?"; }
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT"
pmc h1 $P1 --; pmc h2 $P2 --; pmc h3 $P3 --
pmc c1 $P1 -- --stage0 cic3; pmc c2 $P2 -- --stage0 cic3; pmc c3 $P3 -- --stage0 cic3
python3 - $O > $O/pmc.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for p in ("h1", "h2", "h3", "c1", "c2", "c3"):
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        for k in sorted({r["Kernel_Name"] for r in rows if "nvx_fir_cascade" in r["Kernel_Name"]}):
            kr = [r for r in rows if r["Kernel_Name"] == k]
            last = max(int(r["Dispatch_Id"]) for r in kr)
            acc = collections.OrderedDict()
            for r in kr:
                if int(r["Dispatch_Id"]) == last:
                    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            for c, v in acc.items():
                print(f"{p} {k[:58]:58s} {c:24s} {v:.6g}")
PY
cat $O/pmc.txt
grep -h '"avg_launch_ms"' $O/h1.log $O/c1.log | grep -o '"avg_launch_ms": [0-9.]*'
rm -rf $O/h? $O/c?
