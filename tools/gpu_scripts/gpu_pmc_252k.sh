#!/bin/bash
# PMC counters + kernel-trace stats of the 252 kS/s cascade kernels (Variant A = nvx_fir_cascade<false,1>, wideband =
# <false,2> or the fused wideband kernel), separate passes as the guide prescribes.
#   gpu_pmc_252k.sh TAG [a|w|aw]      -> gpurun_out/pmc252_TAG/{a,w}_{stats.csv,pmc.txt,bench.json}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-now}; WHICH=${2:-aw}
O=$R/gpurun_out/pmc252_$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
run_one() {   # name, bench args
    local n=$1; shift
    local B="python3 bench.py --no-cpu --verify 32 --steps 3 --warmup 1 $*"
    timeout -k 10 300 python3 bench.py --no-cpu --verify 32 --steps 6 --warmup 1 "$@" > $O/${n}_bench.json 2> $O/${n}_bench.err || { echo "$n bench failed"; tail -5 $O/${n}_bench.err; return 1; }
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${n}_kt -- $B > $O/${n}_kt.log 2>&1 || { echo "$n kernel-trace failed"; return 1; }
    cp $(find $O/${n}_kt -name '*kernel_stats.csv' | head -1) $O/${n}_stats.csv
    timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/${n}_p1 -- $B > $O/${n}_p1.log 2>&1 || { echo "$n p1 failed"; return 1; }
    timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/${n}_p2 -- $B > $O/${n}_p2.log 2>&1 || { echo "$n p2 failed"; return 1; }
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${n}_p3 -- $B > $O/${n}_p3.log 2>&1 || { echo "$n p3 failed"; return 1; }
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${n}_p4 -- $B > $O/${n}_p4.log 2>&1 || { echo "$n p4 failed"; return 1; }
    python3 - $O $n "$B" > $O/${n}_pmc.txt <<'PY'
import csv, glob, sys, collections
O, n, B = sys.argv[1:4]
print(f"rocprofv3 PMC, {B}; separate passes; per kernel: LAST dispatch; SQ_* cycle counters are quad-cycles")
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(f"{O}/{n}_{p}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        kernels = sorted({r["Kernel_Name"] for r in rows if "nvx_" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]})
        for k in kernels:
            kr = [r for r in rows if r["Kernel_Name"] == k]
            last = max(int(r["Dispatch_Id"]) for r in kr)
            acc = collections.OrderedDict()
            for r in kr:
                if int(r["Dispatch_Id"]) == last:
                    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            for c, v in acc.items():
                print(f"{p} {k[:60]:60s} {c:24s} {v:.6g}")
PY
    grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"parity": [a-z]*' $O/${n}_bench.json | paste -sd' '
}
case $WHICH in *a*) run_one a --variant-a --frames 96 || exit 1;; esac
case $WHICH in *w*) run_one w --wideband 512 --frames 12 || exit 1;; esac
rm -rf $O/*_kt $O/*_p1 $O/*_p2 $O/*_p3 $O/*_p4
ls $O
