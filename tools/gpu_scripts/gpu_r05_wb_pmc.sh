#!/bin/bash
# PMC counters of the fused wideband kernel for a set of library builds: gpu_r05_wb_pmc.sh TAG=LIB ...   ("-" = the product)
# Two counter passes per build (separate rocprofv3 runs, as the guide prescribes); per kernel the LAST dispatch.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05wbpmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python3 bench.py --no-cpu --no-legs --verify 16 --steps 4 --warmup 1 --wideband 512 --frames 12"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_LDS"
for spec in "$@"; do
    tag=${spec%%=*}; lib=${spec#*=}
    if [ "$lib" = "-" ]; then unset NAVTEX_AMD_LIB; else export NAVTEX_AMD_LIB=$R/$lib; fi
    timeout -k 10 200 rocprofv3 --pmc $P1 --output-format csv -d $O/${tag}_1 -- $B > $O/${tag}_1.log 2>&1; echo "$tag pass 1 rc=$?"
    timeout -k 10 200 rocprofv3 --pmc $P2 --output-format csv -d $O/${tag}_2 -- $B > $O/${tag}_2.log 2>&1; echo "$tag pass 2 rc=$?"
done
python3 - $O "$@" > $O/pmc.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
print("rocprofv3 --pmc, two separate passes per build; bench.py --no-cpu --no-legs --verify 16 --steps 4 --warmup 1 --wideband 512 --frames 12; per kernel: LAST dispatch; SQ_* cycle counters are quad-cycles")
for spec in sys.argv[2:]:
    tag = spec.split("=")[0]
    for p in ("1", "2"):
        for f in glob.glob(f"{O}/{tag}_{p}/**/*counter_collection.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            for k in sorted({r["Kernel_Name"] for r in rows if "nvx_wideband_fused" in r["Kernel_Name"] or "nvx_fir3" in r["Kernel_Name"]}):
                kr = [r for r in rows if r["Kernel_Name"] == k]
                last = max(int(r["Dispatch_Id"]) for r in kr)
                acc = collections.OrderedDict()
                for r in kr:
                    if int(r["Dispatch_Id"]) == last:
                        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                for c, v in acc.items():
                    print(f"{tag:10s} {k[:44]:44s} {c:22s} {v:.6g}")
PY
cat $O/pmc.txt
rm -rf $O/*_1 $O/*_2
