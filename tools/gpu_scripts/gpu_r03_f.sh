#!/bin/bash
# A/B: mixer table in device memory instead of LDS (13 312 B of LDS per wave: 12 waves per CU?) -- headline and Variant A
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
bash tools/gpu_scripts/gpu_abn.sh - tools/_bin/libnavtex_amd_mixg.so -- --no-legs > $O/ab_headline.log 2>&1; echo "ab headline rc=$?"; cat $O/ab_headline.log
bash tools/gpu_scripts/gpu_ab_a.sh "--variant-a --frames 96 --steps 5 --no-legs" - tools/_bin/libnavtex_amd_mixg.so > $O/ab_variant_a.log 2>&1; echo "ab variant-a rc=$?"; grep -E "step" $O/ab_variant_a.log
export TMPDIR=/tmp
NAVTEX_AMD_LIB=$R/tools/_bin/libnavtex_amd_mixg.so timeout -k 10 280 rocprofv3 --pmc SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 bench.py --no-cpu --no-stage0-extra --no-legs --verify 32 --steps 3 --warmup 1 > $O/p1.log 2>&1; echo "p1 rc=$?"
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(f"{O}/p1/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "nvx_fir_cascade" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last and r["Counter_Name"] == "SQ_WAVES": print("SQ_WAVES", r["Counter_Value"], r.get("Grid_Size"), r.get("LDS_Block_Size"))
PY
