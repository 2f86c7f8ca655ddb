#!/bin/bash
# round 3, 252 kS/s family: new boundary / rank tests, then Variant A A/B of read-ahead shapes and its PMC passes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python -m pytest tests/test_multirank.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_scripts/gpu_ab_a.sh "--variant-a --frames 96 --steps 5 --no-legs" - tools/_bin/libnavtex_amd_g8a2.so tools/_bin/libnavtex_amd_g2a6.so tools/_bin/libnavtex_amd_f23a6.so tools/_bin/libnavtex_amd_y2run80.so > $O/ab_variant_a.log 2>&1; echo "ab rc=$?"; cat $O/ab_variant_a.log
B="python3 bench.py --variant-a --frames 96 --no-cpu --no-legs --verify 32 --steps 3 --warmup 1"
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1; echo "p2 rc=$?"
python3 - $O > $O/pmc_variant_a.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for p in ("p1", "p2"):
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
cat $O/pmc_variant_a.txt
