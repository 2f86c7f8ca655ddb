#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/pmc1; mkdir -p $O
export TMPDIR=/tmp
cd $R
rocprofv3 -L > $O/counters.txt 2>&1
B="python3 bench.py --no-cpu --steps 2 --warmup 1"
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1; echo "p2 rc=$?"
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1; echo "p3 rc=$?"
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p4 -- $B > $O/p4.log 2>&1; echo "p4 rc=$?"
find $O -name "*counter_collection.csv" | head
