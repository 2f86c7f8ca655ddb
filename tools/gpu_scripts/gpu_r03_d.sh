#!/bin/bash
# round 3: fused wideband kernel A/B of the read-ahead shapes, group push rate, then the round's artefacts
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
bash tools/gpu_scripts/gpu_ab_a.sh "--wideband 512 --frames 12 --steps 8" - tools/_bin/libnavtex_amd_g8a2.so tools/_bin/libnavtex_amd_f23a6.so > $O/ab_wideband.log 2>&1; echo "ab wideband rc=$?"; grep -E "step" $O/ab_wideband.log
timeout -k 10 300 python tools/push_rate.py --group > $O/push_rate_group.txt 2>&1; echo "group push rc=$?"; cat $O/push_rate_group.txt
timeout -k 10 300 python tools/push_rate.py > $O/push_rate.txt 2>&1; echo "push rc=$?"; cat $O/push_rate.txt
bash tools/gpu_scripts/gpu_r03_final.sh
