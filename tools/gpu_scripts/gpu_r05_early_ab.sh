#!/bin/bash
# early_cascade (the next cascade launch follows its predecessor with nothing in between; the control block is cleared on
# the demodulator's stream): same-box A/B against the old ordering (NVX_EARLY_CASCADE=0), ROUNDS interleaved rounds of the
# headline workload with the third-order leg; then the kernel-trace gaps.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; mkdir -p gpurun_out; L=gpurun_out/early_ab.txt; : > $L
for r in $(seq 1 ${ROUNDS:-6}); do for m in 0 1; do
    NVX_EARLY_CASCADE=$m python3 bench.py --no-cpu --no-legs --verify 32 --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); t=j['stage0_third_order']; print($m, j['ms_per_step'], j['roofline']['avg_launch_ms'], t['ms_per_step'], t['cascade_avg_launch_ms'], j['parity'], t['parity'], j['roofline']['handoff']['units_waited_frac'])" >> $L
done; done
cat $L
python3 - <<PY
import statistics
rows=[l.split() for l in open("$L")]
for m in ("0","1"):
    R=[r for r in rows if r[0]==m]
    print("early",m,"step",statistics.median(float(r[1]) for r in R),"kernel",statistics.median(float(r[2]) for r in R),"cic3 step",statistics.median(float(r[3]) for r in R),"cic3 kernel",statistics.median(float(r[4]) for r in R),"n",len(R), all(r[5]=="True" and r[6]=="True" for r in rows))
PY
bash tools/gpu_scripts/gpu_r05_gaps.sh 2>&1 | tail -3 | cut -c1-330
