#!/bin/bash
# kernel timeline of a small configuration: gpu_trace_small.sh STREAMS FRAMES  (start, duration, queue, kernel of the last steps)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
export TMPDIR=/tmp
S=${1:-1}; F=${2:-62}
O=gpurun_out/trace_small; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --streams $S --frames $F --steps 12 --warmup 3 --no-cpu --no-legs --no-stage0-extra --verify 1 > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_small/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-40:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  q{r.get("Queue_Id")}  {r["Kernel_Name"][:48]}')
PY
