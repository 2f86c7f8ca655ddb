#!/bin/bash
# kernel trace with timestamps of a short bench run -> gpurun_out/trace/ (+ a gap summary)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/trace; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --no-cpu --steps 6 --warmup 2 > $O/trace.log 2>&1; echo "trace rc=$?"
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:40]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.append((name, (e - s) / 1e3, gap))
    prev_end = e
for name, dur, gap in out[-24:]:
    print(f"{name:42s} dur {dur:10.1f} us   gap before {gap:8.1f} us")
PY
