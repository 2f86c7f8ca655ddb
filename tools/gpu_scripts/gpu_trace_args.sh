#!/bin/bash
# kernel trace with timestamps of a short bench run with the given bench arguments -> gpurun_out/trace/ : every kernel of the
# last launches with its start relative to the first one shown, duration, and the stream-order picture (who overlaps whom)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/trace; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --no-cpu --verify 32 --no-legs --no-stage0-extra "$@" > $O/trace.log 2>&1; echo "trace rc=$?"
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nvx_" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-30:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:44]:46s} start {(s - t0) / 1e3:10.1f} us  end {(e - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us")
PY
grep '^{' $O/trace.log | tail -1 | cut -c1-400
