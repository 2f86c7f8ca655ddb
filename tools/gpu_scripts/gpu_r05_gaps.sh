#!/bin/bash
# Where does (step - cascade kernel) go?  Kernel trace with timestamps of a short headline run (no legs): for every
# cascade dispatch of the timed loop the idle time between the end of the previous cascade kernel and its own start,
# and what ran in between on the device.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/gaps; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --no-cpu --no-legs --no-stage0-extra --verify 32 --steps 10 --warmup 3 > $O/trace.log 2>&1; echo "trace rc=$?"
grep -h '^{' $O/trace.log | tail -1 | cut -c1-300
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
casc = [r for r in rows if "nvx_fir_cascade<true, 1>" in r["Kernel_Name"]]
print(len(casc), "cascade dispatches")
for a, b in zip(casc[-8:-1], casc[-7:]):
    ea, sb, eb = int(a["End_Timestamp"]), int(b["Start_Timestamp"]), int(b["End_Timestamp"])
    between = [r for r in rows if int(r["Start_Timestamp"]) >= ea - 2_000_000 and int(r["Start_Timestamp"]) < sb and r is not a]
    names = ", ".join(f"{r['Kernel_Name'][:24]}@{(int(r['Start_Timestamp'])-ea)/1e3:.0f}us+{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.0f}" for r in between[-6:])
    print(f"cascade {(eb-sb)/1e6:.3f} ms; start-to-start {(sb-int(a['Start_Timestamp']))/1e6:.3f} ms; idle gap before it {(sb-ea)/1e3:.1f} us; in between: {names}")
PY
rm -rf $O/t
