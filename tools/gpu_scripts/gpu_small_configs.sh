#!/bin/bash
# BASELINE configs[1] / [2] shapes through bench.py: S channels x 62 frames (19.8 s of signal per launch), S = 1 3 16 64
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/small; rm -rf $O; mkdir -p $O
cd $R
for S in 1 3 16 64; do
    timeout -k 10 300 python3 bench.py --streams $S --frames 62 --steps 200 --warmup 10 --no-cpu --no-legs --no-stage0-extra > $O/s$S.json 2> $O/s$S.err || { echo "S=$S failed"; tail -3 $O/s$S.err; exit 1; }
    python3 - $O/s$S.json $S <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(f"{int(sys.argv[2]):3d} channels x 62 frames: step {j['ms_per_step']:.3f} ms  cascade {j['roofline']['avg_launch_ms']:.3f} ms  {j['value'] / 1e3:.1f} G samples/s  ({j['value'] * 1e6 / 2016000 / int(sys.argv[2]):.0f} x real time per channel)  parity {j['parity']} after timed {j['parity_after_timed']} ({j['parity_after_timed_streams']} streams, {j['parity_after_timed_launches']} launches)  stale {j['roofline']['handoff']['stale_detected']}")
PY
done
