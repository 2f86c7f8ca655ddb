#!/bin/bash
# throughput against the number of streams in a launch (12 frames each)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
L=$R/gpurun_out/streams_curve.log; : > $L
for S in 64 256 512 1024 2048 2816 4096 8192; do
    F=12; [ $S -gt 4096 ] && F=6
    echo "== streams $S frames $F" >> $L
    timeout -k 10 300 python bench.py --streams $S --frames $F --steps 8 --warmup 2 --no-cpu --no-stage0-extra --verify 32 2>/dev/null >> $L || { echo FAILED >> $L; }
done
python - <<PY
import json
tag=None
for line in open("$L"):
    line=line.strip()
    if line.startswith("=="): tag=line[3:]
    elif line.startswith("{"):
        j=json.loads(line); r=j["roofline"]
        print(f"{tag:28s} step {j['ms_per_step']:8.3f} ms  cascade {r['avg_launch_ms']:8.3f} ms  {r['achieved']:7.1f} GB/s  frac {r['frac']:.3f}  value {j['value']/1e6:.3f} T/s parity {j.get('parity')}")
PY
