#!/bin/bash
# gpu_env_a.sh "<bench args>" VAR=val ... : the same library under different environment switches, two rounds
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mode=$1; shift
for round in 1 2; do for kv in "$@"; do
    env $kv python bench.py --warmup 2 --no-cpu --verify 32 $mode 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$kv', j['ms_per_step'], j['roofline']['avg_launch_ms'], j['parity'])"
done; done
