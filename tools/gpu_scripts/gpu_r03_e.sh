#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_independent_streams.py tests/test_gpu_boundary.py tests/test_gpu_group.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/push_rate.py --threads > $O/push_rate_threads.txt 2>&1; echo "threads rc=$?"; cat $O/push_rate_threads.txt
timeout -k 10 300 python tools/push_rate.py --group > $O/push_rate_group.txt 2>&1; echo "group rc=$?"; cat $O/push_rate_group.txt
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json; r=json.load(open('$O/bench.json')); print('step', r['ms_per_step'], 'cascade', r['roofline']['avg_launch_ms'], 'parity', r['parity']); print(json.dumps(r['push_path']))"
