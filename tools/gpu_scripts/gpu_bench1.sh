#!/bin/bash
mkdir -p gpurun_out
echo "== small ==" > gpurun_out/bench1.log
timeout -k 10 300 python bench.py --streams 256 --frames 2 --steps 3 --warmup 1 >> gpurun_out/bench1.log 2>&1 || { echo "small failed rc=$?" >> gpurun_out/bench1.log; tail -30 gpurun_out/bench1.log; exit 1; }
echo "== full ==" >> gpurun_out/bench1.log
timeout -k 10 600 python bench.py >> gpurun_out/bench1.log 2>&1
echo "rc=$?" >> gpurun_out/bench1.log
tail -20 gpurun_out/bench1.log
