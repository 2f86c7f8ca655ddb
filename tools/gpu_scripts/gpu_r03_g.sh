#!/bin/bash
# the demodulator front's tile-parallel form: whole -m gpu suite in the automatic mode and with either form forced,
# then the small BASELINE configs through bench.py with the form off / on
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
for mode in auto 1 0; do
  if [ $mode = auto ]; then unset NVX_DEMOD_TILES; else export NVX_DEMOD_TILES=$mode; fi
  timeout -k 10 500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_independent_streams.py::test_a_paused_capture_ring_does_not_hold_the_other > $O/suite_tiles_$mode.log 2>&1; rc=$?
  echo "suite NVX_DEMOD_TILES=$mode rc=$rc: $(tail -1 $O/suite_tiles_$mode.log)"
  [ $rc -ne 0 ] && { tail -30 $O/suite_tiles_$mode.log; exit $rc; }
done
for mode in 0 1; do
  export NVX_DEMOD_TILES=$mode
  for S in 1 3 16; do
    timeout -k 10 200 python3 bench.py --streams $S --frames 62 --steps 40 --warmup 5 --no-cpu --no-legs --no-stage0-extra 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read().splitlines()[-1]); print('NVX_DEMOD_TILES=$mode streams $S frames 62: step', r['ms_per_step'], 'ms  cascade', r['roofline']['avg_launch_ms'], ' demod span', r['roofline']['demod_span_ms'], ' value', r['value'], 'parity', r['parity'])"
  done
done | tee $O/small_configs.txt
