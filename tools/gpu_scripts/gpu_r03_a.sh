#!/bin/bash
# round 3, first call: the new rank tests and the default bench line with its side legs
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_multirank.py tests/test_gpu_group.py -m gpu -x -q > $O/pytest_ranks.log 2>&1; rc=$?; echo "pytest ranks rc=$rc"; tail -3 $O/pytest_ranks.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"
tail -5 $O/bench.err
python - <<'PY'
import json,sys,os
p=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03a/bench.json"
try:
    r=json.loads(open(p).read().splitlines()[-1])
except Exception as e:
    print("no line", e); sys.exit(1)
print("value", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"], "parity", r["parity"])
print("cpu", {k:v for k,v in r["cpu_baseline"].items() if k!="reference_check"})
for k in ("stage0_third_order","push_path","variant_a","wideband","ranks"):
    print(k, json.dumps(r.get(k))[:900])
PY
