#!/bin/bash
# Round-6 artefacts, one box: the driver's bench command (line with all side legs), rocprofv3 kernel stats of the same
# command, PMC passes (separate, as the guide prescribes) of the headline kernel, of the third-order stage 0's kernel
# (traffic: FETCH_SIZE / WRITE_SIZE) and of the fused wideband kernel + nvx_fir3.  Progress goes to stdout step by step.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06final; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/trace.log 2>&1; echo "trace rc=$?"
grep -h '^{' $O/trace.log | tail -1 > $O/bench_under_rocprof.json
cat $O/trace/*/*kernel_stats.csv > $O/kernel_stats.csv
B="python3 bench.py --no-cpu --no-stage0-extra --no-legs --verify 32 --steps 4 --warmup 1"
pmc() {  # dir, counters..., then -- bench args
    local d=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
    timeout -k 10 280 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $O/$d -- $B "$@" > $O/$d.log 2>&1; echo "$d rc=$?"
}
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_LDS"
pmc h1 $P1 --; pmc h2 $P2 --; pmc h3 FETCH_SIZE --; pmc h4 WRITE_SIZE --
pmc c3 FETCH_SIZE -- --stage0 cic3; pmc c4 WRITE_SIZE -- --stage0 cic3
pmc w1 $P1 -- --wideband 512 --frames 12; pmc w2 $P2 -- --wideband 512 --frames 12; pmc w3 FETCH_SIZE -- --wideband 512 --frames 12; pmc w4 WRITE_SIZE -- --wideband 512 --frames 12
python3 - $O > $O/pmc.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
import importlib.util
spec = importlib.util.spec_from_file_location("bench_for_hash", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print("# kernel sources", b.kernel_source_hash(), "(bench.py: KERNEL_SOURCES)")
print("rocprofv3 --pmc, separate passes; bench.py --no-cpu --no-stage0-extra --no-legs --verify 32 --steps 4 --warmup 1 [+ --stage0 cic3 | --wideband 512 --frames 12]; per kernel: LAST dispatch; SQ_* cycle counters are quad-cycles")
for p in ("h1", "h2", "h3", "h4", "c3", "c4", "w1", "w2", "w3", "w4"):
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        for k in sorted({r["Kernel_Name"] for r in rows if "nvx_" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"] and "demod" not in r["Kernel_Name"]}):
            kr = [r for r in rows if r["Kernel_Name"] == k]
            last = max(int(r["Dispatch_Id"]) for r in kr)
            acc = collections.OrderedDict()
            for r in kr:
                if int(r["Dispatch_Id"]) == last:
                    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            for c, v in acc.items():
                print(f"{p} {k[:58]:58s} {c:22s} {v:.6g}")
PY
cat $O/pmc.txt
rm -rf $O/trace $O/h? $O/c? $O/w?
cat $O/kernel_stats.csv | head -14
cut -c1-1200 $O/bench.json
