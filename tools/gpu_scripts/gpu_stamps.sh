#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
NAVTEX_AMD_LIB=$R/tools/_bin/libnavtex_amd_probe9.so python bench.py --variant-a --frames 12 --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/stamps.log 2>&1
grep "^PH" $R/gpurun_out/stamps.log | python -c "
import sys,re
rows=[list(map(int,re.findall(r'(?<= )\d+(?= |$)', l.split(':',1)[1]))) for l in sys.stdin]
n=len(rows); print('waves printed', n)
names=['loop','inwait','input','fir1','mix','fir23']
tot=[sum(r[i] for r in rows)/n/315 for i in range(6)]
for nm,t in zip(names,tot): print(f'{nm:8s} {t:8.1f} cycles per pass per wave')
print('sum', sum(tot))
"
grep -v "^PH" $R/gpurun_out/stamps.log | tail -2 | cut -c1-300
