#!/usr/bin/env python3
"""More seeds of tests/test_gpu_independent_streams.py::ragged_case (random push-mode handles fed in random order with
silent streams; every chain against the oracle): sweep_ragged.py [first] [count].  Prints one line per case and a
summary by list kernel."""
import collections
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import navtex_amd as nv
import oracle_binding as ob
from test_gpu_independent_streams import ragged_case

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
by = collections.Counter(); partial = 0; t0 = time.time()
for seed in range(first, first + count):
    info = ragged_case(nv, ob, seed)
    by[(info["raw"], info["order"], info["two_chain_kernel"])] += 1
    partial += info["partial_launches"]
    print(info, flush=True)
print(f"{count} cases identical to the oracle in {time.time() - t0:.0f} s; partial launches {partial}; by (raw, stage-0 order, two-chain kernel): {dict(by)}")
