#!/usr/bin/env python3
"""More seeds of tests/test_gpu_independent_streams.py::ragged_case (random push-mode handles fed in random order with
silent streams; every chain against the oracle): sweep_ragged.py [first] [count] [--tails].  Prints one line per case and a
summary by list kernel.  --tails: every stream's input ends with a ragged tail of its own and the handle is ended by
nvx_finish (the last, partial frames at their true lengths)."""
import collections
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import navtex_amd as nv
import oracle_binding as ob
from test_gpu_independent_streams import ragged_case

tails = "--tails" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--tails"]
first = int(argv[0]) if len(argv) > 0 else 100
count = int(argv[1]) if len(argv) > 1 else 60
by = collections.Counter(); partial = 0; t0 = time.time()
for seed in range(first, first + count):
    info = ragged_case(nv, ob, seed, tails=tails)
    by[(info["raw"], info["order"], info["two_chain_kernel"])] += 1
    partial += info["partial_launches"]
    print(info, flush=True)
print(f"{count} cases{' with ragged ends (nvx_finish)' if tails else ''} identical to the oracle in {time.time() - t0:.0f} s; partial launches {partial}; by (raw, stage-0 order, two-chain kernel): {dict(by)}")
