#!/usr/bin/env python3
"""One-off wider sweep of tests/test_gpu_parity.py::test_randomized_configurations: sweep_random_configs.py [first] [count]"""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import navtex_amd as nv, oracle_binding as ob
import test_gpu_parity as T
bad = 0
t0 = time.time()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(first, first + count):
    try:
        T.test_randomized_configurations(nv, ob, seed)
    except AssertionError as e:
        bad += 1; print("FAIL seed", seed, str(e)[:200], flush=True)
print(f"done: {count} seeds from {first}, {bad} failures, {time.time() - t0:.1f} s")
