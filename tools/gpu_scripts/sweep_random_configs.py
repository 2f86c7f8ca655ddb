#!/usr/bin/env python3
"""One-off wider sweep of tests/test_gpu_parity.py::test_randomized_configurations (more seeds)."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import navtex_amd as nv, oracle_binding as ob
import test_gpu_parity as T
bad = 0
t0 = time.time()
for seed in range(100, 140):
    try:
        T.test_randomized_configurations(nv, ob, seed)
    except AssertionError as e:
        bad += 1; print("FAIL seed", seed, str(e)[:200], flush=True)
print(f"done: 40 seeds, {bad} failures, {time.time() - t0:.1f} s")
