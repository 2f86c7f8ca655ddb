#!/usr/bin/env python3
"""One-off differential fuzz of the oracle's decoder restatement (oracle/nvx_oracle.c: nvxo_decode) against the
compiled reference decoder (oracle/_ref/ref_dec) on synthetic 900 S/s inputs of many shapes: the bit strings
must be identical.  Build container only.  usage: tools/fuzz_decoder.py [first_seed] [count]"""
import sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R / "tests"))
import numpy as np
import oracle_binding as ob

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(600, 9000))
    kind = seed % 6
    t = np.arange(n)
    if kind == 0:   y = rng.normal(size=(n, 2)) * float(rng.choice([1e-300, 1e-3, 1.0, 3000.0, 1e12]))
    elif kind == 1: # FSK at +-85 Hz with random bit timing, drifting baud rate and noise
        baud = 100.0 * (1 + rng.uniform(-3e-3, 3e-3)); bits = rng.integers(0, 2, n // 8 + 2)
        f = np.where(bits[np.minimum((t * baud / 900.0 + rng.uniform(0, 1)).astype(int), len(bits) - 1)] == 1, 85.0, -85.0)
        ph = np.cumsum(2 * np.pi * f / 900.0)
        y = np.stack([np.cos(ph), np.sin(ph)], 1) * 4000.0 + rng.normal(size=(n, 2)) * float(rng.choice([0.0, 50.0, 2000.0]))
    elif kind == 2: y = np.zeros((n, 2)); y[rng.integers(0, n, 20)] = rng.normal(size=(20, 2)) * 1e4   # mostly exact zeros
    elif kind == 3: y = np.round(rng.normal(size=(n, 2)) * 3.0)                                          # many ties / repeated values
    elif kind == 4: y = np.stack([np.cos(0.3 * t), np.sin(0.3 * t)], 1) * 1000.0                         # pure tone, constant delta-phi
    else:           y = rng.normal(size=(n, 2)) * np.exp(rng.normal(size=(n, 1)) * 8)                    # wild dynamic range
    y = np.ascontiguousarray(y, dtype=np.float64)
    r = ob.run_ref("dec", y.tobytes())
    got, _ = ob.decode(y)
    if got != r["bits518"].decode():
        bad += 1; print(f"seed {seed} (kind {kind}): DIFFERS", flush=True)
    if (seed - first) % 50 == 49: print(f"{seed - first + 1} cases, {bad} differing", flush=True)
print(f"done: {count} cases from seed {first}, {bad} differing")
