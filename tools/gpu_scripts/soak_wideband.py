#!/usr/bin/env python3
"""One-off soak of the fused wideband kernel's workgroup hand-over: N launches of 512 wideband streams x 12 frames (8192
carriers; from reset each time, three launch partitions), every decoded stream's bits compared with the first launch's
and the first wideband streams with the oracle chain (channeliser restatement -> oracle pipeline)."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import numpy as np
import navtex_amd as nv, oracle_binding as ob, signals
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W, F = 512, 12
n = F * nv.FRAME_RAW
buf = nv.DeviceBuffer(W * n * 4)
nv.synth_device(bench.wideband_streams(nv, signals, 0, W), nv.RATE_RAW, n, buf, n)
t0 = time.time()
bad = 0
with nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=F, char_layer=False) as p:
    first = None
    for i in range(N):
        p.reset()
        if i % 3 == 0:
            p.process_resident(buf, n, 0, F)
        elif i % 3 == 1:
            p.process_resident(buf, n, 0, 5); p.process_resident(buf, n, 5, 7)
        else:
            for f in range(F): p.process_resident(buf, n, f, 1)
        p.fetch()
        bits = [p.bits(s, c) for s in range(8 * W) for c in (0, 1)]
        if first is None:
            first = bits
            nw = 4
            sample = buf.download(nw * n * 4, dtype=np.int16).reshape(nw, n, 2)
            _s, want = ob.bench_wide(sample, nw, n // 8, 16, want_bits=True)
            if bits[: 16 * nw] != want:
                bad += 1; print("FAIL: first launch differs from the oracle chain", flush=True)
        elif bits != first:
            bad += 1
            print(f"FAIL: launch {i} differs from launch 0 in {sum(a != b for a, b in zip(bits, first))} chains", flush=True)
        p._bits.clear()
        if i % 10 == 0:
            print(f"launch {i}: ok so far ({time.time() - t0:.0f} s)", flush=True)
    stale, failures, launches = p.integrity_stats()
print(f"done: {N} runs x {W * F} units (8 sub-bands each), {bad} failures, {time.time() - t0:.0f} s; seals (r4): nine blocks per hand-over, "
      f"{stale} workgroup hand-overs repaired, {failures} launches failed the check of their inherited state, {launches} launches")
sys.exit(1 if bad or stale or failures else 0)
