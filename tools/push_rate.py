#!/usr/bin/env python3
"""Host-fed (PCIe-inclusive) throughput of the push path: pinned staging ->
hipMemcpy2DAsync -> kernels -> bits.  Reported in DESIGN.md; never bench.py's value."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import navtex_amd as nv
import signals

def run(n_streams, raw, frames_per_push, pushes):
    frame = nv.FRAME_RAW if raw else nv.FRAME_IN
    rate = nv.RATE_RAW if raw else nv.RATE_IN
    st, _ = signals.stream_params(nv, 0, rate)
    block = nv.synth_host(st, rate, frames_per_push * frame)
    with nv.Pipeline(n_streams=n_streams, raw_rate=raw, chain_mask=nv.CHAIN_518, max_frames=frames_per_push, push_mode=True, char_layer=True) as p:
        for s in range(n_streams): p.push(s, block)          # warm-up
        p.flush()
        t0 = time.perf_counter()
        for _ in range(pushes):
            for s in range(n_streams): p.push(s, block)
        p.flush()
        dt = time.perf_counter() - t0
    n = n_streams * pushes * block.shape[0]
    print(f"streams {n_streams:4d} raw {int(raw)} frames/push {frames_per_push}: {n/dt/1e6:9.1f} Msamples/s  {4*n/dt/1e9:6.2f} GB/s host->device inclusive", flush=True)

if __name__ == "__main__":
    run(1, False, 1, 20)
    run(1, True, 1, 20)
    run(1, False, 24, 5)          # multi-frame pushes of one stream: the frames of a launch run in parallel
    run(1, True, 12, 5)
    run(8, True, 12, 4)
    run(64, True, 2, 6)
    run(256, True, 2, 3)
