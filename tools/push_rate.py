#!/usr/bin/env python3
"""Host-fed (PCIe-inclusive) throughput of the push path: pinned staging ->
hipMemcpy2DAsync -> kernels -> bits.  Reported in profiles/TUNING.md; never bench.py's value."""
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

def run_threads(n_streams, n_threads, frames_per_push, pushes):
    """ONE handle fed by several threads, each owning a share of the streams: big pushes copy into the pinned staging
    without the handle's lock (nvx_push.cpp), so the threads' copies overlap."""
    import threading
    st, _ = signals.stream_params(nv, 0, nv.RATE_RAW)
    block = nv.synth_host(st, nv.RATE_RAW, frames_per_push * nv.FRAME_RAW)
    with nv.Pipeline(n_streams=n_streams, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=frames_per_push, push_mode=True, char_layer=True) as p:
        def feed(t, rounds):
            for _ in range(rounds):
                for s in range(t, n_streams, n_threads): p.push(s, block)
        for t in range(n_threads): feed(t, 1)
        p.flush()
        threads = [threading.Thread(target=feed, args=(t, pushes)) for t in range(n_threads)]
        t0 = time.perf_counter()
        for t in threads: t.start()
        for t in threads: t.join()
        p.flush()
        dt = time.perf_counter() - t0
    n = n_streams * pushes * block.shape[0]
    print(f"one handle, {n_streams} streams, {n_threads} pusher thread(s), {frames_per_push} frames per push: "
          f"{n/dt/1e6:9.1f} Msamples/s  {4*n/dt/1e9:6.2f} GB/s host->device inclusive", flush=True)


def run_group(n_members, streams_per_member, frames_per_push, pushes):
    """One capture thread per member of a group, each pushing into its own member's streams (ctypes drops the GIL inside
    nvx_group_push_iq; the group holds no lock around it): the aggregate host-fed rate against one member alone."""
    import threading
    st, _ = signals.stream_params(nv, 0, nv.RATE_RAW)
    block = nv.synth_host(st, nv.RATE_RAW, frames_per_push * nv.FRAME_RAW)
    S = n_members * streams_per_member
    with nv.Group([0] * n_members, n_streams=S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=frames_per_push, push_mode=True) as g:
        def feed(m, rounds):
            for _ in range(rounds):
                for s in range(m * streams_per_member, (m + 1) * streams_per_member):
                    g.push(s, block)
        for m in range(n_members): feed(m, 1)                    # warm-up
        g.flush()
        threads = [threading.Thread(target=feed, args=(m, pushes)) for m in range(n_members)]
        t0 = time.perf_counter()
        for t in threads: t.start()
        for t in threads: t.join()
        g.flush()
        dt = time.perf_counter() - t0
    n = S * pushes * block.shape[0]
    print(f"group of {n_members} member(s) on device 0, {streams_per_member} streams each, one pusher thread per member: "
          f"{n/dt/1e6:9.1f} Msamples/s  {4*n/dt/1e9:6.2f} GB/s host->device inclusive", flush=True)
    return 4 * n / dt / 1e9


if __name__ == "__main__":
    if "--threads" in sys.argv:
        for t in (1, 2, 4, 8):
            run_threads(64, t, 2, 6)
        sys.exit(0)
    if "--group" in sys.argv:
        one = run_group(1, 32, 2, 6)
        two = run_group(2, 32, 2, 6)
        four = run_group(4, 32, 2, 6)
        print(f"aggregate of two members / one member: {two / one:.2f}; four: {four / one:.2f}")
        sys.exit(0)
    run(1, False, 1, 20)
    run(1, True, 1, 20)
    run(1, False, 24, 5)          # multi-frame pushes of one stream: the frames of a launch run in parallel
    run(1, True, 12, 5)
    run(8, True, 12, 4)
    run(64, True, 2, 6)
    run(256, True, 2, 3)
