#!/usr/bin/env python3
"""One-off soak of the fence-free unit hand-off: N full-size launches (4096 streams x 12 frames, from reset each
time), every stream's bits compared with the first launch's and a sample of streams with the oracle."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import numpy as np
import navtex_amd as nv, oracle_binding as ob, signals

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S, F = 4096, 12
pitch = F * nv.FRAME_RAW
buf = nv.DeviceBuffer(S * pitch * 4)
streams = [signals.stream_params(nv, s, nv.RATE_RAW)[0] for s in range(S)]
nv.synth_device(streams, nv.RATE_RAW, pitch, buf, pitch)
t0 = time.time()
bad = 0
with nv.Pipeline(n_streams=S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=F, char_layer=False) as p:
    first = None
    for i in range(N):
        p.reset()
        if i % 3 == 0:
            p.process_resident(buf, pitch, 0, F)
        elif i % 3 == 1:
            p.process_resident(buf, pitch, 0, 5); p.process_resident(buf, pitch, 5, 7)
        else:
            for f in range(F): p.process_resident(buf, pitch, f, 1)
        p.fetch()
        bits = [p.bits(s, 0) for s in range(S)]
        if first is None:
            first = bits
            for s in (0, 17, 2047, 4095):
                iq = buf.download(pitch * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)
                ref = ob.Pipe(chain_mask=1, charlayer=False); ref.push_raw(iq)
                assert bits[s] == ref.bits(0), f"stream {s} differs from the oracle"
        else:
            diff = sum(1 for a, b in zip(bits, first) if a != b)
            if diff: bad += 1; print(f"launch {i}: {diff} streams differ", flush=True)
        if i % 20 == 0: print(f"launch {i} ok so far, {time.time() - t0:.0f} s", flush=True)
print(f"done: {N} full-size runs (1 launch / 2 launches / 12 launches in turn), {bad} differing, {time.time() - t0:.1f} s")
