#!/usr/bin/env python3
"""One-off campaign on the third-order stage 0: sweep_cic3.py [first] [count] -- more seeds of
tests/test_stage0_cic3.py::test_randomized_configurations, then 200 launches of 3000 streams x 2 frames with carried
state (hand-over form, the two carried blocks travel through 400 unit boundaries per stream) against the oracle on a sample."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import numpy as np
import navtex_amd as nv, oracle_binding as ob, signals
import test_stage0_cic3 as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad, t0 = 0, time.time()
for seed in range(first, first + count):
    try:
        T.test_randomized_configurations.__wrapped__(nv, ob, seed) if hasattr(T.test_randomized_configurations, "__wrapped__") else T.test_randomized_configurations(nv, ob, seed)
    except AssertionError as e:
        bad += 1; print("FAIL seed", seed, str(e)[:200], flush=True)
print(f"random configurations: {count} seeds from {first}, {bad} failures, {time.time() - t0:.1f} s", flush=True)

S, F, N = 3000, 2, 200
streams = [signals.stream_params(nv, 30000 + s, nv.RATE_RAW)[0] for s in range(S)]
pitch = F * nv.FRAME_RAW
buf = nv.DeviceBuffer(S * pitch * 4)
nv.synth_device(streams, nv.RATE_RAW, pitch, buf, pitch)
check = [0, 1, 777, 1500, 2815, 2816, 2999]
t0 = time.time()
with nv.Pipeline(n_streams=S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=F, char_layer=False, stage0_order=3) as p:
    for i in range(N):
        p.process_resident(buf, pitch, 0, F)
    p.fetch()
    ok = True
    for s in check:
        iq = buf.download(pitch * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)
        ref = ob.Pipe(chain_mask=1, charlayer=False); ref.set_stage0(3)
        for i in range(N): ref.push_raw(iq)
        same = p.bits(s, 0) == ref.bits(0)[-len(p.bits(s, 0)):] and p.bit_count(s, 0) == len(ref.bits(0))
        ok &= same
        print(f"stream {s}: {p.bit_count(s, 0)} bits: {'identical' if same else 'DIFFERENT'}", flush=True)
print(f"carried state over {N} launches x {F} frames: {'ok' if ok else 'FAILED'}, {time.time() - t0:.0f} s")
sys.exit(0 if ok and not bad else 1)
