#!/usr/bin/env python3
"""One-off endurance run (soak_long_run.py [N] [streams] [frames]): the batch (default 4096 streams x 12 frames) fed N times in a row WITHOUT reset
(38 minutes of signal per stream at N = 600, state carried through every launch), then four streams compared
with the oracle fed the same N repetitions: total bit count and the retained bit history must be identical."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import numpy as np
import navtex_amd as nv, oracle_binding as ob, signals
from concurrent.futures import ThreadPoolExecutor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096       # few streams -> the independent-unit form of the cascade
F = int(sys.argv[3]) if len(sys.argv) > 3 else 12
pitch = F * nv.FRAME_RAW
buf = nv.DeviceBuffer(S * pitch * 4)
streams = [signals.stream_params(nv, s, nv.RATE_RAW)[0] for s in range(S)]
nv.synth_device(streams, nv.RATE_RAW, pitch, buf, pitch)
check = sorted({0, S // 3, (2 * S) // 3, S - 1})
t0 = time.time()
with nv.Pipeline(n_streams=S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=F, char_layer=True) as p:
    for i in range(N):
        p.process_resident(buf, pitch, 0, F)
        if i % 50 == 49: p.fetch(); print(f"launch {i + 1}, {time.time() - t0:.0f} s", flush=True)
    p.fetch()
    polls, waited, launches = p.wait_stats()
    counts = {s: p.bit_count(s, 0) for s in check}
    tails = {s: p.bits(s, 0) for s in check}
    n_msgs = len(p.messages)
print(f"GPU: {N} launches in {time.time() - t0:.1f} s, {n_msgs} messages; hand-over: {waited} of {launches * S * F} units waited, {polls} polls")

def oracle_run(s):
    iq = buf.download(pitch * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)
    o = ob.Pipe(chain_mask=1, charlayer=False)
    for _ in range(N): o.push_raw(iq)
    return o.bits(0)

iqs = {s: buf.download(pitch * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2).copy() for s in check}
def oracle_bits(s):
    o = ob.Pipe(chain_mask=1, charlayer=False)
    for _ in range(N): o.push_raw(iqs[s])
    return o.bits(0)
t1 = time.time()
with ThreadPoolExecutor(4) as ex:
    want = dict(zip(check, ex.map(oracle_bits, check)))
bad = 0
for s in check:
    ok = counts[s] == len(want[s]) and want[s].endswith(tails[s]) and len(tails[s]) >= min(65536, counts[s])
    print(f"stream {s}: {counts[s]} bits, history {len(tails[s])}: {'identical' if ok else 'DIFFERS'}")
    bad += not ok
print(f"oracle: {time.time() - t1:.0f} s; {'ok' if not bad else 'FAILED'}")
sys.exit(1 if bad else 0)
