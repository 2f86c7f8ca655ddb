#!/usr/bin/env python3
"""Regression soak of the unit hand-over after the round-3 changes to the state-block addressing: soak_units.py [N] [--cic3]
(--cic3: the third-order stage 0's kernel, whose carried block holds four more integers).
N full-size runs (4096 streams x 12 frames) from reset in three launch partitions (12 / 5+7 / 4+4+4); the bits of all
streams are compared run to run, and a spread sample of 64 against the oracle once per partition."""
import hashlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import navtex_amd as nv
import oracle_binding as ob
import fullsize, signals

CIC3 = "--cic3" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--cic3"]
N = int(argv[0]) if argv else 60
S, F = 4096, 12
pitch = F * nv.FRAME_RAW
buf = nv.DeviceBuffer(S * pitch * 4)
nv.synth_device([signals.stream_params(nv, s, nv.RATE_RAW)[0] for s in range(S)], nv.RATE_RAW, pitch, buf, pitch)
plans = ([12], [5, 7], [4, 4, 4])
ref = None; t0 = time.time(); waited = 0
with nv.Pipeline(n_streams=S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=F, char_layer=False, stage0_order=3 if CIC3 else 1) as p:
    for run in range(N):
        plan = plans[run % 3]
        p.reset(); f0 = 0
        for k in plan:
            p.process_resident(buf, pitch, f0, k); f0 += k
        p.fetch()
        h = hashlib.sha256()
        for s in range(S):
            h.update(p.bits(s, 0).encode()); h.update(b"|")
        d = h.hexdigest()
        if ref is None:
            ref = d
        assert d == ref, f"run {run} (plan {plan}): bits differ from the first run"
        if run < 3:
            checked, bad, _ = fullsize.verify_streams(ob, buf, pitch, pitch, 3 if CIC3 else True, lambda s: p.bits(s, 0), fullsize.spread(S, 64), 16)
            assert not bad, (plan, bad)
        if run % 10 == 9:
            print(f"run {run + 1}: identical; {time.time() - t0:.0f} s", flush=True)
    polls, units, launches = p.wait_stats()
    stale, failures, _ = p.integrity_stats()
print(f"{N} full-size runs{' (third-order stage 0)' if CIC3 else ''} in three launch partitions: all 4096 streams' bits identical run to run, 64 checked against the oracle per partition; "
      f"{units} units pre-rolled or waited over {launches} launches; seals (r4): about {N * S * (F + 1)} hand-overs checked, "
      f"{stale} stale / torn blocks repaired, {failures} launches failed the check of their inherited state")
assert stale == 0 and failures == 0
buf.free()
