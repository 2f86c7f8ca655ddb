#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle): every kernel family that hands filter state from work unit to work unit, with the
hand-over form forced (NVX_INDEPENDENT=0), complete 900 S/s output and bits against the oracle, and the seal counters of
nvx_cascade_integrity_stats.  Run in a process of its own by tests/test_gpu_integrity.py -- with the shipped library (no
stale hand-over may be seen) and with the fault-injection build tests/_variants/libnavtex_amd_inject.so
(NAVTEX_AMD_LIB; -DNVX_INJECT_STALE=n: every n-th hand-over reads the block of the stream's previous launch instead -- the
seal must catch each one, the pre-roll must repair it, and nothing may change in the output).  Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np

import navtex_amd as nv
import oracle_binding as ob
import signals

assert os.environ.get("NVX_INDEPENDENT") == "0", "run with NVX_INDEPENDENT=0: the hand-over form is the one under test"
out = {"lib": os.environ.get("NAVTEX_AMD_LIB", "shipped"), "cases": []}


def u64(a):
    return np.ascontiguousarray(a).view(np.uint64)


# ---- A. resident launches of every stream (no list): both rates, one- and two-chain kernels, both stage-0 forms;
#         10 frames as 7 + 3, then the same again without a reset (positions 30..59 thirds: tags move on)
for raw, masks, order in ((False, [1, 2, 1], 1), (False, [3, 1, 3], 1), (True, [1, 2, 1], 1), (True, [3, 1, 3], 1), (True, [2, 1, 1], 3), (True, [3, 3, 1], 3)):
    rate, frame = (nv.RATE_RAW, nv.FRAME_RAW) if raw else (nv.RATE_IN, nv.FRAME_IN)
    S, F = 3, 20
    iqs = []
    for s in range(S):
        car = [dict(freq_hz=f, bits=nv.sitor_encode(signals.stream_text(700 + s), 10), bit_offset=(977 * (s + 1)) % (rate // 100) | 1,
                    phase0=s * 7654321, amplitude=5000) for c, f in ((0, 14000), (1, -14000)) if (masks[s] >> c) & 1]
        iqs.append(nv.synth_host(nv.make_stream(car, seed=70 + s, noise_amp=1800), rate, F * frame))
    buf = nv.DeviceBuffer(S * F * frame * 4)
    for s in range(S):
        buf.upload(iqs[s], offset=s * F * frame * 4)
    refs = []
    for s in range(S):
        r = ob.Pipe(chain_mask=masks[s], charlayer=False, tap_y3=F * nv.FRAME_Y3)
        if raw:
            r.set_stage0(order); r.push_raw(iqs[s])
        else:
            r.push(iqs[s])
        refs.append(r)
    ok = True
    with nv.Pipeline(n_streams=S, raw_rate=raw, chain_masks=masks, max_frames=7, char_layer=False, stage0_order=order) as p:
        got = {(s, c): [] for s in range(S) for c in range(2) if (masks[s] >> c) & 1}
        f0 = 0
        for k in (7, 3, 6, 4):
            p.process_resident(buf, F * frame, f0, k); f0 += k
            p.fetch()
            for key in got:
                got[key].append(p.debug_y3(*key)[: k * nv.FRAME_Y3].copy())
        for (s, c), parts in got.items():
            ok = ok and np.array_equal(u64(np.concatenate(parts)), u64(refs[s].y3(c))) and p.bits(s, c) == refs[s].bits(c) and len(refs[s].bits(c)) > 300
        stale, failed, launches = p.integrity_stats()
    buf.free()
    out["cases"].append(dict(kind="resident", raw=raw, masks=masks, order=order, ok=bool(ok), stale=stale, failed=failed, launches=launches))

# ---- B. launches that name their streams (the list kernels): random push-mode handles with silent streams
from test_gpu_independent_streams import ragged_case
for seed in (1, 2, 3, 4, 5, 6, 7, 8, 11, 12):
    try:
        info = ragged_case(nv, ob, seed)
        info["ok"] = True
    except AssertionError as e:
        info = dict(seed=seed, ok=False, error=str(e)[:200], stale_repaired=-1)
    out["cases"].append(dict(kind="list", ok=info["ok"], stale=info.get("stale_repaired", -1), **{k: info[k] for k in ("seed", "raw", "order", "two_chain_kernel", "partial_launches") if k in info}))

# ---- B2. ... and deep launches with a list: four streams, the third one silent (declared inactive), six frames at a time
for raw, masks, order in ((False, [1, 2, 1, 1], 1), (False, [3, 1, 2, 3], 1), (True, [1, 1, 2, 1], 1), (True, [3, 1, 1, 3], 3)):
    rate, frame = (nv.RATE_RAW, nv.FRAME_RAW) if raw else (nv.RATE_IN, nv.FRAME_IN)
    S, F = 4, 12
    iqs, refs = [], []
    for s in range(S):
        car = [dict(freq_hz=f, bits=nv.sitor_encode(signals.stream_text(800 + s), 10), bit_offset=(463 * (s + 1)) % (rate // 100) | 1,
                    phase0=s * 1357911, amplitude=5500) for c, f in ((0, 14000), (1, -14000)) if (masks[s] >> c) & 1]
        iqs.append(nv.synth_host(nv.make_stream(car, seed=80 + s, noise_amp=1600), rate, F * frame))
        r = ob.Pipe(chain_mask=masks[s], charlayer=False)
        if raw:
            r.set_stage0(order); r.push_raw(iqs[s])
        else:
            r.push(iqs[s])
        refs.append(r)
    ok = True
    with nv.Pipeline(n_streams=S, raw_rate=raw, chain_masks=masks, max_frames=6, push_mode=True, char_layer=False, stage0_order=order) as p:
        p.set_active(2, False)
        for half in range(2):
            for s in (0, 1, 3):
                p.push(s, iqs[s][half * 6 * frame:(half + 1) * 6 * frame])
        p.push(2, iqs[2])                                  # the late stream: alone, its own parity and position
        p.flush()
        partial = p.stream_stats(0)[2]
        for s in range(S):
            for c in range(2):
                ok = ok and p.bits(s, c) == (refs[s].bits(c) if (masks[s] >> c) & 1 else "")
        stale, failed, launches = p.integrity_stats()
    out["cases"].append(dict(kind="list", deep=True, raw=raw, masks=masks, order=order, ok=bool(ok), stale=stale, failed=failed, launches=launches, partial_launches=partial))

# ---- C. the fused wideband kernel: 3 wideband streams x 12 frames as 12, then 5 + 7
W, F = 3, 12
n = F * nv.FRAME_RAW
rng = np.random.default_rng(77)
raw = rng.integers(-9000, 9000, size=(W, n, 2), dtype=np.int16)
raw[2] = rng.integers(-32768, 32768, size=(n, 2), dtype=np.int16)
for w in range(2):
    car = [dict(freq_hz=(k * 252000 if k < 4 else (k - 8) * 252000) + off, bits=nv.sitor_encode(f"ZCZC IN{k}{c}\nX\nNNNN\n", 8),
                bit_offset=1000 * k + 77 * c + 13 * w + 1, phase0=k * 999 + c, amplitude=1200) for k in range(8) for c, off in ((0, 14000), (1, -14000))]
    raw[w] = np.clip(raw[w].astype(np.int32) + nv.synth_host(nv.make_stream(car, seed=5 + w, noise_amp=0), nv.RATE_RAW, n), -32768, 32767).astype(np.int16)
want = {}
for w in range(W):
    sub = ob.channelise(raw[w])
    for k in range(8):
        ref = ob.Pipe(chain_mask=3, charlayer=False, tap_y3=F * nv.FRAME_Y3)
        ref.push(sub[k])
        for c in range(2):
            want[(8 * w + k, c)] = (u64(ref.y3(c)).copy(), ref.bits(c))
buf = nv.DeviceBuffer(W * n * 4)
buf.upload(raw)
ok = True
with nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=F, char_layer=False) as p:
    stale_total = 0
    for plan in ([F], [5, 7]):
        p.reset()
        got = {key: [] for key in want}
        f0 = 0
        for k in plan:
            p.process_resident(buf, n, f0, k); f0 += k
            p.fetch()
            for key in want:
                got[key].append(p.debug_y3(*key)[: k * nv.FRAME_Y3].copy())
        for key, (y3, bits) in want.items():
            ok = ok and np.array_equal(u64(np.concatenate(got[key])), y3) and p.bits(*key) == bits
    stale, failed, launches = p.integrity_stats()
buf.free()
out["cases"].append(dict(kind="wideband_fused", ok=bool(ok), stale=stale, failed=failed, launches=launches))
print(json.dumps(out))
