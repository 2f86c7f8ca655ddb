"""GPU tests of stream independence (round 3): the streams of one handle are separate receivers -- the chains share no
state (receiver/decoder.h:31-60, receiver/nav_b_sm.h:92-114; FIR1 / mixer per stream: receiver/fir1cpp.C:57-60,
receiver/fir2cpp.C:74-83) -- so a stream that stalls (an unplugged radio: receiver/capt_sched.c:210-212 only prints
sdrplay_api_DeviceRemoved) or runs slow must not hold the others, and must rejoin bit-exactly from its own carried
state.  Launches then cover a LIST of streams, each with its own state-block parity and sample count."""
import ctypes as C
from pathlib import Path
import threading
import time

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


def _oracle_bits(oracle, iq, raw, mask=1):
    ref = oracle.Pipe(chain_mask=mask, charlayer=False)
    (ref.push_raw if raw else ref.push)(iq)
    return [ref.bits(c) for c in (0, 1)]


@pytest.mark.parametrize("raw", [False, True])
def test_a_silent_stream_does_not_hold_the_others_and_rejoins_bit_exactly(nv, oracle, raw):
    """Three streams on a push-mode handle.  Stream 1 delivers nothing while 0 and 2 deliver everything (no NVX_ERR_FULL
    any more: partial launches), then delivers its whole signal late, then all three go on together -- launches that
    cover all streams again, but with per-stream parities and sample counts.  Every stream == the oracle on its own input."""
    rate, frame = (nv.RATE_RAW, nv.FRAME_RAW) if raw else (nv.RATE_IN, nv.FRAME_IN)
    F1, F2 = 9, 5
    iqs = [nv.synth_host(signals.stream_params(nv, 6100 + s, rate, n_phasing=12)[0], rate, (F1 + F2) * frame) for s in range(3)]
    rng = np.random.default_rng(11)
    with nv.Pipeline(n_streams=3, raw_rate=raw, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True, char_layer=False) as p:
        pos = [0, 0, 0]
        while pos[0] < F1 * frame or pos[2] < F1 * frame:              # ragged pushes of streams 0 and 2 only
            s = int(rng.choice([0, 2]))
            if pos[s] >= F1 * frame:
                continue
            m = int(min(F1 * frame - pos[s], rng.integers(1000, 2 * frame)))
            p.push(s, iqs[s][pos[s]:pos[s] + m]); pos[s] += m             # never raises: the others do not wait for stream 1
        p.flush()
        active, frames0, partial = p.stream_stats(0)
        assert frames0 == F1 and partial > 0 and p.stream_stats(1)[1] == 0 and p.stream_stats(2)[1] == F1
        for s in (0, 2):
            assert p.bits(s, 0) == _oracle_bits(oracle, iqs[s][:F1 * frame], raw)[0]
        assert p.bits(1, 0) == ""
        # the late stream: alone in its launches, from its own (reset) state
        while pos[1] < F1 * frame:
            m = int(min(F1 * frame - pos[1], rng.integers(1000, 2 * frame)))
            p.push(1, iqs[1][pos[1]:pos[1] + m]); pos[1] += m
        p.flush()
        assert p.bits(1, 0) == _oracle_bits(oracle, iqs[1][:F1 * frame], raw)[0]
        # together again: every launch covers all three, each with its own parity / sample count
        before = p.stream_stats(0)[2]
        while min(pos) < (F1 + F2) * frame:
            s = int(np.argmin(pos))
            m = int(min((F1 + F2) * frame - pos[s], rng.integers(1000, frame)))
            p.push(s, iqs[s][pos[s]:pos[s] + m]); pos[s] += m
        p.flush()
        assert p.stream_stats(0)[2] == before                           # no partial launch was needed
        for s in range(3):
            assert p.bits(s, 0) == _oracle_bits(oracle, iqs[s], raw)[0] and len(p.bits(s, 0)) > 300
            assert p.stream_stats(s)[1] == F1 + F2


def test_inactive_stream_is_not_waited_for(nv, oracle):
    """nvx_stream_set_active(s, 0): the lock-step trigger stops waiting for s at once (no need to fill a staging set
    first), a push to s makes it active again."""
    F = 4
    iqs = [nv.synth_host(signals.stream_params(nv, 6200 + s, nv.RATE_IN, n_phasing=12)[0], nv.RATE_IN, F * nv.FRAME_IN) for s in range(2)]
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=4, push_mode=True, char_layer=False) as p:
        p.push(0, iqs[0][:nv.FRAME_IN])                                 # one frame staged; stream 1 has nothing: no launch yet
        assert p.stream_stats(0)[1] == 0
        p.set_active(1, False)                                          # ... until stream 1 is declared silent
        assert p.stream_stats(0)[1] == 1 and p.stream_stats(1)[0] is False
        p.push(0, iqs[0][nv.FRAME_IN:2 * nv.FRAME_IN])                  # launches as soon as stream 0 has a frame
        assert p.stream_stats(0)[1] == 2
        p.push(1, iqs[1][:100])                                         # stream 1 is back: waited for again
        assert p.stream_stats(1)[0] is True
        p.push(0, iqs[0][2 * nv.FRAME_IN:3 * nv.FRAME_IN])
        assert p.stream_stats(0)[1] == 2
        p.push(1, iqs[1][100:]); p.push(0, iqs[0][3 * nv.FRAME_IN:])
        p.flush()
        for s in range(2):
            assert p.bits(s, 0) == _oracle_bits(oracle, iqs[s], False)[0]
    # when the LAST active stream goes silent, whole frames that were waiting for company go out at once
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=4, push_mode=True, char_layer=False) as p:
        p.push(0, iqs[0][:nv.FRAME_IN + 500])                           # a frame and a bit, waiting for stream 1
        p.set_active(0, False)                                          # its own radio dies first: still waiting ...
        assert p.stream_stats(0)[1] == 0
        p.set_active(1, False)                                          # ... nobody left to wait for
        assert p.stream_stats(0)[1] == 1


def test_resident_launches_after_divergence(nv, oracle):
    """A push-mode handle whose streams have diverged also takes nvx_process_resident (all streams, each from its own
    state): the launch carries the full list."""
    F = 6
    rate, frame = nv.RATE_IN, nv.FRAME_IN
    iqs = [nv.synth_host(signals.stream_params(nv, 6300 + s, rate, n_phasing=12)[0], rate, F * frame) for s in range(2)]
    # a THIRD frame range, fed resident: stream s gets frames [3, 6) of its own signal after having pushed [0, 3) at different times
    buf = nv.DeviceBuffer(2 * F * frame * 4)
    for s in range(2):
        buf.upload(iqs[s], offset=s * F * frame * 4)
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=3, max_frames=3, push_mode=True, char_layer=False) as p:
        p.push(0, iqs[0][:3 * frame]); p.flush()                        # stream 0 alone (stream 1 inactive by silence: staging full -> partial)
        p.push(1, iqs[1][:3 * frame]); p.flush()
        assert p.stream_stats(0)[2] > 0
        p.process_resident(buf, F * frame, 3, 3); p.fetch()
        for s in range(2):
            want = _oracle_bits(oracle, iqs[s], False, mask=3)
            assert [p.bits(s, 0), p.bits(s, 1)] == want
    buf.free()


def _stats(nv, cap):
    r, d, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    nv.lib.nvx_capture_stats(cap, C.byref(r), C.byref(d), C.byref(c))
    return r.value, d.value, c.value


def test_a_paused_capture_ring_does_not_hold_the_other(nv, oracle):
    """two capture rings on one handle, one radio goes silent for 10 s.  The other stream keeps decoding --
    dropped == 0, its bits == the oracle's and they ARRIVE during the pause -- the silent one is reported
    (nvx_capture_stalled) and resumes bit-exactly from its own carried state."""
    n = 14 * nv.FRAME_IN
    iqs = [nv.synth_host(signals.stream_params(nv, seed, nv.RATE_IN, n_phasing=12)[0], nv.RATE_IN, n) for seed in (911, 912)]
    pause_after, pause_s = 3 * nv.FRAME_IN + 1234, 10.0
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True) as p:
        caps = []
        for s in range(2):
            cap = C.c_void_p()
            assert nv.lib.nvx_capture_start(p._h, s, 2.0, C.byref(cap)) == 0
            caps.append(cap)
        done_at, seen_stalled = {}, []
        abort = threading.Event()

        def vendor(s):
            xi, xq = np.ascontiguousarray(iqs[s][:, 0]), np.ascontiguousarray(iqs[s][:, 1])
            rng = np.random.default_rng(s)
            pos, paused = 0, False
            while pos < n:
                if s == 1 and not paused and pos >= pause_after:        # the radio goes silent
                    paused = True
                    t_end = time.monotonic() + pause_s
                    while time.monotonic() < t_end and not abort.is_set():
                        seen_stalled.append(nv.lib.nvx_capture_stalled(caps[1], None))
                        time.sleep(0.25)
                m = int(min(n - pos, rng.integers(500, 5000)))
                while True:                                             # paced by the ring (2 s deep): never overrun it
                    r, d, c = _stats(nv, caps[s])
                    if r - d - c + m <= 400000 or abort.is_set(): break
                    time.sleep(0.0005)
                nv.lib.nvx_capture_callback(xi[pos:pos + m].ctypes.data, xq[pos:pos + m].ctypes.data, None, m, 0, caps[s])
                pos += m
            done_at[s] = time.monotonic()

        t0 = time.monotonic()
        threads = [threading.Thread(target=vendor, args=(s,)) for s in range(2)]
        for t in threads: t.start()
        stopped = False
        try:
            threads[0].join()
            # stream 0 has delivered everything long before the silent radio comes back; its bits arrive without it
            # (its last frames go out once the silent stream is declared so, 2 s; ring 0's consumer takes the results in)
            deadline = time.monotonic() + 6.0
            want0 = _oracle_bits(oracle, iqs[0], False)[0]
            while p.bit_count(0, 0) < len(want0) - 40 and time.monotonic() < deadline:
                time.sleep(0.05)
            got0_during_pause, t_checked = p.bit_count(0, 0), time.monotonic() - t0
            got1_during_pause = p.bit_count(1, 0)
            threads[1].join()
            events = C.c_uint64()
            assert nv.lib.nvx_capture_stalled(caps[1], C.byref(events)) in (0, 1) and events.value >= 1
            rings = [_stats(nv, caps[s]) for s in range(2)]
            rcs = [nv.lib.nvx_capture_stop(caps[s]) for s in range(2)]
            stopped = True
        finally:
            abort.set()                                                 # (a failure above must not leave the vendor threads waiting)
            for t in threads: t.join()
            if not stopped:
                for s in range(2): nv.lib.nvx_capture_stop(caps[s])
        assert got0_during_pause >= len(want0) - 40, "the running stream's bits must not wait for the silent one"
        assert t_checked < 9.0, "(that check has to have happened DURING the pause)"
        assert got1_during_pause < 200                                  # the silent stream got its first three frames through at most
        assert 1 in seen_stalled, "the silent radio must be reported while it is silent"
        for s in range(2):
            assert rings[s][:2] == (n, 0), f"ring {s}: dropped {rings[s][1]}"
            assert rcs[s] == 0
        assert p.stream_stats(0)[2] > 0                                 # partial launches happened
        for s in range(2):
            assert p.bits(s, 0) == _oracle_bits(oracle, iqs[s], False)[0] and len(p.bits(s, 0)) > 350


def test_wideband_streams_advance_independently(nv, oracle):
    """The same for a wideband handle (fused kernel): two 2.016 MS/s inputs, 16 carriers each; input 1 is late.  The
    launches carry a list of WIDEBAND streams; sub-band state and the 40-sample channeliser halo follow each stream's own parity."""
    F = 4
    n = F * nv.FRAME_RAW
    raws = []
    for w in range(2):
        carriers = []
        for k in range(8):
            centre = k * 252000 if k < 4 else (k - 8) * 252000
            for c, off in ((0, 14000), (1, -14000)):
                cid = 16 * w + 2 * k + c
                carriers.append(dict(freq_hz=centre + off, bits=nv.sitor_encode(signals.stream_text(7000 + cid), 10),
                                     bit_offset=(signals.mix32(cid) % 20160) | 1, phase0=signals.mix32(cid ^ 0x55AA), amplitude=1700))
        raws.append(nv.synth_host(nv.make_stream(carriers, seed=70 + w, noise_amp=500), nv.RATE_RAW, n))
    _secs, want = oracle.bench_wide(np.stack(raws), 2, n // 8, 2, want_bits=True)
    with nv.Pipeline(n_streams=2, wideband=True, chain_mask=3, max_frames=1, push_mode=True, char_layer=False) as p:
        p.push(0, raws[0][:3 * nv.FRAME_RAW])                           # input 0 runs three frames ahead: partial launches
        assert p.stream_stats(0)[2] > 0
        p.push(1, raws[1][:2 * nv.FRAME_RAW])
        p.push(0, raws[0][3 * nv.FRAME_RAW:])
        p.push(1, raws[1][2 * nv.FRAME_RAW:])
        p.flush()
        got = [p.bits(8 * w + k, c) for w in range(2) for k in range(8) for c in (0, 1)]
        assert got == want and all(len(b) > 50 for b in want)


def ragged_case(nv, oracle, seed, tails=False):
    """Random handles (stream count, chain masks, input rate, stage-0 order, max_frames) fed in random order with random
    chunk sizes, streams going silent for a while (explicitly inactive, or simply not fed until another stream's staging
    fills), a reset-free flush in the middle: every list kernel (252 kS/s and raw rate, one and two chains, both stage-0
    forms) meets partial launches, per-stream parities and per-stream sample counts.  Every chain == the oracle.
    tails: every stream's input also ends with a ragged tail of its own (some empty, some shorter than one 900 S/s sample)
    and the handle is ended with nvx_finish instead of flushed: the bits are the oracle's on exactly those samples."""
    rng = np.random.default_rng(1000 + seed)
    raw = bool(rng.integers(0, 2))
    order = int(rng.choice([1, 3])) if raw else 1
    S = int(rng.integers(2, 9))
    F = int(rng.integers(6, 12))
    maxf = int(rng.integers(1, 4))
    masks = [int(rng.choice([1, 2, 3])) for _ in range(S)] if seed % 2 else [int(rng.choice([1, 2]))] * S      # odd seeds: NCH = 2 kernels
    rate, frame = (nv.RATE_RAW, nv.FRAME_RAW) if raw else (nv.RATE_IN, nv.FRAME_IN)
    iqs = []
    for s in range(S):
        carriers = []
        for c, f in ((0, 14000), (1, -14000)):
            if (masks[s] >> c) & 1:
                h = signals.mix32(9000 * seed + 2 * s + c)
                carriers.append(dict(freq_hz=f, bits=nv.sitor_encode(f"ZCZC R{chr(65 + s)}{seed}{c}\nRAGGED {s}\nNNNN\n", 10),
                                     bit_offset=(h % (rate // 100)) | 1, phase0=signals.mix32(h), amplitude=6000))
        iqs.append(nv.synth_host(nv.make_stream(carriers, seed=seed * 100 + s, noise_amp=1200), rate, F * frame))
    total = [F * frame] * S
    if tails:                                                   # (a generator of its own: the cases above keep their draws)
        trng = np.random.default_rng(77000 + seed)
        per_y3 = 2240 if raw else 280
        for s in range(S):
            k = int(trng.integers(0, 5))
            t = 0 if k == 0 else (int(trng.integers(1, per_y3)) if k == 1 else int(trng.integers(per_y3, frame)))
            total[s] += t
            st = nv.make_stream([], seed=seed * 100 + s + 50, noise_amp=3000)
            if t: iqs[s] = np.vstack([iqs[s], nv.synth_host(st, rate, t)])
    with nv.Pipeline(n_streams=S, raw_rate=raw, chain_masks=masks, max_frames=maxf, push_mode=True, char_layer=False, stage0_order=order) as p:
        pos = [0] * S
        asleep = {}                                             # stream -> pushes (of others) until it wakes
        flushed = False
        while any(pos[s] < total[s] for s in range(S)):
            for s in list(asleep):
                asleep[s] -= 1
                if asleep[s] <= 0: del asleep[s]
            live = [s for s in range(S) if pos[s] < total[s] and s not in asleep]
            if not live:
                asleep.clear(); continue
            s = int(rng.choice(live))
            m = int(min(total[s] - pos[s], rng.integers(1, 2 * frame)))
            p.push(s, iqs[s][pos[s]:pos[s] + m]); pos[s] += m
            r = rng.random()
            if r < 0.08 and len(asleep) < S - 1:                # a radio goes quiet: sometimes declared, sometimes just silent
                z = int(rng.integers(0, S))
                asleep[z] = int(rng.integers(3, 25))
                if rng.random() < 0.5: p.set_active(z, False)
            elif r < 0.11 and not flushed:
                p.flush(); flushed = True
        if tails: p.finish()
        else: p.flush()
        partial = p.stream_stats(0)[2]
        stale = p.integrity_stats()[0]
        for s in range(S):
            ref = oracle.Pipe(chain_mask=masks[s], charlayer=False)
            if raw:
                ref.set_stage0(order); ref.push_raw(iqs[s][: total[s] // 8 * 8])
            else:
                ref.push(iqs[s])
            for c in range(2):
                want = ref.bits(c) if (masks[s] >> c) & 1 else ""
                assert p.bits(s, c) == want, f"seed {seed}: stream {s} chain {c} (raw {raw}, order {order}, masks {masks}, max_frames {maxf})"
            assert p.stream_stats(s)[1] == F
    return dict(seed=seed, raw=raw, order=order, streams=S, frames=F, max_frames=maxf, two_chain_kernel=3 in masks, partial_launches=partial, stale_repaired=stale,
                tails=[t - F * frame for t in total] if tails else None)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_randomized_ragged_streams(nv, oracle, seed):
    info = ragged_case(nv, oracle, seed)
    assert info["partial_launches"] > 0, info               # every case really had launches of only some streams


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_randomized_ragged_streams_with_ragged_ends(nv, oracle, seed):
    """The same random handles with every stream's input ending at a place of its own, ended by nvx_finish: streams that
    have come apart in time (per-stream parities and sample counts) AND end raggedly, every list kernel form."""
    info = ragged_case(nv, oracle, seed, tails=True)
    assert info["partial_launches"] > 0 and any(info["tails"]), info


def test_ragged_cases_cover_every_list_kernel(nv, oracle):
    """The seeds above and a few more, by what they exercise: each of the six list kernels (252 kS/s and raw rate x one and
    two chains, third-order stage 0 x one and two chains) must have met partial launches at least once."""
    seen = set()
    for seed in range(1, 40):
        rng = np.random.default_rng(1000 + seed)            # the case's first draws, without running it
        raw = bool(rng.integers(0, 2)); order = int(rng.choice([1, 3])) if raw else 1
        kind = (raw, order, bool(seed % 2))
        if kind in seen:
            continue
        info = ragged_case(nv, oracle, seed)
        if info["partial_launches"] > 0:
            seen.add((info["raw"], info["order"], info["two_chain_kernel"]))
        if len(seen) == 6:
            break
    assert seen == {(False, 1, False), (False, 1, True), (True, 1, False), (True, 1, True), (True, 3, False), (True, 3, True)}, seen


def test_several_threads_push_into_one_handle(nv, oracle):
    """Capture / replay threads of several radios feeding ONE handle at once: pushes of 128 KB and more copy into the pinned
    staging without the handle's lock (a launch first waits for the copies in flight to be committed).  Eight streams,
    four threads with their own pace and push sizes, a flush from the main thread in between: every stream == the oracle."""
    S, F, T = 8, 9, 4
    iqs = [nv.synth_host(signals.stream_params(nv, 6400 + s, nv.RATE_RAW, n_phasing=12)[0], nv.RATE_RAW, F * nv.FRAME_RAW) for s in range(S)]
    errors = []
    with nv.Pipeline(n_streams=S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True, char_layer=True) as p:
        def feed(t):
            rng = np.random.default_rng(t)
            pos = {s: 0 for s in range(t, S, T)}
            try:
                while any(q < F * nv.FRAME_RAW for q in pos.values()):
                    s = int(rng.choice([k for k, q in pos.items() if q < F * nv.FRAME_RAW]))
                    m = int(min(F * nv.FRAME_RAW - pos[s], rng.choice([700, 5000, 40000, 300000, 900000])))
                    p.push(s, iqs[s][pos[s]:pos[s] + m]); pos[s] += m
                    if t == 3 and rng.random() < 0.1: time.sleep(0.002)          # one thread is slower than the rest
            except Exception as e:                                      # noqa: BLE001 -- reported by the main thread
                errors.append(repr(e))
        threads = [threading.Thread(target=feed, args=(t,)) for t in range(T)]
        for th in threads: th.start()
        time.sleep(0.01); p.flush()
        for th in threads: th.join()
        p.flush()
        assert not errors, errors
        for s in range(S):
            assert p.bits(s, 0) == _oracle_bits(oracle, iqs[s], True)[0] and len(p.bits(s, 0)) > 200
            assert p.stream_stats(s)[1] == F


@pytest.mark.parametrize("eager", [False, True], ids=["lock_step", "eager"])
def test_streams_ended_under_their_pushers(nv, oracle, eager):
    """nvx_stream_finish / nvx_finish while other threads are in the middle of push calls on the very streams that end (the
    advisor's r5 finding: a pusher that had released the handle's lock inside its loop went on staging samples into the
    ended stream, which then vetoed every launch of the handle).  A push call is atomic against the end of its stream:
    accepted whole and decoded, or refused whole with NVX_ERR_STATE.  Four streams, a pusher thread each, pushes from a few
    hundred samples to several frames; stream 1 is ended alone while all four run, the others a little later all at once.
    Every stream's bits == the oracle on exactly the samples its accepted calls carried; the streams that were not yet
    ended kept launching in between; only NVX_ERR_STATE was ever seen, and only behind an end."""
    S, F = 4, 40
    iqs = [nv.synth_host(signals.stream_params(nv, 8800 + s, nv.RATE_IN, n_phasing=12)[0], nv.RATE_IN, F * nv.FRAME_IN) for s in range(S)]
    accepted, wrong = [0] * S, []
    with nv.Pipeline(n_streams=S, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True, char_layer=False,
                     eager_launch=eager, stall_timeout_ms=-1) as p:
        def feed(s):
            rng = np.random.default_rng(50 + s)
            pos = 0
            while pos < F * nv.FRAME_IN:
                m = int(min(F * nv.FRAME_IN - pos, rng.choice([300, 4000, 30000, 200000, 3 * nv.FRAME_IN + 17])))
                try:
                    p.push(s, iqs[s][pos:pos + m])
                except nv.NvxError as e:
                    if e.code != -5: wrong.append((s, e.code))       # NVX_ERR_STATE: the stream ended under this pusher
                    return
                pos += m; accepted[s] = pos
                time.sleep(0.002)                                   # (a pace at which the ends below fall into the middle of the inputs)
        threads = [threading.Thread(target=feed, args=(s,)) for s in range(S)]
        for th in threads: th.start()
        time.sleep(0.02)
        p.finish(1)
        frames_then = p.stream_stats(0)[1]
        time.sleep(0.025)
        frames_later = p.stream_stats(0)[1]
        p.finish()
        for th in threads: th.join()
        assert not wrong, wrong
        assert frames_later > frames_then, "the handle made no progress behind the ended stream"
        for s in range(S):
            ref = oracle.Pipe(chain_mask=1, charlayer=False)
            ref.push(iqs[s][:accepted[s]])
            assert p.bits(s, 0) == ref.bits(0), f"stream {s}: {accepted[s]} samples accepted"
        assert accepted[1] < F * nv.FRAME_IN and accepted[1] < max(accepted), accepted        # stream 1 really ended in mid-input, before the others
        assert p.integrity_stats()[:2] == (0, 0)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_randomized_ragged_wideband_inputs(nv, oracle, seed):
    """The fused wideband kernel with participant lists, randomised: two or three 2.016 MS/s inputs (16 carriers each) fed
    in random order with random chunk sizes and silent spells; sub-band filter state and the channeliser's 40-sample halo
    follow each input's own parity.  All carriers == the restatement chain."""
    rng = np.random.default_rng(500 + seed)
    W, F, maxf = int(rng.integers(2, 4)), int(rng.integers(4, 7)), int(rng.integers(1, 3))
    n = F * nv.FRAME_RAW
    raws = []
    for w in range(W):
        carriers = []
        for k in range(8):
            centre = k * 252000 if k < 4 else (k - 8) * 252000
            for c, off in ((0, 14000), (1, -14000)):
                cid = 1000 * seed + 16 * w + 2 * k + c
                carriers.append(dict(freq_hz=centre + off, bits=nv.sitor_encode(signals.stream_text(8000 + cid), 8),
                                     bit_offset=(signals.mix32(cid) % 20160) | 1, phase0=signals.mix32(cid ^ 0x1234), amplitude=1600))
        raws.append(nv.synth_host(nv.make_stream(carriers, seed=seed * 10 + w, noise_amp=500), nv.RATE_RAW, n))
    _secs, want = oracle.bench_wide(np.stack(raws), W, n // 8, 4, want_bits=True)
    with nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=maxf, push_mode=True, char_layer=False) as p:
        pos = [0] * W
        quiet = {}
        while any(q < n for q in pos):
            for w in list(quiet):
                quiet[w] -= 1
                if quiet[w] <= 0: del quiet[w]
            live = [w for w in range(W) if pos[w] < n and w not in quiet]
            if not live:
                quiet.clear(); continue
            w = int(rng.choice(live))
            m = int(min(n - pos[w], rng.integers(1, 2 * nv.FRAME_RAW)))
            p.push(w, raws[w][pos[w]:pos[w] + m]); pos[w] += m
            if rng.random() < 0.15 and len(quiet) < W - 1:
                z = int(rng.integers(0, W)); quiet[z] = int(rng.integers(2, 8))
                if rng.random() < 0.5: p.set_active(z, False)
        p.flush()
        got = [p.bits(8 * w + k, c) for w in range(W) for k in range(8) for c in (0, 1)]
        assert got == want and all(len(b) > 30 for b in want)
        assert p.stream_stats(0)[2] > 0, "the case must have had launches of only some inputs"


def test_ragged_cases_with_the_hand_over_forms_forced(nv, tmp_path):
    """The ragged cases run few streams, so their launches use independent units.  Launches of thousands of streams that
    name their streams hand filter state from unit to unit instead (done[] indexed by list position, state blocks by
    stream and parity): force that form -- with and without the dynamic pre-roll -- in a subprocess and run cases that
    between them cover all six list kernels."""
    import subprocess, sys, os
    root = str(Path(__file__).resolve().parent.parent)
    for env in (dict(NVX_INDEPENDENT="0", NVX_DYNAMIC_PREROLL="1"), dict(NVX_INDEPENDENT="0", NVX_DYNAMIC_PREROLL="0"), dict(NVX_INDEPENDENT="1")):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_scripts", "sweep_ragged.py"), "1", "20"],
                             capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert out.returncode == 0, (env, out.stdout[-1500:], out.stderr[-1500:])
        last = out.stdout.strip().splitlines()[-1]
        assert last.startswith("20 cases identical to the oracle"), last
        kinds = last[last.index("{"):]
        assert kinds.count("(") == 6, (env, last)                 # all six (rate, stage-0 order, chains) list kernels were met


def test_two_pushers_of_one_stream_take_turns_call_by_call(nv, oracle):
    """a push call is atomic with respect to other pushes of the SAME stream -- a second pusher waits for the whole
    call, also while the first one waits for a launch or copies a large chunk without the handle's lock.  Two threads push
    one multi-frame chunk each into stream 0 at the same moment: the stream must read X then Y or Y then X, never a mix."""
    import threading
    F = 5
    st, _ = signals.stream_params(nv, 8080, nv.RATE_RAW)
    xy = nv.synth_host(st, nv.RATE_RAW, 2 * F * nv.FRAME_RAW)
    x, y = xy[: F * nv.FRAME_RAW], xy[F * nv.FRAME_RAW:]
    wants = []
    for order in ((x, y), (y, x)):
        ref = oracle.Pipe(chain_mask=1, charlayer=False)
        ref.push_raw(order[0]); ref.push_raw(order[1])
        wants.append(ref.bits(0))
    assert wants[0] != wants[1]
    seen = set()
    for rep in range(6):
        with nv.Pipeline(n_streams=2, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True, char_layer=False) as p:
            p.set_active(1, False)
            go = threading.Barrier(2)
            def pusher(chunk):
                go.wait(); p.push(0, chunk)
            ths = [threading.Thread(target=pusher, args=(c,)) for c in (x, y)]
            for t in ths: t.start()
            for t in ths: t.join()
            p.flush()
            got = p.bits(0, 0)
            assert got in wants, f"round {rep}: the two pushes were interleaved"
            seen.add(wants.index(got))
    assert seen                                                 # (either order is legitimate; usually both occur)


def test_a_stream_fed_directly_that_goes_quiet_is_not_waited_for(nv, oracle):
    """a stream fed through nvx_push_iq (no capture ring to declare it silent) that stops delivering must not
    keep the others at one launch per FULL staging buffer: after 2 s without a push it is no longer waited for, and the
    healthy stream's frames go out one by one again.  It rejoins bit-exactly."""
    import time
    F, maxf = 14, 8
    iqs = [nv.synth_host(signals.stream_params(nv, 8200 + s, nv.RATE_IN)[0], nv.RATE_IN, F * nv.FRAME_IN) for s in range(2)]
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=maxf, push_mode=True, char_layer=False) as p:
        for s in range(2):
            p.push(s, iqs[s][: 2 * nv.FRAME_IN])                # both deliver two frames: lock-step launches
        assert p.stream_stats(0)[1] == 2 and p.stream_stats(1)[1] == 2
        p.push(0, iqs[0][2 * nv.FRAME_IN: 4 * nv.FRAME_IN])     # stream 1 has gone quiet; stream 0 waits for it at first ...
        assert p.stream_stats(0)[1] == 2
        time.sleep(2.2)
        p.push(0, iqs[0][4 * nv.FRAME_IN: 5 * nv.FRAME_IN])     # ... and after 2 s no longer: everything staged goes out
        assert p.stream_stats(0)[1] == 5, "the healthy stream must not wait for a full staging buffer"
        p.push(0, iqs[0][5 * nv.FRAME_IN: 6 * nv.FRAME_IN])
        assert p.stream_stats(0)[1] == 6                        # frame by frame now (staging holds 9)
        p.push(1, iqs[1][2 * nv.FRAME_IN:])                     # the quiet one comes back with all it has
        p.push(0, iqs[0][6 * nv.FRAME_IN:])
        p.flush()
        for s in range(2):
            ref = oracle.Pipe(chain_mask=1, charlayer=False); ref.push(iqs[s])
            assert p.bits(s, 0) == ref.bits(0) and len(ref.bits(0)) > 300, s
        assert p.integrity_stats()[:2] == (0, 0)


@pytest.mark.parametrize("eager", [False, True], ids=["lockstep", "eager"])
def test_free_running_radios_need_not_wait_for_each_others_frames(nv, oracle, eager):
    """Two radios on one handle, started half a frame apart (free-running SDR clocks: their frames never complete at the
    same moment), fed at the real rate through a capture ring each.  Lock-step launches (the default: fewest, largest
    launches) make the earlier radio's bits wait for the later radio's frame -- about 160 ms here --; with
    cfg.eager_launch a launch goes out as soon as ANY stream has a whole frame, and both radios' frames are decoded and
    delivered within milliseconds.  Same bits either way: the streams are independent receivers."""
    import time
    from fake_sdr import FakeSdr
    n_frames = 8
    iqs = [nv.synth_host(signals.stream_params(nv, 8300 + s, nv.RATE_IN)[0], nv.RATE_IN, n_frames * nv.FRAME_IN) for s in range(2)]
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True, char_layer=False, eager_launch=eager) as p:
        caps = [nv.Capture(p, s, ring_seconds=2.0) for s in range(2)]
        sdrs = [FakeSdr(caps[s], iqs[s], nv.RATE_IN, nv.FRAME_IN, seed=20 + s, packet=(150, 420)) for s in range(2)]
        sdrs[0].start(); time.sleep(0.16); sdrs[1].start()          # radio 1 runs half a frame behind radio 0
        for t in sdrs: t.join()
        time.sleep(0.1)
        lat = [c.latency() for c in caps]
        stats = [c.stats() for c in caps]
        partial = p.stream_stats(0)[2]
        for c in caps: c.stop()
        for s in range(2):
            ref = oracle.Pipe(chain_mask=1, charlayer=False); ref.push(iqs[s])
            assert p.bits(s, 0) == ref.bits(0) and stats[s][1] == 0 and lat[s]["frames"] >= n_frames - 1, s
        assert max(t.late_ms for t in sdrs) < 50.0
        if eager:
            assert lat[0]["max_ms"] < 50.0 and lat[1]["max_ms"] < 50.0 and partial > 0, (lat, partial)
        else:
            assert lat[0]["p50_ms"] > 100.0 and lat[1]["p50_ms"] < 50.0, lat          # the premise: the early radio waits for the late one's frame
