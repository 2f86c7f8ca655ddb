"""GPU tests of the drop-in boundary: the reference-shaped push surface, the
SDRplay-shaped stream callback, the WAV file path, and the golden vectors of
the compiled reference replayed through the HIP path (BASELINE configs 0-2)."""
import ctypes as C
import json
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((Path(__file__).parent / "golden" / "golden.json").read_text())


def gold_messages(rec):
    return sorted(rec["messages"])


def got_messages(p):
    return sorted([f, b, m] for (_s, f, b, m) in p.messages)


@pytest.mark.parametrize("name", sorted(GOLD["iq"]))
def test_reference_bits_reproduced_on_gpu(nv, name):
    """Every golden case (clean, weak, offset, noise, silence, DC, full-scale random, ragged length): the bits and the
    messages the compiled reference produced from exactly these samples -- no more, no fewer.  The input is pushed as it
    is, whatever its length; nvx_finish runs the last, partial frame at its true length (the reference's loop stops with
    its last sample: receiver/capt_sched.c:509-513)."""
    rec = GOLD["iq"][name]
    iq = cases.make_iq(nv, rec["spec"])
    with nv.Pipeline(n_streams=1, raw_rate=False, max_frames=8, push_mode=True) as p:
        p.push(0, iq)
        p.finish()
        for tag, chain in (("518", 0), ("490", 1)):
            assert p.bits(0, chain) == rec[f"bits{tag}"], f"{name}/{tag}"
        assert got_messages(p) == gold_messages(rec)
        with pytest.raises(nv.NvxError):             # the stream has ended: nothing more goes in until a reset
            p.push(0, iq[:100])
        p.reset()
        p.push(0, iq); p.finish()
        assert p.bits(0, 0) == rec["bits518"] and p.bits(0, 1) == rec["bits490"]


@pytest.mark.parametrize("name", ["ragged_length", "two_carrier", "weak_518"])
def test_flush_keeps_a_partial_frame_staged_and_finish_ends_the_stream(nv, name):
    """A streaming handle: arbitrary push sizes, plain nvx_flush calls in between (a partial frame stays staged and the
    stream goes on bit-exactly), nvx_finish at the end -- bits and messages equal to the compiled reference's at every
    cut, and the bits present after each flush are a prefix of the final ones."""
    rec = GOLD["iq"][name]
    iq = cases.make_iq(nv, rec["spec"])
    rng = np.random.default_rng(len(name))
    with nv.Pipeline(n_streams=1, raw_rate=False, max_frames=3, push_mode=True) as p:
        pos, seen = 0, ["", ""]
        while pos < iq.shape[0]:
            m = int(min(iq.shape[0] - pos, rng.integers(1, 2 * nv.FRAME_IN)))
            p.push(0, iq[pos:pos + m]); pos += m
            if rng.integers(0, 3) == 0:
                p.flush()
                for c in (0, 1):
                    now = p.bits(0, c)
                    assert now.startswith(seen[c]); seen[c] = now
        p.finish()
        assert p.bits(0, 0) == rec["bits518"] and p.bits(0, 1) == rec["bits490"]
        assert p.bits(0, 0).startswith(seen[0]) and p.bits(0, 1).startswith(seen[1])
        assert got_messages(p) == gold_messages(rec)


@pytest.mark.parametrize("cut", [0, 1, 279, 280, 281, 2519, 2520, 9 * 280 + 5, 17 * 280, 40000, 80639])
def test_every_kind_of_tail_matches_the_oracle(nv, oracle, cut):
    """The end of a stream at every kind of position: nothing, less than one 900 S/s sample, exactly one, whole and ragged
    bit periods, almost a frame -- after two whole frames and well past the demodulator's priming (the timing filter is
    primed at 900 S/s sample 582, a frame is 288).  Bits == the oracle fed exactly the same samples."""
    import signals
    st, _ = signals.stream_params(nv, 5, nv.RATE_IN)
    n = 3 * nv.FRAME_IN + cut
    iq = nv.synth_host(st, nv.RATE_IN, n)
    ref = oracle.Pipe(chain_mask=3, charlayer=False, tap_y3=4 * nv.FRAME_Y3)
    ref.push(iq)
    with nv.Pipeline(n_streams=1, raw_rate=False, max_frames=2, push_mode=True, char_layer=False) as p:
        p.push(0, iq)
        p.finish()
        assert p.bits(0, 0) == ref.bits(0) and p.bits(0, 1) == ref.bits(1)
        assert len(ref.bits(0)) >= 28
        if cut >= 280:                               # the tail's own launch: exactly the 900 S/s samples the real input produced, bit for bit
            for c in (0, 1):
                want = np.ascontiguousarray(ref.y3(c)[3 * nv.FRAME_Y3:])
                got = p.debug_y3(0, c)
                assert want.shape[0] == cut // 280 and got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_tails_of_many_streams_in_one_launch_raw_rate_and_wideband(nv, oracle):
    """nvx_finish on handles with several streams whose inputs end at different places (one on a frame boundary, one too
    short for a single 900 S/s sample more): ONE launch carries every tail at its own length.  Both stage-0 forms at
    2.016 MS/s, and a wideband handle (8 sub-bands x 2 chains per input).  Every chain == the oracle on the same samples;
    every stream is ended afterwards -- the one that stopped on a frame boundary too (one rule for every length)."""
    import signals
    tails = [0, 2239, 2240, 100001, 300000, 645119]
    for order in (1, 3):
        with nv.Pipeline(n_streams=len(tails), raw_rate=True, chain_mask=3, max_frames=2, push_mode=True, char_layer=False, stage0_order=order) as p:
            want = []
            for s, t in enumerate(tails):
                st, _ = signals.stream_params(nv, 20 + s, nv.RATE_RAW)
                iq = nv.synth_host(st, nv.RATE_RAW, 3 * nv.FRAME_RAW + t)
                ref = oracle.Pipe(chain_mask=3, charlayer=False); ref.set_stage0(order)
                ref.push_raw(iq[: iq.shape[0] // 8 * 8])
                want.append((ref.bits(0), ref.bits(1)))
                p.push(s, iq)
            launches0 = p.stream_stats(0)[2]
            p.finish()
            for s in range(len(tails)):
                assert (p.bits(s, 0), p.bits(s, 1)) == want[s], f"stage0 order {order}, stream {s}"
            for s, t in enumerate(tails):
                with pytest.raises(nv.NvxError):
                    p.push(s, np.zeros((16, 2), dtype=np.int16))
            p.stream_reset(0)                                             # (the one that stopped on a frame boundary starts anew like any other)
            p.push(0, np.zeros((16, 2), dtype=np.int16))
    # wideband: two inputs, different tails
    wt = [7 * 2240 * 9 + 3, 400000]
    with nv.Pipeline(n_streams=2, wideband=True, chain_mask=3, max_frames=2, push_mode=True, char_layer=False) as p:
        raws = []
        for w, t in enumerate(wt):
            car = [dict(freq_hz=(k * 252000 if k < 4 else (k - 8) * 252000) + off, bits=nv.sitor_encode(f"ZCZC WT{k}{c}\nTAIL\nNNNN\n", 4),
                        bit_offset=977 * (2 * k + c + 1) + 31 * w, phase0=k * 1234567 + c, amplitude=1800) for k in range(8) for c, off in ((0, 14000), (1, -14000))]
            raw = nv.synth_host(nv.make_stream(car, seed=90 + w, noise_amp=300), nv.RATE_RAW, 3 * nv.FRAME_RAW + t)
            raws.append(raw)
            p.push(w, raw)
        p.finish()
        for w, raw in enumerate(raws):
            sub = oracle.channelise(raw[: raw.shape[0] // 8 * 8])
            for k in range(8):
                ref = oracle.Pipe(chain_mask=3, charlayer=False)
                ref.push(sub[k])
                for c in (0, 1):
                    assert p.bits(8 * w + k, c) == ref.bits(c), f"wideband input {w} band {k} chain {c}"


@pytest.mark.parametrize("raw", [False, True], ids=["252k", "raw"])
def test_random_tails_of_many_streams_end_in_one_launch(nv, oracle, raw):
    """48 streams of one handle, random chain masks, random lengths (one to three frames and a random tail, some tails empty,
    some shorter than one 900 S/s sample), pushed in random order and chunk sizes with flushes in between; then nvx_finish:
    whole frames first, then ONE launch that carries every tail at its own length.  Every chain of every stream == the
    oracle on exactly the stream's samples."""
    import signals
    rate, frame, per_y3 = (nv.RATE_RAW, nv.FRAME_RAW, 2240) if raw else (nv.RATE_IN, nv.FRAME_IN, 280)
    rng = np.random.default_rng(31 + int(raw))
    S = 48
    masks = [int(rng.choice([1, 2, 3])) for _ in range(S)]
    lens = []
    for s in range(S):
        tail = 0 if s % 11 == 0 else (int(rng.integers(1, per_y3)) if s % 13 == 0 else int(rng.integers(per_y3, frame)))
        lens.append(int(rng.integers(1, 4)) * frame + tail)
    iqs = []
    for s in range(S):
        car = [dict(freq_hz=f, bits=nv.sitor_encode(signals.stream_text(300 + s), 6), bit_offset=(197 * (s + 1)) % (rate // 100) | 1, phase0=s * 424243, amplitude=5000)
               for c, f in ((0, 14000), (1, -14000)) if (masks[s] >> c) & 1]
        iqs.append(nv.synth_host(nv.make_stream(car, seed=300 + s, noise_amp=1200), rate, lens[s]))
    with nv.Pipeline(n_streams=S, raw_rate=raw, chain_masks=masks, max_frames=2, push_mode=True, char_layer=False, stall_timeout_ms=-1) as p:
        pos = [0] * S
        live = list(range(S))
        while live:
            s = int(rng.choice(live))
            m = int(min(lens[s] - pos[s], rng.integers(1, frame)))
            p.push(s, iqs[s][pos[s]:pos[s] + m]); pos[s] += m
            if pos[s] == lens[s]:
                live.remove(s)
            if rng.integers(0, 40) == 0:
                p.flush()
        launches_before = p.wait_stats()[2]
        p.finish()
        for s in range(S):
            ref = oracle.Pipe(chain_mask=masks[s], charlayer=False)
            (ref.push_raw if raw else ref.push)(iqs[s][: lens[s] // 8 * 8] if raw else iqs[s])
            for c in range(2):
                want = ref.bits(c) if (masks[s] >> c) & 1 else ""
                assert p.bits(s, c) == want, f"stream {s} chain {c} mask {masks[s]} length {lens[s]}"
        assert p.integrity_stats()[:2] == (0, 0) and launches_before > 0


def test_empty_inputs(nv, tmp_path):
    """Nothing in, nothing out, no error: a finish on a fresh handle, zero-length pushes, an empty WAV file (the golden
    `empty` case of the WAV boundary) -- and the handle is as good as new afterwards (no stream is ended by it)."""
    rec = GOLD["iq"]["ragged_length"]
    iq = cases.make_iq(nv, rec["spec"])
    path = str(tmp_path / "empty.wav")
    nv.wav_write(path, np.zeros((0, 2), dtype=np.int16), nv.RATE_IN)
    with nv.Pipeline(n_streams=2, raw_rate=False, max_frames=2, push_mode=True) as p:
        p.finish()
        p.push(0, np.zeros((0, 2), dtype=np.int16)); p.push_planar(1, np.zeros(0, dtype=np.int16), np.zeros(0, dtype=np.int16))
        p.flush(); p.finish()
        assert p.decode_wav(path, stream=1) == 0
        assert p.bits(0, 0) == "" and p.bits(1, 1) == "" and p.messages == [] and p.stream_stats(0)[1] == 0
        p.push(0, iq); p.finish(0)
        assert p.bits(0, 0) == rec["bits518"] and p.bits(0, 1) == rec["bits490"]


def test_wav_file_path_config0(nv, tmp_path):
    """configs[0]/[1] plumbing: 2-channel 16-bit 252 kHz WAV -> nvx_decode_wav -> bits and messages, exactly the compiled
    reference's on the same samples (the file's last, partial frame runs at its true length); also a file of ragged length."""
    for name in ("two_carrier", "ragged_length"):
        rec = GOLD["iq"][name]
        iq = cases.make_iq(nv, rec["spec"])
        path = str(tmp_path / f"{name}.wav")
        nv.wav_write(path, iq, nv.RATE_IN)
        with nv.Pipeline(n_streams=1, raw_rate=False, max_frames=4, push_mode=True) as p:
            frames = p.decode_wav(path)
            assert frames == -(-iq.shape[0] // nv.FRAME_IN)
            assert got_messages(p) == gold_messages(rec)
            assert p.bits(0, 0) == rec["bits518"] and p.bits(0, 1) == rec["bits490"]
    with nv.Pipeline(n_streams=1, raw_rate=True, max_frames=1, push_mode=True) as p:
        with pytest.raises(nv.NvxError):          # wrong sample rate for this handle
            p.decode_wav(path)


def test_one_stream_starts_anew_while_the_others_carry_on(nv, oracle, tmp_path):
    """nvx_stream_reset: stream 0 of a two-stream handle decodes one WAV file after another (each ended exactly by
    nvx_decode_wav, then reset for the next) while stream 1 is fed a long signal in pieces in between and never notices:
    every file's bits and messages are the compiled reference's, stream 1's bits the oracle's over its whole input.
    An ended stream refuses input until it is reset; the reset restarts its bit counters."""
    import signals
    files = ["ragged_length", "offset_490", "two_carrier", "ragged_length"]
    st, _ = signals.stream_params(nv, 91, nv.RATE_IN)
    n1 = 9 * nv.FRAME_IN + 12345
    iq1 = nv.synth_host(st, nv.RATE_IN, n1)
    ref1 = oracle.Pipe(chain_mask=3, charlayer=False)
    ref1.push(iq1)
    with nv.Pipeline(n_streams=2, raw_rate=False, max_frames=3, push_mode=True, stall_timeout_ms=-1) as p:
        pos = 0
        for k, name in enumerate(files):
            rec = GOLD["iq"][name]
            path = str(tmp_path / f"{k}_{name}.wav")
            nv.wav_write(path, cases.make_iq(nv, rec["spec"]), nv.RATE_IN)
            step = n1 // len(files) + 1
            p.push(1, iq1[pos:pos + step]); pos += step          # the other receiver keeps running
            p.messages.clear()
            p.decode_wav(path, stream=0)
            assert p.bits(0, 0) == rec["bits518"] and p.bits(0, 1) == rec["bits490"], (k, name)
            assert sorted([f, b, m] for (s_, f, b, m) in p.messages if s_ == 0) == sorted(rec["messages"]), (k, name)
            with pytest.raises(nv.NvxError):                     # ended: nothing more goes in ...
                p.push(0, np.zeros((8, 2), dtype=np.int16))
            p.stream_reset(0)                                    # ... until it starts anew
            assert p.bit_count(0, 0) == 0 and p.bits(0, 0) == ""
        p.finish()
        assert len(ref1.bits(0)) > 200
        assert p.bits(1, 0) == ref1.bits(0) and p.bits(1, 1) == ref1.bits(1)
        assert p.integrity_stats()[:2] == (0, 0)


def _wide_input(nv, seed, n):
    """A 2.016 MS/s input with sixteen carriers (8 sub-bands x +-14 kHz), each with its own text, timing and phase."""
    car = [dict(freq_hz=(k * 252000 if k < 4 else (k - 8) * 252000) + off, bits=nv.sitor_encode(f"ZCZC R{k}{c}{seed % 10}\nRESET {seed}\nNNNN\n", 4),
                bit_offset=(977 * (2 * k + c + 1) + 31 * seed) % 20160 | 1, phase0=k * 1234567 + c + seed, amplitude=1800) for k in range(8) for c, off in ((0, 14000), (1, -14000))]
    return nv.synth_host(nv.make_stream(car, seed=seed, noise_amp=300), nv.RATE_RAW, n)


@pytest.mark.parametrize("kind", ["one_stream_252k", "one_stream_raw", "raw_rate", "raw_rate_cic3", "wideband", "one_wideband"])
def test_stream_reset_on_every_kind_of_handle(nv, oracle, kind):
    """nvx_stream_reset's per-row offsets on the handles the two-stream 252 kS/s test above does not reach (the advisor's r5
    finding): raw-rate handles in both stage-0 forms (the third-order form carries four more integers per stream), a
    wideband handle (eight state blocks, sixteen demodulator slots, the channeliser's halo and FIR3's row prefixes per
    input), and handles with ONE stream (no participant lists: after an odd number of launches the handle reads state
    block 1).  Three inputs of different ragged lengths run through stream 0 one after the other -- pushed in pieces with
    a flush in between, ended by nvx_stream_finish, reset -- while (where there is one) stream 1 is fed a long signal in
    pieces and never notices.  Every chain == the oracle on exactly its samples, every time."""
    import signals
    wide = kind.endswith("wideband")
    raw = kind in ("one_stream_raw", "raw_rate", "raw_rate_cic3")
    order = 3 if kind == "raw_rate_cic3" else 1
    S = 1 if kind.startswith("one_") else 2
    rate, frame = (nv.RATE_RAW, nv.FRAME_RAW) if (raw or wide) else (nv.RATE_IN, nv.FRAME_IN)

    def make(seed, n):
        if wide:
            return _wide_input(nv, seed, n)
        car = [dict(freq_hz=f, bits=nv.sitor_encode(signals.stream_text(seed + c), 6), bit_offset=(977 * (seed + c)) % (rate // 100) | 1,
                    phase0=seed * 424243 + c, amplitude=5000) for c, f in ((0, 14000), (1, -14000))]
        return nv.synth_host(nv.make_stream(car, seed=seed, noise_amp=1200), rate, n)

    def want(iq):
        """[(bits518, bits490)] per decoded stream of one input"""
        if wide:
            sub = oracle.channelise(iq[: iq.shape[0] // 8 * 8])
            out = []
            for k in range(8):
                r = oracle.Pipe(chain_mask=3, charlayer=False); r.push(sub[k]); out.append((r.bits(0), r.bits(1)))
            return out
        r = oracle.Pipe(chain_mask=3, charlayer=False)
        if raw:
            r.set_stage0(order); r.push_raw(iq[: iq.shape[0] // 8 * 8])
        else:
            r.push(iq)
        return [(r.bits(0), r.bits(1))]

    per = 8 if wide else 1
    lengths = [4 * frame + 100001, 5 * frame + 7 * 2240 * 9 + 3, 3 * frame + frame // 2 + 11]
    kw = dict(wideband=True) if wide else dict(raw_rate=raw, stage0_order=order)
    with nv.Pipeline(n_streams=S, chain_mask=3, max_frames=2, push_mode=True, char_layer=False, stall_timeout_ms=-1, **kw) as p:
        other = make(500, 9 * frame + 4321) if S == 2 else None
        pos = 0
        for k, n in enumerate(lengths):
            iq = make(40 + k, n)
            cut = frame + 12345 * (k + 1)                 # a flush in mid-input: whole frames launched, the rest staged
            p.push(0, iq[:cut]); p.flush()
            if S == 2:
                step = other.shape[0] // len(lengths) + 1
                p.push(1, other[pos:pos + step]); pos += step
            p.push(0, iq[cut:]); p.finish(0)
            w = want(iq)
            for d in range(per):
                assert (p.bits(d, 0), p.bits(d, 1)) == w[d], f"{kind}: input {k}, decoded stream {d}"
            assert len(w[0][0]) > 40
            with pytest.raises(nv.NvxError):
                p.push(0, iq[:16])                        # ended ...
            p.stream_reset(0)                             # ... until it starts anew
            assert all(p.bit_count(d, c) == 0 for d in range(per) for c in (0, 1))
        if S == 2:
            p.finish(1)
            w = want(other)
            for d in range(per):
                assert (p.bits(per + d, 0), p.bits(per + d, 1)) == w[d], f"{kind}: the other stream, decoded stream {d}"
        assert p.integrity_stats()[:2] == (0, 0)


def test_stream_callback_shape(nv):
    """nvx_StreamACallback: planar xi/xq, jittered numSamples, cbContext = handle; the input as it is, ended by nvx_finish:
    bits and messages exactly the compiled reference's."""
    rec = GOLD["iq"]["offset_490"]
    iq = cases.make_iq(nv, rec["spec"])
    xi, xq = np.ascontiguousarray(iq[:, 0]), np.ascontiguousarray(iq[:, 1])
    rng = np.random.default_rng(9)
    with nv.Pipeline(n_streams=1, raw_rate=False, max_frames=2, push_mode=True) as p:
        pos = 0
        while pos < xi.size:
            m = int(min(xi.size - pos, rng.integers(1, 4096)))
            a, b = xi[pos:pos + m].copy(), xq[pos:pos + m].copy()      # valid only during the call
            nv.lib.nvx_StreamACallback(a.ctypes.data, b.ctypes.data, None, m, 0, p._h)
            a[:] = 0; b[:] = 0
            pos += m
        p.finish()
        assert p.bits(0, 1) == rec["bits490"] and p.bits(0, 0) == rec["bits518"]
        assert sorted([f, b, m] for (_s, f, b, m) in p.messages) == sorted(rec["messages"])


def test_capture_loop_program_links_and_decodes(nv, tmp_path):
    """The reference-shaped main program (tests/harness/capt_loop.c: capt_sched.c's ring,
    callback and consumer loop, its own add_message) linked against libnavtex_amd.so."""
    rec = GOLD["iq"]["two_carrier"]
    iq = cases.make_iq(nv, rec["spec"])              # as it is: the program ends the stream with nvx_shim_finish
    data = tmp_path / "iq.bin"
    iq.tofile(data)
    exe = tmp_path / "capt_loop"
    lib = ROOT / "navtex_amd"
    subprocess.run(["gcc", "-O2", str(ROOT / "tests" / "harness" / "capt_loop.c"), "-o", str(exe), f"-L{lib}", "-lnavtex_amd",
                    f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe), str(data)], check=True, capture_output=True, text=True, timeout=300).stdout
    got = []
    for line in out.splitlines():
        if "|" in line and line.split("|")[0].isdigit():
            f, b, m = line.split("|", 2)
            got.append([int(f), b, m.replace("\\n", "\n")])
    assert sorted(got) == sorted(rec["messages"])


def test_set_trace_on_a_handle_delivers_the_character_layers_text(nv):
    """nvx_set_trace: the text the reference's character layer prints (receiver/nav_b_sm.C) for the handle's chains, as the
    launches are collected -- equal to the stand-alone character layer's trace on the same bits; NULL turns it off."""
    rec = GOLD["iq"]["weak_518"]
    iq = cases.make_iq(nv, rec["spec"])                  # as it is: ended by nvx_finish at its true length
    text = []
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=3, push_mode=True, char_layer=True) as p:
        p.set_trace(text.append)
        half = (len(iq) // nv.FRAME_IN // 2) * nv.FRAME_IN
        p.push(0, iq[:half]); p.flush()
        n_half = len("".join(text))
        s = nv.Sitor(518, trace=True); s.feed(p.bits(0, 0))
        assert s.trace() == "".join(text) and "phasing detected" in s.trace()       # the text so far, exactly
        p.set_trace(None)
        p.push(0, iq[half:]); p.finish()
        assert len("".join(text)) == n_half and n_half > 0          # nothing more once it is off
        assert p.bits(0, 0) == rec["bits518"]
    # ... and left on to the end of the input: the whole trace of the character layer on the compiled reference's bits
    text = []
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=3, push_mode=True, char_layer=True) as p:
        p.set_trace(text.append)
        p.push(0, iq); p.finish()
        s = nv.Sitor(518, trace=True); s.feed(rec["bits518"])
        assert "".join(text) == s.trace()


def test_singleton_prints_the_reference_trace_when_asked(nv, tmp_path):
    """NAVTEX_AMD_TRACE=1: the reference-shaped surface prints what the reference's character layer prints to stdout
    (receiver/nav_b_sm.C: "phasing detected", "START OF MESSAGE", "line added: ...", "END OF MESSAGE", ...).  With ONE chain
    carrying traffic (golden cases weak_518, offset_490) the program's output, its own message lines removed, is exactly
    the character layer's trace on the golden bits -- which tests/test_host_layer.py pins to the compiled reference's stdout;
    with two carriers the chains' texts come frame by frame and all the reference's landmarks are there.  Off by default."""
    import re
    exe = tmp_path / "capt_loop"
    lib = ROOT / "navtex_amd"
    subprocess.run(["gcc", "-O2", str(ROOT / "tests" / "harness" / "capt_loop.c"), "-o", str(exe), f"-L{lib}", "-lnavtex_amd",
                    f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)

    def run(case, trace):
        rec = GOLD["iq"][case]
        data = tmp_path / f"{case}.bin"
        cases.make_iq(nv, rec["spec"]).tofile(data)
        env = dict(os.environ, NAVTEX_AMD_TRACE="1") if trace else {k: v for k, v in os.environ.items() if k != "NAVTEX_AMD_TRACE"}
        return rec, subprocess.run([str(exe), str(data)], check=True, capture_output=True, text=True, timeout=300, env=env).stdout

    for case, tag, freq in (("weak_518", "bits518", 518), ("offset_490", "bits490", 490)):
        rec, out = run(case, True)
        s = nv.Sitor(freq, trace=True)
        s.feed(rec[tag])
        want = s.trace()
        got = re.sub(r"(?m)^\d+\|[^\n]*\n", "", out)            # without the program's own "freq|bbbb|text" lines
        # (the program ends its file with nvx_shim_finish, at its true length: exactly the reference's bits, so exactly its text)
        assert "phasing detected" in want and got == want, case
    rec, out = run("two_carrier", True)
    for landmark in ("phasing detected", "START OF MESSAGE", "line added", "END OF MESSAGE"):
        assert out.count(landmark) >= 2, landmark
    _rec, quiet = run("two_carrier", False)
    assert "phasing detected" not in quiet and re.sub(r"(?m)^\d+\|[^\n]*\n", "", quiet) == ""


def test_poll_takes_in_finished_work_without_waiting(nv, oracle):
    """nvx_poll: results of launches that have finished reach the host (bits, messages) without a fetch, a flush or a
    further launch; a poll right behind a launch returns at once whether or not the GPU is done."""
    import time
    import signals
    st, _ = signals.stream_params(nv, 4300, nv.RATE_IN)
    n_frames = 75
    iq = nv.synth_host(st, nv.RATE_IN, n_frames * nv.FRAME_IN)
    buf = nv.DeviceBuffer(iq.nbytes); buf.upload(iq)
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=n_frames) as p:
        p.process_resident(buf, n_frames * nv.FRAME_IN, 0, n_frames)
        t0 = time.perf_counter(); p.poll(); dt = time.perf_counter() - t0
        assert dt < 0.05                                    # never waits
        deadline = time.time() + 5.0
        while p.bit_count(0, 0) == 0 and time.time() < deadline:
            time.sleep(0.01); p.poll()
        ref = oracle.Pipe(chain_mask=1, charlayer=False); ref.push(iq)
        assert p.bits(0, 0) == ref.bits(0) and len(p.messages) == 1          # all of it, and the message, with no fetch
        p.poll()                                            # nothing in flight: a no-op
    buf.free()


def test_messages_reach_add_message_without_a_flush(nv, tmp_path):
    """An unmodified capt_sched.c never calls nvx_shim_flush: the same program, ending WITHOUT the
    flush, still sees every message of the frames that were launched -- the singleton's housekeeping thread takes in
    finished launches every 50 ms (nvx_poll).  Two frames of silence behind the signal stand for the band going quiet."""
    rec = GOLD["iq"]["two_carrier"]
    iq = cases.make_iq(nv, rec["spec"])
    # (the silence is the scenario -- the band goes quiet and no flush ever comes -- not padding: what completes a message
    # must have been LAUNCHED, and without a flush only whole frames are)
    iq = np.vstack([iq, np.zeros((3 * nv.FRAME_IN + 4096, 2), dtype=np.int16)])
    data = tmp_path / "iq.bin"
    iq.tofile(data)
    exe = tmp_path / "capt_loop"
    lib = ROOT / "navtex_amd"
    subprocess.run(["gcc", "-O2", str(ROOT / "tests" / "harness" / "capt_loop.c"), "-o", str(exe), f"-L{lib}", "-lnavtex_amd",
                    f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    def messages(*extra, env=None):
        out = subprocess.run([str(exe), str(data), *extra], check=True, capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, **(env or {}))).stdout
        return sorted([int(l.split("|")[0]), l.split("|")[1], l.split("|", 2)[2].replace("\\n", "\n")]
                      for l in out.splitlines() if "|" in l and l.split("|")[0].isdigit())
    assert messages("noflush") == sorted(rec["messages"])
    # (without the housekeeping -- NAVTEX_AMD_NO_KEEPER=1 -- the results of the last launches wait for a later launch or a
    # flush; whether the messages are among them depends on timing, so only the switch itself is exercised here)
    assert len(messages("noflush", env={"NAVTEX_AMD_NO_KEEPER": "1"})) <= len(rec["messages"])


def test_three_carriers_config2(nv, oracle):
    """configs[2]: stream A carries 518 (+14 k) and 490 (-14 k); stream B carries the
    4209.5 kHz service on its +14 k chain, labelled 4209 (the reference has no such label)."""
    n_frames = 66
    b518 = nv.sitor_encode("ZCZC EA01\nFIVE ONE EIGHT\nNNNN\n", 40)
    b490 = nv.sitor_encode("ZCZC LB02\nFOUR NINE ZERO\nNNNN\n", 42)
    b4209 = nv.sitor_encode("ZCZC XC03\nFOUR TWO ZERO NINE DECIMAL FIVE\nNNNN\n", 44)
    sa = nv.make_stream([dict(freq_hz=14000, bits=b518, bit_offset=301, phase0=1), dict(freq_hz=-14000, bits=b490, bit_offset=1701, phase0=2)],
                        seed=11, noise_amp=1200)
    sb = nv.make_stream([dict(freq_hz=14000, bits=b4209, bit_offset=997, phase0=3)], seed=12, noise_amp=1200)
    n = n_frames * nv.FRAME_IN
    ia, ib = nv.synth_host(sa, nv.RATE_IN, n), nv.synth_host(sb, nv.RATE_IN, n)
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_masks=[3, 1], labels=[[518, 490], [4209, 0]], max_frames=3, push_mode=True) as p:
        for k in range(0, n, 50000):                  # interleaved pushes of the two streams
            p.push(0, ia[k:k + 50000]); p.push(1, ib[k:k + 50000])
        p.flush()
        got = sorted((s, f, b) for (s, f, b, _m) in p.messages)
        assert got == [(0, 490, "LB02"), (0, 518, "EA01"), (1, 4209, "XC03")]
        for s, iq, mask in ((0, ia, 3), (1, ib, 1)):
            ref = oracle.Pipe(chain_mask=mask, charlayer=False)
            ref.push(iq)
            for c in range(2):
                assert p.bits(s, c) == (ref.bits(c) if (mask >> c) & 1 else "")


def test_launch_partitioning_is_invisible(nv):
    """Size-independent property: one launch of 6 frames == six launches of 1 frame ==
    2+4, on device-resident data; reset reproduces the run exactly."""
    import signals
    n_streams, frames = 40, 6
    streams = [signals.stream_params(nv, 1000 + s, nv.RATE_RAW)[0] for s in range(n_streams)]
    streams[7] = streams[3]                           # identical streams must give identical bits
    pitch = frames * nv.FRAME_RAW
    buf = nv.DeviceBuffer(n_streams * pitch * 4)
    nv.synth_device(streams, nv.RATE_RAW, pitch, buf, pitch)
    outs = []
    with nv.Pipeline(n_streams=n_streams, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=frames, char_layer=False) as p:
        for plan in ([6], [1] * 6, [2, 4], [6]):
            p.reset()
            f0 = 0
            for k in plan:
                p.process_resident(buf, pitch, f0, k); f0 += k
            p.fetch()
            outs.append([p.bits(s, 0) for s in range(n_streams)])
    assert outs[0] == outs[1] == outs[2] == outs[3]
    assert outs[0][7] == outs[0][3] and len(outs[0][0]) > 100
    buf.free()


def test_results_arrive_behind_the_launches_without_a_fetch(nv):
    """A receiver that only ever calls process_resident (no fetch, no poll) still gets its messages: a launch takes in
    every earlier result that has finished, and a full result ring takes in its oldest one only.  Same bits and messages
    as one launch followed by a fetch."""
    import time
    import signals
    n_streams, frames = 3, 75                                      # 24 s: the whole message of every stream
    streams = [signals.stream_params(nv, 4200 + s, nv.RATE_RAW)[0] for s in range(n_streams)]
    pitch = frames * nv.FRAME_RAW
    buf = nv.DeviceBuffer(n_streams * pitch * 4)
    nv.synth_device(streams, nv.RATE_RAW, pitch, buf, pitch)
    with nv.Pipeline(n_streams=n_streams, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=frames) as p:
        p.process_resident(buf, pitch, 0, frames)
        p.fetch()
        want_bits = [p.bits(s, 0) for s in range(n_streams)]
        want_msgs = sorted(p.messages)
        assert len(want_msgs) == n_streams
        p.reset(); p.messages.clear()
        seen_before_fetch = 0
        for f in range(frames):                                    # 75 launches through a ring of four results
            p.process_resident(buf, pitch, f, 1)
            if f >= frames - 3: time.sleep(0.05)                   # the earlier launches have certainly finished by now
            seen_before_fetch = len(p.messages)
        assert seen_before_fetch == n_streams, "messages must not wait for a fetch"
        p.fetch()
        assert [p.bits(s, 0) for s in range(n_streams)] == want_bits
        assert sorted(p.messages) == want_msgs
    buf.free()


def test_callback_from_a_producer_thread_while_polling(nv, oracle):
    """SURVEY 8(f) rank 1 in miniature: a 'vendor' thread delivers jittered callbacks while
    the consumer thread polls bits -- the library's locking must keep the stream intact."""
    import threading
    import signals
    st, _ = signals.stream_params(nv, 77, nv.RATE_IN)
    n = 30 * nv.FRAME_IN
    iq = nv.synth_host(st, nv.RATE_IN, n)
    xi, xq = np.ascontiguousarray(iq[:, 0]), np.ascontiguousarray(iq[:, 1])
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True) as p:
        def producer():
            rng = np.random.default_rng(5)
            pos = 0
            while pos < n:
                m = int(min(n - pos, rng.integers(200, 5000)))
                nv.lib.nvx_StreamACallback(xi[pos:pos + m].ctypes.data, xq[pos:pos + m].ctypes.data, None, m, 0, p._h)
                pos += m
        t = threading.Thread(target=producer)
        t.start()
        seen = ""
        while t.is_alive():
            seen = p.bits(0, 0)                       # concurrent nvx_poll_bits
        t.join()
        p.flush()
        ref = oracle.Pipe(chain_mask=1, charlayer=False)
        ref.push(iq)
        assert p.bits(0, 0) == ref.bits(0) and ref.bits(0).startswith(seen)


def test_error_paths(nv):
    import signals
    with nv.Pipeline(n_streams=2, raw_rate=False, max_frames=1, push_mode=False) as p:
        with pytest.raises(nv.NvxError) as e:
            p.push(0, np.zeros((16, 2), dtype=np.int16))
        assert e.value.code == -5                      # NVX_ERR_STATE: not a push-mode handle
        buf = nv.DeviceBuffer(2 * nv.FRAME_IN * 4 * 2)
        with pytest.raises(nv.NvxError) as e:
            p.process_resident(buf, 2 * nv.FRAME_IN, 0, 2)      # n_frames > max_frames
        assert e.value.code == -1
        with pytest.raises(nv.NvxError) as e:
            p.process_resident(buf, 2 * nv.FRAME_IN + 2, 0, 1)  # pitch not a multiple of 4
        assert e.value.code == -1
        # a launch whose last stream would read past the end of the caller's buffer is refused BEFORE it is launched (a
        # faulting kernel can take the whole node down): the buffer holds 2 streams x 2 frames at this pitch
        p.process_resident(buf, 2 * nv.FRAME_IN, 1, 1); p.fetch()           # frames [1, 2) of both streams: the last byte of the buffer, fine
        for pitch, first in ((2 * nv.FRAME_IN, 2), (2 * nv.FRAME_IN + 4, 1), (4 * nv.FRAME_IN, 0)):
            with pytest.raises(nv.NvxError, match="leave the allocation") as e:
                p.process_resident(buf, pitch, first, 1)
            assert e.value.code == -1
        p.process_resident(buf, 2 * nv.FRAME_IN, 0, 1); p.fetch()           # ... and the handle is none the worse for it
        st, _ = signals.stream_params(nv, 1, nv.RATE_IN)
        with pytest.raises(nv.NvxError, match="leave the allocation"):
            nv.synth_device([st, st, st], nv.RATE_IN, 2 * nv.FRAME_IN, buf, 2 * nv.FRAME_IN)     # three rows into a buffer of two
        sub = nv.DeviceBuffer(8 * 64 * 4)
        assert nv.lib.nvx_channelise_resident(0, buf.ptr, 1024, 0, 1, 128, None, None, sub.ptr, 64, 0, None) == -1      # 128 outputs per sub-band into rows of 64
        assert b"leave the allocation" in nv.lib.nvx_last_error()
        sub.free()
        buf.free()
    with nv.Pipeline(n_streams=2, raw_rate=False, max_frames=1, push_mode=True) as p:
        blk = np.zeros((nv.FRAME_IN, 2), dtype=np.int16)
        assert nv.lib.nvx_push_iq(p._h, 0, None, 16) == -1 and nv.lib.nvx_push_planar(p._h, 0, None, blk.ctypes.data, 16) == -1      # null samples
        assert nv.lib.nvx_push_iq(p._h, 0, None, 0) == 0                                                                          # (nothing to read: fine)
        p.push(0, blk); p.push(0, blk)                 # staging holds max_frames + 1 frames per stream
        p.push(0, blk)                                  # stream 1 never delivered: since round 3 stream 0 goes on without it
        assert p.stream_stats(0)[1] >= 1 and p.stream_stats(1)[1] == 0 and p.stream_stats(0)[2] >= 1
        p.push(1, blk); p.push(1, blk); p.push(1, blk) # the lagging stream catches up
        p.flush()
        assert p.bit_count(0, 0) == p.bit_count(1, 0)
    with pytest.raises(nv.NvxError):
        nv.Pipeline(n_streams=1, chain_mask=0)
    with pytest.raises(nv.NvxError):
        nv.Pipeline(n_streams=1, device=99)


def _capture(nv, p, ring_seconds):
    cap = C.c_void_p()
    assert nv.lib.nvx_capture_start(p._h, 0, ring_seconds, C.byref(cap)) == 0
    return cap


def _stats(nv, cap):
    r, d, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    nv.lib.nvx_capture_stats(cap, C.byref(r), C.byref(d), C.byref(c))
    return r.value, d.value, c.value


def test_live_capture_ring_fake_sdr(nv, oracle):
    """the reference's producer / ring / consumer structure with a fake SDR
    thread: jittered numSamples, many ring wrap-arounds, no overrun -> bits identical."""
    import threading
    import time
    import signals
    st, _ = signals.stream_params(nv, 4242, nv.RATE_IN)
    n = 25 * nv.FRAME_IN
    iq = nv.synth_host(st, nv.RATE_IN, n)
    xi, xq = np.ascontiguousarray(iq[:, 0]), np.ascontiguousarray(iq[:, 1])
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True) as p:
        cap = _capture(nv, p, 0.5)                        # 126 000-sample ring: wraps ~16 times
        def vendor_thread():
            rng = np.random.default_rng(1)
            pos = 0
            while pos < n:
                m = int(min(n - pos, rng.integers(100, 3000)))
                # never outrun the consumer in this test: wait while the ring is nearly full
                while True:
                    r, d, c = _stats(nv, cap)
                    if r - d - c + m <= 100000: break
                    time.sleep(0.0005)
                nv.lib.nvx_capture_callback(xi[pos:pos + m].ctypes.data, xq[pos:pos + m].ctypes.data, None, m, 0, cap)
                pos += m
        t = threading.Thread(target=vendor_thread); t.start(); t.join()
        r, d, c = _stats(nv, cap)
        assert nv.lib.nvx_capture_stop(cap) == 0
        assert (r, d) == (n, 0)
        ref = oracle.Pipe(chain_mask=1, charlayer=False); ref.push(iq)
        assert p.bits(0, 0) == ref.bits(0) and len(ref.bits(0)) > 600


def test_live_capture_overrun_is_counted_not_silent(nv, oracle):
    """Overrun accounting: with the consumer stalled, samples beyond the ring are dropped and
    counted, and what is decoded is exactly the stream without them."""
    import signals
    st, _ = signals.stream_params(nv, 4343, nv.RATE_IN)
    n = 12 * nv.FRAME_IN
    iq = nv.synth_host(st, nv.RATE_IN, n)
    xi, xq = np.ascontiguousarray(iq[:, 0]), np.ascontiguousarray(iq[:, 1])
    def cb(cap, a, b):
        nv.lib.nvx_capture_callback(xi[a:b].ctypes.data, xq[a:b].ctypes.data, None, b - a, 0, cap)
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=1, push_mode=True) as p:
        cap = _capture(nv, p, 0.05)                       # 12 600 samples
        nv.lib.nvx_capture_pause(cap, 1)
        cb(cap, 0, 10000)                                 # fits
        cb(cap, 10000, 15000)                             # 2 600 fit, 2 400 are dropped
        r, d, c = _stats(nv, cap)
        assert (r, d, c) == (15000, 2400, 0)
        nv.lib.nvx_capture_pause(cap, 0)
        import time
        pos = 15000
        while pos < n:                                    # the rest at a pace the consumer follows
            m = min(4000, n - pos)
            while True:
                r, d, c = _stats(nv, cap)
                if r - d - c + m <= 12000: break
                time.sleep(0.0005)
            cb(cap, pos, pos + m); pos += m
        assert nv.lib.nvx_capture_stop(cap) == 0          # (drains the ring, flushes whole frames: the partial one stays staged)
        p.finish()                                        # ... and the capture is over: its last frame at its true length
        kept = np.vstack([iq[:12600], iq[15000:]])
        ref = oracle.Pipe(chain_mask=1, charlayer=False); ref.push(kept)
        assert p.bits(0, 0) == ref.bits(0) and len(ref.bits(0)) > 300


def test_a_stream_ended_under_a_running_capture_ring(nv, oracle):
    """The advisor's r5 scenario as it would happen in a receiver: nvx_stream_finish on a stream whose capture ring's consumer
    is still feeding it.  The consumer's push in progress runs to its end and is decoded, its next push is refused whole
    (NVX_ERR_STATE), the consumer stops and says why (nvx_capture_error); the bits are the oracle's on exactly the samples
    the ring had handed on; the other stream of the handle, fed directly, never notices."""
    import time
    import signals
    sts = [signals.stream_params(nv, 4500 + s, nv.RATE_IN)[0] for s in range(2)]
    n = 30 * nv.FRAME_IN
    iqs = [nv.synth_host(st, nv.RATE_IN, n) for st in sts]
    xi, xq = np.ascontiguousarray(iqs[0][:, 0]), np.ascontiguousarray(iqs[0][:, 1])
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True, char_layer=False, stall_timeout_ms=-1) as p:
        cap = _capture(nv, p, 0.5)
        pos = 0
        for k in range(10):                               # the first ten frames, at a pace the consumer follows; stream 1 beside it
            for _ in range(8):
                m = nv.FRAME_IN // 8
                while True:
                    r, d, c = _stats(nv, cap)
                    if r - d - c + m <= 100000: break
                    time.sleep(0.0005)
                nv.lib.nvx_capture_callback(xi[pos:pos + m].ctypes.data, xq[pos:pos + m].ctypes.data, None, m, 0, cap); pos += m
            p.push(1, iqs[1][k * nv.FRAME_IN:(k + 1) * nv.FRAME_IN])
        m = 1234                                          # (a ragged end, as a radio's last callback leaves it)
        nv.lib.nvx_capture_callback(xi[pos:pos + m].ctypes.data, xq[pos:pos + m].ctypes.data, None, m, 0, cap); pos += m
        deadline = time.time() + 5.0
        while _stats(nv, cap)[2] < pos and time.time() < deadline:
            time.sleep(0.002)
        p.finish(0)                                       # ... ended while the ring runs
        _r, _d, consumed_then = _stats(nv, cap)
        assert consumed_then == pos
        for _ in range(8):                                # the radio goes on delivering
            m = nv.FRAME_IN // 8
            nv.lib.nvx_capture_callback(xi[pos:pos + m].ctypes.data, xq[pos:pos + m].ctypes.data, None, m, 0, cap); pos += m
        deadline = time.time() + 5.0
        while nv.lib.nvx_capture_error(cap) == 0 and time.time() < deadline:
            time.sleep(0.005)
        assert nv.lib.nvx_capture_error(cap) == -5        # NVX_ERR_STATE: its stream has ended
        _r, d, consumed = _stats(nv, cap)
        assert d == 0 and consumed == consumed_then       # nothing was staged behind the end
        assert nv.lib.nvx_capture_stop(cap) == -5
        p.push(1, iqs[1][10 * nv.FRAME_IN:]); p.finish(1)                # the other stream goes on to its own end
        ref0 = oracle.Pipe(chain_mask=1, charlayer=False); ref0.push(iqs[0][:consumed])
        ref1 = oracle.Pipe(chain_mask=1, charlayer=False); ref1.push(iqs[1])
        assert p.bits(0, 0) == ref0.bits(0) and len(ref0.bits(0)) > 200
        assert p.bits(1, 0) == ref1.bits(0) and len(ref1.bits(0)) > 850
        assert p.integrity_stats()[:2] == (0, 0)


def test_empty_and_tiny_pushes(nv):
    with nv.Pipeline(n_streams=1, raw_rate=False, max_frames=1, push_mode=True) as p:
        p.push(0, np.zeros((0, 2), dtype=np.int16))      # empty push is a no-op
        p.flush()
        assert p.bits(0, 0) == "" and p.bit_count(0, 0) == 0
        p.push(0, np.zeros((1, 2), dtype=np.int16))      # less than a frame: staged, nothing launched
        p.flush()
        assert p.bits(0, 0) == ""


def test_bit_history_is_bounded_and_counts_stay_absolute(nv):
    """A receiver runs for weeks: the poll history per chain is capped (cfg.bit_history),
    the total count keeps running, a slow reader resumes at the oldest bit still held and
    what it reads is the tail of the full bit string; the character layer misses nothing."""
    import signals
    frames = 16                                      # 512 bit periods, the first 66 go to priming
    s = signals.stream_params(nv, 4242, nv.RATE_RAW)[0]
    pitch = frames * nv.FRAME_RAW
    buf = nv.DeviceBuffer(pitch * 4)
    nv.synth_device([s], nv.RATE_RAW, pitch, buf, pitch)
    hist = 64
    with nv.Pipeline(n_streams=1, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=1, char_layer=True) as full, \
         nv.Pipeline(n_streams=1, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=1, char_layer=True,
                     bit_history=hist) as capped:
        early = ""
        for f in range(frames):
            for p in (full, capped):
                p.process_resident(buf, pitch, f, 1)
                p.fetch()
            if f == 3:
                early = capped.bits(0, 0)            # a reader that keeps up loses nothing
        want = full.bits(0, 0)
        assert early == want[: len(early)] and len(early) > 0
        assert capped.bit_count(0, 0) == full.bit_count(0, 0) == len(want) > 4 * hist
        got = capped.bits(0, 0)                      # early bits + what is still held after the gap
        tail = got[len(early):]
        assert hist <= len(tail) <= 2 * hist + 32 and want.endswith(tail)
        assert capped.messages == full.messages
    buf.free()


def test_messages_land_in_the_sqlite_database(nv, tmp_path):
    """IQ in, rows out: the C character layer writes straight into the database the
    reference's web server reads (nvx_store_on_message as the handle's sink)."""
    import sqlite3
    rec = GOLD["iq"]["two_carrier"]
    iq = cases.make_iq(nv, rec["spec"])
    db = str(tmp_path / "Navtex.db")
    with nv.Store(db) as st, nv.Pipeline(n_streams=1, raw_rate=False, max_frames=4, push_mode=True, store=st) as p:
        p.push(0, iq)
        p.finish()
        assert st.stats() == (len(rec["messages"]), 0) and p.messages == []
    con = sqlite3.connect(db)
    got = con.execute("select freq,bbbb,message,age from messages order by id").fetchall()
    con.close()
    assert sorted([f, b, m] for (f, b, m, _a) in got) == sorted(rec["messages"])
    assert all(a == "NEW" for (*_x, a) in got) and len(got) >= 2


def test_capture_debug_recording_reproduces_the_input(nv, tmp_path):
    """The reference's debug_mode (capt_sched.c:87-101, 516): what the consumer hands to the
    DSP is also written to a 252 kHz 2-channel WAV; replaying that file gives the same bits."""
    import signals
    st, _ = signals.stream_params(nv, 77, nv.RATE_IN)
    n = 6 * nv.FRAME_IN
    iq = nv.synth_host(st, nv.RATE_IN, n)
    xi, xq = np.ascontiguousarray(iq[:, 0]), np.ascontiguousarray(iq[:, 1])
    path = str(tmp_path / "NTcapture.wav")
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True) as p:
        cap = _capture(nv, p, 8.0)                        # the reference's 8 s ring: no overrun here
        assert nv.lib.nvx_capture_record(cap, str(tmp_path / "no_dir" / "x.wav").encode()) < 0
        assert nv.lib.nvx_capture_record(cap, path.encode()) == 0
        for pos in range(0, n, 2520):
            m = min(2520, n - pos)
            nv.lib.nvx_capture_callback(xi[pos:pos + m].ctypes.data, xq[pos:pos + m].ctypes.data, None, m, 0, cap)
        assert nv.lib.nvx_capture_stop(cap) == 0          # drains, flushes, closes the recording
        live = p.bits(0, 0)
    got, rate = nv.wav_read(path)
    assert rate == nv.RATE_IN and np.array_equal(got, iq)
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True) as p:
        assert p.decode_wav(path) == 6
        assert p.bits(0, 0) == live and len(live) > 100


def test_a_capture_that_ends_is_finished_exactly(nv):
    """The live path's end: a capture ring fed the `ragged_length` golden input (1 s + 1237 samples) in callback-sized
    pieces, nvx_capture_stop (drains the ring, flushes whole frames; the partial frame stays staged), then nvx_finish:
    bits and messages are exactly the compiled reference's -- and the same for a recording of two carriers."""
    for name in ("ragged_length", "two_carrier"):
        rec = GOLD["iq"][name]
        iq = cases.make_iq(nv, rec["spec"])
        xi, xq = np.ascontiguousarray(iq[:, 0]), np.ascontiguousarray(iq[:, 1])
        rng = np.random.default_rng(3)
        with nv.Pipeline(n_streams=1, raw_rate=False, max_frames=2, push_mode=True) as p:
            cap = nv.Capture(p, 0, 8.0)
            pos = 0
            while pos < xi.size:
                m = int(min(xi.size - pos, rng.integers(200, 3000)))
                cap.feed(xi[pos:pos + m], xq[pos:pos + m]); pos += m
            cap.stop()
            early = p.bits(0, 0)                         # whole frames only so far: a proper prefix of the reference's bits
            assert rec["bits518"].startswith(early) and len(early) < len(rec["bits518"])
            p.finish()
            assert p.bits(0, 0) == rec["bits518"] and p.bits(0, 1) == rec["bits490"], name
            assert got_messages(p) == gold_messages(rec)


def test_two_receivers_two_capture_rings_one_handle(nv, oracle):
    """Two SDRs on one GPU handle: each has its own ring and vendor thread (jittered callbacks,
    different signals); a launch happens whenever both streams have a whole frame staged."""
    import threading
    import time
    import signals
    n = 10 * nv.FRAME_IN
    sigs = [signals.stream_params(nv, seed, nv.RATE_IN)[0] for seed in (901, 902)]
    iqs = [nv.synth_host(s, nv.RATE_IN, n) for s in sigs]
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True) as p:
        caps = []
        for s in range(2):
            cap = C.c_void_p()
            assert nv.lib.nvx_capture_start(p._h, s, 2.0, C.byref(cap)) == 0
            caps.append(cap)
        def vendor(s):
            xi, xq = np.ascontiguousarray(iqs[s][:, 0]), np.ascontiguousarray(iqs[s][:, 1])
            rng = np.random.default_rng(s)
            pos = 0
            while pos < n:
                m = int(min(n - pos, rng.integers(500, 5000)))
                # neither radio runs more than about a frame ahead of the other (they share a sample clock)
                while pos - min(progress) > nv.FRAME_IN:
                    time.sleep(0.0005)
                nv.lib.nvx_capture_callback(xi[pos:pos + m].ctypes.data, xq[pos:pos + m].ctypes.data, None, m, 0, caps[s])
                pos += m
                progress[s] = pos
        progress = [0, 0]
        threads = [threading.Thread(target=vendor, args=(s,)) for s in range(2)]
        for t in threads: t.start()
        for t in threads: t.join()
        for s in range(2):
            r, d, c = _stats(nv, caps[s])
            assert (r, d) == (n, 0)
            assert nv.lib.nvx_capture_stop(caps[s]) == 0
        for s in range(2):
            ref = oracle.Pipe(chain_mask=1, charlayer=False); ref.push(iqs[s])
            assert p.bits(s, 0) == ref.bits(0) and len(ref.bits(0)) > 200
        assert not np.array_equal(iqs[0], iqs[1])        # different signals (the first seconds are phasing either way)
