"""GPU tests of the multi-device group (header section C') on ONE GPU: two or three members on device 0.
The claim under test: sharding streams over members changes nothing -- bits and messages are those of one handle
holding all the streams (the chains share no state: receiver/decoder.h:31-60, receiver/nav_b_sm.h:92-114)."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


def _short_text(i):
    return f"ZCZC {chr(65 + i % 26)}{chr(65 + (i // 26) % 26)}{i % 100:02d}\nGRP {i}\nNNNN\n"


def _batch(nv, n_streams, n_frames, masks):
    """252 kS/s streams with a carrier on every decoded chain; returns (DeviceBuffer, pitch, list of host IQ)."""
    pitch = n_frames * nv.FRAME_IN
    streams = []
    for s in range(n_streams):
        carriers = []
        for c, f in ((0, 14000), (1, -14000)):
            if (masks[s] >> c) & 1:
                h = signals.mix32(7000 + 2 * s + c)
                carriers.append(dict(freq_hz=f, bits=nv.sitor_encode(_short_text(2 * s + c), 12), bit_offset=(h % 2520) | 1,
                                     phase0=signals.mix32(h), amplitude=6000))
        streams.append(nv.make_stream(carriers, seed=900 + s, noise_amp=1200))
    buf = nv.DeviceBuffer(n_streams * pitch * 4)
    nv.synth_device(streams, nv.RATE_IN, pitch, buf, pitch)
    return buf, pitch, streams


def test_group_of_two_members_equals_one_handle(nv):
    S, F = 24, 26
    masks = [(1, 2, 3)[s % 3] for s in range(S)]
    labels = [[1000 + s, 2000 + s] for s in range(S)]
    buf, pitch, _ = _batch(nv, S, F, masks)
    plan = [7, 7, 7, 5]
    with nv.Pipeline(n_streams=S, raw_rate=False, chain_masks=masks, labels=labels, max_frames=7) as one:
        f0 = 0
        for k in plan:
            one.process_resident(buf, pitch, f0, k); f0 += k
        one.fetch()
        want_bits = {(s, c): one.bits(s, c) for s in range(S) for c in range(2)}
        want_msgs = list(one.messages)
    assert len(want_msgs) >= S and any(len(b) > 700 for b in want_bits.values())
    with nv.Group([0, 0], n_streams=S, raw_rate=False, chain_masks=masks, labels=labels, max_frames=7) as g:
        assert g.members == [(0, 0, 12), (0, 12, 12)]
        ptrs = [buf.ptr + first * pitch * 4 for (_d, first, _n) in g.members]
        for rep in range(2):                                  # the second round after a reset: same result again
            f0 = 0
            for k in plan:
                g.process_resident(ptrs, pitch, f0, k); f0 += k
            g.fetch()
            got = {(s, c): g.bits(s, c) for s in range(S) for c in range(2)}
            assert got == want_bits, f"round {rep}: bits differ between the group and one handle"
            assert g.messages == want_msgs, f"round {rep}: messages (or their order) differ"
            g.reset()
    buf.free()


def test_uneven_split_three_members_and_the_index_map(nv, oracle):
    S, F = 7, 4
    masks = [1, 2, 1, 1, 2, 1, 2]
    buf, pitch, streams = _batch(nv, S, F, masks)
    with nv.Group([0, 0, 0], n_streams=S, raw_rate=False, chain_masks=masks, max_frames=F, char_layer=False) as g:
        assert g.members == [(0, 0, 3), (0, 3, 2), (0, 5, 2)]
        assert [g.member_of(s) for s in range(S)] == [0, 0, 0, 1, 1, 2, 2] and g.member_of(S) == -1 and g.member_of(-1) == -1
        g.process_resident([buf.ptr + f * pitch * 4 for (_d, f, _n) in g.members], pitch, 0, F)
        g.fetch()
        for s in range(S):
            ref = oracle.Pipe(chain_mask=masks[s], charlayer=False)
            ref.push(nv.synth_host(streams[s], nv.RATE_IN, pitch))
            c = 0 if masks[s] == 1 else 1
            assert g.bits(s, c) == ref.bits(c) and g.bits(s, 1 - c) == "" and g.bit_count(s, c) == len(ref.bits(c))
    buf.free()


def test_group_host_input_and_errors(nv, oracle):
    S, F = 4, 3
    masks = [3, 1, 2, 3]
    _buf, pitch, streams = _batch(nv, S, F, masks)
    _buf.free()
    iqs = [nv.synth_host(st, nv.RATE_IN, pitch) for st in streams]
    with nv.Group([0, 0], n_streams=S, raw_rate=False, chain_masks=masks, max_frames=2, push_mode=True, char_layer=False) as g:
        rng = np.random.default_rng(5)
        pos = [0] * S
        while any(p < pitch for p in pos):                     # streams advance in ragged steps, as capture threads would
            s = int(rng.integers(0, S))
            if pos[s] >= pitch:
                continue
            m = int(min(pitch - pos[s], rng.integers(1000, 60000)))
            try:
                g.push(s, iqs[s][pos[s]:pos[s] + m]); pos[s] += m
            except nv.NvxError as e:                           # a stream a whole staging set ahead of its member's slowest: back off
                assert e.code == -7
        g.flush()
        for s in range(S):
            ref = oracle.Pipe(chain_mask=masks[s], charlayer=False); ref.push(iqs[s])
            for c in range(2):
                assert g.bits(s, c) == (ref.bits(c) if (masks[s] >> c) & 1 else "")
        with pytest.raises(nv.NvxError) as e:
            g.process_resident([0, 0], pitch, 0, 1)
        assert e.value.code == -1
        with pytest.raises(nv.NvxError):
            g.push(S, iqs[0][:16])
    with pytest.raises(nv.NvxError):
        nv.Group([0, 0, 0], n_streams=2)                       # fewer streams than members


def test_group_finish_ends_every_members_streams_exactly(nv, oracle):
    """nvx_group_finish: five streams over two members, every stream's input ending at a place of its own (one on a frame
    boundary); pushes by global stream id, one finish for the whole group: every chain == the oracle on exactly its
    samples; every stream is ended afterwards and refuses more input -- the one that stopped on a frame boundary too."""
    import signals
    S = 5
    masks = [3, 1, 2, 3, 1]
    tails = [12345, 0, 279, 80639, 40000]
    iqs = []
    for s in range(S):
        car = [dict(freq_hz=f, bits=nv.sitor_encode(signals.stream_text(700 + s), 8), bit_offset=(311 * (s + 1)) | 1, phase0=s * 7919, amplitude=5000)
               for c, f in ((0, 14000), (1, -14000)) if (masks[s] >> c) & 1]
        iqs.append(nv.synth_host(nv.make_stream(car, seed=700 + s, noise_amp=1000), nv.RATE_IN, 3 * nv.FRAME_IN + tails[s]))
    with nv.Group([0, 0], n_streams=S, raw_rate=False, chain_masks=masks, max_frames=2, push_mode=True, char_layer=False) as g:
        for s in range(S):
            for pos in range(0, iqs[s].shape[0], 50000):
                g.push(s, iqs[s][pos:pos + 50000])
        g.finish()
        for s in range(S):
            ref = oracle.Pipe(chain_mask=masks[s], charlayer=False); ref.push(iqs[s])
            for c in range(2):
                assert g.bits(s, c) == (ref.bits(c) if (masks[s] >> c) & 1 else ""), (s, c)
            with pytest.raises(nv.NvxError):
                g.push(s, iqs[s][:16])


def test_member_threads_bind_to_the_device_numa_node(nv):
    """nvx_bind_thread_to_device: never widens the affinity mask, never fails on a box without NUMA information."""
    import os, threading
    before = os.sched_getaffinity(0)
    out = {}
    def run():
        out["n"] = nv.lib.nvx_bind_thread_to_device(0)
        out["after"] = os.sched_getaffinity(threading.get_native_id())
    t = threading.Thread(target=run); t.start(); t.join()
    assert out["n"] >= 0 and out["after"] <= before
    assert out["n"] in (0, len(out["after"]))
    assert os.sched_getaffinity(0) == before                   # the calling thread of the test is untouched


def test_group_two_physical_devices(nv):
    """Members on devices 0 and 1 == one handle of all the streams on device 0: the per-device launch cache
    (nvx_cascade.hip), the per-member hipSetDevice discipline and the members' NUMA binding where they matter.
    Skipped on a one-GPU box; any multi-GPU box the suite lands on runs it."""
    if nv.device_count() < 2:
        pytest.skip("needs two physical devices")
    S, F = 26, 14
    masks = [(1, 2, 3)[s % 3] for s in range(S)]
    labels = [[1000 + s, 2000 + s] for s in range(S)]
    buf, pitch, streams = _batch(nv, S, F, masks)
    plan = [5, 5, 4]
    with nv.Pipeline(n_streams=S, raw_rate=False, chain_masks=masks, labels=labels, max_frames=5) as one:
        f0 = 0
        for k in plan:
            one.process_resident(buf, pitch, f0, k); f0 += k
        one.fetch()
        want_bits = {(s, c): one.bits(s, c) for s in range(S) for c in range(2)}
        want_msgs = list(one.messages)
    with nv.Group([0, 1], n_streams=S, raw_rate=False, chain_masks=masks, labels=labels, max_frames=5) as g:
        assert g.members == [(0, 0, 13), (1, 13, 13)]
        # member 1's shard lives on ITS device, generated there from the same descriptors
        b1 = nv.DeviceBuffer(13 * pitch * 4, device=1)
        nv.synth_device(streams[13:], nv.RATE_IN, pitch, b1, pitch)
        ptrs = [buf.ptr, b1.ptr]
        for rep in range(2):
            f0 = 0
            for k in plan:
                g.process_resident(ptrs, pitch, f0, k); f0 += k
            g.fetch()
            got = {(s, c): g.bits(s, c) for s in range(S) for c in range(2)}
            assert got == want_bits, f"round {rep}: bits differ between two devices and one"
            assert g.messages == want_msgs, f"round {rep}: messages (or their order) differ"
            g.reset()
        b1.free()
    buf.free()
