"""GPU half of tests/test_deviations.py: what the LIBRARY does where it deliberately differs from the reference
(include/navtex_amd.h section A).  The reference's side of each row is pinned in the build
container by the `ref`-gated tests there; here the comparison is with the oracle, which those tests tie to it."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((Path(__file__).parent / "golden" / "golden.json").read_text())


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    """tests/harness/shim_probe.c linked against libnavtex_amd.so alone (the reference-shaped surface, section A)."""
    exe = tmp_path_factory.mktemp("probe") / "shim_probe"
    lib = ROOT / "navtex_amd"
    subprocess.run(["gcc", "-O2", str(ROOT / "tests" / "harness" / "shim_probe.c"), "-o", str(exe), f"-L{lib}", "-lnavtex_amd",
                    f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)

    def run(*args):
        r = subprocess.run([str(exe), *[str(a) for a in args]], check=True, capture_output=True, text=True, timeout=300)
        out = {"bits518": [], "bits490": [], "msg": [], "stderr": r.stderr}
        for line in r.stdout.splitlines():
            key, _, rest = line.partition(" ")
            if key in ("bits518", "bits490", "msg"):
                out[key].append(rest)
            elif key == "stats":
                out["stats"] = tuple(int(v) for v in rest.split())
        return out
    return run


def test_init_fir_filter1_starts_a_new_stream(nv, oracle, probe, tmp_path):
    """Deviation 1.  The reference's init_fir_filter1() / init_fir2_wrapper() called again in mid-stream clear FIR1 and the
    518 chain's FIR2 and go on (receiver/fir1cpp.C:65-77, fir2cpp.C:90-110; tests/test_deviations.py pins that against the
    compiled reference).  Here they start a NEW stream: the bits afterwards are exactly those of a fresh receiver on the
    samples behind the call -- on both chains -- and nothing of the old stream's undecoded samples leaks into them."""
    rec = GOLD["iq"]["two_carrier"]
    iq = cases.make_iq(nv, rec["spec"])
    n = iq.shape[0] // 2 + 1237
    data = tmp_path / "iq.bin"; iq.tofile(data)
    got = probe("reinit", data, n)
    fresh = oracle.Pipe(chain_mask=3, charlayer=False)
    fresh.push(iq[n:])
    assert got["bits518"] == [fresh.bits(0)] and got["bits490"] == [fresh.bits(1)]
    assert len(fresh.bits(0)) > 900 and got["stats"] == (iq.shape[0], 0)


def test_off_domain_samples_are_rounded_and_counted(nv, probe, tmp_path):
    """Deviation 2.  sample_in_1 is handed value +- 0.25 on every 1000th sample: not int16 values, so outside the input
    domain (capt_sched.c:511 passes int16) -- each is rounded back to the int16 it came from and counted; bits and messages
    are the compiled reference's on the clean input, and the first violation leaves a line on stderr."""
    rec = GOLD["iq"]["offset_490"]
    iq = cases.make_iq(nv, rec["spec"])
    data = tmp_path / "iq.bin"; iq.tofile(data)
    got = probe("domain", data, 1000)
    assert got["bits518"] == [rec["bits518"]] and got["bits490"] == [rec["bits490"]]
    assert got["stats"] == (iq.shape[0], -(-iq.shape[0] // 1000))
    assert sorted(got["msg"]) == sorted(f"{f}|{b}" for f, b, _m in rec["messages"])
    assert got["stderr"].count("not an int16 value") == 1
    clean = probe("domain", data, 0)
    assert clean["stats"] == (iq.shape[0], 0) and clean["stderr"] == ""


def test_a_sample_behind_shim_finish_starts_a_new_stream(nv, probe, tmp_path):
    """nvx_shim_finish ends the input; the next sample_in_1 -- with no init_fir_filter1 in between -- is the first sample of
    a new stream (it used to abort the process: the ended stream refused the push).  The same file decoded twice in a
    row gives the compiled reference's bits and messages twice, also when the file is a whole number of frames long."""
    for name in ("ragged_length", "weak_518"):
        rec = GOLD["iq"][name]
        iq = cases.make_iq(nv, rec["spec"])
        data = tmp_path / f"{name}.bin"; iq.tofile(data)
        got = probe("refinish", data)
        assert got["bits518"] == [rec["bits518"]] * 2 and got["bits490"] == [rec["bits490"]] * 2, name
        assert sorted(got["msg"]) == sorted([f"{f}|{b}" for f, b, _m in rec["messages"]] * 2)
    whole = np.ascontiguousarray(iq[: 3 * nv.FRAME_IN])
    data = tmp_path / "whole.bin"; whole.tofile(data)
    got = probe("refinish", data)
    assert got["bits518"][0] == got["bits518"][1] and len(got["bits518"][0]) > 20


@pytest.mark.parametrize("power,blocks_before", [(31, 40), (32, 34)], ids=["2^31_int", "2^32_unsigned"])
@pytest.mark.parametrize("max_frames", [1, 12], ids=["front_walk", "front_tiles"])
def test_a_stream_walks_across_sample_2_to_31(nv, oracle, max_frames, power, blocks_before):
    """Deviation 3.  The reference's `int bd_seq_nbr` passes INT_MAX after 2^31 samples at 900 S/s = 27.6 days and its
    decoder falls silent (decoder.h:60, decoder.C:75,85; tests/test_deviations.py shows it on the compiled reference).
    Here the stream's sample clock is 64 bits wide everywhere it is used.  480 frames of signal, then the clock is put
    10.4 frames below 2^31 (nvx_debug_advance_clock: a whole number of periods of everything derived from it, nothing
    else touched), then 24 more frames: the stream crosses sample 2 147 483 648 in the middle of a frame and every bit of
    both chains is the oracle's on the uninterrupted signal -- in both forms of the demodulator's front (the walk that
    short launches use, head + tiles for long launches of few chains).  The hook's own seal re-tag is judged by the
    kernels: no launch failure, no repaired hand-over.  The same across 2^32 (55 days), where an unsigned 32-bit count
    would wrap."""
    import signals
    block_frames, blocks_after = 12, 2
    b518, b490 = nv.sitor_encode(signals.stream_text(4242), 40), nv.sitor_encode(signals.stream_text(4243), 40)
    st = nv.make_stream([dict(freq_hz=14000, bits=b518, bit_offset=1201, phase0=77, amplitude=7000),
                         dict(freq_hz=-14000, bits=b490, bit_offset=333, phase0=99, amplitude=6000)], seed=4242, noise_amp=1500)
    iq = nv.synth_host(st, nv.RATE_IN, block_frames * nv.FRAME_IN)
    ref = oracle.Pipe(chain_mask=3, charlayer=False)
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=3, max_frames=max_frames, push_mode=True, char_layer=False) as p:
        for _k in range(blocks_before):
            p.push(0, iq); ref.push(iq)
        p.flush()
        g = p.stream_stats(0)[1] * nv.FRAME_Y3
        assert g == block_frames * blocks_before * nv.FRAME_Y3
        periods = (2 ** power - g) // p.CLOCK_PERIOD
        p.debug_advance_clock(0, periods)
        g_new = p.stream_stats(0)[1] * nv.FRAME_Y3
        assert g_new == g + periods * p.CLOCK_PERIOD and 0 < 2 ** power - g_new < 11 * nv.FRAME_Y3
        for _k in range(blocks_after):
            p.push(0, iq); ref.push(iq)
        p.flush()
        assert p.stream_stats(0)[1] * nv.FRAME_Y3 > 2 ** power + 12 * nv.FRAME_Y3       # well across
        for c in (0, 1):
            want = ref.bits(c)
            assert len(want) > 32 * block_frames * (blocks_before + blocks_after) - 100 and p.bits(0, c) == want, f"chain {c}"
        assert p.integrity_stats()[:2] == (0, 0)


def test_clock_hook_refuses_what_it_cannot_do(nv):
    with nv.Pipeline(n_streams=1, wideband=True, chain_mask=3, max_frames=1, push_mode=True, char_layer=False) as p:
        with pytest.raises(nv.NvxError):
            p.debug_advance_clock(0, 1)
    with nv.Pipeline(n_streams=2, raw_rate=False, max_frames=1, push_mode=True, char_layer=False) as p:
        with pytest.raises(nv.NvxError):
            p.debug_advance_clock(2, 1)
        with pytest.raises(nv.NvxError):
            p.debug_advance_clock(0, 2 ** 63)                 # would pass 2^64


def test_one_of_two_streams_far_ahead_in_time(nv, oracle):
    """The same hook on ONE stream of a two-stream handle: the streams are then 27 days apart, launches carry lists with
    each stream's own clock, both decode the oracle's bits.  (A stream that is still priming is refused: the priming
    thresholds are the one thing derived from the clock that is not periodic.)"""
    import signals
    iqs, refs = [], []
    for s in range(2):
        st, _ = signals.stream_params(nv, 900 + s, nv.RATE_IN)
        iqs.append(nv.synth_host(st, nv.RATE_IN, 8 * nv.FRAME_IN))
        r = oracle.Pipe(chain_mask=1, charlayer=False); r.push(iqs[s]); refs.append(r)
    with nv.Pipeline(n_streams=2, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True, char_layer=False) as p:
        with pytest.raises(nv.NvxError):
            p.debug_advance_clock(1, 13150)                   # position 0: still priming
        head = 3 * nv.FRAME_IN
        p.push(0, iqs[0][:head]); p.push(1, iqs[1][:head]); p.flush()
        p.debug_advance_clock(1, 13150)
        for k in range(head, 8 * nv.FRAME_IN, 50000):
            p.push(0, iqs[0][k:k + 50000]); p.push(1, iqs[1][k:k + 50000])
        p.flush()
        assert p.stream_stats(1)[1] - p.stream_stats(0)[1] == 13150 * 567
        assert p.bits(0, 0) == refs[0].bits(0) and p.bits(1, 0) == refs[1].bits(0) and len(refs[1].bits(0)) > 150
        assert p.integrity_stats()[:2] == (0, 0)


def test_decode_wav_ends_its_stream_whatever_the_files_length(nv, tmp_path):
    """A file has an end: nvx_decode_wav leaves its stream ended also when the file is a whole number of frames long (the
    advisor's r5 finding: such a file used to leave the stream live, and a second file was silently decoded as the
    continuation of the first).  An empty file ends nothing."""
    rec = GOLD["iq"]["weak_518"]
    iq = np.ascontiguousarray(cases.make_iq(nv, rec["spec"])[: 6 * nv.FRAME_IN])
    path = str(tmp_path / "whole.wav"); nv.wav_write(path, iq, nv.RATE_IN)
    empty = str(tmp_path / "empty.wav"); nv.wav_write(empty, np.zeros((0, 2), dtype=np.int16), nv.RATE_IN)
    with nv.Pipeline(n_streams=1, raw_rate=False, max_frames=2, push_mode=True, char_layer=False) as p:
        assert p.decode_wav(empty) == 0
        assert p.decode_wav(path) == 6
        first = p.bits(0, 0)
        with pytest.raises(nv.NvxError):
            p.decode_wav(path)                                # ended: the second file is refused, not appended
        with pytest.raises(nv.NvxError):
            p.push(0, iq[:100])
        p.stream_reset(0)
        assert p.decode_wav(path) == 6 and p.bits(0, 0) == first and len(first) > 100
