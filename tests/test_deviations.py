"""Where the product deliberately does NOT behave like the reference (include/navtex_amd.h
section A): each deviation is pinned here from BOTH sides -- what the compiled reference does (oracle/_ref, build
container only: the `ref`-gated tests) and what the library documents and does (the GPU half is
tests/test_gpu_deviations.py).

  1. re-initialisation in mid-stream   receiver/fir1cpp.C:65-77, receiver/fir2cpp.C:90-110
  2. the input domain of sample_in_1   receiver/capt_sched.c:511 (the only caller passes int16 values)
  3. the 27.6-day sample counter       receiver/decoder.h:60, receiver/decoder.C:75,85
  4. the layout guard of nvx_config    (ABI hygiene, no reference counterpart)
"""
import ctypes as C
import json
import math
from pathlib import Path

import numpy as np
import pytest

import cases
import oracle_binding as ob

GOLD = json.loads((Path(__file__).parent / "golden" / "golden.json").read_text())
needs_ref = pytest.mark.skipif(not ob.have_ref(), reason="compiled reference (oracle/_ref) not present")
INT_MAX = 2 ** 31 - 1


# ------------------------------------------------------------------------------------------ 1. re-initialisation
@needs_ref
@pytest.mark.parametrize("which", [1, 2, 3], ids=["init_fir_filter1", "init_fir2_wrapper", "both"])
def test_reinit_in_mid_stream_is_a_partial_reset_in_the_reference(nv, oracle, which):
    """The reference's init functions called AGAIN in front of sample n (capt_sched.c calls each once, :552-555, :612):
    init_fir_filter1 clears FIR1's ring and counter only, init_fir2_wrapper the 518 chain's FIR2 ring and the mixer index
    -- FIR3, the 490 chain's FIR2 statics, both decoders and both character layers go on as they were.
    (a) the oracle's statement of exactly that (nvxo_pipe_reinit) gives the compiled reference's bits on both chains;
    (b) so the reference does NOT start a new stream there: it keeps emitting bits across the call -- within a bit of the
        uninterrupted run -- while a FRESH reference on the samples behind n needs the demodulator's priming again
        (63 bit periods of timing history, decoder.h:23) and is ~65 bits short;
    (c) the library follows the second reading: init_fir_filter1() is ALWAYS a new stream (header section A, DESIGN 4.4;
        tests/test_gpu_deviations.py::test_init_fir_filter1_starts_a_new_stream holds it to the fresh reference's bits)."""
    rec = GOLD["iq"]["two_carrier"]
    iq = cases.make_iq(nv, rec["spec"])
    n = iq.shape[0] // 2 + 1237                          # nowhere in particular: no multiple of 4, 28, 280, 2520
    reinit = ob.run_ref("bits", iq.tobytes(), ("reinit", n, which))
    p = oracle.Pipe(chain_mask=3, charlayer=False)
    p.push(iq[:n]); p.reinit(which); p.push(iq[n:])
    head, fresh = ob.run_ref("bits", iq[:n].tobytes()), ob.run_ref("bits", iq[n:].tobytes())
    for c, tag in ((0, "bits518"), (1, "bits490")):
        got = reinit[tag].decode()
        assert p.bits(c) == got, f"(a) {tag}"
        whole = rec[tag]                                 # the uninterrupted run (golden = compiled reference)
        m = min(len(got), len(whole)) - 1
        assert abs(len(got) - len(whole)) <= 1 and got[:m] == whole[:m], f"(b) {tag}: not a continuation"
        behind = len(got) - len(head[tag])               # bits the re-initialised reference decoded behind n
        assert behind - len(fresh[tag]) >= 60, f"(b) {tag}: {behind} vs {len(fresh[tag])} of a fresh start"


@needs_ref
def test_once_before_the_first_sample_every_reading_is_the_same(nv, oracle):
    """... and called once, before the first sample -- all capt_sched.c ever does -- "FIR1 cleared", "FIR2-518 cleared" and
    "everything zero" are one state: the reference re-initialised in front of sample 0 is the reference."""
    rec = GOLD["iq"]["offset_490"]
    iq = cases.make_iq(nv, rec["spec"])
    again = ob.run_ref("bits", iq.tobytes(), ("reinit", 0, 3))
    assert again["bits518"].decode() == rec["bits518"] and again["bits490"].decode() == rec["bits490"]


# ------------------------------------------------------------------------------------------ 2. input domain
def test_sample_in_1_input_domain_is_a_checked_conversion(nv):
    """nvx_sample_to_int16 -- the conversion sample_in_1 applies (nvx_shim.cpp): every int16 value comes back exactly
    and is reported in-domain (capt_sched.c:511 passes nothing else); anything else is rounded to the nearest int16,
    ties to even, clipped at the rails, NaN to 0, and reported off-domain -- defined for every double, where a plain
    (int16_t) cast is undefined behaviour outside the range.  No device needed."""
    out = C.c_int16(0)
    conv = lambda v: (nv.lib.nvx_sample_to_int16(C.c_double(v), C.byref(out)), out.value)
    for v in range(-32768, 32768):
        assert conv(float(v)) == (1, v)
    assert conv(-0.0) == (1, 0)
    for v, want in [(0.25, 0), (0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (32766.5, 32766), (32766.51, 32767),
                    (32767.2, 32767), (32768.0, 32767), (1e300, 32767), (math.inf, 32767), (-32768.4, -32768),
                    (-32769.0, -32768), (-1e300, -32768), (-math.inf, -32768), (math.nan, 0), (4.9e-324, 0)]:
        assert conv(v) == (0, want), v
    assert nv.lib.nvx_sample_to_int16(C.c_double(7.0), None) == 1          # the out pointer is optional
    rng = np.random.default_rng(5)
    for v in rng.uniform(-40000, 40000, 2000):
        ok, got = conv(float(v))
        assert ok == 0 and got == int(np.clip(np.rint(v), -32768, 32767))


# ------------------------------------------------------------------------------------------ 3. the 27.6-day counter
def _y3_of_a_long_clean_signal(nv, oracle, seconds=40):
    import signals
    st, _ = signals.stream_params(nv, 77, nv.RATE_IN)
    iq = nv.synth_host(st, nv.RATE_IN, 252000 * seconds)
    p = oracle.Pipe(chain_mask=1, charlayer=False, tap_y3=seconds * 900 + 10)
    p.push(iq)
    return p.y3(0).copy(), p.bits(0)


@pytest.mark.parametrize("margin", [200, 2000])
def test_the_references_sample_counter_overflows_after_27_days_and_silences_it(nv, oracle, margin):
    """decoder.h:60 `int bd_seq_nbr`, decoder.C:75 `bd_seq_nbr ++` for ever, decoder.C:85 `bd_seq_nbr % 9 == bit_sync_offset`:
    after 2^31 samples at 900 S/s (27.6 days) the counter passes INT_MAX, the remainders turn -8..0 and the test fails for
    every sync offset but 0.  The state is reached here by setting the counter `margin` samples below INT_MAX (same phase
    mod 9, so nothing else changes) in the middle of a clean 40 s signal:
    * the oracle (which keeps the `int`) decodes exactly the uninterrupted run's bits up to the overflow and then NOTHING
      for the remaining 25 s of signal;
    * [ref] the compiled reference, its private member set through the decoder seam, does the same, bit for bit;
    * the product has no such counter: its bit phase is the sample's position in its bit period (nvx_fsm.h), its
      bit-timing filter indexes by a 64-bit sample clock (nvx_demod.hip) -- nvx_fsm_selftest walks 3 million bit periods
      (8.3 hours of signal) here, tests/test_gpu_deviations.py walks a stream across sample 2^31 on the GPU."""
    y3, control = _y3_of_a_long_clean_signal(nv, oracle)
    assert len(control) > 3900
    k0 = 9000                                            # 10 s in: the decoder is synchronised and bits are flowing
    v = INT_MAX - margin
    v -= (v - k0) % 9                                    # bd_seq_nbr == k0 there: keep the phase
    bits, was = oracle.decode_inject(y3, k0, v)
    assert was == k0
    overflow_at = k0 + (INT_MAX - v)                     # the sample whose increment passes INT_MAX
    assert control.startswith(bits) and abs(len(bits) - (overflow_at // 9 - 66)) <= 3
    assert len(control) - len(bits) > 2700               # ... and 25 s of good signal decode to nothing
    if ob.have_ref():
        ref = ob.run_ref("dec", np.ascontiguousarray(y3).tobytes(), ("inject", k0, v))["bits518"].decode()
        assert ref == bits, "the compiled reference and the oracle disagree about the overflow"
    assert nv.lib.nvx_fsm_selftest(12345, 3_000_000) == 0


def test_the_counter_is_harmless_below_the_overflow(nv, oracle):
    """The probe itself changes nothing: the same injection 10 000 samples below INT_MAX, where the signal ends before the
    counter gets there, decodes the uninterrupted run's bits."""
    y3, control = _y3_of_a_long_clean_signal(nv, oracle, seconds=20)
    v = INT_MAX - 100_000
    v -= (v - 9000) % 9
    assert oracle.decode_inject(y3, 9000, v)[0] == control


# ------------------------------------------------------------------------------------------ 4. layout guard
def test_a_caller_built_against_another_layout_is_refused_not_read_past(nv):
    """nvx_config begins with struct_size (ABI 2): nvx_create / nvx_group_create compare it with their own sizeof before
    reading anything else, so a caller compiled against another navtex_amd.h -- or one that did not start from
    nvx_config_default -- gets NVX_ERR_ARG and a text that says so, on any machine (no device is touched before the check)."""
    N = nv.N if hasattr(nv, "N") else __import__("navtex_amd._native", fromlist=["x"])
    cfg = N.Config()
    nv.lib.nvx_config_default(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(N.Config) and nv.lib.nvx_abi_version() == 2
    assert nv.lib.nvx_version().decode().startswith("navtex_amd 2.")
    h = C.c_void_p()
    for bad in (0, cfg.struct_size - 4, cfg.struct_size + 8):
        cfg.struct_size = bad
        assert nv.lib.nvx_create(C.byref(cfg), C.byref(h)) == N.ERR_ARG and not h.value
        assert b"struct_size" in nv.lib.nvx_last_error()
        g = C.c_void_p()
        dev = (C.c_int * 1)(0)
        assert nv.lib.nvx_group_create(dev, 1, C.byref(cfg), C.byref(g)) == N.ERR_ARG and not g.value
