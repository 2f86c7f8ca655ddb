"""Shared, fully deterministic test-case definitions.

Used by tests/golden/make_golden.py (to record what the compiled reference does
with them) and by the tests (to regenerate the same inputs anywhere, without
the reference).  Only integer arithmetic and individually-rounded IEEE
operations are used, so every machine regenerates bit-identical inputs."""
from __future__ import annotations

import numpy as np

RATE_IN = 252000

# --------------------------------------------------------------------------
# full-path IQ cases at 252 kS/s (what the reference consumes)
# --------------------------------------------------------------------------
IQ_CASES = {
    # SURVEY probe scenario: both NAVTEX carriers in one stream, different texts
    "two_carrier": {
        "kind": "synth", "seconds": 22, "seed": 7, "noise_amp": 1500,
        "carriers": [
            {"freq_hz": 14000, "text": "ZCZC EA01\nTEST MESSAGE 123 OK\nNNNN\n", "n_phasing": 40,
             "bit_offset": 777, "phase0": 12345678, "amplitude": 8000},
            {"freq_hz": -14000, "text": "ZCZC GB42\nGALE WARNING 7/8 NW-LY.\nNNNN\n", "n_phasing": 45,
             "bit_offset": 1999, "phase0": 987654321, "amplitude": 6000},
        ]},
    # weak carrier: decision errors, RX/DX fall-backs and '*' marks
    "weak_518": {
        "kind": "synth", "seconds": 26, "seed": 99, "noise_amp": 9000,
        "carriers": [
            {"freq_hz": 14000, "text": "ZCZC KA77\nWEAK SIGNAL TEST 0123456789 (A/B) = ?\nSECOND LINE: 50-10N 002-30W\nNNNN\n",
             "n_phasing": 50, "bit_offset": 1301, "phase0": 55555, "amplitude": 1500},
        ]},
    # frequency error and a strong signal next to full scale
    "offset_490": {
        "kind": "synth", "seconds": 20, "seed": 1234, "noise_amp": 3000,
        "carriers": [
            {"freq_hz": -13988, "text": "ZCZC PD15\nCARRIER 12 HZ HIGH\nNNNN\n", "n_phasing": 40,
             "bit_offset": 2519, "phase0": 4000000000, "amplitude": 28000},
        ]},
    "noise_only": {"kind": "synth", "seconds": 6, "seed": 31337, "noise_amp": 12000, "carriers": []},
    "silence": {"kind": "zeros", "seconds": 3},
    "constant": {"kind": "const", "seconds": 3, "i": -5000, "q": -3000},
    "full_scale_random": {"kind": "random", "seconds": 3, "seed": 5},
    "ragged_length": {     # not a multiple of anything: 1 s + 1237 samples
        "kind": "synth", "samples": 252000 + 1237, "seed": 8, "noise_amp": 500,
        "carriers": [{"freq_hz": 14000, "text": "ZCZC AA00\nX\nNNNN\n", "n_phasing": 10, "bit_offset": 3,
                      "phase0": 1, "amplitude": 9000}]},
}


def make_iq(nv, spec) -> np.ndarray:
    n = spec.get("samples", spec.get("seconds", 0) * RATE_IN)
    kind = spec["kind"]
    if kind == "zeros":
        return np.zeros((n, 2), dtype=np.int16)
    if kind == "const":
        a = np.empty((n, 2), dtype=np.int16); a[:, 0] = spec["i"]; a[:, 1] = spec["q"]
        return a
    if kind == "random":
        rng = np.random.default_rng(spec["seed"])
        return rng.integers(-32768, 32768, size=(n, 2), dtype=np.int16)
    carriers = []
    for c in spec["carriers"]:
        carriers.append(dict(freq_hz=c["freq_hz"], bits=nv.sitor_encode(c["text"], c["n_phasing"]),
                             bit_offset=c["bit_offset"], phase0=c["phase0"], amplitude=c["amplitude"]))
    st = nv.make_stream(carriers, seed=spec["seed"], noise_amp=spec["noise_amp"])
    return nv.synth_host(st, RATE_IN, n)


# --------------------------------------------------------------------------
# decoder-only cases: 900 S/s complex inputs
# --------------------------------------------------------------------------
DECODER_CASES = {
    "dyadic_noise": {"kind": "noise", "n": 9000, "seed": 11},
    "fsk_rotor": {"kind": "rotor", "n": 18000, "seed": 12, "noise": 64},
    "fsk_rotor_noisy": {"kind": "rotor", "n": 18000, "seed": 13, "noise": 700},
    "zeros_then_noise": {"kind": "zeros_noise", "n": 5000, "seed": 14},
    "tiny_values": {"kind": "noise", "n": 4000, "seed": 15, "scale_exp": -300},
}

# one 900 S/s step of +-85 Hz: cos/sin(2 pi 85/900) as fixed literals
_ROT_C = float.fromhex("0x1.a8794265ef1cap-1")
_ROT_S = float.fromhex("0x1.1e4b5a2e7f2ebp-1")


def make_y3(spec) -> np.ndarray:
    rng = np.random.default_rng(spec["seed"])
    n = spec["n"]
    kind = spec["kind"]
    if kind in ("noise", "zeros_noise"):
        y = rng.integers(-(1 << 20), 1 << 20, size=(n, 2)).astype(np.float64) / 1024.0
        if "scale_exp" in spec:
            y = np.ldexp(y, spec["scale_exp"])
        if kind == "zeros_noise":
            y[: n // 3] = 0.0
            y[n // 3: n // 3 + 40, 1] = 0.0          # purely real stretch: atan2 axis cases
            y[n // 3 + 40: n // 3 + 60, 0] *= -1.0
        return y
    # rotor: a unit phasor advanced by +-85 Hz per sample, 9 samples per bit, random bits,
    # every operation a single IEEE multiply/add in Python floats
    bits = rng.integers(0, 2, size=n // 9 + 2)
    noise = rng.integers(-spec["noise"], spec["noise"] + 1, size=(n, 2))
    re, im = 4096.0, 0.0
    y = np.empty((n, 2))
    for k in range(n):
        s = _ROT_S if bits[(k + 4) // 9] else -_ROT_S
        re, im = re * _ROT_C - im * s, re * s + im * _ROT_C
        y[k, 0] = re + float(noise[k, 0])
        y[k, 1] = im + float(noise[k, 1])
    return y


# --------------------------------------------------------------------------
# character-layer cases: 'B'/'Y' strings
# --------------------------------------------------------------------------
def _codes(bits: str):
    return [bits[i:i + 7] for i in range(0, len(bits) - 6, 7)]


def _code_bits(code: int) -> str:
    return "".join("Y" if (code >> i) & 1 else "B" for i in range(6, -1, -1))


CHAR_CASES = {
    "clean": {"kind": "text", "text": "ZCZC EA01\nTEST MESSAGE 123 OK\nNNNN\n", "n_phasing": 40},
    "figures": {"kind": "text", "n_phasing": 12,
                "text": "ZCZC QX09\n011200 UTC JAN 24 = WIND: NW 7/8, (GUSTS 45 KT) 'SEA' 3.5 M + SWELL? YES-NO\nA1B2C3 D4 E5\nNNNN\n"},
    "two_messages": {"kind": "text", "n_phasing": 20,
                     "text": "ZCZC AB12\nFIRST\nNNNN\nZCZC CD34\nSECOND MESSAGE\nNNNN\n"},
    "double_header": {"kind": "text", "n_phasing": 20,
                      "text": "ZCZC AB12\nFIRST WITHOUT END\nZCZC CD34\nSECOND\nNNNN\nZCZC EF56\nTHIRD\nNNNN\n"},
    "mangled_markers": {"kind": "text", "n_phasing": 20,
                        "text": "ZXZC  GH78\nSOM WITH ONE BAD LETTER AND TWO BLANKS\nNXNN\nCZC IJ90\nSHORT SOM\nNNXN TAIL\n"
                                "ZCXC KL11\nTHIRD FORM\nNNN\nZCZC MN\nNO DIGITS\nZCZC OP2\nNNNN\n"},
    "rx_errors": {"kind": "text_flip", "n_phasing": 30, "seed": 21, "flip_rx": 0.3, "flip_dx": 0.0,
                  "text": "ZCZC RX01\nERRORS IN THE REPEAT ONLY 0123456789\nNNNN\n"},
    "dx_errors": {"kind": "text_flip", "n_phasing": 30, "seed": 22, "flip_rx": 0.0, "flip_dx": 0.3,
                  "text": "ZCZC DX01\nERRORS IN THE FIRST COPY ONLY 0123456789\nNNNN\n"},
    "both_errors": {"kind": "text_flip", "n_phasing": 30, "seed": 23, "flip_rx": 0.12, "flip_dx": 0.12,
                    "text": "ZCZC BE01\nERRORS IN BOTH COPIES GIVE STARS 0123456789 ABCDEFGHIJKLMNOPQRSTUVWXYZ\nNNNN\n"},
    "error_abort": {"kind": "text_then_random", "n_phasing": 30, "seed": 24, "random_bits": 4000,
                    "text": "ZCZC AB99\nTHIS MESSAGE IS CUT BY NOISE\n"},
    "all_codes": {"kind": "all_codes", "n_phasing": 16},
    "random_bits": {"kind": "random", "seed": 25, "n": 30000},
    "phasing_variants": {"kind": "phasing_variants"},
    "long_message": {"kind": "text", "n_phasing": 40,
                     "text": "ZCZC LM55\n" + "".join(f"LINE {i:03d} THE QUICK BROWN FOX JUMPS OVER THE LAZY DOG 0123456789\n"
                                                  for i in range(30)) + "NNNN\n"},
    "rephase_after_mute": {"kind": "repeat_text", "n_phasing": 40, "times": 3,
                           "text": "ZCZC RP0%d\nTRANSMISSION NUMBER %d\nNNNN\n"},
}


def make_bits(nv, spec) -> str:
    kind = spec["kind"]
    if kind == "text":
        return nv.sitor_encode(spec["text"], spec["n_phasing"])
    if kind == "repeat_text":
        return "".join(nv.sitor_encode(spec["text"] % (i, i), spec["n_phasing"]) + "B" * 37 for i in range(spec["times"]))
    if kind == "text_flip":
        bits = nv.sitor_encode(spec["text"], spec["n_phasing"])
        rng = np.random.default_rng(spec["seed"])
        codes = _codes(bits)
        out = []
        for j, c in enumerate(codes):
            in_phasing = j < 2 * spec["n_phasing"]
            p = 0.0 if in_phasing else (spec["flip_dx"] if j % 2 == 0 else spec["flip_rx"])
            if p and rng.random() < p:
                k = int(rng.integers(0, 7))
                c = c[:k] + ("B" if c[k] == "Y" else "Y") + c[k + 1:]
            out.append(c)
        return "".join(out)
    if kind == "text_then_random":
        bits = nv.sitor_encode(spec["text"], spec["n_phasing"])
        bits = bits[: len(bits) - 6 * 7]          # drop the closing idle pairs
        rng = np.random.default_rng(spec["seed"])
        return bits + "".join("BY"[int(b)] for b in rng.integers(0, 2, size=spec["random_bits"]))
    if kind == "all_codes":
        e = 0x4a        # 'E' / '3'
        seq = [0x4c, 0x07] * spec["n_phasing"]
        for shift in (0x5a, 0x49):                # letters, then figures
            seq += [shift, 0x07, e, shift]        # establish the shift in both slots
            for c in range(128):
                seq += [c, e, e]                  # period 3: c lands on DX and RX slots alternately
        seq += [0x07] * 8
        return "".join(_code_bits(c) for c in seq)
    if kind == "random":
        rng = np.random.default_rng(spec["seed"])
        return "".join("BY"[int(b)] for b in rng.integers(0, 2, size=spec["n"]))
    if kind == "phasing_variants":
        pat = "BBBBBBYYYYBBYYBBBBBBYYYYBBYYBB"
        msg = nv.sitor_encode("ZCZC PV01\nOK\nNNNN\n", 3)
        parts = [
            "Y" * 5 + "B" * 11 + pat[6:],                  # extra B's absorbed by the six-B state
            "YYY" + pat[:17] + "B" + pat,                  # broken attempt immediately followed by a good one
            pat[:29] + "Y" + pat,                          # fails on the very last bit
            "BYBYBY" + pat[:10] + "YY",                    # partial
        ]
        # each variant followed by enough traffic to see whether byte reception started
        return "".join(p + msg + "B" * 1200 for p in parts)
    raise ValueError(kind)


# --------------------------------------------------------------------------
# WAV boundary cases (receiver/wav.c as receiver/capt_sched.c:87-101,516 uses it): frames of int16 I,Q
# --------------------------------------------------------------------------
WAV_CASES = {
    "capture_252k": {"frames": 5000, "rate": 252000, "seed": 11},     # the reference's own capture format
    "one_frame": {"frames": 1, "rate": 252000, "seed": 12},
    "empty": {"frames": 0, "rate": 252000, "seed": 13},
    "raw_rate": {"frames": 4097, "rate": 2016000, "seed": 14},       # the build's wideband recordings
}


def make_wav_frames(spec) -> np.ndarray:
    rng = np.random.default_rng(spec["seed"])
    return rng.integers(-32768, 32768, size=(spec["frames"], 2), dtype=np.int16)
