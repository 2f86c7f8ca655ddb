"""The oracle (oracle/nvx_oracle.c) against the golden vectors recorded from the
compiled reference (tests/golden/golden.json, made by make_golden.py).  Runs on
CPU anywhere; this is what pins the oracle ("parity pinned")."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

import cases

GOLD = json.loads((Path(__file__).parent / "golden" / "golden.json").read_text())


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_seam(arr: np.ndarray, rec: dict, what: str):
    flat = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
    assert flat.size == rec["n_doubles"], what
    step = rec["sparse"]["step"]
    got = [float(v).hex() for v in flat[::step]]
    assert got == rec["sparse"]["values"], f"{what}: sparse samples differ"
    assert sha(flat) == rec["sha256"], f"{what}: stream hash differs"


@pytest.mark.parametrize("name", sorted(GOLD["iq"]))
def test_full_path_stage_by_stage(nv, oracle, name):
    rec = GOLD["iq"][name]
    iq = cases.make_iq(nv, rec["spec"])
    assert sha(iq) == rec["iq_sha256"], "test input is not what the golden run used"
    y1 = oracle.fir1(iq)
    check_seam(y1, rec["y1"], "FIR1 out")
    for chain, tag in ((0, "518"), (1, "490")):
        y2 = oracle.fir2(oracle.mix(y1, chain))
        check_seam(y2, rec[f"y2_{tag}"], f"FIR2 out {tag}")
        y3 = oracle.fir3(y2)
        check_seam(y3, rec[f"y3_{tag}"], f"FIR3 out {tag}")
        bits, _ = oracle.decode(y3)
        assert bits == rec[f"bits{tag}"], f"bit stream {tag}"


@pytest.mark.parametrize("name", sorted(GOLD["iq"]))
def test_streaming_pipeline_any_chunking(nv, oracle, name):
    """The block-based streaming form (the timed CPU baseline) with ragged chunk sizes."""
    rec = GOLD["iq"][name]
    iq = cases.make_iq(nv, rec["spec"])
    p = oracle.Pipe(chain_mask=3)
    rng = np.random.default_rng(1)
    pos = 0
    while pos < iq.shape[0]:
        m = int(min(iq.shape[0] - pos, rng.integers(1, 40000)))
        p.push(iq[pos:pos + m]); pos += m
    assert p.bits(0) == rec["bits518"] and p.bits(1) == rec["bits490"]
    assert [list(m) for m in p.messages] == rec["messages"]


@pytest.mark.parametrize("name", sorted(GOLD["decoder"]))
def test_decoder_alone(oracle, name):
    rec = GOLD["decoder"][name]
    bits, _ = oracle.decode(cases.make_y3(rec["spec"]))
    assert bits == rec["bits"]


@pytest.mark.parametrize("name", sorted(GOLD["charlayer"]))
def test_charlayer_restatement(nv, oracle, name):
    rec = GOLD["charlayer"][name]
    bits = cases.make_bits(nv, rec["spec"])
    assert hashlib.sha256(bits.encode()).hexdigest() == rec["bits_sha256"]
    cl = oracle.CharLayer(518)
    cl.feed(bits)
    assert [list(m) for m in cl.messages] == rec["messages"]
    assert cl.trace() == rec["stdout"], "printf-visible trace differs from the reference's stdout"


def test_stage0_definition(oracle):
    """Build-owned stage 0: (sum of 8 + 4) >> 3 with floor semantics, per component."""
    rng = np.random.default_rng(0)
    raw = rng.integers(-32768, 32768, size=(8 * 5000, 2), dtype=np.int16)
    raw[:8] = 32767; raw[8:16] = -32768; raw[16:24, 0] = -1; raw[16:24, 1] = 1
    got = oracle.stage0(raw)
    want = np.floor((raw.astype(np.int64).reshape(-1, 8, 2).sum(axis=1) + 4) / 8.0).astype(np.int16)
    assert np.array_equal(got, want)
    assert got[0, 0] == 32767 and got[1, 0] == -32768
