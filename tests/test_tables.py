"""Constant tables: the product's hex-float copies and the oracle's decimal
copies against the values the reference spells (golden.json "tables"), and the
baked libm-derived tables against libm on this host."""
import json
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((Path(__file__).parent / "golden" / "golden.json").read_text())


def c_array(text: str, name: str):
    i = text.index(name + "[")
    j = text.index("{", i); k = text.index("}", j)
    toks = re.findall(r"-?0x[0-9a-fA-F.]+p[+-]?\d+f?|-?\d+\.\d+(?:[eE][+-]?\d+)?", text[j:k])
    return [float.fromhex(t.rstrip("f")) if "0x" in t else float(t) for t in toks]


def test_product_taps_equal_reference_values():
    src = (ROOT / "navtex_amd" / "csrc" / "nvx_tables.h").read_text()
    for name, key, n in (("NVX_H1", "h1", 37), ("NVX_H2", "h2", 47), ("NVX_H3", "h3", 71)):
        got = c_array(src, name)
        assert len(got) == n
        assert [v.hex() for v in got] == GOLD["tables"][key]


def test_oracle_taps_equal_reference_values():
    src = (ROOT / "oracle" / "nvx_oracle_tables.h").read_text()
    for name, key in (("NVXO_H1", "h1"), ("NVXO_H2", "h2"), ("NVXO_H3", "h3")):
        assert [v.hex() for v in c_array(src, name)] == GOLD["tables"][key]


def test_taps_are_not_exactly_symmetric():
    """Why no kernel folds symmetric taps (hard part 1)."""
    h1 = [float.fromhex(v) for v in GOLD["tables"]["h1"]]
    assert h1[5] != h1[31] and h1[4] != h1[32]


def test_baked_mixer_and_bitfilter_tables_match_libm(oracle):
    src = (ROOT / "navtex_amd" / "csrc" / "nvx_tables.h").read_text()
    cr, ci = oracle.mixer_table()                 # cos / -sin via this host's libm, reference expression
    assert [v.hex() for v in c_array(src, "NVX_MIX_CR")] == [float(v).hex() for v in cr]
    got_ci = c_array(src, "NVX_MIX_CI")
    assert [v.hex() for v in got_ci] == [float(v).hex() for v in ci]
    assert np.signbit(got_ci[0])                  # -sin(0) = -0.0, kept
    fr, fi = oracle.bitfilter_table()
    assert np.array_equal(np.array(c_array(src, "NVX_BF_R"), dtype=np.float32), fr)
    assert np.array_equal(np.array(c_array(src, "NVX_BF_I"), dtype=np.float32), fi)
