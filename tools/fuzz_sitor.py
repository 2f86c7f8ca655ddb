#!/usr/bin/env python3
"""One-off differential fuzz of the product character layer (navtex_amd/csrc/nvx_sitor.c) against the
compiled reference (oracle/_ref/ref_sm): messages AND the printf-visible trace must be identical.
Build container only (needs oracle/_ref).  usage: tools/fuzz_sitor.py [first_seed] [count]"""
import sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import numpy as np
import navtex_amd as nv, oracle_binding as ob

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ALPHA = list("ABCDEFGHIJKLMNOPQRSTUVWXYZ 0123456789.,/-:'()?+=\n\r")
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(int(rng.integers(1, 4))):                      # 1..3 transmissions back to back
        kind = rng.integers(0, 6)
        hdr = "ZCZC " + "".join(rng.choice(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ"), 2)) + f"{int(rng.integers(0, 100)):02d}"
        if kind == 1: hdr = hdr.replace("ZCZC", "ZCZ")            # damaged start-of-message forms the regex still takes / rejects
        if kind == 2: hdr = "ZCC" + hdr[4:]
        body = "".join(rng.choice(ALPHA, int(rng.integers(0, 400))))
        tail = ["\nNNNN\n", "\nNNN\n", "\nNN N\n", "\n", ""][int(rng.integers(0, 5))]
        text = hdr + "\n" + body + tail
        bits = list(nv.sitor_encode(text, int(rng.integers(3, 45))))
        rate = float(rng.choice([0.0, 0.0, 0.002, 0.01, 0.05, 0.2]))
        for k in rng.integers(0, len(bits), size=int(len(bits) * rate)):
            bits[k] = "B" if bits[k] == "Y" else "Y"
        if kind == 3: bits = bits[: int(len(bits) * rng.uniform(0.2, 0.9))]       # transmission cut short
        if kind == 4: del bits[int(rng.integers(0, len(bits)))]                    # a bit slip
        parts.append("".join(bits))
        parts.append("".join(rng.choice(["B", "Y"], int(rng.integers(0, 2500)))))  # noise between transmissions
        if kind == 5: parts.append("B" * int(rng.integers(10, 300)))               # idle carrier
    bits = "".join(parts)
    freq = int(rng.choice([518, 490]))
    r = ob.run_ref("sm", bits.encode())
    want_msgs = [tuple(m) for m in ob.parse_messages(r["messages"])]
    want_trace = r["stdout"].decode("latin1")
    s = nv.Sitor(518, trace=True); s.feed(bits)                 # ref_sm runs the 518 instance
    if s.messages != want_msgs or s.trace() != want_trace:
        bad += 1
        print(f"seed {seed}: DIFFERS (messages equal: {s.messages == want_msgs})", flush=True)
    if (seed - first) % 50 == 49: print(f"{seed - first + 1} cases, {bad} differing", flush=True)
print(f"done: {count} cases from seed {first}, {bad} differing")
