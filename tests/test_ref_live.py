"""Oracle and product host code against the LIVE compiled reference
(oracle/_ref/ref_*), on fresh random cases beyond the committed goldens.
Only runs where the reference seams were built (the build container);
skipped on the GPU box, where /root/reference does not exist."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.skipif(not ob.have_ref(), reason="compiled reference (oracle/_ref) not present")


def u64(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_fir_seams_random_iq(oracle, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(30000, 90000))
    iq = rng.integers(-32768, 32768, size=(n, 2), dtype=np.int16)
    data = iq.tobytes()
    y1 = oracle.fir1(iq)
    assert np.array_equal(u64(y1), np.frombuffer(ob.run_ref("fir1", data)["y1"], dtype=np.uint64).reshape(-1, 2))
    f2, f3 = ob.run_ref("fir2", data), ob.run_ref("fir3", data)
    for chain, tag in ((0, "518"), (1, "490")):
        y2 = oracle.fir2(oracle.mix(y1, chain))
        assert np.array_equal(u64(y2).reshape(-1), np.frombuffer(f2[f"y2_{tag}"], dtype=np.uint64))
        y3 = oracle.fir3(y2)
        assert np.array_equal(u64(y3).reshape(-1), np.frombuffer(f3[f"y3_{tag}"], dtype=np.uint64))


@pytest.mark.parametrize("seed", [7, 8])
def test_full_path_random_signal(nv, oracle, seed):
    import signals
    rng = np.random.default_rng(seed)
    st, _ = signals.stream_params(nv, 500 + seed, nv.RATE_IN, freq_hz=int(rng.choice([14000, -14000])) + int(rng.integers(-15, 16)),
                                  noise_amp=int(rng.integers(500, 6000)), amplitude=int(rng.integers(2000, 12000)), n_phasing=30)
    iq = nv.synth_host(st, nv.RATE_IN, 252000 * 18 + int(rng.integers(0, 5000)))
    data = iq.tobytes()
    bits = ob.run_ref("bits", data)
    full = ob.run_ref("full", data)
    p = oracle.Pipe(chain_mask=3)
    p.push(iq)
    assert p.bits(0) == bits["bits518"].decode() and p.bits(1) == bits["bits490"].decode()
    assert [tuple(m) for m in p.messages] == [tuple(m) for m in ob.parse_messages(full["messages"])]
    # product character layer on the reference's own bits
    got = []
    for tag, freq in (("bits518", 518), ("bits490", 490)):
        s = nv.Sitor(freq); s.feed(bits[tag].decode()); got += s.messages
    assert sorted(got) == sorted(tuple(m) for m in ob.parse_messages(full["messages"]))


@pytest.mark.parametrize("cut", [0, 1, 279, 280, 281, 2519, 2520, 9 * 280 + 5, 17 * 280, 40000, 80639])
def test_the_end_of_an_input_is_where_the_reference_stops(nv, oracle, cut):
    """What the GPU's end-of-stream tests (tests/test_gpu_boundary.py::test_every_kind_of_tail_matches_the_oracle) compare
    with is pinned here to the COMPILED REFERENCE: the same signal cut at the same places (three frames and a tail of
    nothing, less than one 900 S/s sample, exactly one, whole and ragged bit periods, almost a frame) -- the reference's
    loop (receiver/capt_sched.c:509-513) hands its objects exactly these samples and stops; the oracle's bits on both
    chains are the reference's, bit for bit, at every cut."""
    import signals
    st, _ = signals.stream_params(nv, 5, nv.RATE_IN)
    iq = nv.synth_host(st, nv.RATE_IN, 3 * nv.FRAME_IN + cut)
    bits = ob.run_ref("bits", iq.tobytes())
    p = oracle.Pipe(chain_mask=3, charlayer=False)
    p.push(iq)
    assert p.bits(0) == bits["bits518"].decode() and p.bits(1) == bits["bits490"].decode()
    assert len(p.bits(0)) >= 28


@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_character_layer_random_bits_and_flips(nv, oracle, seed):
    """Random garbage and randomly damaged traffic: messages and full trace, reference vs both implementations."""
    rng = np.random.default_rng(seed)
    text = "ZCZC " + "".join(rng.choice(list("ABCDEFGH"), 2)) + f"{int(rng.integers(0, 100)):02d}\n" + \
           "".join(rng.choice(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ 0123456789.,/-"), 200)) + "\nNNNN\n"
    bits = list(nv.sitor_encode(text, 25))
    for k in rng.integers(0, len(bits), size=int(len(bits) * rng.choice([0.0, 0.005, 0.03]))):
        bits[k] = "B" if bits[k] == "Y" else "Y"
    bits = "".join(bits) + "".join(rng.choice(["B", "Y"], 3000))
    r = ob.run_ref("sm", bits.encode())
    want_msgs, want_trace = [tuple(m) for m in ob.parse_messages(r["messages"])], r["stdout"].decode("latin1")
    o = oracle.CharLayer(518); o.feed(bits)
    assert o.messages == want_msgs and o.trace() == want_trace
    s = nv.Sitor(518, trace=True); s.feed(bits)
    assert s.messages == want_msgs and s.trace() == want_trace


def test_decoder_random_inputs(oracle):
    rng = np.random.default_rng(77)
    y3 = rng.normal(size=(6000, 2)) * 3000.0
    r = ob.run_ref("dec", np.ascontiguousarray(y3).tobytes())
    bits, _ = oracle.decode(y3)
    assert bits == r["bits518"].decode() and len(bits) > 500


@pytest.mark.skipif(not ob.have_ref_wav(), reason="compiled reference wav.c (oracle/_ref/ref_wav) not present")
@pytest.mark.parametrize("seed,n,rate", [(1, 3, 252000), (2, 70001, 252000), (3, 12345, 2016000), (4, 0, 252000)])
def test_wav_boundary_both_directions_live(nv, tmp_path, seed, n, rate):
    """reference wav_write -> product nvx_wav_read, and product nvx_wav_write -> reference wav_read
    (receiver/wav.c:469-528), on fresh random frames."""
    frames = np.random.default_rng(seed).integers(-32768, 32768, size=(n, 2), dtype=np.int16)
    blob = ob.ref_wav_write(frames, rate)
    f = tmp_path / "ref.wav"; f.write_bytes(blob)
    back, got_rate = nv.wav_read(str(f))
    assert got_rate == rate and np.array_equal(back.reshape(-1, 2), frames)
    p = str(tmp_path / "product.wav")
    nv.wav_write(p, frames, rate)
    assert open(p, "rb").read() == blob, "product and reference files differ"
    meta, data = ob.ref_wav_read(p)
    assert meta == (1, 2, rate, 2, n) and data == frames.tobytes()


def test_capt_sched_dsp_symbols_resolve_from_the_library(nv):
    """Link-level drop-in check against the reference's own capture program, WITHOUT building it: receiver/capt_sched.c
    needs the vendor's sdrplay_api.h, which this image lacks, and a stand-in header would make it a reference build
    resting on our own guesses (not allowed, and not evidence).  What can be checked from its text and from the
    library's dynamic symbol table: every function capt_sched.c forward-declares for itself (capt_sched.c:17-19) -- the
    whole DSP surface it links against -- is a defined, exported, C-linkage text symbol of libnavtex_amd.so; the calls it
    makes (:511, :554, :612) name only those; and the library itself leaves nothing undefined that the six replaced
    reference objects (fir1cpp, fir2cpp, fir3cpp, decoder, nav_b_sm, nav_sched) used to provide."""
    import re
    import subprocess
    from pathlib import Path
    src = Path("/root/reference/receiver/capt_sched.c")
    if not src.exists():
        pytest.skip("reference sources not present")
    text = src.read_text(errors="replace")
    head = text[: text.index("///////////// NAVTEX/SITOR-B")]
    protos = re.findall(r"^\s*void\s+(\w+)\s*\(([^)]*)\)\s*;", head, flags=re.M)
    assert [p[0] for p in protos] == ["init_fir_filter1", "sample_in_1", "init_fir2_wrapper"]
    assert protos[1][1].replace(" ", "") == "doublesample_I,doublesample_Q"
    lib = Path(nv.lib._name)
    defined = {l.split()[-1]: l.split()[-2] for l in subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True).stdout.splitlines() if l.strip()}
    for name, _args in protos:
        assert defined.get(name) == "T", f"{name} is not an exported text symbol of {lib.name}"
        assert re.search(rf"\b{name}\s*\(", text[len(head):]), f"capt_sched.c never calls {name}"
    assert defined.get("add_message") == "W"                 # weak: the receiver's message_store.o overrides it
    # calls into the DSP from the rest of the file: exactly those three names
    body = text[len(head):]
    for callee in ("sample_in_2", "fir_in_2", "init_fir_filter2", "receive_bit", "bs_decoded_sample_in"):
        assert not re.search(rf"\b{callee}\s*\(", body)
    undefined = subprocess.run(["nm", "-D", "--undefined-only", str(lib)], capture_output=True, text=True, check=True).stdout
    for sym in ("sample_in_2", "init_fir_filter2", "fir_filter3", "decoder", "byte_state_machine", "_Z10fir_in_2"):
        assert sym not in undefined


def test_capt_scheds_own_declarations_and_calls_link_against_the_library(nv, tmp_path):
    """The literal form of the link claim that CAN be made here.  receiver/capt_sched.c itself cannot be compiled in this
    image (it includes the vendor's sdrplay_api.h; a hand-written stand-in header would turn the proof into a guess about
    that header -- not allowed, and not evidence).  But the part of the file that touches the DSP can be used AS IT IS:
    at test time this reads capt_sched.c's own three declaration lines (:17-19), its own three call statements (:511,
    :554, :612) and message_store.h's add_message prototype (:7), puts them -- verbatim, nothing of ours in between but
    `short sample_buffer[2]; int index = 0;` and the braces -- into a C file under tmp_path, compiles it with gcc as the
    reference's build would (C, not C++: the declarations are K&R-style `()`), and links it against -lnavtex_amd and
    nothing else.  The link succeeds; in the program the three symbols are undefined and resolve from libnavtex_amd.so
    (DT_NEEDED), and its own add_message overrides the library's weak one."""
    import re
    import subprocess
    from pathlib import Path
    ref = Path("/root/reference/receiver")
    if not (ref / "capt_sched.c").exists():
        pytest.skip("reference sources not present")
    text = (ref / "capt_sched.c").read_text(errors="replace").splitlines()
    decls = [l for l in text[:40] if re.match(r"^\s*void\s+(init_fir_filter1|sample_in_1|init_fir2_wrapper)\s*\(", l)]
    assert len(decls) == 3
    calls = {}
    for i, l in enumerate(text):
        m = re.match(r"^\s*(init_fir_filter1|sample_in_1|init_fir2_wrapper)\s*\(.*\)\s*;\s*$", l)
        if m and i > 40:
            calls.setdefault(m.group(1), l)
    assert set(calls) == {"init_fir_filter1", "sample_in_1", "init_fir2_wrapper"}
    assert "sample_buffer[index]" in calls["sample_in_1"] and "(double)" in calls["sample_in_1"]
    proto = [l for l in (ref / "message_store.h").read_text().splitlines() if re.match(r"^\s*int\s+add_message\s*\(", l)]
    assert len(proto) == 1
    src = tmp_path / "drop_in.c"
    src.write_text("\n".join(
        ["/* generated by tests/test_ref_live.py at test time from the reference's own lines; never committed */"] + decls + proto +
        ["int add_message(char *bbbb,char *message, int freq) { (void)bbbb; (void)message; return freq; }",
         "short sample_buffer[2]; int index = 0;", "int main(void) {", calls["init_fir_filter1"], calls["init_fir2_wrapper"], calls["sample_in_1"], "return 0; }"]) + "\n")
    libdir = Path(nv.lib._name).parent
    exe = tmp_path / "drop_in"
    r = subprocess.run(["gcc", "-O2", "-Wall", str(src), "-o", str(exe), f"-L{libdir}", "-lnavtex_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    syms = subprocess.run(["nm", "-D", str(exe)], capture_output=True, text=True, check=True).stdout
    for name in ("init_fir_filter1", "sample_in_1", "init_fir2_wrapper"):
        assert re.search(rf"^\s+U {name}$", syms, flags=re.M), f"{name} is not an undefined dynamic symbol of the program"
    assert re.search(r"^[0-9a-f]+ T add_message$", syms, flags=re.M)          # the program's own sink, exported: it overrides the library's weak one
    needed = subprocess.run(["readelf", "-d", str(exe)], capture_output=True, text=True, check=True).stdout
    assert "libnavtex_amd.so" in needed
    # (it is not RUN here: this container has no GPU, and the library has no CPU path -- tests/test_gpu_boundary.py runs
    # the look-alike program tests/harness/capt_loop.c, which keeps capt_sched.c's ring, callback and consumer loop)


def test_replay_is_what_the_reference_does_with_the_same_frames_again(nv, oracle):
    """bench.py's parity gate AFTER its timed region rests on nvxo_replay: the same frames pushed `loops` times into one
    pipe, state carried from repeat to repeat.  The reference carries that state in its statics (receiver/fir1cpp.C:51-60,
    receiver/fir2cpp.C:74-83, receiver/fir3cpp.h:90-95, receiver/decoder.h:31-60): fed the frames repeated as ONE stream,
    the compiled reference itself must produce the replay's bits, on both chains."""
    import signals
    n = 3 * nv.FRAME_IN
    car = [dict(freq_hz=f, bits=nv.sitor_encode(signals.stream_text(71), 8), bit_offset=517, phase0=777 * k, amplitude=5500) for k, f in enumerate((14000, -14000))]
    iq = nv.synth_host(nv.make_stream(car, seed=71, noise_amp=1800), nv.RATE_IN, n)
    loops = 4
    _s, got = oracle.replay(iq[None], 1, n, False, 3, 1, loops)
    ref = ob.run_ref("bits", np.concatenate([iq] * loops).tobytes())
    assert got[0] == [ref["bits518"].decode(), ref["bits490"].decode()] and len(got[0][0]) > 300
