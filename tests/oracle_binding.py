"""ctypes binding of the oracle (oracle/_build/libnvx_oracle.so) and runner for
the compiled-reference seam binaries (oracle/_ref/ref_*).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the navtex_amd package.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import tempfile
from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "_build" / "libnvx_oracle.so"
REF = ROOT / "oracle" / "_ref"


def _build():
    subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "oracle"], check=True)


if not LIB.exists():
    _build()
L = C.CDLL(str(LIB))

MSG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int)
_vp, _sz, _i = C.c_void_p, C.c_size_t, C.c_int
for name, res, args in [
    ("nvxo_stage0", None, [_vp, _sz, _vp]), ("nvxo_channelise", None, [_vp, _sz, _vp, _vp]),
    ("nvxo_stage0_cic3", None, [_vp, _sz, _vp, _vp]), ("nvxo_pipe_set_stage0", None, [_vp, _i]),
    ("nvxo_fir1", _sz, [_vp, _sz, _vp]), ("nvxo_mix", None, [_vp, _sz, _i, _vp]),
    ("nvxo_fir2", _sz, [_vp, _sz, _vp]), ("nvxo_fir3", _sz, [_vp, _sz, _vp]),
    ("nvxo_mixer_table", None, [_vp, _vp]), ("nvxo_bitfilter_table", None, [_vp, _vp]),
    ("nvxo_decode", _sz, [_vp, _sz, _vp, _vp]), ("nvxo_decode_inject", _sz, [_vp, _sz, _vp, _sz, _i, C.POINTER(_i)]),
    ("nvxo_pipe_reinit", None, [_vp, _i]), ("nvxo_decode_with", _sz, [_vp, _sz, _vp, _vp, C.POINTER(_sz)]),
    ("nvxo_sm_new", _vp, [_i, MSG_FN, _vp]), ("nvxo_sm_free", None, [_vp]), ("nvxo_sm_bit", None, [_vp, C.c_char]),
    ("nvxo_sm_trace", C.c_char_p, [_vp, C.POINTER(_sz)]),
    ("nvxo_pipe_new", _vp, [_i, _i, _i, MSG_FN, _vp]), ("nvxo_pipe_free", None, [_vp]),
    ("nvxo_pipe_push", None, [_vp, _vp, _sz]), ("nvxo_pipe_push_raw", None, [_vp, _vp, _sz]),
    ("nvxo_pipe_bits", C.c_char_p, [_vp, _i, C.POINTER(_sz)]),
    ("nvxo_pipe_tap_y3", None, [_vp, _i, _vp, _sz, C.POINTER(_sz)]),
    ("nvxo_pipe_set_charlayer", None, [_vp, _i]),
    ("nvxo_bench", C.c_double, [_vp, _sz, _sz, _i, _i, _i, _i, _vp, _sz]), ("nvxo_max_threads", _i, []),
    ("nvxo_bench_wide", C.c_double, [_vp, _sz, _sz, _i, _i, _vp, _sz]),
    ("nvxo_replay", C.c_double, [_vp, _sz, _sz, _i, _i, _i, _i, _vp, _sz]),
    ("nvxo_replay_wide", C.c_double, [_vp, _sz, _sz, _i, _i, _vp, _sz]),
]:
    fn = getattr(L, name)
    fn.restype, fn.argtypes = res, args


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def stage0(raw_iq: np.ndarray) -> np.ndarray:
    raw_iq = np.ascontiguousarray(raw_iq, dtype=np.int16).reshape(-1, 2)
    n_out = raw_iq.shape[0] // 8
    out = np.empty((n_out, 2), dtype=np.int16)
    L.nvxo_stage0(_p(raw_iq), n_out, _p(out))
    return out


def stage0_cic3(raw_iq: np.ndarray, hist14: np.ndarray | None = None) -> np.ndarray:
    """Third-order stage 0; hist14 ([14, 2] int16, the samples in front; updated in place) or None = silence, not carried."""
    raw_iq = np.ascontiguousarray(raw_iq, dtype=np.int16).reshape(-1, 2)
    n_out = raw_iq.shape[0] // 8
    out = np.empty((n_out, 2), dtype=np.int16)
    if hist14 is not None:
        assert hist14.dtype == np.int16 and hist14.shape == (14, 2) and hist14.flags.c_contiguous
    L.nvxo_stage0_cic3(_p(raw_iq), n_out, _p(hist14) if hist14 is not None else None, _p(out))
    return out


def channelise(raw_iq: np.ndarray, hist40: np.ndarray | None = None) -> np.ndarray:
    """Wideband front-end: [n*8, 2] int16 at 2.016 MS/s -> [8, n, 2] int16 at 252 kS/s."""
    raw_iq = np.ascontiguousarray(raw_iq, dtype=np.int16).reshape(-1, 2)
    n_out = raw_iq.shape[0] // 8
    out = np.empty((8, n_out, 2), dtype=np.int16)
    h = None if hist40 is None else np.ascontiguousarray(hist40, dtype=np.int16).reshape(40, 2)
    L.nvxo_channelise(_p(raw_iq), n_out, _p(h) if h is not None else None, _p(out))
    return out


def fir1(iq: np.ndarray) -> np.ndarray:
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
    y = np.empty((iq.shape[0] // 4, 2))
    n = L.nvxo_fir1(_p(iq), iq.shape[0], _p(y))
    return y[:n]


def mix(y1: np.ndarray, chain: int) -> np.ndarray:
    y1 = np.ascontiguousarray(y1, dtype=np.float64)
    u = np.empty_like(y1)
    L.nvxo_mix(_p(y1), y1.shape[0], chain, _p(u))
    return u


def fir2(u: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(u, dtype=np.float64)
    y = np.empty((u.shape[0] // 7, 2))
    n = L.nvxo_fir2(_p(u), u.shape[0], _p(y))
    return y[:n]


def fir3(y2: np.ndarray) -> np.ndarray:
    y2 = np.ascontiguousarray(y2, dtype=np.float64)
    y = np.empty((y2.shape[0] // 10, 2))
    n = L.nvxo_fir3(_p(y2), y2.shape[0], _p(y))
    return y[:n]


def decode(y3: np.ndarray) -> Tuple[str, np.ndarray]:
    y3 = np.ascontiguousarray(y3, dtype=np.float64).reshape(-1, 2)
    bits = C.create_string_buffer(y3.shape[0] + 1)
    dphi = np.empty(y3.shape[0])
    n = L.nvxo_decode(_p(y3), y3.shape[0], bits, _p(dphi))
    return bits.raw[:n].decode("ascii"), dphi


def decode_inject(y3: np.ndarray, at: int, value: int) -> Tuple[str, int]:
    """nvxo_decode with the decoder's `int bd_seq_nbr` (decoder.h:60) set to `value` in front of sample `at`.
    Returns (bits, the value the counter had there)."""
    y3 = np.ascontiguousarray(y3, dtype=np.float64).reshape(-1, 2)
    bits = C.create_string_buffer(y3.shape[0] + 1)
    was = _i(0)
    n = L.nvxo_decode_inject(_p(y3), y3.shape[0], bits, at, value, C.byref(was))
    return bits.raw[:n].decode("ascii"), was.value


def decode_with(y3: np.ndarray, atan2_fn_ptr) -> Tuple[str, int]:
    """Decoder restatement with a caller-supplied atan2 (C function pointer, or None for libm).
    Returns (bits, number of samples whose delta-phi differed from libm's)."""
    y3 = np.ascontiguousarray(y3, dtype=np.float64).reshape(-1, 2)
    bits = C.create_string_buffer(y3.shape[0] + 1)
    mism = _sz(0)
    n = L.nvxo_decode_with(_p(y3), y3.shape[0], bits, atan2_fn_ptr, C.byref(mism))
    return bits.raw[:n].decode("ascii"), mism.value


def mixer_table() -> Tuple[np.ndarray, np.ndarray]:
    cr, ci = np.empty(9), np.empty(9)
    L.nvxo_mixer_table(_p(cr), _p(ci))
    return cr, ci


def bitfilter_table() -> Tuple[np.ndarray, np.ndarray]:
    r, i = np.empty(5, dtype=np.float32), np.empty(5, dtype=np.float32)
    L.nvxo_bitfilter_table(_p(r), _p(i))
    return r, i


class CharLayer:
    """nvxo_sm: literal restatement of the reference's byte_state_machine."""

    def __init__(self, freq: int = 518):
        self.messages: List[Tuple[int, str, str]] = []
        self._cb = MSG_FN(lambda u, b, m, f: self.messages.append((f, b.decode("latin1"), m.decode("latin1"))))
        self._h = L.nvxo_sm_new(freq, self._cb, None)

    def feed(self, bits: str) -> None:
        for ch in bits.encode("ascii"):
            L.nvxo_sm_bit(self._h, C.c_char(bytes([ch])))

    def trace(self) -> str:
        n = _sz()
        p = L.nvxo_sm_trace(self._h, C.byref(n))
        return C.string_at(p, n.value).decode("latin1")

    def __del__(self):
        if self._h:
            L.nvxo_sm_free(self._h)
            self._h = None


class Pipe:
    """nvxo_pipe: streaming oracle for one IQ stream."""

    def __init__(self, chain_mask: int = 3, freqs=(518, 490), charlayer: bool = True, tap_y3: int = 0):
        self.messages: List[Tuple[int, str, str]] = []
        self._cb = MSG_FN(lambda u, b, m, f: self.messages.append((f, b.decode("latin1"), m.decode("latin1"))))
        self._h = L.nvxo_pipe_new(chain_mask, freqs[0], freqs[1], self._cb, None)
        L.nvxo_pipe_set_charlayer(self._h, int(charlayer))
        self._taps = {}
        if tap_y3:
            for c in range(2):
                buf = np.zeros((tap_y3, 2))
                cnt = _sz(0)
                L.nvxo_pipe_tap_y3(self._h, c, _p(buf), tap_y3, C.byref(cnt))
                self._taps[c] = (buf, cnt)

    def push(self, iq252: np.ndarray) -> None:
        iq252 = np.ascontiguousarray(iq252, dtype=np.int16).reshape(-1, 2)
        L.nvxo_pipe_push(self._h, _p(iq252), iq252.shape[0])

    def push_raw(self, raw: np.ndarray) -> None:
        raw = np.ascontiguousarray(raw, dtype=np.int16).reshape(-1, 2)
        assert raw.shape[0] % 8 == 0
        L.nvxo_pipe_push_raw(self._h, _p(raw), raw.shape[0] // 8)

    def reinit(self, which: int) -> None:
        """The reference's init functions called again in mid-stream (1: init_fir_filter1, 2: init_fir2_wrapper, 3: both)."""
        L.nvxo_pipe_reinit(self._h, which)

    def set_stage0(self, order: int) -> None:
        L.nvxo_pipe_set_stage0(self._h, order)

    def bits(self, chain: int = 0) -> str:
        return L.nvxo_pipe_bits(self._h, chain, None).decode("ascii")

    def y3(self, chain: int = 0) -> np.ndarray:
        buf, cnt = self._taps[chain]
        return buf[: cnt.value]

    def __del__(self):
        if self._h:
            L.nvxo_pipe_free(self._h)
            self._h = None


def bench(iq: np.ndarray, nstreams: int, n: int, raw: bool, chain_mask: int, nthreads: int, want_bits: bool = False,
          repeat: int = 1):
    """Timed CPU baseline; iq is [nstreams, n*(8 if raw else 1), 2] int16 (raw: False / True = stage 0 as integrate-and-dump /
    3 = its third-order form).  Returns (seconds, bits list)."""
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    cap = n // 2520 + 64
    buf = C.create_string_buffer(nstreams * cap) if want_bits else None
    secs = L.nvxo_bench(_p(iq), nstreams, n, int(raw), chain_mask, nthreads, repeat, buf, cap)
    bits = None
    if want_bits:
        bits = [buf.raw[s * cap:(s + 1) * cap].split(b"\0")[0].decode("ascii") for s in range(nstreams)]
    return secs, bits


def bench_wide(raw: np.ndarray, nwide: int, n_out: int, nthreads: int, repeat: int = 1, want_bits: bool = False):
    """Timed wideband CPU baseline; raw is [nwide, n_out*8, 2] int16.  Returns (seconds, bits[nwide*16])."""
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    cap = n_out // 2520 + 64
    buf = C.create_string_buffer(nwide * 16 * cap) if want_bits else None
    secs = L.nvxo_bench_wide(_p(raw), nwide, n_out, nthreads, repeat, buf, cap)
    bits = [buf.raw[i * cap:(i + 1) * cap].split(b"\0")[0].decode("ascii") for i in range(nwide * 16)] if want_bits else None
    return secs, bits


def replay(iq: np.ndarray, nstreams: int, n: int, raw, chain_mask: int, nthreads: int, loops: int):
    """What a benchmark loop over a resident batch computes: iq ([nstreams, n*(8 if raw else 1), 2] int16) pushed `loops` times
    into ONE pipe per stream, state carried from repeat to repeat.  Returns (seconds, bits): one string per stream, or
    [chain 0, chain 1] per stream when chain_mask is 3."""
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    nch = 2 if chain_mask == 3 else 1
    cap = loops * (n // 2520) + 64
    buf = C.create_string_buffer(nstreams * nch * cap)
    secs = L.nvxo_replay(_p(iq), nstreams, n, int(raw), chain_mask, nthreads, loops, buf, cap)
    flat = [buf.raw[i * cap:(i + 1) * cap].split(b"\0")[0].decode("ascii") for i in range(nstreams * nch)]
    return secs, (flat if nch == 1 else [flat[2 * s:2 * s + 2] for s in range(nstreams)])


def replay_wide(raw: np.ndarray, nwide: int, n_out: int, nthreads: int, loops: int):
    """The same for wideband streams ([nwide, n_out*8, 2] int16): (seconds, bits[nwide*16]) with index (w*8 + k)*2 + chain."""
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    cap = loops * (n_out // 2520) + 64
    buf = C.create_string_buffer(nwide * 16 * cap)
    secs = L.nvxo_replay_wide(_p(raw), nwide, n_out, nthreads, loops, buf, cap)
    return secs, [buf.raw[i * cap:(i + 1) * cap].split(b"\0")[0].decode("ascii") for i in range(nwide * 16)]


# --------------------------------------------------- compiled reference seams
def have_ref() -> bool:
    return (REF / "ref_full").exists()


def run_ref(seam: str, data: bytes, probe: Tuple = (), ref_dir: Path | None = None) -> Dict[str, bytes]:
    """Run oracle/_ref/ref_<seam> on `data`; returns {output name: bytes, 'stdout': bytes}.
    probe: ("reinit", n, which) for the full / bits seams, ("inject", n, value) for the decoder seam
    (oracle/ref_seams/ref_harness.cpp); ref_dir: seams built elsewhere (another set of compiler flags)."""
    with tempfile.TemporaryDirectory() as td:
        inp = Path(td) / "in.bin"
        inp.write_bytes(data)
        r = subprocess.run([str((ref_dir or REF) / f"ref_{seam}"), str(inp), str(Path(td) / "o"), *[str(a) for a in probe]], check=True, capture_output=True)
        out = {"stdout": r.stdout}
        for f in Path(td).glob("o.*.bin"):
            out[f.name[2:-4]] = f.read_bytes()
        return out


def have_ref_wav() -> bool:
    return (REF / "ref_wav").exists()


def ref_wav_write(frames: np.ndarray, rate: int, ref_dir: Path | None = None) -> bytes:
    """The reference's wav.c writes `frames` ([n, 2] int16) the way capt_sched.c:91-95,516 does; returns the file bytes."""
    with tempfile.TemporaryDirectory() as td:
        f = Path(td) / "frames.bin"
        f.write_bytes(np.ascontiguousarray(frames, dtype=np.int16).tobytes())
        subprocess.run([str((ref_dir or REF) / "ref_wav"), "write", str(f), str(Path(td) / "o.wav"), str(rate)], check=True)
        return (Path(td) / "o.wav").read_bytes()


def ref_wav_read(path: str, ref_dir: Path | None = None):
    """The reference's wav_open/wav_get_*/wav_read (wav.c:469-528) on a file; returns ((format, channels, rate,
    sample_size, length), frame bytes)."""
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "frames.bin"
        r = subprocess.run([str((ref_dir or REF) / "ref_wav"), "read", str(path), str(out)], check=True, capture_output=True)
        return tuple(int(v) for v in r.stdout.split()), out.read_bytes()


def parse_messages(blob: bytes) -> List[Tuple[int, str, str]]:
    out, pos = [], 0
    while pos < len(blob):
        freq = int.from_bytes(blob[pos:pos + 4], "little", signed=True); pos += 4
        n = int.from_bytes(blob[pos:pos + 4], "little"); pos += 4
        bbbb = blob[pos:pos + n].decode("latin1"); pos += n
        n = int.from_bytes(blob[pos:pos + 4], "little"); pos += 4
        msg = blob[pos:pos + n].decode("latin1"); pos += n
        out.append((freq, bbbb, msg))
    return out
