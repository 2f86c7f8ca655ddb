"""navtex_amd -- MI355X-native NAVTEX demodulation hot path.

The product is the C-ABI shared library `libnavtex_amd.so` (HIP kernels for
gfx950 + host C/C++, see include/navtex_amd.h).  This Python package is a thin
ctypes mirror of that ABI for tests, bench.py and scripting; it contains no
signal processing and no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from ._native import (CHAIN_490, CHAIN_518, FRAME_BITS, FRAME_IN, FRAME_RAW, FRAME_Y3, RATE_IN, RATE_RAW,
                      NvxError, lib)

__all__ = ["Pipeline", "Group", "Capture", "Sitor", "sitor_encode", "make_stream", "synth_host", "synth_device", "device_count",
           "DeviceBuffer", "channelise", "channelise_time_stats", "Store", "wav_write", "wav_read", "NvxError", "lib",
           "CHAIN_518", "CHAIN_490", "FRAME_BITS", "FRAME_IN", "FRAME_RAW", "FRAME_Y3", "RATE_IN", "RATE_RAW"]


def device_count() -> int:
    return lib.nvx_device_count()


# ----------------------------------------------------------------- SITOR-B
def sitor_encode(text: str, n_phasing: int = 40) -> str:
    """SITOR-B transmit framing of `text` as a 'B'/'Y' string (nvx_sitor_encode)."""
    t = text.encode("ascii")
    n = lib.nvx_sitor_encode(t, n_phasing, None, 0)
    buf = C.create_string_buffer(n + 1)
    lib.nvx_sitor_encode(t, n_phasing, buf, n)
    return buf.raw[:n].decode("ascii")


class Sitor:
    """Host character layer (nvx_sitor_*): bits in, (bbbb, message, freq) out."""

    def __init__(self, freq: int = 518, trace: bool = False):
        self.messages: List[Tuple[int, str, str]] = []
        self.trace_text: List[str] = []
        self._cb = N.SITOR_MSG_FN(lambda u, b, m, f: self.messages.append((f, b.decode("latin1"), m.decode("latin1"))))
        self._h = lib.nvx_sitor_new(freq, self._cb, None)
        if trace:
            self._tcb = N.SITOR_TRACE_FN(lambda u, t: self.trace_text.append(t.decode("latin1")))
            lib.nvx_sitor_set_trace(self._h, self._tcb, None)

    def feed(self, bits: str) -> None:
        b = bits.encode("ascii")
        lib.nvx_sitor_receive_bits(self._h, b, len(b))

    def trace(self) -> str:
        return "".join(self.trace_text)

    def close(self) -> None:
        if self._h:
            lib.nvx_sitor_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


# --------------------------------------------------------- synthetic source
def make_stream(carriers: Sequence[dict], seed: int = 1, noise_amp: int = 1500) -> N.SynthStream:
    """carriers: dicts with freq_hz, bits ('B'/'Y' str) and optional shift_hz (85),
    amplitude (8000), phase0 (0), bit_offset (0)."""
    s = N.SynthStream()
    s.seed, s.noise_amp, s.n_carriers = seed & 0xFFFFFFFF, noise_amp, len(carriers)
    s._keep = []
    for i, c in enumerate(carriers):
        bits = c["bits"].encode("ascii")
        s._keep.append(bits)
        cr = s.carrier[i]
        cr.freq_hz, cr.shift_hz = int(c["freq_hz"]), int(c.get("shift_hz", 85))
        cr.amplitude, cr.phase0 = int(c.get("amplitude", 8000)), int(c.get("phase0", 0)) & 0xFFFFFFFF
        cr.bit_offset, cr.n_bits, cr.bits = int(c.get("bit_offset", 0)), len(bits), bits
    return s


def synth_host(stream: N.SynthStream, rate: int, n: int, n0: int = 0) -> np.ndarray:
    """n complex samples as int16 [n, 2] (I, Q) from the host generator."""
    out = np.empty((n, 2), dtype=np.int16)
    N.check(lib.nvx_synth_host(C.byref(stream), rate, n0, n, N.as_ptr(out)), "nvx_synth_host")
    return out


class DeviceBuffer:
    """hipMalloc'ed bytes owned by Python (through the C ABI, no HIP binding needed)."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = device, nbytes
        self.ptr = lib.nvx_device_alloc(device, nbytes)
        if not self.ptr:
            raise NvxError(N.ERR_NOMEM, f"nvx_device_alloc({nbytes})")

    def upload(self, a: np.ndarray, offset: int = 0) -> None:
        a = np.ascontiguousarray(a)
        N.check(lib.nvx_memcpy_h2d(self.device, self.ptr + offset, N.as_ptr(a), a.nbytes), "nvx_memcpy_h2d")

    def download(self, nbytes: int, offset: int = 0, dtype=np.uint8) -> np.ndarray:
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        N.check(lib.nvx_memcpy_d2h(self.device, N.as_ptr(out), self.ptr + offset, out.nbytes), "nvx_memcpy_d2h")
        return out

    def free(self) -> None:
        if self.ptr:
            lib.nvx_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


def synth_device(streams: Sequence[N.SynthStream], rate: int, n: int, buf: DeviceBuffer, pitch: int) -> None:
    arr = (N.SynthStream * len(streams))(*streams)
    arr._keep = list(streams)
    N.check(lib.nvx_synth_device(buf.device, arr, len(streams), rate, n, buf.ptr, pitch), "nvx_synth_device")


def channelise(raw: "DeviceBuffer", pitch_raw: int, first_sample: int, n_wide: int, n_out: int, sub: "DeviceBuffer", pitch_sub: int,
               sub_first: int = 0, hist_in: Optional["DeviceBuffer"] = None, hist_out: Optional["DeviceBuffer"] = None,
               hip_stream: Optional[int] = None) -> None:
    """Wideband front-end (nvx_channelise_resident): n_wide streams at 2.016 MS/s -> 8*n_wide sub-bands at 252 kS/s.
    Without hip_stream it runs on the null stream and the device is synchronised afterwards; with the
    stream of a Pipeline (Pipeline.hip_stream) it is simply ordered in front of that pipeline's launches."""
    N.check(lib.nvx_channelise_resident(raw.device, raw.ptr, pitch_raw, first_sample, n_wide, n_out,
                                        hist_in.ptr if hist_in else None, hist_out.ptr if hist_out else None,
                                        sub.ptr, pitch_sub, sub_first, hip_stream), "nvx_channelise_resident")
    if hip_stream is None:
        N.check(lib.nvx_device_sync(raw.device), "nvx_device_sync")


def channelise_time_stats(reset: bool = False):
    s, n = C.c_double(), C.c_uint64()
    N.check(lib.nvx_channelise_time_stats(C.byref(s), C.byref(n), int(reset)), "nvx_channelise_time_stats")
    return s.value, n.value


# ------------------------------------------------------------------ pipeline
class HandleStats:
    """Instrumentation of one nvx_handle (self._h): what a Pipeline offers about itself and what a Group offers about each
    of its members (Group.member_view)."""
    _h = None

    def wait_stats(self, reset: bool = False) -> Tuple[int, int, int]:
        """(polls, units that waited, launches) of the cascade's unit hand-over since the last reset."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        N.check(lib.nvx_cascade_wait_stats(self._h, C.byref(a), C.byref(b), C.byref(c), int(reset)), "nvx_cascade_wait_stats")
        return a.value, b.value, c.value

    def integrity_stats(self, reset: bool = False) -> Tuple[int, int, int]:
        """(hand-overs whose state block failed its seal and were repaired by a pre-roll, launches whose inherited state
        failed it, launches collected) -- nvx_cascade_integrity_stats; the first two are expected to be 0."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        N.check(lib.nvx_cascade_integrity_stats(self._h, C.byref(a), C.byref(b), C.byref(c), int(reset)), "nvx_cascade_integrity_stats")
        return a.value, b.value, c.value

    def tie_stats(self) -> Tuple[int, int, float]:
        """(near ties, evaluations, smallest relative margin) of the bit-timing arg-max since create / reset."""
        a, b, m = C.c_uint64(), C.c_uint64(), C.c_double()
        N.check(lib.nvx_demod_tie_stats(self._h, C.byref(a), C.byref(b), C.byref(m)), "nvx_demod_tie_stats")
        return a.value, b.value, m.value

    def enable_timing(self, on: bool = True) -> None:
        lib.nvx_enable_timing(self._h, int(on))

    def kernel_ms(self, which: int) -> float:
        return float(lib.nvx_last_kernel_ms(self._h, which))

    def kernel_time_stats(self, which: int, reset: bool = False):
        """(sum of HIP-event ms, number of launches) for kernel `which` (0 cascade, 1 demodulator, 2 nvx_fir3 of a wideband handle)."""
        s, n = C.c_double(), C.c_uint64()
        N.check(lib.nvx_kernel_time_stats(self._h, which, C.byref(s), C.byref(n), int(reset)), "nvx_kernel_time_stats")
        return s.value, n.value



class Pipeline(HandleStats):
    """nvx_handle wrapper: the GPU receive pipeline for n_streams IQ streams."""

    def __init__(self, n_streams: int = 1, raw_rate: bool = False, chain_mask: int = CHAIN_518 | CHAIN_490,
                 chain_masks: Optional[Iterable[int]] = None, labels: Optional[Sequence[Sequence[int]]] = None,
                 max_frames: int = 1, char_layer: bool = True, push_mode: bool = False, device: int = 0,
                 wideband: bool = False, bit_history: int = 0, store: "Optional[Store]" = None, stage0_order: int = 1,
                 eager_launch: bool = False, stall_timeout_ms: int = 0):
        self.messages: List[Tuple[int, int, str, str]] = []          # (stream, freq, bbbb, text)
        cfg = N.Config()
        lib.nvx_config_default(C.byref(cfg))
        cfg.device, cfg.n_streams, cfg.raw_rate = device, n_streams, int(raw_rate)
        cfg.stage0_order = int(stage0_order)
        cfg.eager_launch = int(eager_launch)
        cfg.stall_timeout_ms = int(stall_timeout_ms)
        cfg.chain_mask, cfg.max_frames, cfg.char_layer, cfg.push_mode = chain_mask, max_frames, int(char_layer), int(push_mode)
        cfg.wideband = int(wideband)
        cfg.bit_history = int(bit_history)
        if chain_masks is not None:
            chain_masks = list(chain_masks)
            self._masks = (C.c_uint8 * len(chain_masks))(*chain_masks)
            cfg.chain_masks = self._masks
        if labels is not None:
            flat = [int(v) for pair in labels for v in pair]
            self._labels = (C.c_int * len(flat))(*flat)
            cfg.labels = self._labels
        self._cb = N.MESSAGE_FN(lambda u, s, b, m, f: self.messages.append((s, f, b.decode("latin1"), m.decode("latin1"))))
        cfg.on_message = self._cb
        if store is not None:                        # messages go straight from the C character layer into SQLite
            cfg.on_message = C.cast(lib.nvx_store_on_message, N.MESSAGE_FN)
            cfg.user = store._s
            self._store = store
        self.n_streams, self.raw_rate, self.max_frames, self.device = n_streams, bool(raw_rate), max_frames, device
        self.wideband = bool(wideband)
        self.frame = FRAME_RAW if (raw_rate or wideband) else FRAME_IN
        h = C.c_void_p()
        N.check(lib.nvx_create(C.byref(cfg), C.byref(h)), "nvx_create")
        self._h = h
        self._bits = {}

    # host input ---------------------------------------------------------
    def push(self, stream: int, iq: np.ndarray) -> None:
        iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
        N.check(lib.nvx_push_iq(self._h, stream, N.as_ptr(iq), iq.shape[0]), "nvx_push_iq")

    def push_planar(self, stream: int, xi: np.ndarray, xq: np.ndarray) -> None:
        xi = np.ascontiguousarray(xi, dtype=np.int16); xq = np.ascontiguousarray(xq, dtype=np.int16)
        N.check(lib.nvx_push_planar(self._h, stream, N.as_ptr(xi), N.as_ptr(xq), xi.size), "nvx_push_planar")

    def flush(self) -> None:
        N.check(lib.nvx_flush(self._h), "nvx_flush")

    def finish(self, stream: Optional[int] = None) -> None:
        """End of the input (nvx_finish / nvx_stream_finish): flush, then the last, partial frame of every stream (or of
        `stream`) at its TRUE length -- afterwards the bits are exactly the reference's on the same samples, whatever
        their number (receiver/capt_sched.c:509-513 stops with its last sample).  Such a stream is ended: reset() starts anew."""
        if stream is None:
            N.check(lib.nvx_finish(self._h), "nvx_finish")
        else:
            N.check(lib.nvx_stream_finish(self._h, stream), "nvx_stream_finish")

    def poll(self) -> None:
        """Take in whatever launched work has already finished; never waits (nvx_poll)."""
        N.check(lib.nvx_poll(self._h), "nvx_poll")

    def set_active(self, stream: int, active: bool) -> None:
        """A silent stream (active=False) is one the launches of a push-mode handle no longer wait for."""
        N.check(lib.nvx_stream_set_active(self._h, stream, int(active)), "nvx_stream_set_active")

    def stream_stats(self, stream: int = 0) -> Tuple[bool, int, int]:
        """(active, frames the stream has been through, launches of the handle that covered only some streams)."""
        a, f, p = C.c_int(), C.c_uint64(), C.c_uint64()
        N.check(lib.nvx_stream_stats(self._h, stream, C.byref(a), C.byref(f), C.byref(p)), "nvx_stream_stats")
        return bool(a.value), f.value, p.value

    def set_trace(self, on_text: Optional[Callable[[str], None]]) -> None:
        """Route the character layers' trace (what the reference prints to stdout: nav_b_sm.C) to on_text, or turn it off (None)."""
        self._trace_cb = N.SITOR_TRACE_FN(lambda u, t: on_text(t.decode("latin1"))) if on_text else C.cast(None, N.SITOR_TRACE_FN)
        N.check(lib.nvx_set_trace(self._h, self._trace_cb, None), "nvx_set_trace")

    def decode_wav(self, path: str, stream: int = 0) -> int:
        return N.check(lib.nvx_decode_wav(self._h, stream, path.encode()), "nvx_decode_wav")

    # resident input -------------------------------------------------------
    def process_resident(self, buf: DeviceBuffer, pitch: int, first_frame: int, n_frames: int, hip_stream: Optional[int] = None) -> None:
        N.check(lib.nvx_process_resident(self._h, buf.ptr, pitch, first_frame, n_frames, hip_stream or None),
                "nvx_process_resident")

    @property
    def hip_stream(self) -> int:
        return lib.nvx_handle_stream(self._h)

    def fetch(self) -> None:
        N.check(lib.nvx_fetch_bits(self._h), "nvx_fetch_bits")

    # results ---------------------------------------------------------------
    def bits(self, stream: int = 0, chain: int = 0) -> str:
        """All bits decoded so far on (stream, chain)."""
        key = (stream, chain)
        cap = 1 << 16
        buf = C.create_string_buffer(cap)
        acc = self._bits.get(key, "")
        while True:
            n = lib.nvx_poll_bits(self._h, stream, chain, buf, cap)
            acc += buf.raw[:n].decode("ascii")
            if n < cap:
                break
        self._bits[key] = acc
        return acc

    def bit_count(self, stream: int = 0, chain: int = 0) -> int:
        return lib.nvx_bit_count(self._h, stream, chain)

    def reset(self) -> None:
        N.check(lib.nvx_reset(self._h), "nvx_reset")
        self._bits.clear()
        self.messages.clear()

    def stream_reset(self, stream: int) -> None:
        """One stream starts anew (nvx_stream_reset): its filters, demodulator, character layers and bit counters; the other
        streams keep everything they carry.  What an ended stream (finish / decode_wav) needs before its next input."""
        N.check(lib.nvx_stream_reset(self._h, stream), "nvx_stream_reset")
        per = 8 if self.wideband else 1
        for key in [k for k in self._bits if per * stream <= k[0] < per * (stream + 1)]:
            del self._bits[key]

    def enable_debug(self, on: bool = True) -> None:
        N.check(lib.nvx_enable_debug(self._h, int(on)), "nvx_enable_debug")

    STATE_BLOCK_BYTES = 4352

    def debug_state(self, stream: int = 0) -> np.ndarray:
        """The carried FIR state block the stream's next launch will read, as 544 uint64 (nvx_debug_cascade_state)."""
        out = np.empty(self.STATE_BLOCK_BYTES // 8, dtype=np.uint64)
        N.check(lib.nvx_debug_cascade_state(self._h, stream, N.as_ptr(out), out.nbytes, 0), "nvx_debug_cascade_state")
        return out

    def debug_set_state(self, stream: int, block: np.ndarray) -> None:
        block = np.ascontiguousarray(block, dtype=np.uint64)
        N.check(lib.nvx_debug_cascade_state(self._h, stream, N.as_ptr(block), block.nbytes, 1), "nvx_debug_cascade_state")

    CLOCK_PERIOD = 163296

    def debug_advance_clock(self, stream: int, periods: int) -> None:
        """Move the stream's 900 S/s sample clock forward by periods x CLOCK_PERIOD without touching its state
        (nvx_debug_advance_clock): decoding must go on as if nothing had happened."""
        N.check(lib.nvx_debug_advance_clock(self._h, stream, periods), "nvx_debug_advance_clock")

    def debug_y3(self, stream: int = 0, chain: int = 0) -> np.ndarray:
        out = np.empty((self.max_frames * FRAME_Y3, 2), dtype=np.float64)
        n = lib.nvx_debug_y3(self._h, stream, chain, N.as_ptr(out), out.shape[0])
        return out[:n]

    def debug_dphi(self, stream: int = 0, chain: int = 0) -> np.ndarray:
        out = np.empty(self.max_frames * FRAME_Y3, dtype=np.float64)
        n = lib.nvx_debug_dphi(self._h, stream, chain, N.as_ptr(out), out.size)
        return out[:n]

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib.nvx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        self.close()


class Capture:
    """nvx_capture wrapper: the live-capture ring of header section B' on one stream of a push-mode Pipeline.
    feed(xi, xq) is the vendor callback (receiver/capt_sched.c:105): planar int16 arrays, copied during the call."""

    def __init__(self, pipe: Pipeline, stream: int = 0, ring_seconds: float = 8.0):
        c = C.c_void_p()
        N.check(lib.nvx_capture_start(pipe._h, stream, ring_seconds, C.byref(c)), "nvx_capture_start")
        self._c, self.pipe, self.stream = c, pipe, stream

    def feed(self, xi: np.ndarray, xq: np.ndarray) -> None:
        lib.nvx_capture_callback(xi.ctypes.data, xq.ctypes.data, None, xi.shape[0], 0, self._c)

    def stats(self) -> Tuple[int, int, int]:
        """(samples offered by the producer, dropped on overrun, handed to the pipeline)"""
        r, d, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib.nvx_capture_stats(self._c, C.byref(r), C.byref(d), C.byref(c))
        return r.value, d.value, c.value

    def latency(self, reset: bool = False) -> dict:
        """Decode latency of the frames collected so far (nvx_capture_latency): callback entry of a frame's last sample ->
        bits pollable and messages delivered, in ms."""
        n, p50, p99, mx, last = C.c_uint64(), C.c_double(), C.c_double(), C.c_double(), C.c_double()
        N.check(lib.nvx_capture_latency(self._c, C.byref(n), C.byref(p50), C.byref(p99), C.byref(mx), C.byref(last), int(reset)), "nvx_capture_latency")
        return {"frames": n.value, "p50_ms": p50.value, "p99_ms": p99.value, "max_ms": mx.value, "last_ms": last.value}

    def stalled(self) -> Tuple[bool, int]:
        ev = C.c_uint64()
        return bool(lib.nvx_capture_stalled(self._c, C.byref(ev))), ev.value

    def stop(self) -> None:
        """Drain the ring, stop the consumer thread, flush the pipeline."""
        if self._c:
            c, self._c = self._c, None
            N.check(lib.nvx_capture_stop(c), "nvx_capture_stop")


class Group:
    """nvx_group wrapper: n_streams sharded over `devices` (one handle + host thread per member, no collective)."""

    def __init__(self, devices: Sequence[int], n_streams: int, raw_rate: bool = False, chain_mask: int = CHAIN_518 | CHAIN_490,
                 chain_masks: Optional[Iterable[int]] = None, labels: Optional[Sequence[Sequence[int]]] = None, max_frames: int = 1,
                 char_layer: bool = True, push_mode: bool = False, host_threads: int = 0):
        self.messages: List[Tuple[int, int, str, str]] = []          # (global stream, freq, bbbb, text)
        cfg = N.Config()
        lib.nvx_config_default(C.byref(cfg))
        cfg.n_streams, cfg.raw_rate, cfg.chain_mask = n_streams, int(raw_rate), chain_mask
        cfg.max_frames, cfg.char_layer, cfg.push_mode, cfg.host_threads = max_frames, int(char_layer), int(push_mode), host_threads
        if chain_masks is not None:
            chain_masks = list(chain_masks)
            self._masks = (C.c_uint8 * len(chain_masks))(*chain_masks)
            cfg.chain_masks = self._masks
        if labels is not None:
            flat = [int(v) for pair in labels for v in pair]
            self._labels = (C.c_int * len(flat))(*flat)
            cfg.labels = self._labels
        self._cb = N.MESSAGE_FN(lambda u, s, b, m, f: self.messages.append((s, f, b.decode("latin1"), m.decode("latin1"))))
        cfg.on_message = self._cb
        devs = (C.c_int * len(devices))(*devices)
        g = C.c_void_p()
        N.check(lib.nvx_group_create(devs, len(devices), C.byref(cfg), C.byref(g)), "nvx_group_create")
        self._g = g
        self.n_streams = n_streams
        self.members = []                                            # (device, first stream, stream count)
        for m in range(lib.nvx_group_size(g)):
            d, f, n = C.c_int(), C.c_int(), C.c_int()
            N.check(lib.nvx_group_member(g, m, C.byref(d), C.byref(f), C.byref(n), None), "nvx_group_member")
            self.members.append((d.value, f.value, n.value))
        self._bits = {}

    def member_of(self, stream: int) -> int:
        return lib.nvx_group_member_of(self._g, stream)

    def member_view(self, m: int) -> HandleStats:
        """Instrumentation of member m's handle (kernel timing, hand-over and seal statistics); the group keeps ownership."""
        h = C.c_void_p()
        N.check(lib.nvx_group_member(self._g, m, None, None, None, C.byref(h)), "nvx_group_member")
        v = HandleStats()
        v._h = h
        return v

    def process_resident(self, ptrs: Sequence[int], pitch: int, first_frame: int, n_frames: int) -> None:
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        N.check(lib.nvx_group_process_resident(self._g, arr, pitch, first_frame, n_frames), "nvx_group_process_resident")

    def fetch(self) -> None:
        N.check(lib.nvx_group_fetch_bits(self._g), "nvx_group_fetch_bits")

    def push(self, stream: int, iq: np.ndarray) -> None:
        iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
        N.check(lib.nvx_group_push_iq(self._g, stream, N.as_ptr(iq), iq.shape[0]), "nvx_group_push_iq")

    def flush(self) -> None:
        N.check(lib.nvx_group_flush(self._g), "nvx_group_flush")

    def finish(self) -> None:
        """End of every stream's input (nvx_group_finish = nvx_finish of every member)."""
        N.check(lib.nvx_group_finish(self._g), "nvx_group_finish")

    def reset(self) -> None:
        N.check(lib.nvx_group_reset(self._g), "nvx_group_reset")
        self._bits.clear(); self.messages.clear()

    def bits(self, stream: int, chain: int = 0) -> str:
        cap = 1 << 16
        buf = C.create_string_buffer(cap)
        acc = self._bits.get((stream, chain), "")
        while True:
            n = lib.nvx_group_poll_bits(self._g, stream, chain, buf, cap)
            acc += buf.raw[:n].decode("ascii")
            if n < cap:
                break
        self._bits[(stream, chain)] = acc
        return acc

    def bit_count(self, stream: int, chain: int = 0) -> int:
        return lib.nvx_group_bit_count(self._g, stream, chain)

    def close(self) -> None:
        if getattr(self, "_g", None):
            lib.nvx_group_destroy(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        self.close()


# ----------------------------------------------------------------------- WAV
class Store:
    """nvx_store wrapper: the SQLite sink the reference's web server reads (message_store.c:59-97)."""

    def __init__(self, path: str, create_schema: bool = True):
        s = C.c_void_p()
        N.check(lib.nvx_store_open(str(path).encode(), int(create_schema), C.byref(s)), "nvx_store_open")
        self._s = s

    def add_message(self, bbbb: str, message: str, freq: int) -> int:
        return lib.nvx_store_add_message(self._s, bbbb.encode("latin1"), message.encode("latin1"), freq)

    def purge(self, max_age_seconds: int = 0) -> int:
        return N.check(lib.nvx_store_purge(self._s, max_age_seconds), "nvx_store_purge")

    def set_time(self, unix_seconds: int) -> None:
        lib.nvx_store_set_time(self._s, unix_seconds)

    def stats(self) -> Tuple[int, int]:
        a, f = C.c_uint64(), C.c_uint64()
        lib.nvx_store_stats(self._s, C.byref(a), C.byref(f))
        return a.value, f.value

    def close(self) -> None:
        if self._s:
            lib.nvx_store_close(self._s)
            self._s = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def wav_write(path: str, iq: np.ndarray, rate: int = RATE_IN) -> None:
    """2-channel 16-bit PCM, as receiver/capt_sched.c:87-96 configures its capture file."""
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
    w = lib.nvx_wav_open(path.encode(), 2)
    if not w:
        raise IOError(lib.nvx_wav_err().decode())
    lib.nvx_wav_set_format(w, 1); lib.nvx_wav_set_num_channels(w, 2)
    lib.nvx_wav_set_sample_rate(w, rate); lib.nvx_wav_set_sample_size(w, 2)
    n = lib.nvx_wav_write(w, N.as_ptr(iq), iq.shape[0])
    lib.nvx_wav_close(w)
    if n != iq.shape[0]:
        raise IOError("short WAV write")


def wav_read(path: str) -> Tuple[np.ndarray, int]:
    w = lib.nvx_wav_open(path.encode(), 1)
    if not w:
        raise IOError(lib.nvx_wav_err().decode())
    n, ch, rate = lib.nvx_wav_get_length(w), lib.nvx_wav_get_num_channels(w), lib.nvx_wav_get_sample_rate(w)
    out = np.empty((n, ch), dtype=np.int16)
    got = lib.nvx_wav_read(w, N.as_ptr(out), n)
    lib.nvx_wav_close(w)
    return out[:got], rate
