"""Deterministic test signals shared by the golden generator, the CPU tests, the
GPU tests and bench.py.  Everything derives from (seed, text, parameters) through
the product's integer generator (nvx_synth_host / nvx_synth_device), so the
build container and the GPU box regenerate bit-identical IQ."""
from __future__ import annotations

GLOBAL_SEED = 0x4E415654        # "NAVT"


def mix32(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def stream_text(stream_id: int) -> str:
    b1 = chr(ord("A") + stream_id % 26)
    b2 = chr(ord("A") + (stream_id // 26) % 26)
    return (f"ZCZC {b1}{b2}{stream_id % 100:02d}\n"
            f"NAVTEX AMD TEST STREAM {stream_id} = GALE WARNING 7/8 NW-LY. POSITION 51-30N 003-15E\n"
            f"NNNN\n")


def stream_params(nv, stream_id: int, rate: int, freq_hz: int = 14000, n_phasing: int = 40,
                  noise_amp: int = 1500, amplitude: int = 8000, text: str | None = None):
    """One-carrier stream with per-stream text, timing offset, phase and seed."""
    spb = rate // 100
    h = mix32(GLOBAL_SEED ^ mix32(stream_id + 1))
    bits = nv.sitor_encode(text if text is not None else stream_text(stream_id), n_phasing)
    # keep the bit boundary off the exact middle between two 900 S/s sampling instants
    off = (mix32(h ^ 0xA5A5A5A5) % spb) | 1
    return nv.make_stream([dict(freq_hz=freq_hz, bits=bits, bit_offset=off % spb, phase0=mix32(h ^ 0x3C3C3C3C),
                                amplitude=amplitude)], seed=h, noise_amp=noise_amp), bits
