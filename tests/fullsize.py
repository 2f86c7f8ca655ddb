"""Total parity of a resident batch: EVERY stream's decoded bits against the oracle.

TEST INFRASTRUCTURE (uses the oracle): called from tests/ and from bench.py's parity gate, never from the product.
The batch stays where it is (HBM); it is copied back a chunk of streams at a time, the oracle decodes the chunk with
OpenMP over the streams, and the bit strings are compared.  At BASELINE configs[3] size (4096 streams x 12 frames,
127 GB) that is 64 chunks of 2 GB."""
from __future__ import annotations

import time
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np


def verify_streams(ob, buf, pitch: int, n_samples: int, raw: bool, gpu_bits: Callable[[int], str], streams: Sequence[int],
                   ncpu: int, chain_mask: int = 1, chunk: int = 64) -> Tuple[int, List[int], float]:
    """Compare gpu_bits(s) with the oracle's bits for every s in `streams` (indices into buf's [stream][pitch] layout of
    packed int16 IQ; n_samples complex samples per stream are decoded, from reset state).  One chain per stream
    (chain_mask 1 = 518, 2 = 490).  Returns (streams checked, list of differing streams, seconds spent)."""
    t0 = time.perf_counter()
    n252 = n_samples // 8 if raw else n_samples
    bad: List[int] = []
    checked = 0
    streams = list(streams)
    for c0 in range(0, len(streams), chunk):
        ids = streams[c0:c0 + chunk]
        sample = np.empty((len(ids), n_samples, 2), dtype=np.int16)
        contiguous = ids == list(range(ids[0], ids[0] + len(ids))) and pitch == n_samples
        if contiguous:                               # one copy for the whole chunk
            sample[:] = buf.download(len(ids) * n_samples * 4, offset=ids[0] * pitch * 4, dtype=np.int16).reshape(len(ids), n_samples, 2)
        else:
            for k, s in enumerate(ids):
                sample[k] = buf.download(n_samples * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)
        _secs, want = ob.bench(sample, len(ids), n252, raw, chain_mask, ncpu, want_bits=True)
        for k, s in enumerate(ids):
            got = gpu_bits(s)
            if got != want[k] or not want[k]:
                bad.append(s)
        checked += len(ids)
    return checked, bad, time.perf_counter() - t0


def _same_or_retained_tail(got: str, want: str, count: Optional[int]) -> bool:
    """got == want -- or, when the handle has trimmed its poll history (cfg.bit_history: a receiver runs for weeks), got is the
    END of want, at least 4096 bits of it, and the handle's total bit count equals len(want)."""
    if got == want:
        return bool(want)
    return bool(want) and count == len(want) and len(got) >= 4096 and want.endswith(got)


def verify_replay(ob, buf, pitch: int, n_samples: int, raw, gpu_bits: Callable[[int], object], streams: Sequence[int], ncpu: int,
                  loops: int, chain_mask: int = 1, chunk: int = 64, gpu_count: Optional[Callable[[int], int]] = None) -> Tuple[int, List[int], float]:
    """What a benchmark loop leaves behind: gpu_bits(s) -- everything the handle has decoded on stream s since its reset,
    after `loops` launches over the SAME n_samples of the resident batch -- against the oracle fed those samples `loops`
    times through one pipe per stream (state carried from repeat to repeat, as the handle carries it from launch to
    launch).  chain_mask 3: gpu_bits(s) returns [chain 0, chain 1].  gpu_count(s) (optional): the handle's total bit count of
    stream s -- lets a handle whose poll history has been trimmed pass on the retained tail.  Returns (checked, differing
    streams, seconds)."""
    t0 = time.perf_counter()
    n252 = n_samples // 8 if raw else n_samples
    bad: List[int] = []
    streams = list(streams)
    for c0 in range(0, len(streams), chunk):
        ids = streams[c0:c0 + chunk]
        sample = np.empty((len(ids), n_samples, 2), dtype=np.int16)
        for k, s in enumerate(ids):
            sample[k] = buf.download(n_samples * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)
        _secs, want = ob.replay(sample, len(ids), n252, raw, chain_mask, ncpu, loops)
        for k, s in enumerate(ids):
            got = gpu_bits(s)
            w = want[k]
            if isinstance(w, str):
                ok = _same_or_retained_tail(got, w, gpu_count(s) if gpu_count else None)
            else:
                ok = got == w and all(w)
            if not ok:
                bad.append(s)
    return len(streams), bad, time.perf_counter() - t0


def spread(n_total: int, n_pick: int) -> List[int]:
    """n_pick stream indices spread over 0..n_total-1, first and last included."""
    if n_pick >= n_total:
        return list(range(n_total))
    return sorted({int(round(i * (n_total - 1) / (n_pick - 1))) for i in range(n_pick)})
