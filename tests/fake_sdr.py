"""A stand-in for the SDRplay API's streaming thread (TEST INFRASTRUCTURE; bench.py's live_latency leg and the GPU tests
use it): hands a recorded IQ stream to a capture ring in the vendor callback's shape (receiver/capt_sched.c:105: planar
int16 xi / xq, a few hundred to a few thousand samples per call, arrays valid only during the call) AT THE REAL RATE, with
jittered packet sizes, and remembers when each frame's last sample was handed over."""
from __future__ import annotations

import threading
import time

import numpy as np


class FakeSdr(threading.Thread):
    def __init__(self, capture, iq: np.ndarray, rate: int, frame: int, seed: int = 1, packet=(1000, 1700), speed: float = 1.0):
        super().__init__(daemon=True)
        self.cap, self.rate, self.frame, self.speed = capture, rate, frame, speed
        self.xi = np.ascontiguousarray(iq[:, 0]); self.xq = np.ascontiguousarray(iq[:, 1])
        self.rng = np.random.default_rng(seed)
        self.packet = packet
        self.frame_done_at = []                 # time.monotonic() when the callback that carried frame k's last sample was ENTERED
        self.late_ms = 0.0                      # how far behind its schedule the thread ever was (a starved host shows here)
        self.t0 = None
        self.packets = 0

    def run(self):
        n, pos = self.xi.shape[0], 0
        sizes = self.rng.integers(self.packet[0], self.packet[1], size=n // self.packet[0] + 2)
        self.t0 = t0 = time.monotonic()
        k = 0
        while pos < n:
            m = int(min(n - pos, sizes[k])); k += 1
            due = t0 + (pos + m) / (self.rate * self.speed)          # a packet is delivered when its last sample has been sampled
            now = time.monotonic()
            if due > now:
                time.sleep(due - now)
            else:
                self.late_ms = max(self.late_ms, (now - due) * 1e3)
            t_enter = time.monotonic()
            self.cap.feed(self.xi[pos:pos + m], self.xq[pos:pos + m])
            f0, f1 = pos // self.frame, (pos + m) // self.frame
            self.frame_done_at.extend([t_enter] * (f1 - f0))
            pos += m
            self.packets += 1
