#!/usr/bin/env python3
"""What the device-span guard in front of every launch over caller memory costs (nvx_check_device_span, nvx_api.cpp):
hipMemGetAddressRange on a pointer inside a large allocation, with a few thousand other allocations alive."""
import ctypes as C
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import navtex_amd as nv

hip = C.CDLL("libamdhip64.so")
big = nv.DeviceBuffer(8 << 30)
others = [nv.DeviceBuffer(1 << 16) for _ in range(4000)]
base, size = C.c_void_p(), C.c_size_t()
hip.hipMemGetAddressRange.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
p = C.c_void_p(big.ptr + (5 << 30) + 12345 * 4)
for n in (1000, 100000):
    t0 = time.perf_counter()
    for _ in range(n):
        rc = hip.hipMemGetAddressRange(C.byref(base), C.byref(size), p)
    dt = time.perf_counter() - t0
    print(f"{n} calls: {dt / n * 1e6:.2f} us per call (python loop included), rc {rc}, base ok {base.value == big.ptr}, size {size.value}")
for b in others: b.free()
big.free()
