"""ctypes binding of libnavtex_amd.so (the C ABI in include/navtex_amd.h).

This module is plumbing for tests and bench.py: it adds no signal processing of
its own and has no fallback -- if the native library is missing, import fails.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

# NAVTEX_AMD_LIB: load another build of the same library (A/B runs of kernel variants)
_LIB_PATH = Path(os.environ.get("NAVTEX_AMD_LIB") or (Path(__file__).resolve().parent / "libnavtex_amd.so"))

OK, ERR_ARG, ERR_NODEV, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_IO = 0, -1, -2, -3, -4, -5, -6
RATE_RAW, RATE_IN = 2016000, 252000
FRAME_BITS, FRAME_IN, FRAME_RAW, FRAME_Y3 = 32, 80640, 645120, 288
CHAIN_518, CHAIN_490 = 1, 2

MESSAGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int)
SITOR_MSG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int)
SITOR_TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int), ("n_streams", C.c_int), ("raw_rate", C.c_int), ("chain_mask", C.c_uint32),
        ("chain_masks", C.POINTER(C.c_uint8)), ("labels", C.POINTER(C.c_int)), ("max_frames", C.c_int),
        ("char_layer", C.c_int), ("on_message", MESSAGE_FN), ("user", C.c_void_p), ("push_mode", C.c_int),
        ("wideband", C.c_int),
        ("bit_history", C.c_int),
        ("host_threads", C.c_int),
        ("stage0_order", C.c_int),
        ("eager_launch", C.c_int),
        ("stall_timeout_ms", C.c_int),
    ]


class Carrier(C.Structure):
    _fields_ = [
        ("freq_hz", C.c_int32), ("shift_hz", C.c_int32), ("amplitude", C.c_int32), ("phase0", C.c_uint32),
        ("bit_offset", C.c_uint32), ("n_bits", C.c_uint32), ("bits", C.c_char_p),
    ]


class SynthStream(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("noise_amp", C.c_int32), ("n_carriers", C.c_int32), ("carrier", Carrier * 16)]


def _load() -> C.CDLL:
    if not _LIB_PATH.exists():
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `python navtex_amd/build.py` "
            "(hipcc, gfx950). navtex_amd has no pure-Python or CPU fallback.")
    lib = C.CDLL(str(_LIB_PATH))
    vp, sz, i, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
    sig = {
        "nvx_last_error": (C.c_char_p, []), "nvx_version": (C.c_char_p, []), "nvx_abi_version": (i, []),
        "nvx_sample_to_int16": (i, [C.c_double, C.POINTER(C.c_int16)]), "nvx_shim_stats": (i, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "nvx_debug_advance_clock": (i, [vp, i, C.c_uint64]),
        "init_fir_filter1": (None, []), "sample_in_1": (None, [C.c_double, C.c_double]), "init_fir2_wrapper": (None, []),
        "nvx_set_trace": (i, [vp, SITOR_TRACE_FN, vp]),
        "nvx_shim_latency": (i, [C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), i]),
        "nvx_shim_flush": (i, []), "nvx_shim_finish": (i, []), "nvx_shim_bits": (sz, [i, C.c_char_p, sz]),
        "nvx_StreamACallback": (None, [vp, vp, vp, C.c_uint, C.c_uint, vp]),
        "nvx_capture_start": (i, [vp, i, C.c_double, C.POINTER(vp)]), "nvx_capture_callback": (None, [vp, vp, vp, C.c_uint, C.c_uint, vp]),
        "nvx_capture_latency": (i, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), i]),
        "nvx_capture_stop": (i, [vp]), "nvx_capture_record": (i, [vp, C.c_char_p]), "nvx_fsm_selftest": (i, [u32, i]),
        "nvx_cascade_wait_stats": (i, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i]),
        "nvx_debug_cascade_state": (i, [vp, i, vp, sz, i]),
        "nvx_cascade_integrity_stats": (i, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i]), "nvx_capture_pause": (None, [vp, i]),
        "nvx_capture_stats": (None, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "nvx_config_default": (None, [C.POINTER(Config)]),
        "nvx_create": (i, [C.POINTER(Config), C.POINTER(vp)]), "nvx_destroy": (None, [vp]), "nvx_reset": (i, [vp]), "nvx_stream_reset": (i, [vp, i]),
        "nvx_push_iq": (i, [vp, i, vp, sz]), "nvx_push_planar": (i, [vp, i, vp, vp, sz]), "nvx_flush": (i, [vp]),
        "nvx_finish": (i, [vp]), "nvx_stream_finish": (i, [vp, i]),
        "nvx_poll_bits": (sz, [vp, i, i, C.c_char_p, sz]),
        "nvx_process_resident": (i, [vp, vp, sz, sz, i, vp]), "nvx_fetch_bits": (i, [vp]),
        "nvx_bit_count": (sz, [vp, i, i]),
        "nvx_last_kernel_ms": (C.c_float, [vp, i]), "nvx_enable_timing": (None, [vp, i]), "nvx_enable_debug": (i, [vp, i]),
        "nvx_group_create": (i, [C.POINTER(C.c_int), i, C.POINTER(Config), C.POINTER(vp)]), "nvx_group_destroy": (None, [vp]),
        "nvx_group_reset": (i, [vp]), "nvx_group_size": (i, [vp]),
        "nvx_group_member": (i, [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(vp)]), "nvx_group_member_of": (i, [vp, i]),
        "nvx_group_process_resident": (i, [vp, C.POINTER(vp), sz, sz, i]), "nvx_group_fetch_bits": (i, [vp]),
        "nvx_group_push_iq": (i, [vp, i, vp, sz]), "nvx_group_flush": (i, [vp]), "nvx_group_finish": (i, [vp]),
        "nvx_group_poll_bits": (sz, [vp, i, i, C.c_char_p, sz]), "nvx_group_bit_count": (sz, [vp, i, i]),
        "nvx_bind_thread_to_device": (i, [i]),
        "nvx_demod_tie_stats": (i, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
        "nvx_capture_error": (i, [vp]),
        "nvx_capture_stalled": (i, [vp, C.POINTER(C.c_uint64)]), "nvx_capture_set_stall_timeout": (None, [vp, C.c_double]),
        "nvx_stream_set_active": (i, [vp, i, i]), "nvx_poll": (i, [vp]),
        "nvx_stream_stats": (i, [vp, i, C.POINTER(i), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "nvx_kernel_time_stats": (i, [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_uint64), i]),
        "nvx_debug_y3": (sz, [vp, i, i, vp, sz]), "nvx_debug_dphi": (sz, [vp, i, i, vp, sz]),
        "nvx_device_count": (i, []), "nvx_device_alloc": (vp, [i, sz]), "nvx_device_free": (None, [i, vp]),
        "nvx_memcpy_h2d": (i, [i, vp, vp, sz]), "nvx_memcpy_d2h": (i, [i, vp, vp, sz]), "nvx_device_sync": (i, [i]),
        "nvx_stream_create": (vp, [i]), "nvx_stream_destroy": (None, [i, vp]),
        "nvx_sitor_new": (vp, [i, SITOR_MSG_FN, vp]), "nvx_sitor_set_trace": (None, [vp, SITOR_TRACE_FN, vp]),
        "nvx_sitor_free": (None, [vp]), "nvx_sitor_reset": (None, [vp]),
        "nvx_sitor_receive_bit": (None, [vp, C.c_char]), "nvx_sitor_receive_bits": (None, [vp, C.c_char_p, sz]),
        "nvx_wav_open": (vp, [C.c_char_p, u32]), "nvx_wav_close": (i, [vp]),
        "nvx_wav_read": (sz, [vp, vp, sz]), "nvx_wav_write": (sz, [vp, vp, sz]),
        "nvx_wav_set_format": (None, [vp, C.c_uint16]), "nvx_wav_set_num_channels": (None, [vp, C.c_uint16]),
        "nvx_wav_set_sample_rate": (None, [vp, u32]), "nvx_wav_set_sample_size": (None, [vp, sz]),
        "nvx_wav_get_format": (C.c_uint16, [vp]), "nvx_wav_get_num_channels": (C.c_uint16, [vp]),
        "nvx_wav_get_sample_rate": (u32, [vp]), "nvx_wav_get_sample_size": (sz, [vp]), "nvx_wav_get_length": (sz, [vp]),
        "nvx_wav_err": (C.c_char_p, []), "nvx_decode_wav": (i, [vp, i, C.c_char_p]),
        "nvx_sitor_encode": (sz, [C.c_char_p, i, C.c_char_p, sz]),
        "nvx_synth_host": (i, [C.POINTER(SynthStream), u32, C.c_uint64, sz, vp]),
        "nvx_synth_device": (i, [i, C.POINTER(SynthStream), i, u32, sz, vp, sz]),
        "nvx_atan2_host": (C.c_double, [C.c_double, C.c_double]),
        "nvx_channelise_resident": (i, [i, vp, sz, sz, i, sz, vp, vp, vp, sz, sz, vp]),
        "nvx_handle_stream": (vp, [vp]), "nvx_channelise_timing": (None, [i]),
        "nvx_channelise_time_stats": (i, [C.POINTER(C.c_double), C.POINTER(C.c_uint64), i]),
        "nvx_store_open": (i, [C.c_char_p, i, C.POINTER(vp)]), "nvx_store_close": (None, [vp]),
        "nvx_store_add_message": (i, [vp, C.c_char_p, C.c_char_p, i]),
        "nvx_store_on_message": (None, [vp, i, C.c_char_p, C.c_char_p, i]),
        "nvx_store_purge": (i, [vp, C.c_long]), "nvx_store_set_time": (None, [vp, C.c_int64]),
        "nvx_store_stats": (None, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if os.environ.get("NAVTEX_AMD_LIB"):         # an older build in an A/B run may lack the newest entry points
                continue
            raise ImportError(f"{_LIB_PATH} does not export {name}: rebuild it (python navtex_amd/build.py)")
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
EXPORTS = None  # filled lazily by tests from include/navtex_amd.h


class NvxError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        super().__init__(f"{where}: error {code}: {lib.nvx_last_error().decode(errors='replace')}")


def check(rc: int, where: str) -> int:
    if rc < 0:
        raise NvxError(rc, where)
    return rc


def as_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)
