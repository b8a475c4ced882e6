"""ctypes binding of include/voice_synth.h (the C ABI of libvoicesynth.so).

Python is only the test / bench driver of this repository; the host side of the product is C
(voice_synth_amd/csrc, voice_synth_amd/cli).  This module therefore does nothing but declare
the entry points and fail loudly when the library is missing: there is no Python or CPU
fallback for any of them.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("VS_LIB", "libvoicesynth.so"))  # VS_LIB: A/B builds in tools/

VS_ORDER = 22
VS_NCOEF = 23
VS_MAX_ORDER = 40
VS_MAX_NCOEF = 41

VS_OK = 0
VS_ERR_ARG = -1
VS_ERR_RANGE = -2
VS_ERR_UNSUPPORTED = -3
VS_ERR_HIP = -4
VS_ERR_NOMEM = -5
VS_ERR_NODEVICE = -6
VS_ERR_IO = -7
VS_USAGE = -8
VS_ERR_INTERNAL = -9

VS_FLAG_JITTER = 0x1
VS_FLAG_SHIMMER = 0x2
VS_FLAG_NOISE = 0x4

VS_ARITH_EXACT = 0
VS_ARITH_FMA = 1
VS_ARITH_F32 = 2

VS_KIND_SYNTH = 0
VS_KIND_SOURCE = 1
VS_KIND_FILTER = 2


class Lane(C.Structure):
    """struct vs_lane"""

    _fields_ = [
        ("jitter", C.c_float),
        ("cq", C.c_float),
        ("K", C.c_float),
        ("Fg", C.c_float),
        ("F0", C.c_float),
        ("DC", C.c_float),
        ("noise", C.c_float),
        ("Kvar", C.c_float),
        ("shimmer", C.c_float),
        ("fs", C.c_int32),
        ("amp", C.c_int32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
        ("gain", C.c_float),
        ("pre_emphasis", C.c_float),
        ("vowel", C.c_int32),
        ("out_snr", C.c_float),
        ("A", C.c_double * VS_MAX_NCOEF),
        ("order", C.c_int32),
        ("reserved_", C.c_int32),
        ("out_seed", C.c_uint64),
    ]


class CycleRec(C.Structure):
    """struct vs_cycle_rec"""

    _fields_ = [("S", C.c_float), ("x_pow", C.c_float), ("w_pow", C.c_float), ("T", C.c_int32)]


class FlowgenCmd(C.Structure):
    _fields_ = [("lane", Lane), ("dur", C.c_float), ("wav_arg", C.c_int)]


class VowelCmd(C.Structure):
    _fields_ = [
        ("gain", C.c_float),
        ("pre_emphasis", C.c_float),
        ("snr", C.c_float),
        ("vowel", C.c_int),
        ("input_arg", C.c_int),
        ("output_arg", C.c_int),
        ("noise_arg", C.c_int),
    ]


class DevLane(C.Structure):
    """struct VsDevLane of csrc/vs_device.h (host expansion of a lane; CPU-side tests only)."""

    _fields_ = [
        ("gain", C.c_double),
        ("pre", C.c_double),
        ("jitter", C.c_float),
        ("shimmer", C.c_float),
        ("K", C.c_float),
        ("Kvar", C.c_float),
        ("DC", C.c_float),
        ("noise", C.c_float),
        ("t_hi", C.c_float),
        ("t_lo", C.c_float),
        ("a_hi", C.c_float),
        ("a_lo", C.c_float),
        ("amp", C.c_int32),
        ("P", C.c_int32),
        ("T2", C.c_int32),
        ("tab_off", C.c_int32),
        ("tbound", C.c_int32),
        ("dcs", C.c_int32),
        ("flags", C.c_uint32),
        ("key0", C.c_uint32),
        ("key1", C.c_uint32),
        ("row", C.c_int32),
        ("out_snr", C.c_float),
        ("Lframe", C.c_int32),
        ("okey0", C.c_uint32),
        ("okey1", C.c_uint32),
        ("thr", C.c_int32),
        ("ready_min", C.c_int32),
        ("tap_row", C.c_int32),
        ("lframe_magic", C.c_uint32),
    ]


class Tuning(C.Structure):
    """struct vs_tuning (all zero = the library's own choices)"""

    _fields_ = [
        ("kernel", C.c_int32),
        ("ring_slots", C.c_int32),
        ("ready_min", C.c_int32),
        ("ws_pairs", C.c_int32),
        ("gen_low", C.c_int32),
        ("gen_min", C.c_int32),
        ("spin_limit", C.c_int32),
        ("fault", C.c_int32),
        ("ws_filter_prio", C.c_int32),
        ("ws_roles", C.c_int32),
        ("mixed_rings", C.c_int32),
    ]


VS_KERNEL_AUTO = 0
VS_KERNEL_SINGLE = 1
VS_KERNEL_WS = 2
VS_FAULT_WITHHOLD_PROGRESS = 1
VS_FAULT_SHORT_COS_ROWS = 2
VS_FAULT_SHARD_PREPARE = 3
VS_FAULT_SHARD_HANDOVER = 4
VS_FAULT_SIMD_DEALING = 5
VS_FAULT_REROUND = 6
VS_DF_FAST = 0x8


# every symbol include/voice_synth.h declares: (restype, argtypes)
_P = C.POINTER
_vp = C.c_void_p
SYMBOLS = {
    "vs_lane_defaults": (C.c_int, [_P(Lane)]),
    "vs_num_samples": (C.c_int, [C.c_int32, C.c_float, _P(C.c_uint64)]),
    "vs_row_pitch": (C.c_size_t, [C.c_size_t]),
    "vs_vowel_coefficients": (C.c_int, [C.c_int, _P(C.c_double)]),
    "vs_lane_order": (C.c_int, [_P(Lane), _P(C.c_int)]),
    "vs_vowel_name": (C.c_char_p, [C.c_int]),
    "vs_lane_validate": (C.c_int, [_P(Lane)]),
    "vs_strerror": (C.c_char_p, [C.c_int]),
    "vs_flowgen_parse": (C.c_int, [C.c_int, _P(C.c_char_p), _P(FlowgenCmd)]),
    "vs_vowel_parse": (C.c_int, [C.c_int, _P(C.c_char_p), _P(VowelCmd)]),
    "vs_wav_header_write": (C.c_int, [_vp, C.c_int, C.c_int32, C.c_float]),
    "vs_wav_header_read": (
        C.c_int,
        [_vp, C.c_size_t, _P(C.c_int32), _P(C.c_int), _P(C.c_int), _P(C.c_uint64)],
    ),
    "vs_ctx_create": (C.c_int, [C.c_int, _P(_vp)]),
    "vs_ctx_destroy": (None, [_vp]),
    "vs_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "vs_ctx_set_arith": (C.c_int, [_vp, C.c_int]),
    "vs_ctx_last_hip_error": (C.c_int, [_vp]),
    "vs_ctx_set_tuning": (C.c_int, [_vp, _P(Tuning)]),
    "vs_ctx_selftest": (C.c_int, [_vp, _P(C.c_uint64)]),
    "vs_ctx_simd_dealing": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int)]),
    "vs_plan_roles": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "vs_ctx_device_info": (C.c_int, [_vp, C.c_char_p, C.c_size_t, _P(C.c_int)]),
    "vs_ctx_device_pci": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "vs_plan_create": (C.c_int, [_vp, _P(Lane), C.c_size_t, C.c_size_t, _P(_vp)]),
    "vs_plan_destroy": (None, [_vp]),
    "vs_plan_launch": (
        C.c_int,
        [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp],
    ),
    "vs_ctx_synchronize": (C.c_int, [_vp]),
    "vs_ctx_timer_mark": (C.c_int, [_vp, C.c_int]),
    "vs_ctx_timer_elapsed": (C.c_int, [_vp, _P(C.c_double)]),
    "vs_plan_status": (C.c_int, [_vp, _P(C.c_int)]),
    "vs_plan_reseed": (C.c_int, [_vp, _vp, _vp]),
    "vs_plan_info": (C.c_int, [_vp, _P(C.c_size_t), _P(C.c_size_t), _P(C.c_size_t)]),
    "vs_synth": (C.c_int, [_vp, _P(Lane), C.c_size_t, C.c_size_t, _vp]),
    "vs_synth_rows": (C.c_int, [_vp, _P(Lane), C.c_size_t, C.c_size_t, _vp, _vp]),
    "vs_node_create": (C.c_int, [_P(C.c_int), C.c_int, _P(_vp)]),
    "vs_node_destroy": (None, [_vp]),
    "vs_node_shards": (C.c_int, [_vp]),
    "vs_node_ctx": (C.c_int, [_vp, C.c_int, _P(_vp)]),
    "vs_node_set_arith": (C.c_int, [_vp, C.c_int]),
    "vs_node_shard_range": (C.c_int, [_vp, C.c_size_t, C.c_int, _P(C.c_size_t), _P(C.c_size_t)]),
    "vs_node_set_transport": (C.c_int, [_vp, C.c_int]),
    "vs_node_link": (C.c_int, [_vp, C.c_int]),
    "vs_node_last_rccl_error": (C.c_int, [_vp]),
    "vs_node_rccl_ranks": (C.c_int, [_vp, C.c_int]),
    "vs_node_synth_gather": (
        C.c_int,
        [_vp, _P(Lane), C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_int, _P(C.c_double), _P(C.c_double)],
    ),
    "vs_node_synth_rows": (C.c_int, [_vp, _P(Lane), C.c_size_t, C.c_size_t, _vp, _vp]),
    "vs_host_alloc": (C.c_int, [_vp, C.c_size_t, _P(_vp)]),
    "vs_host_free": (C.c_int, [_vp, _vp]),
    "vs_ctx_trim": (C.c_int, [_vp]),
    "vs_plan_timing": (C.c_int, [_vp, _P(C.c_double), _P(C.c_double)]),
    "vs_plan_kernel_name": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_size_t]),
    "vs_source": (
        C.c_int,
        [_vp, _P(Lane), C.c_size_t, C.c_size_t, _vp, _vp, C.c_size_t, _vp],
    ),
    "vs_filter": (C.c_int, [_vp, _P(Lane), C.c_size_t, C.c_size_t, _vp, _vp]),
    "vs_dev_alloc": (C.c_int, [_vp, C.c_size_t, _P(_vp)]),
    "vs_dev_free": (C.c_int, [_vp, _vp]),
    "vs_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "vs_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "vs_version": (C.c_char_p, []),
}

# internal host helpers exported for CPU-side tests of the plan builder (not in the public header)
INTERNAL_SYMBOLS = {
    "vs_expand_lane": (C.c_int, [_P(Lane), C.c_int32, _P(DevLane)]),
    "vs_cos_row": (None, [C.c_int, _P(C.c_double)]),
    "vs_ring_slots_for": (C.c_int, [C.c_int, _P(C.c_int)]),
    "vs_tap_table_build": (C.c_int, [_P(Lane), _P(DevLane), C.c_size_t, C.c_size_t, _P(_P(C.c_double)), _P(C.c_size_t)]),
    "vs_vowel_index": (C.c_int, [C.c_int]),
    "vs_vowel_by_index": (C.c_int, [C.c_int]),
}

_lib = None


def load():
    """Load libvoicesynth.so (built in-tree by `make` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvoicesynth.so is not built (%s). Run `make` or __graft_entry__.build(); "
            "there is no fallback implementation." % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for table in (SYMBOLS, INTERNAL_SYMBOLS):
        for name, (res, args) in table.items():
            if "VS_LIB" in os.environ and not hasattr(lib, name):
                continue             # an A/B build of an earlier round (tools/): it simply lacks the newer entries
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


ROWS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p)


class VsError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        try:
            msg = load().vs_strerror(code).decode()
        except Exception:  # pragma: no cover
            msg = "error"
        super().__init__("%s: %s (%d)" % (where, msg, code))


def check(code, where=""):
    if code != VS_OK:
        raise VsError(code, where)
    return code
