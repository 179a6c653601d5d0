"""ctypes binding of include/doppler_hip.h, doppler_hip_host.h and doppler_hip_debug.h (doppler_amd/lib/libdoppler_hip.so).

Loading never falls back to anything: a missing library or a missing symbol
raises ImportError naming the build command.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdoppler_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")
HEADER_PATH = os.path.join(INCLUDE_DIR, "doppler_hip.h")                # the boundary proper
HEADERS = ("doppler_hip.h", "doppler_hip_host.h", "doppler_hip_debug.h")   # everything the library exports

FMT_I16, FMT_F32 = 0, 1
BUFFER_SIZE = 8192

OK, ERR_ARG, ERR_BLOCK_LEN, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_PLAN = 0, -1, -2, -3, -4, -5, -6


class Segment(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("shift_hz", C.c_float)]


class Stretch(C.Structure):
    _fields_ = [("first", C.c_uint64), ("count", C.c_uint64), ("ratio", C.c_float),
                ("n_start", C.c_uint32), ("period", C.c_uint32), ("lut_len", C.c_uint32)]


class Layout(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("rows_samples", C.c_uint64), ("walk_samples", C.c_uint64),
                ("tile_samples", C.c_uint64), ("single_samples", C.c_uint64), ("table_entries", C.c_uint64),
                ("n_stretches", C.c_uint32), ("rows_launches", C.c_uint32), ("tile_launches", C.c_uint32),
                ("walk_launches", C.c_uint32), ("walk_matrices", C.c_uint32), ("walk_workgroups", C.c_uint32),
                ("leftover_ranges", C.c_uint32), ("leftover_workgroups", C.c_uint32),
                ("f32_i16_by_tiles", C.c_uint32), ("reserved", C.c_uint32)]


class StreamStats(C.Structure):
    _fields_ = [("slabs", C.c_uint64), ("plans_reused", C.c_uint64), ("plan_us", C.c_double), ("upload_us", C.c_double),
                ("enqueue_us", C.c_double), ("total_us", C.c_double)]


class StreamOptions(C.Structure):
    _fields_ = [("path", C.c_uint32), ("in_host_flags", C.c_uint32), ("out_host_flags", C.c_uint32), ("gather", C.c_uint32)]


STREAM_PATHS = {"default": 0, "direct": 1, "staged": 2, "direct_in": 3, "direct_out": 4, "staged_per_slab": 5}
STREAM_COPY_ONLY = 0x100
STREAM_UNPACED = 0x200
STREAM_NO_PROBE = 0x400
STREAM_SHARED_QUEUE = 0x800
STREAM_DESCRIBE_RCCL = 0x1000
STREAM_GATHER = {None: 0, "d2h": 0, "rccl": 1}
STREAM_GATHER_SELF = 0x100


class ResidentCounters(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("blocks", C.c_uint64), ("stops", C.c_uint64), ("idle_exits", C.c_uint64),
                ("running", C.c_uint32), ("tickets_in_flight", C.c_uint32), ("slots_parked", C.c_uint32), ("reserved", C.c_uint32)]


class Options(C.Structure):
    _fields_ = [("rows_mult", C.c_uint32), ("rows_maxl", C.c_uint32), ("rows_r", C.c_uint32), ("rows_compute", C.c_uint32),
                ("walk_waves", C.c_uint32), ("walk_span", C.c_uint32), ("walk_flags", C.c_uint32), ("sub_lg", C.c_uint32),
                ("walk_tilemin", C.c_uint64)]


def make_options(opts):
    """None or a dict of dpx_options fields -> pointer argument (None = defaults)."""
    if not opts:
        return None
    o = Options()
    for k, v in opts.items():
        if k not in dict(Options._fields_):
            raise KeyError("unknown option %r" % k)
        setattr(o, k, int(v))
    return C.byref(o)


def segments_array(segments):
    """(n_samples, shift_hz) pairs -> (dpx_segment array pointer, count, keep-alive).  Through numpy: a replay's 600 segments
    cost 0.4 ms as ctypes structure stores, 0.03 ms this way — it is on the one-shot path bench.py times as plan_ms."""
    import numpy as np
    segs = segments if isinstance(segments, (list, tuple)) else list(segments)
    if len(segs) <= 4:                       # const mode's one segment per slab: plain structure stores are cheaper than numpy
        small = (Segment * max(1, len(segs)))()
        for i, (n, hz) in enumerate(segs):
            small[i].n_samples = int(n)
            small[i].shift_hz = float(hz)
        return small, len(segs), small
    arr = np.zeros(max(1, len(segs)), dtype=np.dtype([("n_samples", "<u8"), ("shift_hz", "<f4"), ("pad", "<u4")]))
    if segs:
        arr["n_samples"] = [s[0] for s in segs]
        arr["shift_hz"] = [s[1] for s in segs]
    return C.cast(arr.ctypes.data, C.POINTER(Segment)), len(segs), arr


def declared_symbols(header=None):
    """Every function the public headers declare (parsed from the header text); header: one of HEADERS, default all."""
    names = set()
    for h in ([header] if header else HEADERS):
        with open(os.path.join(INCLUDE_DIR, h)) as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(dpx_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


_vp, _sz, _u32, _u64, _i, _f = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_float
_P = C.POINTER

_SIGNATURES = {
    "dpx_abi_version": (_i, []),
    "dpx_last_error": (C.c_char_p, []),
    "dpx_device_count": (_i, [_P(_i)]),
    "dpx_ctx_create": (_i, [_i, _P(_vp)]),
    "dpx_ctx_destroy": (None, [_vp]),
    "dpx_convert_iqi16_to_complex": (_i, [_vp, _vp, _sz, _vp, _sz, _P(_sz)]),
    "dpx_convert_iqf32_to_complex": (_i, [_vp, _vp, _sz, _vp, _sz, _P(_sz)]),
    "dpx_shift_frequency": (_i, [_vp, _vp, _sz, _P(_u32), _f, _u32, _vp]),
    "dpx_pack_iqi16": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "dpx_shift_block": (_i, [_vp, _vp, _sz, _i, _vp, _sz, _i, _P(_u32), _f, _u32, _P(_sz)]),
    "dpx_shift_blocks": (_i, [_vp, _vp, _sz, _i, _vp, _sz, _i, _P(_u32), _vp, _sz, _u32, _P(_sz)]),
    "dpx_shift_block_async": (_i, [_vp, _vp, _sz, _i, _i, _P(_u32), _f, _u32, _P(_u32)]),
    "dpx_wait": (_i, [_vp, _u32, _vp, _sz, _P(_sz)]),
    "dpx_ccexpf": (_i, [_vp, _vp, _sz]),
    "dpx_ccexpf_imag": (_i, [_vp, _vp, _sz]),
    "dpx_find_reset": (_i, [_f, _u32, _u32, _u64, _P(_u32), _P(_i)]),
    "dpx_find_reset_scan": (_i, [_f, _u32, _u32, _u64, _P(_u32), _P(_i)]),
    "dpx_samplenum_after": (_i, [_f, _u32, _u32, _u64, _P(_u32)]),
    "dpx_samplenum_after_segments": (_i, [_vp, _sz, _u32, _u32, _P(_u32)]),
    "dpx_plan_describe": (_i, [_P(Segment), _sz, _u32, _u32, _i, _P(Stretch), _sz, _P(_sz), _P(_u32)]),
    "dpx_plan_simulate": (_i, [_P(Segment), _sz, _u32, _u32, _i, _i, _i, _vp, _i, _i, _vp, _vp, _u64]),
    "dpx_plan_layout": (_i, [_P(Segment), _sz, _u32, _u32, _i, _i, _i, _vp, _P(Layout)]),
    "dpx_set_options": (_i, [_vp, _vp]),
    "dpx_track_schedule": (_i, [_vp, _sz, _u32, _u32, C.c_int32, _i, _i, _u64, _vp, _sz, _P(_sz)]),
    "dpx_orbit_observe": (_i, [C.c_char_p, C.c_char_p, C.c_double, C.c_double, C.c_double, C.c_double, _vp]),
    "dpx_orbit_propagate": (_i, [C.c_char_p, C.c_char_p, C.c_double, _vp]),
    "dpx_plan_const": (_i, [_vp, _f, _u32, _u32, _u64, _P(_vp)]),
    "dpx_plan_segments": (_i, [_vp, _P(Segment), _sz, _u32, _u32, _P(_vp)]),
    "dpx_plan_n_samples": (_i, [_vp, _P(_u64)]),
    "dpx_plan_timing": (_i, [_vp, _P(C.c_double)]),
    "dpx_set_resident": (_i, [_vp, _i]),
    "dpx_resident_stats": (_i, [_vp, _P(_u64), _P(_u64)]),
    "dpx_resident_info": (_i, [_vp, _vp]),
    "dpx_plan_final_samplenum": (_i, [_vp, _P(_u32)]),
    "dpx_plan_destroy": (None, [_vp]),
    "dpx_stream_create": (_i, [_vp, _i, _i, _u32, _u32, _sz, _i, _P(_vp)]),
    "dpx_stream_create_multi": (_i, [_P(_vp), _i, _i, _i, _u32, _u32, _sz, _i, _P(_vp)]),
    "dpx_stream_create_opts": (_i, [_P(_vp), _i, _i, _i, _u32, _u32, _sz, _i, _vp, _P(_vp)]),
    "dpx_stream_describe": (_i, [_vp, _P(_u32), _P(_i), _sz, _P(_sz)]),
    "dpx_stream_acquire": (_i, [_vp, _P(_vp), _P(_sz)]),
    "dpx_stream_submit": (_i, [_vp, _sz, _P(Segment), _sz]),
    "dpx_stream_pending": (_i, [_vp, _P(_i)]),
    "dpx_stream_next": (_i, [_vp, _P(_vp), _P(_sz)]),
    "dpx_stream_release": (_i, [_vp]),
    "dpx_stream_samplenum": (_i, [_vp, _P(_u32)]),
    "dpx_stream_get_stats": (_i, [_vp, _vp]),
    "dpx_stream_destroy": (None, [_vp]),
    "dpx_run_device": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "dpx_debug_copy": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "dpx_set_tuning": (_i, [_vp, _i, _i, _i]),
    "dpx_set_libm_contraction": (_i, [_vp, _i]),
    "dpx_set_i16_cast": (_i, [_vp, _i]),
    "dpx_malloc": (_i, [_vp, _sz, _P(_vp)]),
    "dpx_free": (_i, [_vp, _vp]),
    "dpx_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "dpx_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "dpx_synchronize": (_i, [_vp]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "doppler_amd: %s is missing. Build it with `make lib` (hipcc --offload-arch=gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError("doppler_amd: %s does not export %s; rebuild with `make lib`" % (LIB_PATH, name))
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
