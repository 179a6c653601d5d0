"""TEST INFRASTRUCTURE — ctypes/numpy front-end of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product package (doppler_amd/) never does.

Function names follow the reference (/root/reference/src/dsp.rs:85,101,117 and
the `shift` closure of /root/reference/src/main.rs:62-99).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libcomplex.so")

I16, F32 = 0, 1
BUFFER_SIZE = 8192  # main.rs:49
_FMT = {"i16": I16, "f32": F32, I16: I16, F32: F32}
_BPS = {I16: 4, F32: 8}

complex32 = np.dtype([("re", "<f4"), ("im", "<f4")])


def build():
    """(Re)build liboracle.so, check_sincosf and, when the reference is mounted, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    u8p, f32p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    lib.orc_convert_iqi16_to_complex.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_convert_iqi16_to_complex.restype = C.c_long
    lib.orc_convert_iqf32_to_complex.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_convert_iqf32_to_complex.restype = C.c_long
    lib.orc_shift_frequency.argtypes = [C.c_void_p, C.c_size_t, u32p, C.c_float, C.c_uint32, C.c_void_p]
    lib.orc_shift_frequency.restype = None
    lib.orc_advance_samplenum.argtypes = [u32p, C.c_float, C.c_uint32, C.c_uint64]
    lib.orc_advance_samplenum.restype = None
    lib.orc_pack_i16.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_pack_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_shift_block.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, u32p, C.c_float,
                                    C.c_uint32, C.c_void_p, C.POINTER(C.c_size_t)]
    lib.orc_shift_block.restype = C.c_int
    lib.orc_const_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int32, C.c_uint32,
                                     u32p, C.c_void_p]
    lib.orc_const_stream.restype = C.c_long
    lib.orc_const_stream_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int32, C.c_uint32,
                                        C.c_void_p, C.c_int]
    lib.orc_const_stream_mt.restype = C.c_long
    lib.orc_segments_stream_mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, u32p,
                                           C.c_void_p, C.c_int]
    lib.orc_segments_stream_mt.restype = C.c_long
    lib.orc_track_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                     C.c_int32, C.c_int, C.c_void_p, C.c_size_t, u32p, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_size_t)]
    lib.orc_track_stream.restype = C.c_long
    lib.orc_ccexpf.argtypes = [C.c_void_p]
    lib.orc_set_ccexpf.argtypes = [C.c_void_p]
    lib.orc_set_corrector_mode.argtypes = [C.c_int]
    lib.orc_get_corrector_mode.restype = C.c_int
    lib.orc_ccexpf_imag_array.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    lib.orc_ccexpf_imag_array.restype = None
    lib.orc_ccexpf_array.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    lib.orc_ccexpf_array.restype = None
    lib.orc_expf_glibc235.argtypes = [C.c_float, C.c_int]
    lib.orc_expf_glibc235.restype = C.c_float
    lib.orc_sincosf_glibc235.argtypes = [C.c_float, f32p, f32p, C.c_int]
    lib.orc_cexpf_imag_glibc235.argtypes = [C.c_float, f32p, f32p, C.c_int]
    lib.orc_detect_libm_variant.restype = C.c_int
    return lib


lib = _load()
_ref = None
if os.path.exists(_REF_PATH):
    # the reference's own complex.c, compiled in place by oracle/Makefile
    _ref = C.CDLL(_REF_PATH)
    lib.orc_set_ccexpf(C.cast(_ref.ccexpf, C.c_void_p))


def have_ref():
    return _ref is not None


class OracleError(AssertionError):
    """Raised where the reference would panic (dsp.rs:87, dsp.rs:103)."""


def set_corrector_mode(mode):
    """0 = libm cexpf / reference complex.c (default); 1 = restated glibc sincosf (FMA build);
    2 = restated glibc sincosf (non-FMA build)."""
    lib.orc_set_corrector_mode(int(mode))


def set_i16_cast(mode):
    """`(x * 32767.0) as i16` of main.rs:77-78: 0 = saturating with NaN -> 0 (Rust >= 1.45, default);
    1 = truncate, keep the low 16 bits (CVTTSS2SI, what a 2016 rustc emitted on x86-64)."""
    lib.orc_set_i16_cast(int(mode))


def libm_variant():
    return int(lib.orc_detect_libm_variant())


def _bytes(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(-1)


def ccexpf(re, im):
    """complex.c:33-39 on one value (through oracle/_ref when present)."""
    z = np.array([(re, im)], dtype=complex32)
    fn = _ref.ccexpf if _ref is not None else lib.orc_ccexpf
    fn(C.c_void_p(z.ctypes.data))
    return np.float32(z["re"][0]), np.float32(z["im"][0])


def ccexpf_imag_array(theta, mode=0):
    """ccexpf(0 + i*theta) per element. mode 0: libm / reference complex.c; 1: restated glibc (FMA build);
    2: restated glibc (SSE2 build)."""
    t = np.ascontiguousarray(theta, dtype=np.float32)
    out = np.empty(t.size, dtype=complex32)
    lib.orc_ccexpf_imag_array(t.ctypes.data, t.size, out.ctypes.data, int(mode))
    return out


def ccexpf_array(z, mode=0):
    """General ccexpf (complex.c:33-39) per element. mode 0: libm / reference complex.c; 1: restated glibc
    cexpf with the FMA builds of expf/sincosf; 2: with the SSE2 builds."""
    a = np.ascontiguousarray(z, dtype=complex32).reshape(-1)
    out = np.empty(a.size, dtype=complex32)
    lib.orc_ccexpf_array(a.ctypes.data, a.size, out.ctypes.data, int(mode))
    return out


def convert_iqi16_to_complex(inbuf):
    b = _bytes(inbuf)
    out = np.empty(b.size // 4 + 1, dtype=complex32)
    n = lib.orc_convert_iqi16_to_complex(b.ctypes.data, b.size, out.ctypes.data)
    if n < 0:
        raise OracleError("assertion failed: inbuf.len() % 4 == 0")
    return out[:n]


def convert_iqf32_to_complex(inbuf):
    b = _bytes(inbuf)
    out = np.empty(b.size // 8 + 1, dtype=complex32)
    n = lib.orc_convert_iqf32_to_complex(b.ctypes.data, b.size, out.ctypes.data)
    if n < 0:
        raise OracleError("assertion failed: inbuf.len() % 8 == 0")
    return out[:n]


def shift_frequency(inbuf, samplenum, shift_hz, samplerate):
    """dsp.rs:117-134. Returns (output, new_samplenum)."""
    a = np.ascontiguousarray(inbuf, dtype=complex32)
    out = np.empty(a.size, dtype=complex32)
    sn = C.c_uint32(samplenum)
    lib.orc_shift_frequency(a.ctypes.data, a.size, C.byref(sn), C.c_float(shift_hz), samplerate,
                            out.ctypes.data)
    return out, sn.value


def advance_samplenum(samplenum, shift_hz, samplerate, n):
    sn = C.c_uint32(samplenum)
    lib.orc_advance_samplenum(C.byref(sn), C.c_float(shift_hz), samplerate, n)
    return sn.value


def pack_i16(samples):
    a = np.ascontiguousarray(samples, dtype=complex32)
    out = np.empty(a.size * 4, dtype=np.uint8)
    lib.orc_pack_i16(a.ctypes.data, a.size, out.ctypes.data)
    return out


def pack_f32(samples):
    a = np.ascontiguousarray(samples, dtype=complex32)
    out = np.empty(a.size * 8, dtype=np.uint8)
    lib.orc_pack_f32(a.ctypes.data, a.size, out.ctypes.data)
    return out


def shift_block(inbytes, intype, outtype, samplenum, shift_hz, samplerate):
    """One call of the `shift` closure (main.rs:62-99). Returns (out_bytes, stop, count, samplenum)."""
    b = _bytes(inbytes)
    it, ot = _FMT[intype], _FMT[outtype]
    out = np.empty(b.size // _BPS[it] * _BPS[ot] + 8, dtype=np.uint8)
    sn = C.c_uint32(samplenum)
    cnt = C.c_size_t(0)
    r = lib.orc_shift_block(b.ctypes.data, b.size, it, ot, C.byref(sn), C.c_float(shift_hz), samplerate,
                            out.ctypes.data, C.byref(cnt))
    if r < 0:
        raise OracleError("reference would panic (block length / arguments), code %d" % r)
    return out[: cnt.value * _BPS[ot]], bool(r), cnt.value, sn.value


def const_stream(inbytes, intype, outtype, shift, samplerate, samplenum=0, threads=1, out=None):
    """`doppler const` (main.rs:102-119) over an in-memory stream. Returns (out_bytes, samplenum).
    `out`: optional preallocated (and pre-touched) uint8 buffer, for timing runs."""
    b = _bytes(inbytes)
    it, ot = _FMT[intype], _FMT[outtype]
    need = b.size // _BPS[it] * _BPS[ot] + 8
    if out is None:
        out = np.empty(need, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size >= need and out.flags["C_CONTIGUOUS"]
    if threads > 1:
        assert samplenum == 0
        r = lib.orc_const_stream_mt(b.ctypes.data, b.size, it, ot, shift, samplerate, out.ctypes.data, threads)
        sn_val = None
    else:
        sn = C.c_uint32(samplenum)
        r = lib.orc_const_stream(b.ctypes.data, b.size, it, ot, shift, samplerate, C.byref(sn), out.ctypes.data)
        sn_val = sn.value
    if r < 0:
        raise OracleError("reference would panic (trailing partial sample), code %d" % r)
    return out[:r], sn_val


class _Segment(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("shift_hz", C.c_float)]


def segments_stream(inbytes, intype, outtype, segments, samplerate, samplenum=0, threads=1, out=None):
    """A stream of (n_samples, shift_hz) runs, counter carried in from `samplenum` and on through every run with the
    sequential rule of dsp.rs:125-130 (also to find each thread's starting counter).  Returns (out_bytes, samplenum)."""
    b = _bytes(inbytes)
    it, ot = _FMT[intype], _FMT[outtype]
    segs = [(int(n), float(hz)) for n, hz in segments if n]
    total = sum(n for n, _ in segs)
    assert b.size == total * _BPS[it], (b.size, total)
    arr = (_Segment * max(1, len(segs)))()
    for i, (n, hz) in enumerate(segs):
        arr[i].n_samples, arr[i].shift_hz = n, hz
    if out is None:
        out = np.empty(total * _BPS[ot] + 8, dtype=np.uint8)
    sn = C.c_uint32(samplenum)
    r = lib.orc_segments_stream_mt(b.ctypes.data, it, ot, samplerate, arr, len(segs), C.byref(sn), out.ctypes.data,
                                   int(threads))
    if r < 0:
        raise OracleError("bad arguments, code %d" % r)
    return out[:r], sn.value


def track_stream(inbytes, intype, outtype, samplerate, frequency_hz, range_rate_km_s, offset_hz=None,
                 samplenum=0):
    """`doppler track --time` replay (main.rs:156-184) with range rate supplied per whole second.
    Returns (out_bytes, samplenum, per_block_shift_hz)."""
    b = _bytes(inbytes)
    it, ot = _FMT[intype], _FMT[outtype]
    rr = np.ascontiguousarray(range_rate_km_s, dtype=np.float64)
    out = np.empty(b.size // _BPS[it] * _BPS[ot] + 8, dtype=np.uint8)
    log = np.empty(b.size // BUFFER_SIZE + 2, dtype=np.float32)
    nb = C.c_size_t(0)
    sn = C.c_uint32(samplenum)
    r = lib.orc_track_stream(b.ctypes.data, b.size, it, ot, samplerate, frequency_hz,
                             0 if offset_hz is None else int(offset_hz), 0 if offset_hz is None else 1,
                             rr.ctypes.data, rr.size, C.byref(sn), out.ctypes.data, log.ctypes.data,
                             C.byref(nb))
    if r < 0:
        raise OracleError("reference would panic, code %d" % r)
    return out[:r], sn.value, log[: nb.value].copy()


def sincosf_glibc235(y, fma=True):
    s, c = C.c_float(), C.c_float()
    lib.orc_sincosf_glibc235(C.c_float(y), C.byref(s), C.byref(c), 1 if fma else 0)
    return np.float32(s.value), np.float32(c.value)
