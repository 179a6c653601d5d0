"""A second, independently written restatement of the reference's hot path — numpy float32, written from the Rust text
alone (reference src/dsp.rs:85-134, src/main.rs:62-99, src/main.rs:156-184), sharing no code with oracle/.

Why: the C oracle and the golden vectors have one author and one implementation language; a slip in the reading of the
unpack, the complex multiply, the counter rule or the pack would be reproduced by both.  This file reads the same lines
again in another language, and tests/test_restatement.py compares the two on every golden case and on clipping inputs.
The only thing borrowed is libm: `ccexpf` is the reference's own src/complex.c, compiled in place into
oracle/_ref/libcomplex.so (or, on a box without /root/reference, libm's cexpf through ctypes — the same function
complex.c:35 calls).

Every arithmetic step is a numpy float32 operation (one IEEE rounding each, never fused), in the association the Rust
expressions have.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "oracle", "_ref", "libcomplex.so")


class _RustComplex(C.Structure):                      # complex.c:28-31
    _fields_ = [("real", C.c_float), ("imag", C.c_float)]


def _load_ccexpf():
    if os.path.exists(_REF):
        lib = C.CDLL(_REF)
        lib.ccexpf.argtypes = [C.POINTER(_RustComplex)]
        lib.ccexpf.restype = None
        return lib.ccexpf, "reference src/complex.c"
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    libm.sincosf.restype = None

    def via_libm(z):          # cexpf(0 + i y) = (cosf y, sinf y) for a zero real part (glibc s_cexp_template.c)
        s, c = C.c_float(), C.c_float()
        libm.sincosf(z.contents.imag, C.byref(s), C.byref(c))
        z.contents.real, z.contents.imag = c.value, s.value
    return via_libm, "libm sincosf"


_ccexpf, CCEXPF_SOURCE = _load_ccexpf()


def convert_iqi16_to_complex(inbuf):
    """dsp.rs:85-99: ((b[1] as i16) << 8 | b[0] as i16) as f32 / 32768."""
    b = np.frombuffer(bytes(inbuf), dtype=np.uint8)
    assert b.size % 4 == 0                                            # dsp.rs:87
    b = b.astype(np.int32).reshape(-1, 4)
    def le16(lo, hi):
        w = ((hi << 8) | lo) & 0xffff                                 # i16 arithmetic: bits past 15 fall off the shift
        return np.where(w >= 0x8000, w - 0x10000, w)                  # ... and the result is read as i16
    i = le16(b[:, 0], b[:, 1]).astype(F32) / F32(32768.0)
    q = le16(b[:, 2], b[:, 3]).astype(F32) / F32(32768.0)
    return i, q


def convert_iqf32_to_complex(inbuf):
    """dsp.rs:101-115: transmute of the little-endian u32."""
    b = np.frombuffer(bytes(inbuf), dtype=np.uint8)
    assert b.size % 8 == 0                                            # dsp.rs:103
    w = b.astype(np.uint32).reshape(-1, 8)
    i = ((w[:, 3] << 24) | (w[:, 2] << 16) | (w[:, 1] << 8) | w[:, 0]).astype(np.uint32).view(F32)
    q = ((w[:, 7] << 24) | (w[:, 6] << 16) | (w[:, 5] << 8) | w[:, 4]).astype(np.uint32).view(F32)
    return i, q


def _counters(n_samples, samplenum, shift_hz, samplerate):
    """The counter each sample uses, and the counter afterwards — dsp.rs:125-130, one sample after the other:
    if (shift_hz / samplerate as f32 * *samplenum as f32).fract() == 0.0 { 1 } else { += 1 }."""
    ratio = F32(shift_hz) / F32(samplerate)                           # `samplerate as f32`, then one f32 division
    used = np.empty(n_samples, dtype=np.uint32)
    sn = int(samplenum)
    with np.errstate(invalid="ignore", over="ignore"):
        for k in range(n_samples):
            used[k] = sn
            p = ratio * F32(np.uint32(sn))                            # `*samplenum as f32` rounds to nearest even
            fract = p - np.trunc(p)                                   # f32::fract: self - self.trunc(); inf -> NaN
            sn = 1 if fract == F32(0.0) else (sn + 1) & 0xffffffff    # u32 `+= 1` (release build: wraps)
    return used, sn


def shift_frequency(i, q, samplenum, shift_hz, samplerate):
    """dsp.rs:117-134."""
    used, sn = _counters(i.size, samplenum, shift_hz, samplerate)
    ratio = F32(shift_hz) / F32(samplerate)
    with np.errstate(invalid="ignore", over="ignore"):
        p = ratio * used.astype(F32)                                  # shift_hz / samplerate as f32 * (*samplenum) as f32
        theta = (F32(-2.0) * F32(np.pi)) * p                          # -2. * PI * (...): left to right
    c = np.empty(i.size, dtype=F32)
    s = np.empty(i.size, dtype=F32)
    z = _RustComplex()
    for k in range(i.size):                                           # Complex::new(0.0, theta); ccexpf(&mut corrector)
        z.real, z.imag = 0.0, float(theta[k])
        _ccexpf(C.pointer(z))
        c[k], s[k] = z.real, z.imag
    with np.errstate(invalid="ignore", over="ignore"):
        # num-complex 0.1.35 Mul: Complex::new(self.re*other.re - self.im*other.im, self.re*other.im + self.im*other.re)
        re = i * c - q * s
        im = i * s + q * c
    return re, im, sn


def _as_i16(x, legacy):
    """`as i16` of main.rs:77-78.  Rust >= 1.45: truncate toward zero, saturate, NaN -> 0.  legacy: the x86-64 code of a
    2016 rustc — CVTTSS2SI into 32 bits (NaN and |x| >= 2^31: 0x80000000), low 16 bits kept."""
    x = np.asarray(x, dtype=F32)
    with np.errstate(invalid="ignore"):
        t = np.trunc(x.astype(np.float64))
    if legacy:
        bad = np.isnan(x) | (t >= 2.0 ** 31) | (t < -(2.0 ** 31))
        w = np.where(bad, -(2 ** 31), np.where(bad, 0, t)).astype(np.int64)
        return (w & 0xffff).astype(np.uint16).view(np.int16)
    t = np.where(np.isnan(x), 0.0, np.clip(t, -32768.0, 32767.0))
    return t.astype(np.int16)


def pack_i16(re, im, legacy=False):
    """main.rs:72-87: i = (sample.re * 32767.0) as i16; bytes i & 0xFF, (i >> 8) & 0xFF, then q."""
    with np.errstate(invalid="ignore", over="ignore"):
        i = _as_i16(re * F32(32767.0), legacy).astype(np.int32)
        q = _as_i16(im * F32(32767.0), legacy).astype(np.int32)
    out = np.empty((re.size, 4), dtype=np.uint8)
    out[:, 0] = i & 0xFF
    out[:, 1] = (i >> 8) & 0xFF
    out[:, 2] = q & 0xFF
    out[:, 3] = (q >> 8) & 0xFF
    return out.reshape(-1)


def pack_f32(re, im):
    """main.rs:89-93: the Complex<f32> array as bytes."""
    out = np.empty((re.size, 2), dtype=F32)
    out[:, 0], out[:, 1] = re, im
    return out.view(np.uint8).reshape(-1)


def shift_block(inbytes, intype, outtype, samplenum, shift_hz, samplerate, legacy_cast=False):
    """One call of the `shift` closure (main.rs:62-99) on the bytes it would have read.  Returns (bytes, samplenum)."""
    i, q = convert_iqi16_to_complex(inbytes) if intype == "i16" else convert_iqf32_to_complex(inbytes)
    re, im, sn = shift_frequency(i, q, samplenum, shift_hz, samplerate)
    return (pack_i16(re, im, legacy_cast) if outtype == "i16" else pack_f32(re, im)), sn


def const_stream(inbytes, intype, outtype, shift, samplerate, samplenum=0, legacy_cast=False):
    """`doppler const` (main.rs:102-119): 8192-byte blocks until a short (or empty) read; shift = args.shift as f32."""
    b = bytes(inbytes)
    out, sn, pos = [], samplenum, 0
    while True:
        blk = b[pos:pos + 8192]
        o, sn = shift_block(blk, intype, outtype, sn, F32(shift), samplerate, legacy_cast)
        out.append(o)
        pos += 8192
        if len(blk) != 8192:                                          # main.rs:98: invec.len() != BUFFER_SIZE
            break
    return np.concatenate(out), sn


def track_stream(inbytes, intype, outtype, samplerate, frequency_hz, range_rate_km_s, offset_hz=None, legacy_cast=False):
    """`doppler track --time` replay, main.rs:156-184, with the range rate per whole second of stream as a table
    (entry t = predict.sat.range_rate_km_sec after predict.update(start_time + t s); the last entry is held) — the orbit
    library is not part of the reference tree.  Line by line:
        predict.update(Some(start_time + dt));                       dt is still the PREVIOUS pass's value (one-block lag)
        doppler_hz = (range_rate * 1000_f64 / C) * frequency as f64 * (-1.0);
        dt = Duration::seconds((sample_count as f32 / samplerate as f32) as i64);
        shift(intype, doppler_hz as f32 + offset.unwrap_or(0) as f32, samplerate);
        sample_count += count;
    Returns (bytes, samplenum, the shift of every pass)."""
    b = bytes(inbytes)
    out, sn, pos, log = [], 0, 0, []
    sample_count = 0
    dt = 0                                                            # whole seconds
    while True:
        rr = float(range_rate_km_s[min(dt, len(range_rate_km_s) - 1)])
        doppler_hz = (rr * 1000.0 / 299792458.0) * float(frequency_hz) * (-1.0)           # f64 throughout (main.rs:163)
        dt = int(F32(sample_count) / F32(samplerate))                 # `as i64` truncates toward zero
        shift = F32(doppler_hz) + F32(0 if offset_hz is None else offset_hz)              # f64 -> f32, then an f32 addition
        blk = b[pos:pos + 8192]
        o, sn = shift_block(blk, intype, outtype, sn, shift, samplerate, legacy_cast)
        out.append(o)
        log.append(shift)
        pos += 8192
        if len(blk) != 8192:
            break
        sample_count += len(o) // (4 if outtype == "i16" else 8)      # count = output.len()
    return np.concatenate(out), sn, np.array(log, dtype=F32)
