"""Host-side mirror of the reference's operator module `doppler::dsp` (reference src/dsp.rs).

Same names, argument meaning and failure behaviour as the Rust functions; every call goes
through the C ABI (include/doppler_hip.h) to the HIP kernels — nothing is computed here.

    convert_iqi16_to_complex(inbuf)                       src/dsp.rs:85-99
    convert_iqf32_to_complex(inbuf)                       src/dsp.rs:101-115
    shift_frequency(inbuf, samplenum, shift_hz, samplerate)   src/dsp.rs:117-134
    shift_block(...)                                      body of the `shift` closure, src/main.rs:62-99
    ccexpf(z)                                             src/complex.c:33-39

Rust's `&mut u32` samplenum becomes an extra return value.
Where the reference panics on a ragged byte length (`assert!`, dsp.rs:87/103) these raise
DspError (an AssertionError) with code DPX_ERR_BLOCK_LEN.
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import BYTES_PER_SAMPLE, DspError, as_bytes, check, complex32, default_context, fmt_code

__all__ = ["convert_iqi16_to_complex", "convert_iqf32_to_complex", "shift_frequency", "shift_block",
           "pack_iqi16", "ccexpf", "complex32", "DspError", "shift_block_async", "wait"]


def _ctx(ctx):
    return ctx if ctx is not None else default_context()


def convert_iqi16_to_complex(inbuf, ctx=None):
    """LE i16 IQ bytes -> Complex<f32> array (each component / 32768)."""
    ctx = _ctx(ctx)
    b = as_bytes(inbuf)
    out = np.empty(b.size // 4 + 1, dtype=complex32)
    n = C.c_size_t()
    check(ctx._lib.dpx_convert_iqi16_to_complex(ctx.handle, b.ctypes.data, b.size, out.ctypes.data, out.size,
                                                C.byref(n)))
    return out[: n.value]


def convert_iqf32_to_complex(inbuf, ctx=None):
    """LE f32 IQ bytes -> Complex<f32> array (bit-for-bit)."""
    ctx = _ctx(ctx)
    b = as_bytes(inbuf)
    out = np.empty(b.size // 8 + 1, dtype=complex32)
    n = C.c_size_t()
    check(ctx._lib.dpx_convert_iqf32_to_complex(ctx.handle, b.ctypes.data, b.size, out.ctypes.data, out.size,
                                                C.byref(n)))
    return out[: n.value]


def shift_frequency(inbuf, samplenum, shift_hz, samplerate, ctx=None):
    """Complex<f32> array in, frequency-shifted Complex<f32> array out. Returns (output, samplenum)."""
    ctx = _ctx(ctx)
    a = np.ascontiguousarray(inbuf, dtype=complex32)
    out = np.empty(a.size, dtype=complex32)
    sn = C.c_uint32(samplenum)
    check(ctx._lib.dpx_shift_frequency(ctx.handle, a.ctypes.data, a.size, C.byref(sn), float(shift_hz),
                                       int(samplerate), out.ctypes.data))
    return out, sn.value


def pack_iqi16(samples, ctx=None):
    """Complex<f32> array -> LE i16 IQ bytes, (x * 32767.0) as i16 (src/main.rs:72-87)."""
    ctx = _ctx(ctx)
    a = np.ascontiguousarray(samples, dtype=complex32)
    out = np.empty(a.size * 4, dtype=np.uint8)
    check(ctx._lib.dpx_pack_iqi16(ctx.handle, a.ctypes.data, a.size, out.ctypes.data, out.size))
    return out


def shift_block(inbytes, intype, outtype, samplenum, shift_hz, samplerate, ctx=None):
    """unpack -> shift -> pack in one fused kernel. Returns (out_bytes, n_samples, samplenum)."""
    ctx = _ctx(ctx)
    b = as_bytes(inbytes)
    it, ot = fmt_code(intype), fmt_code(outtype)
    out = np.empty(b.size // BYTES_PER_SAMPLE[it] * BYTES_PER_SAMPLE[ot] + 8, dtype=np.uint8)
    sn = C.c_uint32(samplenum)
    n = C.c_size_t()
    check(ctx._lib.dpx_shift_block(ctx.handle, b.ctypes.data, b.size, it, out.ctypes.data, out.size, ot,
                                   C.byref(sn), float(shift_hz), int(samplerate), C.byref(n)))
    return out[: n.value * BYTES_PER_SAMPLE[ot]], n.value, sn.value


def shift_block_async(inbytes, intype, outtype, samplenum, shift_hz, samplerate, ctx=None):
    """dpx_shift_block_async: enqueue one block, return (ticket, samplenum after the block) at once."""
    ctx = _ctx(ctx)
    b = as_bytes(inbytes)
    sn = C.c_uint32(samplenum)
    t = C.c_uint32()
    check(ctx._lib.dpx_shift_block_async(ctx.handle, b.ctypes.data, b.size, fmt_code(intype), fmt_code(outtype), C.byref(sn),
                                         float(shift_hz), int(samplerate), C.byref(t)))
    return t.value, sn.value


def wait(ticket, outtype, max_samples=8192, ctx=None):
    """dpx_wait: the output bytes of the block behind `ticket`."""
    ctx = _ctx(ctx)
    out = np.empty(max_samples * BYTES_PER_SAMPLE[fmt_code(outtype)] + 8, dtype=np.uint8)
    n = C.c_size_t()
    check(ctx._lib.dpx_wait(ctx.handle, int(ticket), out.ctypes.data, out.size, C.byref(n)))
    return out[: n.value * BYTES_PER_SAMPLE[fmt_code(outtype)]]


def shift_blocks(inbytes, intype, outtype, samplenum, shift_hz_per_block, samplerate, ctx=None):
    """Many 8192-byte blocks in one call, one shift per block (dpx_shift_blocks). Returns (out_bytes, n_samples, samplenum)."""
    ctx = _ctx(ctx)
    b = as_bytes(inbytes)
    it, ot = fmt_code(intype), fmt_code(outtype)
    hz = np.ascontiguousarray(shift_hz_per_block, dtype=np.float32)
    out = np.empty(b.size // BYTES_PER_SAMPLE[it] * BYTES_PER_SAMPLE[ot] + 8, dtype=np.uint8)
    sn = C.c_uint32(samplenum)
    n = C.c_size_t()
    check(ctx._lib.dpx_shift_blocks(ctx.handle, b.ctypes.data, b.size, it, out.ctypes.data, out.size, ot, C.byref(sn),
                                    hz.ctypes.data, hz.size, int(samplerate), C.byref(n)))
    return out[: n.value * BYTES_PER_SAMPLE[ot]], n.value, sn.value


def ccexpf(z, ctx=None):
    """src/complex.c:33-39: cexpf(z.re + i*z.im) for each element (returned; the C function works in place)."""
    ctx = _ctx(ctx)
    a = np.array(z, dtype=complex32, copy=True).reshape(-1)
    check(ctx._lib.dpx_ccexpf(ctx.handle, a.ctypes.data, a.size))
    return a
