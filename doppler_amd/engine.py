"""Context / Plan objects over the C ABI (include/doppler_hip.h)."""
import ctypes as C

import numpy as np

from . import _lib

complex32 = np.dtype([("re", "<f4"), ("im", "<f4")])
_FMT = {"i16": _lib.FMT_I16, "f32": _lib.FMT_F32, _lib.FMT_I16: _lib.FMT_I16, _lib.FMT_F32: _lib.FMT_F32}
BYTES_PER_SAMPLE = {_lib.FMT_I16: 4, _lib.FMT_F32: 8}


class DspError(AssertionError):
    """The C ABI returned an error. `code` is the dpx_status value; DPX_ERR_BLOCK_LEN is what
    the reference reports as an `assert!` panic (src/dsp.rs:87, src/dsp.rs:103)."""

    def __init__(self, code, message):
        super().__init__("dpx error %d: %s" % (code, message))
        self.code = code


def _lib_handle():
    from . import lib
    return lib


def check(rc):
    if rc != _lib.OK:
        raise DspError(rc, _lib_handle().dpx_last_error().decode("utf-8", "replace"))


def fmt_code(fmt):
    try:
        return _FMT[fmt]
    except KeyError:
        raise DspError(_lib.ERR_ARG, "unknown IQ format %r (use 'i16' or 'f32')" % (fmt,))


def as_bytes(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(-1)


class Context:
    """One GPU. Creation fails (DspError, DPX_ERR_NO_DEVICE) when no gfx950 device is usable."""

    def __init__(self, device=0):
        self._lib = _lib_handle()
        self._h = C.c_void_p()
        check(self._lib.dpx_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.dpx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_tuning(self, block=0, vecs=0, variant=3):
        """block: lanes per workgroup (128/256); vecs: 4-sample groups per lane (1/2); 0 keeps the current value,
        block = vecs = -1 returns to the geometry chosen per launch (the default until one is named);
        variant: 3 auto, 1 sincos per sample, 2 tabulated correctors. Applies to plans created afterwards."""
        check(self._lib.dpx_set_tuning(self._h, block, vecs, variant))

    def set_options(self, **opts):
        """Kernel-shape knobs (include/doppler_hip_debug.h, dpx_options: rows_mult, rows_maxl, rows_r, rows_compute, walk_waves,
        walk_span, walk_flags, walk_tilemin); no arguments restores the defaults. Applies to plans created afterwards."""
        check(self._lib.dpx_set_options(self._h, _lib.make_options(opts)))

    def set_resident(self, on=True):
        """The resident block kernel behind shift_block / shift_block_async (default on); off: a launch per block.
        While it is resident (until 2 ms after the last block) launches of OTHER software on this device — torch in the same
        process — may queue behind it: turn it off where the device is shared with such work."""
        check(self._lib.dpx_set_resident(self._h, 1 if on else 0))

    def resident_stats(self):
        """(kernel launches, blocks served through doorbells) of the resident block kernel so far."""
        a, b = C.c_uint64(), C.c_uint64()
        check(self._lib.dpx_resident_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def resident_info(self):
        """dict of dpx_resident_counters: launches, blocks, stops, idle_exits, running, tickets_in_flight, slots_parked;
        launches == stops + idle_exits + running whenever no call is in progress."""
        st = _lib.ResidentCounters()
        check(self._lib.dpx_resident_info(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _lib.ResidentCounters._fields_ if k != "reserved"}

    def set_libm_contraction(self, fma=True):
        check(self._lib.dpx_set_libm_contraction(self._h, 1 if fma else 0))

    def set_i16_cast(self, legacy_x86=False):
        """Meaning of `(x * 32767.0) as i16` (main.rs:77-78) outside the i16 range: saturate (Rust >= 1.45, default) or
        truncate-and-wrap (the x86-64 code of a 2016 rustc). Applies to plans created afterwards."""
        check(self._lib.dpx_set_i16_cast(self._h, 1 if legacy_x86 else 0))

    # ---- device memory helpers (for callers without torch)
    def malloc(self, nbytes):
        p = C.c_void_p()
        check(self._lib.dpx_malloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        check(self._lib.dpx_free(self._h, C.c_void_p(ptr)))

    def h2d(self, dptr, host):
        b = as_bytes(host)
        check(self._lib.dpx_memcpy_h2d(self._h, C.c_void_p(dptr), b.ctypes.data, b.size))

    def d2h(self, host, dptr):
        assert host.flags["C_CONTIGUOUS"]
        check(self._lib.dpx_memcpy_d2h(self._h, host.ctypes.data, C.c_void_p(dptr), host.nbytes))

    def synchronize(self):
        check(self._lib.dpx_synchronize(self._h))

    def debug_copy(self, d_in, d_out, nbytes, stream=0):
        """Calibration: one-shot 16-byte non-temporal copy with the fused kernel's access pattern."""
        check(self._lib.dpx_debug_copy(self._h, C.c_void_p(d_in), C.c_void_p(d_out), nbytes, C.c_void_p(stream)))

    # ---- plans
    def plan_const(self, shift_hz, samplerate, n_samples, samplenum=0):
        return Plan(self, [(int(n_samples), float(shift_hz))], samplerate, samplenum)

    def plan_segments(self, segments, samplerate, samplenum=0):
        """segments: iterable of (n_samples, shift_hz)."""
        return Plan(self, list(segments), samplerate, samplenum)


class Plan:
    """Closed-form counter description of a stream, resident on the GPU; run() launches the fused kernel."""

    def __init__(self, ctx, segments, samplerate, samplenum=0):
        self._lib = _lib_handle()
        self.ctx = ctx
        arr, n_segs, _keep = _lib.segments_array(segments)
        self._h = C.c_void_p()
        check(self._lib.dpx_plan_segments(ctx.handle, arr, n_segs, int(samplerate), int(samplenum),
                                          C.byref(self._h)))
        n = C.c_uint64()
        check(self._lib.dpx_plan_n_samples(self._h, C.byref(n)))
        self.n_samples = n.value
        sn = C.c_uint32()
        check(self._lib.dpx_plan_final_samplenum(self._h, C.byref(sn)))
        self.final_samplenum = sn.value

    def timing(self):
        """Microseconds dpx_plan_segments spent on: the stretch list, the launch layout, the device image (upload + wait)."""
        t = (C.c_double * 3)()
        check(self._lib.dpx_plan_timing(self._h, t))
        return {"stretch_list_us": round(t[0], 1), "layout_us": round(t[1], 1), "device_image_us": round(t[2], 1)}

    def run(self, d_in, in_fmt, d_out, out_fmt, stream=0):
        """Asynchronous launch on `stream` (a hipStream_t value; 0 = default stream)."""
        check(self._lib.dpx_run_device(self._h, C.c_void_p(d_in), fmt_code(in_fmt), C.c_void_p(d_out),
                                       fmt_code(out_fmt), C.c_void_p(stream)))

    def run_tensors(self, x, out, in_fmt, out_fmt):
        """Convenience for torch tensors on this context's GPU: launches on torch's current stream."""
        import torch
        assert x.is_cuda and out.is_cuda and x.is_contiguous() and out.is_contiguous()
        need_in = self.n_samples * BYTES_PER_SAMPLE[fmt_code(in_fmt)]
        need_out = self.n_samples * BYTES_PER_SAMPLE[fmt_code(out_fmt)]
        if x.numel() * x.element_size() < need_in or out.numel() * out.element_size() < need_out:
            raise DspError(_lib.ERR_CAPACITY, "tensor smaller than the plan's %d samples" % self.n_samples)
        self.run(x.data_ptr(), in_fmt, out.data_ptr(), out_fmt, torch.cuda.current_stream(x.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.dpx_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """Ring of pinned host slabs (dpx_stream_*): fill a slab, submit it with its constant-shift segments, collect
    the outputs in order.  The sample counter is carried from slab to slab like `samplenr` (main.rs:60)."""

    def __init__(self, ctx, in_fmt, out_fmt, samplerate, samplenum=0, slab_bytes=8 << 20, n_slabs=3, path=None,
                 copy_only=False, in_host_flags=0, out_host_flags=0, unpaced=False, no_probe=False, gather=None, gather_self=False):
        """ctx: one Context, or a list of Contexts (one per GPU; slab k runs on context k mod len(ctx), n_slabs slabs
        per context).  path / copy_only / *_host_flags: dpx_stream_options (measurement only; None = the library's default).
        gather="rccl": outputs of GPUs 1..N-1 travel over RCCL into the first GPU and leave from there (distinct devices only)."""
        self._lib = _lib_handle()
        self.ctxs = list(ctx) if isinstance(ctx, (list, tuple)) else [ctx]
        self.ctx = self.ctxs[0]
        self.in_fmt, self.out_fmt = fmt_code(in_fmt), fmt_code(out_fmt)
        self._h = C.c_void_p()
        arr = (C.c_void_p * len(self.ctxs))(*[c.handle.value for c in self.ctxs])
        if path is None and not copy_only and not in_host_flags and not out_host_flags and not unpaced and not no_probe and not gather:
            check(self._lib.dpx_stream_create_multi(arr, len(self.ctxs), self.in_fmt, self.out_fmt, int(samplerate),
                                                    int(samplenum), int(slab_bytes), int(n_slabs), C.byref(self._h)))
        else:
            opt = _lib.StreamOptions(_lib.STREAM_PATHS[path or "default"] | (_lib.STREAM_COPY_ONLY if copy_only else 0) | (_lib.STREAM_UNPACED if unpaced else 0) | (_lib.STREAM_NO_PROBE if no_probe else 0),
                                     int(in_host_flags), int(out_host_flags),
                                     _lib.STREAM_GATHER[gather] | (_lib.STREAM_GATHER_SELF if gather_self else 0))
            check(self._lib.dpx_stream_create_opts(arr, len(self.ctxs), self.in_fmt, self.out_fmt, int(samplerate),
                                                   int(samplenum), int(slab_bytes), int(n_slabs), C.byref(opt), C.byref(self._h)))

    def describe(self):
        """{'path': name, 'copy_only': bool, 'numa_nodes': [node of every slab's pinned buffers, -1 = the caller's policy]}"""
        path, n = C.c_uint32(), C.c_size_t()
        check(self._lib.dpx_stream_describe(self._h, C.byref(path), None, 0, C.byref(n)))
        nodes = (C.c_int * max(1, n.value))()
        check(self._lib.dpx_stream_describe(self._h, C.byref(path), nodes, n.value, C.byref(n)))
        names = {v: k for k, v in _lib.STREAM_PATHS.items()}
        return {"path": names[path.value & 0xff], "copy_only": bool(path.value & _lib.STREAM_COPY_ONLY),
                "unpaced": bool(path.value & _lib.STREAM_UNPACED), "probe_rounds": (path.value >> 16) & 0xf,
                "streams_share_a_queue": bool(path.value & _lib.STREAM_SHARED_QUEUE),
                "gather": "rccl" if path.value & _lib.STREAM_DESCRIBE_RCCL else "d2h", "numa_nodes": list(nodes[: n.value])}

    def next_view(self):
        """Waits for the oldest submitted slab; returns a VIEW of its pinned output (valid until release())."""
        p, nb = C.c_void_p(), C.c_size_t()
        check(self._lib.dpx_stream_next(self._h, C.byref(p), C.byref(nb)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(nb.value, 1),))[: nb.value]

    def release(self):
        check(self._lib.dpx_stream_release(self._h))

    def acquire(self):
        """numpy uint8 view of the next free pinned input slab (raises DspError ERR_PLAN when all are in flight)."""
        p, cap = C.c_void_p(), C.c_size_t()
        check(self._lib.dpx_stream_acquire(self._h, C.byref(p), C.byref(cap)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cap.value,))

    def submit(self, in_bytes, segments):
        arr, n_segs, _keep = _lib.segments_array(segments)
        check(self._lib.dpx_stream_submit(self._h, int(in_bytes), arr, n_segs))

    def pending(self):
        n = C.c_int()
        check(self._lib.dpx_stream_pending(self._h, C.byref(n)))
        return n.value

    def next(self):
        """Waits for the oldest submitted slab; returns a COPY of its output bytes and frees the slab."""
        p, nb = C.c_void_p(), C.c_size_t()
        check(self._lib.dpx_stream_next(self._h, C.byref(p), C.byref(nb)))
        out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(nb.value, 1),))[: nb.value].copy()
        check(self._lib.dpx_stream_release(self._h))
        return out

    @property
    def samplenum(self):
        n = C.c_uint32()
        check(self._lib.dpx_stream_samplenum(self._h, C.byref(n)))
        return n.value

    def stats(self):
        """Host cost of submit() so far (dpx_stream_stats): slabs, plans_reused, and microseconds by part."""
        st = _lib.StreamStats()
        check(self._lib.dpx_stream_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _lib.StreamStats._fields_}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.dpx_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = None


def default_context():
    """Process-wide context on device LOCAL_RANK (or 0), created on first use."""
    global _default
    if _default is None:
        import os
        _default = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default


def plan_describe(segments, samplerate, samplenum=0, variant=0):
    """Host-only: the closed-form stretches for (n_samples, shift_hz) segments.
    Returns (list of dict, final_samplenum). Needs no GPU."""
    lib = _lib_handle()
    segs = list(segments)
    arr, _n, _keep = _lib.segments_array(segs)
    n_out = C.c_size_t()
    fin = C.c_uint32()
    cap = 64
    while True:
        out = (_lib.Stretch * cap)()
        check(lib.dpx_plan_describe(arr, len(segs), int(samplerate), int(samplenum), int(variant), out, cap,
                                    C.byref(n_out), C.byref(fin)))
        if n_out.value <= cap:
            break
        cap = n_out.value
    res = [dict(first=s.first, count=s.count, ratio=np.float32(s.ratio), n_start=s.n_start, period=s.period,
                lut_len=s.lut_len) for s in out[: n_out.value]]
    return res, fin.value


def plan_simulate(segments, samplerate, samplenum=0, block=0, vecs=0, variant=3, options=None, pair=("i16", "i16")):
    """Host-only: per-sample counter values the launch list of a plan selects, and per-sample write
    counts (must all be 1). Returns (counters uint32[n], writes uint8[n]). Needs no GPU.
    pair: the format pair of the launch being mirrored (a span launch cuts its grid per pair)."""
    lib = _lib_handle()
    segs = list(segments)
    arr, _n, _keep = _lib.segments_array(segs)
    n = sum(int(cnt) for cnt, _ in segs)
    counters = np.zeros(n, dtype=np.uint32)
    writes = np.zeros(n, dtype=np.uint8)
    check(lib.dpx_plan_simulate(arr, len(segs), int(samplerate), int(samplenum), block, vecs, variant,
                                _lib.make_options(options), _FMT[pair[0]], _FMT[pair[1]], counters.ctypes.data, writes.ctypes.data, n))
    return counters, writes


def plan_layout(segments, samplerate, samplenum=0, block=0, vecs=0, variant=3, options=None):
    """Host-only: how a plan is laid out over the kernels (dict of the dpx_layout fields). Needs no GPU."""
    lib = _lib_handle()
    segs = list(segments)
    arr, _n, _keep = _lib.segments_array(segs)
    lay = _lib.Layout()
    check(lib.dpx_plan_layout(arr, len(segs), int(samplerate), int(samplenum), block, vecs, variant,
                              _lib.make_options(options), C.byref(lay)))
    return {name: getattr(lay, name) for name, _ in _lib.Layout._fields_}


def find_reset(shift_hz, samplerate, n_start, max_scan):
    lib = _lib_handle()
    n = C.c_uint32()
    found = C.c_int()
    check(lib.dpx_find_reset(float(shift_hz), int(samplerate), int(n_start), int(max_scan), C.byref(n), C.byref(found)))
    return n.value if found.value else None


def find_reset_scan(shift_hz, samplerate, n_start, max_scan):
    """The same by trying every candidate (the definition find_reset is held against)."""
    lib = _lib_handle()
    n = C.c_uint32()
    found = C.c_int()
    check(lib.dpx_find_reset_scan(float(shift_hz), int(samplerate), int(n_start), int(max_scan), C.byref(n), C.byref(found)))
    return n.value if found.value else None


def samplenum_after_segments(segments, samplerate, samplenum0=0):
    """Counter after a list of (n_samples, shift_hz) segments (closed form per segment, one call)."""
    lib = _lib_handle()
    arr, n_segs, _keep = _lib.segments_array(segments)
    n = C.c_uint32()
    check(lib.dpx_samplenum_after_segments(arr, n_segs, int(samplerate), int(samplenum0), C.byref(n)))
    return n.value


def samplenum_after(shift_hz, samplerate, samplenum0, k):
    lib = _lib_handle()
    n = C.c_uint32()
    check(lib.dpx_samplenum_after(float(shift_hz), int(samplerate), int(samplenum0), int(k), C.byref(n)))
    return n.value
