"""Soak and leak tests: the loop of reference src/main.rs:113-118 runs until its input ends — for days on a live receiver —
so objects of the C ABI are made and destroyed by the thousand, the ring runs with a new ratio on every slab, the resident
block kernel is started and stopped hundreds of times, and device memory must come back to where it was.  Kept to about a
minute in total; tests/extended/stress_slabs.py (slab cuts at random block boundaries) runs here at reduced size."""
import ctypes as C
import time

import numpy as np
import pytest

from helpers import assert_same_bytes, make_iq

pytestmark = pytest.mark.gpu

GRANULE = 4 << 20        # hipMalloc takes memory from the driver in 2 MiB granules; pools may keep one or two


def free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def settled_free(ctx):
    ctx.synchronize()
    return free_bytes()


def test_plans_streams_and_contexts_by_the_thousand_leave_no_memory_behind(ctx, orc):
    import doppler_amd
    rate = 1024000
    n = 1 << 16
    x = make_iq("i16", n, 3)
    d_in, d_out = ctx.malloc(4 * n), ctx.malloc(4 * n)
    try:
        ctx.h2d(d_in, x)
        out = np.empty(4 * n, dtype=np.uint8)
        # one of everything first: code objects, pools and staging buffers the runtime keeps are not leaks
        p = ctx.plan_const(5000.0, rate, n)
        p.run(d_in, "i16", d_out, "i16")
        p.close()
        doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=1 << 20, n_slabs=2).close()
        doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=4 << 20, n_slabs=2).close()
        doppler_amd.Context(0).close()
        base = settled_free(ctx)
        # 2000 plans: created, run, destroyed — table plans, span plans, per-sample plans, segment lists
        t0 = time.perf_counter()
        shifts = [5000.0, 5001.0, 3.0, -7777.77, 815000.0, 0.0]
        last = {}
        for i in range(2000):
            hz = shifts[i % len(shifts)]
            if i % 5 == 4:
                p = ctx.plan_segments([(n // 4, hz), (n // 2, hz + 1.0), (n - n // 4 - n // 2, hz - 2.0)], rate, samplenum=i)
            else:
                p = ctx.plan_const(hz, rate, n, samplenum=i % 1000)
            p.run(d_in, "i16", d_out, "i16")
            if i >= 1994:
                ctx.synchronize()
                ctx.d2h(out, d_out)
                last[i] = (hz, out.copy(), p.final_samplenum)
            p.close()
        t_plans = time.perf_counter() - t0
        after_plans = settled_free(ctx)
        assert abs(base - after_plans) <= GRANULE, (base, after_plans)
        for i, (hz, got, fin) in last.items():              # ... and the last ones still compute what the oracle does
            if i % 5 == 4:
                want, sn = orc.segments_stream(x, "i16", "i16", [(n // 4, hz), (n // 2, hz + 1.0), (n - n // 4 - n // 2, hz - 2.0)], rate, samplenum=i)
            else:
                want, sn = orc.segments_stream(x, "i16", "i16", [(n, hz)], rate, samplenum=i % 1000)
            assert fin == sn
            assert_same_bytes(got, want, "i16", "plan %d of the soak" % i)
        # 1000 rings (every path; pinned slabs, HBM staging, streams, events, the staged path's probe), each used once
        t0 = time.perf_counter()
        paths = ["direct", "staged", "direct_out", "direct_in", "staged_per_slab"]
        for i in range(1000):
            st = doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=1 << 18, n_slabs=2, path=paths[i % len(paths)])
            b = st.acquire()
            b[: 4 * n] = x
            st.submit(4 * n, [(n, 5000.0)])
            if i % 3:
                st.next()                                   # two in three are drained; the others are destroyed with a slab in flight
            st.close()
        t_rings = time.perf_counter() - t0
        after_rings = settled_free(ctx)
        assert abs(base - after_rings) <= GRANULE, (base, after_rings)
        # 300 contexts (each: a stream, the warm-up allocation, staging of one operator call), two alive at a time
        t0 = time.perf_counter()
        from doppler_amd import dsp
        for i in range(300):
            c = doppler_amd.Context(0)
            o, cnt, sn = dsp.shift_block(x[:8192], "i16", "i16", 0, 5001.0, rate, ctx=c)       # resident kernel up ...
            if i % 2:
                c.plan_const(5000.0, rate, 4096).close()                                         # ... and asked to leave again
            c.close()
        t_ctx = time.perf_counter() - t0
        after_ctx = settled_free(ctx)
        assert abs(base - after_ctx) <= GRANULE, (base, after_ctx)
        print("soak: 2000 plans %.1f s, 1000 rings %.1f s, 300 contexts %.1f s; free memory %d -> %d -> %d -> %d" %
              (t_plans, t_rings, t_ctx, base, after_plans, after_rings, after_ctx))
    finally:
        ctx.free(d_in)
        ctx.free(d_out)


def test_ring_with_a_new_ratio_on_every_slab_overruns_the_period_cache(ctx, orc):
    """A live `doppler track` brings a new shift per block; the context's period cache (dpx_planner.h: kMaxEntries = 16384) is
    emptied when full.  60 000 slabs of one 8 KiB block, every one its own ratio, through the ring: the cache is overrun at
    least once, bytes and counter equal the oracle's sequential pass, device memory stays put."""
    import doppler_amd
    rate = 1024000
    n_slabs_total, per = 60000, 2048
    x = make_iq("i16", n_slabs_total * per, 17, full_scale=True)
    segs = [(per, float(np.float32(-40000.0 + 1.3 * k))) for k in range(n_slabs_total)]
    st = doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=8192, n_slabs=8)
    base = None
    try:
        outs = []
        t0 = time.perf_counter()
        for k, sg in enumerate(segs):
            if st.pending() == 8:
                outs.append(st.next())
            buf = st.acquire()
            buf[:] = x[k * 8192:(k + 1) * 8192]
            st.submit(8192, [sg])
            if k == 2000:
                base = settled_free(ctx)
        while st.pending():
            outs.append(st.next())
        dt = time.perf_counter() - t0
        got = np.concatenate(outs)
        assert abs(settled_free(ctx) - base) <= GRANULE
        want, sn = orc.segments_stream(x, "i16", "i16", segs, rate, threads=32)
        assert st.samplenum == sn
        assert_same_bytes(got, want, "i16", "ring with %d ratios" % n_slabs_total)
        s = st.stats()
        assert s["slabs"] == n_slabs_total and s["plans_reused"] == 0
        print("soak: %d one-block slabs with distinct ratios in %.1f s (%.1f us of submit per slab)" % (n_slabs_total, dt, s["total_us"] / s["slabs"]))
    finally:
        st.close()


def test_resident_kernel_started_and_stopped_2000_times_across_two_contexts(orc):
    """Two contexts of one device take turns: a block through the resident kernel of one, then work that makes it leave (the other
    context's block, a plan, a synchronize).  After 2000 cycles every launch has ended in exactly one of the two ways, on both."""
    import doppler_amd
    from doppler_amd import dsp
    a, b = doppler_amd.Context(0), doppler_amd.Context(0)
    try:
        rate = 1024000
        x = make_iq("i16", 2048 * 8, 91, full_scale=True)
        want, _ = orc.const_stream(x, "i16", "i16", 5001, rate)
        base = None
        sn = {a: 0, b: 0}
        blk = {a: 0, b: 0}
        for cycle in range(2000):
            c = a if cycle % 2 == 0 else b
            k = blk[c] % 8
            if k == 0:
                sn[c] = 0
            o, _, sn[c] = dsp.shift_block(x[k * 8192:(k + 1) * 8192], "i16", "i16", sn[c], 5001.0, rate, ctx=c)
            assert_same_bytes(o, want[k * 8192:(k + 1) * 8192], "i16", "cycle %d" % cycle)
            blk[c] += 1
            if cycle % 5 == 3:
                c.plan_const(5000.0, rate, 4096).close()        # a launch of the library: the kernel leaves first
            elif cycle % 5 == 4:
                c.synchronize()
            if cycle == 50:
                base = free_bytes()
        for c in (a, b):
            info = c.resident_info()
            assert info["launches"] == info["stops"] + info["idle_exits"] + info["running"], info
            assert info["tickets_in_flight"] == 0
        ia, ib = a.resident_info(), b.resident_info()
        assert ia["launches"] + ib["launches"] >= 1600, (ia, ib)     # nearly every cycle handed the kernel over
        assert ia["running"] + ib["running"] <= 1                    # one resident kernel per device and process
        a.synchronize()
        assert abs(free_bytes() - base) <= GRANULE
    finally:
        a.close()
        b.close()


def test_slab_cuts_at_random_block_boundaries(ctx, orc):
    """tests/extended/stress_slabs.py at reduced size: the golden-style track replay cut into slabs at random block boundaries
    (what pipe timing does to the `doppler` command), 25 cut patterns, every one equal to the oracle's block-by-block pass."""
    import doppler_amd
    rate, freq, off = 256000, 437505000, -2500
    rr = 6.8 * np.tanh((np.arange(8) - 3.0) / 2.0)
    n = rate * 3 + 2048 * 2 + 55
    x = make_iq("i16", n, 5)
    want, _, log = orc.track_stream(x, "i16", "i16", rate, freq, rr, offset_hz=off)
    want = np.asarray(want)
    nblocks = (x.size + 8191) // 8192
    rng = np.random.default_rng(2)
    for trial in range(25):
        k = int(rng.integers(0, 12))
        cuts = sorted(set(int(c) for c in rng.integers(1, nblocks, size=k))) if k else []
        bounds = [0] + cuts + [nblocks]
        st = doppler_amd.Stream(ctx, "i16", "i16", rate, 0, slab_bytes=4 << 20, n_slabs=3)
        try:
            outs = []
            for lo_b, hi_b in zip(bounds[:-1], bounds[1:]):
                lo, hi = lo_b * 8192, min(hi_b * 8192, x.size)
                if st.pending() == 3:
                    outs.append(st.next())
                buf = st.acquire()
                buf[: hi - lo] = x[lo:hi]
                segs = []
                for blk in range(lo_b, hi_b):
                    cnt = min(2048, (x.size - blk * 8192) // 4)
                    hz = float(log[blk])
                    if segs and segs[-1][1] == hz:
                        segs[-1] = (segs[-1][0] + cnt, hz)
                    else:
                        segs.append((cnt, hz))
                st.submit(hi - lo, segs)
            while st.pending():
                outs.append(st.next())
            assert_same_bytes(np.concatenate(outs), want, "i16", "cuts %r" % (cuts,))
        finally:
            st.close()
