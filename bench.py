"""bench.py — headline benchmark of the MI355X Doppler-correction hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): `doppler const -s 1024000 -i i16
--shift 5000` on 1 GiB (268 435 456 samples) of synthetic interleaved i16 IQ per GPU, input and
output resident in HBM.  One step = one pass of the fused kernel over the rank's 1 GiB chunk.
With N GPUs the stream is N GiB long; rank r owns time-chunk r and seeds its sample counter from
the closed form (no data-path collective: weak scaling).  The RCCL ordered gather to rank 0 that
a stdout consumer would need is measured once after the timed region and reported separately.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

T_START = time.perf_counter()
T_CTX = T_START
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_SAMPLES = 268435456          # 1 GiB of i16 IQ = 131072 reference blocks of 8192 bytes
SHIFT = 5000
RATE = 1024000
BYTES_PER_SAMPLE = 8           # algorithmic: 4 read + 4 written
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
CPU_SAMPLE = N_SAMPLES         # cpu_baseline: the whole 1 GiB stream (about 10 s on one core)


def cpu_baseline(x_host, gpu_out_host):
    """Oracle (CPU restatement of the reference, reference complex.c linked when oracle/_ref exists)
    on a bounded sample of the same stream; also a bit-exactness check of the GPU output."""
    import numpy as np
    from oracle import oracle as orc
    out = {"unit": "Msamples/s", "kind": "port"}
    xb = x_host.view(np.uint8)
    # output buffers touched beforehand and a short warm pass, so that page faults and lazy
    # binding are not timed
    buf1 = np.ones(xb.size + 8, dtype=np.uint8)
    buf2 = np.ones(xb.size + 8, dtype=np.uint8)
    orc.const_stream(xb[: 8192 * 64], "i16", "i16", SHIFT, RATE)
    t = time.perf_counter()
    ref, _ = orc.const_stream(xb, "i16", "i16", SHIFT, RATE, out=buf1)
    dt = time.perf_counter() - t
    LEGS["cpu_1_core"] = round(dt, 3)
    n = x_host.size // 2
    out["value"] = round(n / dt / 1e6, 3)
    out["cores"] = 1
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 64)
    t = time.perf_counter()
    ref_mt, _ = orc.const_stream(xb, "i16", "i16", SHIFT, RATE, threads=threads, out=buf2)
    dt_mt = time.perf_counter() - t
    LEGS["cpu_all_cores"] = round(dt_mt, 3)
    out["all_cores"] = {"value": round(n / dt_mt / 1e6, 3), "cores": threads}
    out["sample"] = ("first %d samples (%d MiB in) of the same stream, memory to memory, 8192-byte blocks, "
                     "unpack/mix/pack passes and one cexpf per sample as in the reference; "
                     "ccexpf from %s" % (n, xb.size >> 20,
                                         "reference src/complex.c (oracle/_ref)" if orc.have_ref() else "the oracle's restatement"))
    same = bool(np.array_equal(ref, gpu_out_host.view(np.uint8)) and np.array_equal(ref_mt, ref))
    out["gpu_output_bit_exact_on_sample"] = same
    return out


DIST_ON = False        # a process group exists (world > 1, or DPX_BENCH_FORCE_DIST=1)
LEGS = {}              # seconds per leg of the run (rank 0), printed as `legs_s`


TLE1 = b"1 39161U 13021C   15022.00000000  .00000500  00000-0  80000-4 0  9990"   # synthetic ESTCUBE-1-like set
TLE2 = b"2 39161  98.1000  95.0000 0010000  90.0000 270.0000 14.69000000 90000"   # (tests/golden/estcube1_synthetic.tle)


def track_segments(seconds, rate, in_fmt, start_unix, frequency=437505000, offset=5000):
    """(n_samples, shift_hz) segments of `doppler track --time` over `seconds` of stream: host SGP4 range rates
    per whole second -> the reference's per-block schedule (main.rs:156-184) -> runs of equal shift."""
    import ctypes as C
    import numpy as np
    import doppler_amd
    lib = doppler_amd.lib
    out = (C.c_double * 4)()
    rr = np.empty(seconds + 2, dtype=np.float64)
    for t in range(seconds + 2):
        assert lib.dpx_orbit_observe(TLE1, TLE2, 58.26541, 26.46667, 76.0, float(start_unix + t), out) == 0
        rr[t] = out[3]
    bps = 4 if in_fmt == "i16" else 8
    nbytes = seconds * rate * bps
    cap = nbytes // 8192 + 2
    hz = np.empty(cap, dtype=np.float32)
    nb = C.c_size_t()
    assert lib.dpx_track_schedule(rr.ctypes.data, rr.size, rate, frequency, offset, 1, 0 if in_fmt == "i16" else 1,
                                  nbytes, hz.ctypes.data, cap, C.byref(nb)) == 0
    hz = hz[: nbytes // 8192]
    spb = 8192 // bps
    edges = np.flatnonzero(hz[1:].view(np.uint32) != hz[:-1].view(np.uint32)) + 1
    starts = np.concatenate([[0], edges])
    ends = np.concatenate([edges, [hz.size]])
    return [(int((e - s0) * spb), float(hz[s0])) for s0, e in zip(starts, ends)]


REPLAYS = {
    # name -> (seconds, samplerate, in, out, start (UTC), offset_hz, what)
    "track": (600, 1024000, "i16", "i16", (2015, 1, 22, 19, 48, 0), 5000,
              "doppler track -s 1024000 -i i16 (synthetic ESTCUBE-1-like TLE, --time replay of a 10 min overpass, --offset 5000): BASELINE.json configs[2]"),
    "track_256k": (600, 256000, "i16", "i16", (2015, 1, 22, 19, 48, 0), -2500,
                   "doppler track -s 256000 -i i16 --offset -2500, --time replay of the same 10 min overpass: the reference README's own "
                   "recording recipe (README.md:60)"),
    "config4_chunk": (3600, 1024000, "f32", "i16", (2015, 1, 22, 19, 23, 0), 5000,
                      "doppler track -s 1024000 -i f32 -o i16, 1 h replay: rank 3 of 8's time chunk (460.8 M samples, counter seeded from "
                      "the closed form): one GPU's share of BASELINE.json configs[4]"),
}
SUSTAIN_S = 5.5      # headline: seconds of back-to-back launches after the timed region (roofline.frac_sustained): longer than the 5 s period of the driver's gpu_busy sampler
SETTLE_S = 0.15     # untimed launches before the settled measurement of a secondary workload (clocks: profiles/r02_walk.md)


def run_track(args, world, rank, dev, ctx, steps=None, warmup=None, emit=True, which="track"):
    """Secondary workloads (never the default `value`): `track` = BASELINE.json configs[2] at N=1 and configs[4] sharded at
    N>1; `track_256k` and `config4_chunk` (N=1 only) are the replays the kernels were NOT tuned on.
    Every line carries TWO fractions of the same K launches' worth of work: `frac` — K launches right after the
    W warm-up launches, the headline's own rule (`value` and `ms_per_step` are these launches') — and `frac_settled` — K
    launches after a further SETTLE_S of untimed launches (`untimed_settling_ms`), the steady state a long stream sees."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    import calendar
    from doppler_amd import shard
    seconds, rate, it, ot, start, offset, name = REPLAYS[which]
    if which == "track" and world > 1:
        seconds, rate, it, ot, start, offset, name = REPLAYS["config4_chunk"]
        name = "doppler track -s 1024000 -i f32 -o i16, 1 h replay sharded in time chunks over %d GPUs, --offset 5000" % world
    bi, bo = (4 if it == "i16" else 8), (4 if ot == "i16" else 8)
    segs = track_segments(seconds, rate, it, calendar.timegm(start), offset=offset)
    total = sum(c for c, _ in segs)
    shards, shard_rank = (8, 3) if which == "config4_chunk" else (world, rank)
    lo, hi = shard.chunk_bounds(total, shards, shard_rank, bytes_per_sample=bi)
    before, inside = shard.segments_for_chunk(segs, lo, hi)
    import doppler_amd
    n = hi - lo
    x = (torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev) if it == "i16"
         else torch.rand(2 * n, dtype=torch.float32, device=dev) * 2 - 1)
    out = torch.empty(n * bo, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    seed = shard.seed_for_segments(before, rate)
    plan = ctx.plan_segments(inside, rate, samplenum=seed)
    plan_ms = (time.perf_counter() - t0) * 1e3
    plan_parts = plan.timing()
    t1 = time.perf_counter()
    plan.run(x.data_ptr(), it, out.data_ptr(), ot, stream.cuda_stream)
    torch.cuda.synchronize(dev)
    one_shot_ms = plan_ms + (time.perf_counter() - t1) * 1e3
    layout = doppler_amd.plan_layout(inside, rate, seed)

    def barrier():
        torch.cuda.synchronize(dev)
        if DIST_ON:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed():
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(steps):
            plan.run(x.data_ptr(), it, out.data_ptr(), ot, stream.cuda_stream)
        ev1.record(stream)
        barrier()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) / steps

    for _ in range(warmup):
        plan.run(x.data_ptr(), it, out.data_ptr(), ot, stream.cuda_stream)
    elapsed, kms_cold = timed()                # the headline's rule: W warm-up launches, then K timed ones -> value, frac
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < SETTLE_S:
        for _ in range(10):
            plan.run(x.data_ptr(), it, out.data_ptr(), ot, stream.cuda_stream)
        torch.cuda.synchronize(dev)
    settle_ms = (time.perf_counter() - t_s) * 1e3
    _, kms = timed()
    if DIST_ON:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    line = None
    if rank == 0:
        alg = n * (bi + bo)
        frac = lambda ms: round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        prof = profiled(which) if world == 1 else None
        by_tiles = it != ot and layout.get("f32_i16_by_tiles")            # the mixed pairs of a many-matrix plan run on the tile kernel
        roof = {"bound": "hbm", "achieved": round(alg / (kms_cold * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": frac(kms_cold), "frac_settled": frac(kms), "achieved_settled": round(alg / (kms * 1e-3) / 1e9, 1),
                "frac_is": "K launches right after the W warm-up launches — the headline's rule (rounds 1-4's `frac`, round 5's `frac_cold`); "
                           "frac_settled: K launches after a further %d ms of untimed launches (round 5's `frac`)" % round(settle_ms),
                "traffic": prof["hbm_bytes_per_launch"] if prof and "hbm_bytes_per_launch" in prof else None,
                "kernel": "dpx::span_kernel" if layout["walk_launches"] and not by_tiles else "dpx::tile_kernel", "layout": layout,
                "avg_launch_ms": round(kms_cold, 4), "avg_launch_ms_settled": round(kms, 4), "algorithmic_bytes_per_launch": alg}
        if prof and prof.get("avg_launch_us_kernel_trace"):
            # from the committed rocprofv3 kernel trace of this workload (same kernel sources): ALL its launches, and the ones
            # that start more than 160 ms after the first
            roof["frac_rocprof"] = round(alg / (prof["avg_launch_us_kernel_trace"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
            if prof.get("settled_avg_us"):
                roof["frac_rocprof_settled"] = round(alg / (prof["settled_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
        line = {
            "metric": "Msamples/s IQ throughput + %% HBM roofline (%s replay, secondary workload)" % which,
            "value": round((total if which != "config4_chunk" else n) * steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "untimed_settling_ms": round(settle_ms, 1), "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "samples_total": total, "samples_per_gpu": n, "segments_total": len(segs),
                       "segments_this_rank": len(inside), "plan_ms": round(plan_ms, 2), "one_shot_ms": round(one_shot_ms, 2),
                       "plan_ms_is": "closed-form seed of the chunk + dpx_plan_segments (Python marshaling included); parts inside the library, us",
                       "plan_parts_us": plan_parts,
                       "in": it, "out": ot, "samplerate": rate, "offset_hz": offset},
            "roofline": roof,
        }
        if emit:
            print(json.dumps(line), flush=True)
    plan.close()
    del x, out
    return line


RING_SLAB_BYTES = 32 << 20      # the product ring's configuration in the driver's line: 4 slabs of 32 MiB (profiles/r06_ring.md)
RING_SLABS = 4
RING_GIB = 16                   # input through the ring per measurement (>= 4 GiB: VERDICT r05 item 1)
RING_WARM_S = 0.6               # untimed cycling before a ring measurement
PRODUCT_RING_SLAB_BYTES = 64 << 20   # N > 1: the N-device ring rank 0 runs after the gather leg (gather.product_ring)
PRODUCT_RING_SLABS = 3               # per GPU
PRODUCT_RING_S = 2.0
PRODUCT_RING_DEADLINE_S = 60         # per form of the shipped ring (neither has ever run between two physical devices)


def link_duplex(dev, nbytes, reps):
    """What the PCIe link gives two free-running copy streams of pinned memory — H2D on one stream against D2H on another,
    `reps` transfers of `nbytes` each, no ring, no dependencies: GB/s per direction (events on each stream)."""
    h_up, h_dn = torch.empty(nbytes, dtype=torch.uint8).pin_memory(), torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d_up, d_dn = torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    s_up, s_dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    out = {}
    for both in (False, True):
        for _ in range(2):                       # first pass untimed
            torch.cuda.synchronize(dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(s_up)
            ev[2].record(s_dn)
            for _ in range(reps):
                with torch.cuda.stream(s_up):
                    d_up.copy_(h_up, non_blocking=True)
                if both:
                    with torch.cuda.stream(s_dn):
                        h_dn.copy_(d_dn, non_blocking=True)
            ev[1].record(s_up)
            ev[3].record(s_dn)
            torch.cuda.synchronize(dev)
        if both:
            out["h2d_GB_per_s_duplex"] = round(nbytes * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9, 2)
            out["d2h_GB_per_s_duplex"] = round(nbytes * reps / (ev[2].elapsed_time(ev[3]) * 1e-3) / 1e9, 2)
        else:
            out["h2d_GB_per_s_alone"] = round(nbytes * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9, 2)
    torch.cuda.synchronize(dev)
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(reps):
            with torch.cuda.stream(s_dn):
                h_dn.copy_(d_dn, non_blocking=True)
        torch.cuda.synchronize(dev)
        out["d2h_GB_per_s_alone"] = round(nbytes * reps / (time.perf_counter() - t0) / 1e9, 2)
    return out


def drive_ring(ctxs, slab_bytes, n_slabs, total_bytes, fill, check=None, warm_s=RING_WARM_S, **stream_kw):
    """The C-ABI slab ring (dpx_stream_create ... _release) driven from pinned memory as a producer with data at hand
    would: every slab filled once by fill(j), one untimed lap (plans, device images, first touch of the output slabs;
    check(j, view) sees its outputs), warm_s seconds of untimed cycling (a process's first 0.2-0.3 s of heavy PCIe traffic
    hold one 30-50 ms stall, whatever the path: profiles/r06_ring.md), then next -> release -> acquire -> submit until
    total_bytes of input have gone through.  Returns (seconds from the first timed submit to the last output handed back,
    bytes, describe(), stats, GB/s of input in windows of 50 ms: (min, median))."""
    import numpy as np
    import doppler_amd
    ctxs = ctxs if isinstance(ctxs, (list, tuple)) else [ctxs]
    st = doppler_amd.Stream(list(ctxs), "i16", "i16", RATE, slab_bytes=slab_bytes, n_slabs=n_slabs, **stream_kw)
    try:
        ring = n_slabs * len(ctxs)
        segs = [(slab_bytes // 4, float(SHIFT))]
        for j in range(ring):
            fill(j, st.acquire())
        for _ in range(ring):
            st.submit(slab_bytes, segs)
        for j in range(ring):
            v = st.next_view()
            if check is not None:
                check(j, v)
            st.release()

        def cycle(stop):
            """keeps the ring full until stop(done, seconds) says so, then drains it; returns the times at which the slabs came back"""
            for _ in range(ring):
                st.acquire()
            t0 = time.perf_counter()
            for _ in range(ring):
                st.submit(slab_bytes, segs)
            stamps, submitted = [], ring
            while len(stamps) < submitted:
                st.next_view()
                st.release()
                stamps.append(time.perf_counter() - t0)
                if not stop(len(stamps), stamps[-1]):
                    st.acquire()
                    st.submit(slab_bytes, segs)
                    submitted += 1
            return stamps

        if warm_s > 0:
            cycle(lambda done, t: t >= warm_s)
        laps = max(ring, total_bytes // slab_bytes)
        stamps = cycle(lambda done, t: done + ring > laps)
        dt = stamps[-1]
        w = 0.05
        cnt, _ = np.histogram(np.array(stamps), np.arange(0.0, dt, w)) if dt > 3 * w else (np.array([len(stamps)]), None)
        rates = cnt * slab_bytes / (w if dt > 3 * w else dt) / 1e9
        return dt, len(stamps) * slab_bytes, st.describe(), st.stats(), (round(float(rates.min()), 2), round(float(np.median(rates)), 2))
    finally:
        st.close()


def product_ring(devices, seconds=PRODUCT_RING_S, gather=None):
    """gather.product_ring (N > 1): dpx_stream_create_multi over all N devices from ONE process, driven from pinned memory
    for about `seconds`: aggregate rate, the rate per device, where every slab's pinned buffers live."""
    import numpy as np
    import doppler_amd
    ctxs = [doppler_amd.Context(d) for d in devices]
    try:
        slab = PRODUCT_RING_SLAB_BYTES
        pat = np.random.default_rng(3).integers(-23170, 23171, size=slab // 2, dtype=np.int16).view(np.uint8)

        def fill(j, buf):
            buf[:] = pat

        # sized by time: about `seconds` at the single-GPU ring's rate per device
        total = int(seconds * 45e9 * len(set(devices))) // slab * slab
        kw = {"gather": gather} if gather else {}
        if gather == "rccl":
            total = int(seconds * 45e9) // slab * slab          # every output byte leaves through the first GPU's link
        dt, nb, desc, stats, win = drive_ring(ctxs, slab, PRODUCT_RING_SLABS, total, fill, None, warm_s=RING_WARM_S, **kw)
        n = len(ctxs)
        return {"what": "dpx_stream_create_multi over %d device(s) from one process (what `doppler --gpus N` runs on): %d slabs of %d MiB per "
                        "GPU in pinned host memory, headline shift, i16 -> i16, slab k on GPU k mod N, per-GPU D2H, outputs in order" %
                        (n, PRODUCT_RING_SLABS, slab >> 20),
                "devices": devices, "Msamples_per_s": round(nb / 4 / dt / 1e6, 1), "seconds": round(dt, 3),
                "GB_per_s_each_way_aggregate": round(nb / dt / 1e9, 2), "GB_per_s_each_way_per_device": round(nb / dt / 1e9 / n, 2),
                "GB_per_s_in_50ms_windows": {"min": win[0], "median": win[1]},
                "gather": desc.get("gather"), "path": desc["path"], "stream_probe_rounds": desc.get("probe_rounds"), "streams_share_a_queue": desc.get("streams_share_a_queue"),
                "slab_numa_nodes": desc["numa_nodes"], "submit_us_per_slab": round(stats["total_us"] / max(1, stats["slabs"]), 2),
                "producer": "one Python thread: acquire / submit / next / release per slab (%.0f us of dpx_stream_submit per slab)" %
                            (stats["total_us"] / max(1, stats["slabs"]))}
    finally:
        for c in ctxs:
            c.close()


def stream_ring(ctx, dev, x, out):
    """extra.stream_ring: the product's streaming path (dpx_stream_*: what `doppler const < in > out` runs on, reference
    src/main.rs:57-99 with the blocks gathered into slabs) on ITS roofline, the PCIe link: headline shift, i16 -> i16, slabs
    in pinned host memory, RING_GIB GiB of input through the ring.  `peak` = the slower direction of two free-running copy
    streams (H2D against D2H, same transfer size, no ring) measured in the same run; the same ring without arithmetic
    (DPX_STREAM_COPY_ONLY) is reported beside it.  x / out: the headline's device input and output (out is compared with the
    oracle in the cpu_baseline leg of this run): the ring's first lap must reproduce out's first slabs byte for byte."""
    import numpy as np
    ring = RING_SLABS
    xh = x[: ring * RING_SLAB_BYTES // 2].cpu().numpy().view(np.uint8)
    oh = out[: ring * RING_SLAB_BYTES // 2].cpu().numpy().view(np.uint8)
    same = []

    def fill(j, buf):
        buf[:] = xh[j * RING_SLAB_BYTES:(j + 1) * RING_SLAB_BYTES]

    def check(j, v):
        same.append(bool(np.array_equal(v, oh[j * RING_SLAB_BYTES:(j + 1) * RING_SLAB_BYTES])))

    total = RING_GIB << 30
    dt, nb, desc, stats, win = drive_ring(ctx, RING_SLAB_BYTES, ring, total, fill, check)
    dt_copy, nb_copy, desc_copy, _, _ = drive_ring(ctx, RING_SLAB_BYTES, ring, total, fill, None, path=desc["path"], copy_only=True)
    link = link_duplex(dev, RING_SLAB_BYTES, 64)
    peak = min(link["h2d_GB_per_s_duplex"], link["d2h_GB_per_s_duplex"])
    ach = nb / dt / 1e9
    res = {
        "what": "dpx_stream_* ring from pinned host memory: %d slabs of %d MiB, %d GiB of i16 IQ in and as much out, headline shift; "
                "wall time from the first timed submit to the last output handed back (one untimed lap and %.1f s of untimed cycling before)" %
                (ring, RING_SLAB_BYTES >> 20, RING_GIB, RING_WARM_S),
        "Msamples_per_s": round(nb / 4 / dt / 1e6, 1), "seconds": round(dt, 4), "path": desc["path"],
        "slab_bytes": RING_SLAB_BYTES, "slabs_in_flight": ring, "stream_probe_rounds": desc.get("probe_rounds"),
        "streams_share_a_queue": desc.get("streams_share_a_queue"),
        "first_lap_equals_the_device_resident_output": bool(same) and all(same),
        "submit_us_per_slab": round(stats["total_us"] / max(1, stats["slabs"]), 2), "plans_reused": stats["plans_reused"],
        "GB_per_s_in_50ms_windows": {"min": win[0], "median": win[1]},
        "roofline": {"bound": "pcie", "unit": "GB/s per direction", "achieved": round(ach, 2), "peak": peak,
                     "frac": round(ach / peak, 4),
                     "peak_is": "the slower direction of H2D against D2H on two free-running streams of pinned memory, %d MiB transfers, "
                                "no ring and no dependencies, same run" % (RING_SLAB_BYTES >> 20),
                     "link": link,
                     "copy_only_ring_GB_per_s": round(nb_copy / dt_copy / 1e9, 2),
                     "frac_of_copy_only_ring": round(ach / (nb_copy / dt_copy / 1e9), 4),
                     "algorithmic_bytes_per_sample_per_direction": 4},
    }
    return res


def device_identity(dev_index):
    """What tells two ranks' GPUs apart in a SCALE record: PCI bus id (hipDeviceGetPCIBusId), the GPU's NUMA node, name."""
    info = {"device": dev_index}
    try:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, dev_index) == 0:
            info["pci_bus_id"] = buf.value.decode()
            numa = "/sys/bus/pci/devices/%s/numa_node" % info["pci_bus_id"].lower()
            if os.path.exists(numa):
                info["numa_node"] = int(open(numa).read().strip())
    except Exception as e:   # identification only
        info["pci_bus_id_error"] = str(e)[:100]
    try:
        info["name"] = torch.cuda.get_device_properties(dev_index).name
    except Exception:
        pass
    return info


def per_rank_report(rank, dev_index, avg_kernel_ms, elapsed_s):
    """Every rank's own figures, gathered on rank 0 (all_gather_object): a SCALE line then shows the skew between the
    GPUs, not only the slowest one."""
    mine = dict(device_identity(dev_index), rank=rank, avg_kernel_ms=round(avg_kernel_ms, 4), timed_region_s=round(elapsed_s, 5),
                pid=os.getpid(), cpu_affinity=len(os.sched_getaffinity(0)))
    if not DIST_ON:
        return [mine]
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    return everyone


GATHER_TIMEOUT_S = 240
PMC_PROFILE = "r06_pmc_traffic.json"     # tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE passes + a kernel-trace pass
# everything that decides what a launch IS: the device code, and the host code that picks row lengths, spans, sub-launches and routing
KERNEL_SOURCES = ["doppler_amd/csrc/dpx_kernels.hip", "doppler_amd/csrc/dpx_sincos.h", "doppler_amd/csrc/dpx_types.h",
                  "doppler_amd/csrc/dpx_planner.cpp", "doppler_amd/csrc/dpx_planner.h", "doppler_amd/csrc/dpx_context.cpp",
                  "doppler_amd/csrc/dpx_plans.cpp"]


def kernel_source_sha():
    """Hash of the kernel sources: profile-derived numbers are only quoted for the code they were measured on."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def profiled(workload):
    """PMC traffic and rocprofv3 kernel-trace duration of the dominant kernel from profiles/ (tools/profile_round.sh),
    or None when the committed profile was taken on different kernel sources."""
    path = os.path.join(ROOT, "profiles", PMC_PROFILE)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        pj = json.load(f)
    if pj.get("kernel_source_sha") != kernel_source_sha():
        return None
    return pj.get(workload)


def build_result(args, world, n, elapsed, avg_kernel_ms, gather, plan_ms=None, one_shot_ms=None, per_rank=None, sustained=None):
    """The one JSON line of the headline workload (rank 0)."""
    achieved = n * BYTES_PER_SAMPLE / (avg_kernel_ms * 1e-3) / 1e9
    prof = profiled("const")
    value = world * n * args.steps / elapsed / 1e6
    roof = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": prof["hbm_bytes_per_launch"] if prof else None,
        "kernel": "dpx::rows_kernel<i16,i16>", "avg_launch_ms": round(avg_kernel_ms, 4),
        "algorithmic_bytes_per_launch": n * BYTES_PER_SAMPLE,
        "timing": "one HIP event pair on the launch stream around the K timed launches / K (includes the ~1.5 us inter-launch gap)",
        "traffic_source": ("profiles/%s (same kernel sources: sha %s)" % (PMC_PROFILE, kernel_source_sha())) if prof else
                          "none: no PMC profile of these kernel sources is committed (sha %s)" % kernel_source_sha(),
    }
    if sustained:
        roof["frac_sustained"] = sustained["frac"]
        roof["sustained_s"] = sustained["sustained_s"]
        roof["sustained_launches"] = sustained["launches"]
        roof["avg_launch_ms_sustained"] = sustained["avg_launch_ms"]
    if prof and prof.get("avg_launch_us_kernel_trace"):
        us = prof["avg_launch_us_kernel_trace"]          # the committed kernel trace's one-shot + warm-up + timed launches
        roof["frac_rocprof"] = round(n * BYTES_PER_SAMPLE / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
        roof["avg_launch_us_rocprof"] = us
        if prof.get("sustained_avg_us"):                 # the committed trace's launches of the sustained leg
            roof["frac_rocprof_sustained"] = round(n * BYTES_PER_SAMPLE / (prof["sustained_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
    result = {
        "metric": "Msamples/s IQ throughput + % HBM roofline, 1 GB i16 stream, 1/2/4/8 GPUs",
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "doppler const -s 1024000 -i i16 --shift 5000, 1 GiB (268435456 samples) synthetic i16 IQ "
                        "per GPU, device-resident in and out, i16 out (BASELINE.json configs[1])",
            "samples_per_gpu": n, "in": "i16", "out": "i16", "shift_hz": SHIFT, "samplerate": RATE,
            "sharding": "independent time-chunk per rank, counter seeded from the closed form; no data-path collective",
            "i16_cast": "`(x * 32767.0) as i16` saturating with NaN -> 0 (Rust >= 1.45 semantics; a 2016 rustc wrapped) — "
                        "the headline inputs stay inside full scale, so the corner is not exercised here",
            "plan_ms": plan_ms, "one_shot_ms": one_shot_ms,
        },
        "roofline": roof,
    }
    # what the process group really was: the driver's SCALE record can show that RCCL saw N ranks
    result["backend"] = dist.get_backend() if DIST_ON else None
    result["world_size_seen"] = dist.get_world_size() if DIST_ON else 1
    if per_rank:
        # rank by rank: the launch duration each GPU measured with its own HIP events, its wall time of the timed region, which
        # physical device it was — `value` above is the max-over-ranks figure, this is what it is the max OF
        result["per_rank"] = per_rank
        ks = [r["avg_kernel_ms"] for r in per_rank if r]
        result["per_rank_kernel_ms"] = ks
        if ks:
            result["kernel_ms_spread"] = {"min": min(ks), "max": max(ks), "max_over_min": round(max(ks) / min(ks), 4)}
    if gather:
        result["gather"] = gather
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-sustain", action="store_true", help="skip the sustained leg (roofline.frac_sustained)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (replay) workloads appended under `extra`")
    ap.add_argument("--workload", default="const", choices=["const"] + sorted(REPLAYS),
                    help="const = the headline (default); track / track_256k / config4_chunk = secondary replay workloads")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # DPX_BENCH_SHARE_GPU=1 (development only): every rank uses GPU 0 and gloo, to exercise the
    # multi-rank code path on a one-GPU box; the numbers of such a run mean nothing.
    share = os.environ.get("DPX_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # DPX_BENCH_FORCE_DIST=1: take the multi-rank branch (RCCL process group, barrier, all_reduce(MAX), ordered gather,
    # per-GPU D2H) even with ONE rank, so that the code the driver's 2/4/8-GPU runs execute has run under RCCL on a one-GPU box
    global DIST_ON
    DIST_ON = world > 1 or os.environ.get("DPX_BENCH_FORCE_DIST") == "1"
    if DIST_ON:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import doppler_amd
    from doppler_amd import shard

    ctx = doppler_amd.Context(dev_index)
    global T_CTX
    T_CTX = time.perf_counter()
    if args.workload != "const":
        if args.workload != "track" and world > 1:
            raise SystemExit("--workload %s is a one-GPU workload" % args.workload)
        run_track(args, world, rank, dev, ctx, which=args.workload)
        if DIST_ON:
            dist.barrier()
            dist.destroy_process_group()
        return
    n = N_SAMPLES
    # rank r owns global samples [r*n, (r+1)*n) of the world*n-sample stream: block-aligned chunk,
    # counter seeded from the closed form of dsp.rs:125-130
    lo, hi = shard.chunk_bounds(world * n, world, rank)
    assert (lo, hi) == (rank * n, (rank + 1) * n)
    t_leg = time.perf_counter()
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xD0BB1E5 + rank)
    x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev, generator=gen)
    out = torch.empty(2 * n, dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream(dev)
    torch.cuda.synchronize(dev)
    LEGS["generate"] = round(time.perf_counter() - t_leg, 3)
    # plan (host: closed-form seed + period scan + launch list; device: descriptor upload, corrector table) and the very
    # first launch, timed once: what a one-shot caller pays; the timed region below re-launches the resident plan
    t_plan = time.perf_counter()
    sn0 = shard.chunk_seed(float(SHIFT), RATE, lo)
    plan = ctx.plan_const(float(SHIFT), RATE, n, samplenum=sn0)
    plan_ms = (time.perf_counter() - t_plan) * 1e3
    plan.run(x.data_ptr(), "i16", out.data_ptr(), "i16", stream.cuda_stream)
    torch.cuda.synchronize(dev)
    one_shot_ms = (time.perf_counter() - t_plan) * 1e3

    def step():
        plan.run(x.data_ptr(), "i16", out.data_ptr(), "i16", stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if DIST_ON:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    # One HIP event pair on the launch stream brackets the K timed launches (no per-launch events: an event
    # record between two launches costs 2-4 us of idle GPU, which would be charged to every step).
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step()
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    LEGS["timed"] = round(elapsed, 4)
    if DIST_ON:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    avg_kernel_ms = ev0.elapsed_time(ev1) / args.steps      # launch duration incl. the inter-launch gap
    per_rank = per_rank_report(rank, dev_index, avg_kernel_ms, LEGS["timed"])

    # ---- outside the timed region: SUSTAIN_S seconds of back-to-back headline launches under one event pair — the settled
    # figure of the headline (the secondary workloads carry theirs as frac_settled), and long enough for an outside sampler
    # (the driver's gpu_busy, 5 s period) to see the GPU at work.  `value` / `ms_per_step` stay the K timed launches above.
    sustained = None
    if not args.no_sustain:
        t_leg = time.perf_counter()
        n_sus = max(args.steps, int(SUSTAIN_S / (avg_kernel_ms * 1e-3)) + 1)
        es0, es1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        es0.record(stream)
        for _ in range(n_sus):
            step()
        es1.record(stream)
        torch.cuda.synchronize(dev)
        sus_ms = es0.elapsed_time(es1)
        sustained = {"launches": n_sus, "sustained_s": round(sus_ms * 1e-3, 3), "avg_launch_ms": round(sus_ms / n_sus, 4),
                     "frac": round(n * BYTES_PER_SAMPLE / (sus_ms / n_sus * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        LEGS["sustained"] = round(time.perf_counter() - t_leg, 3)

    # ---- outside the timed region: ordered gather (multi-GPU), host round trip, CPU baseline
    hard_exit = False
    gather = None
    t_leg = time.perf_counter()
    if DIST_ON:
        # The gather is reported beside `value`, never inside it.  A watchdog keeps a stuck transfer from
        # swallowing the measurement: after GATHER_TIMEOUT_S every rank gives up and rank 0 prints its line without it.
        import threading
        gather_state = {"done": False}

        def give_up():
            if gather_state["done"]:
                return
            if rank == 0:
                line = result_line(None)
                line["gather"] = {"error": "ordered gather did not finish within %d s; skipped" % GATHER_TIMEOUT_S}
                print(json.dumps(line), flush=True)
            os._exit(0)

        def result_line(g):
            return build_result(args, world, n, elapsed, avg_kernel_ms, g, round(plan_ms, 3), round(one_shot_ms, 3), per_rank, sustained)

        timer = threading.Timer(GATHER_TIMEOUT_S, give_up)
        timer.daemon = True
        timer.start()
        try:
            # (development mode on a shared GPU: gloo stages device tensors through host sockets at ~0.1 GB/s — a 64 MiB piece
            # per rank exercises the same code in seconds; under RCCL the whole chunk moves)
            ng = n if not share else min(n, 1 << 24)
            og = out[: 2 * ng]
            barrier()
            tg = time.perf_counter()
            buf = shard.ordered_gather(og, [2 * ng] * world, dst=0)
            barrier()
            tg = time.perf_counter() - tg
            gather = {"what": "RCCL send/recv of every rank's output chunk into rank 0, in rank order (not in `value`)",
                      "bytes_per_rank": 4 * ng,
                      "ms": round(tg * 1e3, 3), "GB_per_s_into_rank0": round((world - 1) * 4 * ng / tg / 1e9, 2)}
            # peer by peer, one transfer at a time: what each link into rank 0 gives on its own (a slow or indirect xGMI
            # path shows here; the batch above shows what they give together)
            per_peer = []
            for peer in range(1, world):
                barrier()
                tp = time.perf_counter()
                if rank == peer:
                    dist.send(og, 0)
                elif rank == 0:
                    dist.recv(buf[2 * ng * peer:2 * ng * (peer + 1)], peer)
                barrier()
                tp = time.perf_counter() - tp
                per_peer.append({"peer": peer, "ms": round(tp * 1e3, 3), "GB_per_s": round(4 * ng / tp / 1e9, 2)})
            gather["per_peer"] = per_peer
            del buf
        except Exception as e:   # reported, not fatal: the headline does not depend on the gather
            gather = {"error": str(e)[:300]}
        try:
            # the alternative a host-side consumer (stdout) really wants: every GPU copies its own chunk to pinned host
            # memory over its own PCIe link, all at once (what `doppler --gpus N` / dpx_stream_create_multi does)
            ho = None
            try:
                ho = torch.empty(2 * n, dtype=torch.int16).pin_memory()
            except Exception:
                ho = torch.empty(2 * n, dtype=torch.int16)          # pageable: slower, but every rank still takes part
            barrier()
            td = time.perf_counter()
            ho.copy_(out, non_blocking=True)
            barrier()
            td = time.perf_counter() - td
            gather["per_gpu_d2h"] = {"what": "every rank copies its chunk to %s host memory concurrently (no gather through one GPU)"
                                             % ("pinned" if ho.is_pinned() else "pageable"),
                                     "ms": round(td * 1e3, 3), "GB_per_s_aggregate": round(world * 4 * n / td / 1e9, 2)}
            del ho
        except Exception as e:
            gather["per_gpu_d2h"] = {"error": str(e)[:300]}
        # ---- what SHIPS for a host consumer (README.md's stated deviation): `doppler --gpus N` / dpx_stream_create_multi, ONE
        # process feeding every GPU's slab ring from pinned memory, per-GPU D2H, outputs in order.  The other ranks free
        # their buffers and wait at the barrier while rank 0 opens a context on every device and runs that ring.
        if world > 1 or os.environ.get("DPX_BENCH_FORCE_PRODUCT_RING") == "1":
            if gather is None:
                gather = {}
            if rank != 0:
                del x, out
                x = out = None
                torch.cuda.empty_cache()
            barrier()
            # The other ranks wait on the process group's STORE (host side), not in a collective: a barrier kernel spinning on
            # GPUs 1..N-1 would sit beside the ring's launches there.  Rank 0 always reaches the key, whatever the ring did.
            store = None
            try:
                store = dist.distributed_c10d._get_default_store()
            except Exception:
                store = None
            if rank == 0:
                devices = [0] * world if share else list(range(world))

                def guarded(key, **kw):
                    """one form of the shipped ring in a thread of its own with a deadline: neither form has ever run between two
                    physical devices — a ring that does not come back is reported as such and the line is printed without it"""
                    box = {}

                    def work():
                        try:
                            box["r"] = product_ring(devices, **kw)
                        except Exception as e:
                            box["r"] = {"error": str(e)[:300]}

                    th = threading.Thread(target=work, daemon=True)
                    th.start()
                    th.join(PRODUCT_RING_DEADLINE_S)
                    if th.is_alive():
                        gather[key] = {"error": "did not come back within %d s" % PRODUCT_RING_DEADLINE_S}
                        return False
                    gather[key] = box["r"]
                    return True

                ok = guarded("product_ring")
                # the same ring with the gather BASELINE.json's north_star names (`doppler --gather rccl`): outputs of GPUs
                # 1..N-1 over RCCL into GPU 0, from there to the host.  Needs distinct devices (one communicator rank each).
                if not ok:
                    gather["product_ring_rccl"] = {"skipped": "the per-GPU form did not come back"}
                elif len(set(devices)) == len(devices):
                    ok = guarded("product_ring_rccl", gather="rccl")
                else:
                    gather["product_ring_rccl"] = {"skipped": "the ranks share one GPU (development mode): RCCL needs one device per rank"}
                hard_exit = not ok            # a ring thread is still stuck inside the runtime: no collective after this, the process leaves by os._exit
                if store is not None:
                    if hard_exit:
                        store.set("dpx_hard_exit", "1")
                    store.set("dpx_product_ring_done", "1")
            elif store is not None:
                import datetime
                try:
                    store.wait(["dpx_product_ring_done"], datetime.timedelta(seconds=GATHER_TIMEOUT_S))
                    hard_exit = store.check(["dpx_hard_exit"])
                except Exception:
                    pass
            if not hard_exit:
                barrier()
        gather_state["done"] = True
        timer.cancel()
        LEGS["gather"] = round(time.perf_counter() - t_leg, 3)

    result = None
    if rank == 0:
        result = build_result(args, world, n, elapsed, avg_kernel_ms, gather, round(plan_ms, 3), round(one_shot_ms, 3), per_rank, sustained)
        if world == 1:
            # PCIe-inclusive figure (never `value`): pinned host -> HBM -> kernel -> pinned host
            t_leg = time.perf_counter()
            try:
                hx = torch.empty(2 * n, dtype=torch.int16).pin_memory()
                ho = torch.empty(2 * n, dtype=torch.int16).pin_memory()
                hx.copy_(x)
                torch.cuda.synchronize(dev)
                th = time.perf_counter()
                x.copy_(hx, non_blocking=True)
                step()
                ho.copy_(out, non_blocking=True)
                torch.cuda.synchronize(dev)
                th = time.perf_counter() - th
                result["host_round_trip"] = {"what": "pinned host -> H2D -> kernel -> D2H -> pinned host, one pass, not overlapped",
                                             "ms": round(th * 1e3, 2), "Msamples_per_s": round(n / th / 1e6, 1)}
                del hx, ho
            except Exception as e:  # pinning 2 GiB can fail on small hosts; the headline does not depend on it
                result["host_round_trip"] = {"error": str(e)[:200]}
            LEGS["host_round_trip"] = round(time.perf_counter() - t_leg, 3)     # pinning 2 x 1 GiB is most of it
            stream_ring_res = None
            if not args.no_extra:
                t_leg = time.perf_counter()
                try:
                    stream_ring_res = stream_ring(ctx, dev, x, out)
                except Exception as e:
                    stream_ring_res = {"error": str(e)[:300]}
                LEGS["stream_ring"] = round(time.perf_counter() - t_leg, 3)
            if not args.no_cpu:
                t_leg = time.perf_counter()
                m = min(CPU_SAMPLE, n)
                xh = x[: 2 * m].cpu().numpy()
                oh = out[: 2 * m].cpu().numpy()
                result["cpu_baseline"] = cpu_baseline(xh, oh)
                LEGS["cpu_baseline"] = round(time.perf_counter() - t_leg, 3)
            del x, out
            torch.cuda.empty_cache()
            t_leg = time.perf_counter()
            if not args.no_extra:
                # secondary workloads, after the timed region: they ride along in the driver's record.  BASELINE.json
                # configs[2], the reference README's 256 ksps recording recipe, one GPU's chunk of configs[4].
                result["extra"] = {"stream_ring": stream_ring_res}
                for which in ("track", "track_256k", "config4_chunk"):
                    try:
                        result["extra"][which] = run_track(args, 1, 0, dev, ctx, steps=min(args.steps, 20),
                                                           warmup=min(args.warmup, 3), emit=False, which=which)
                    except Exception as e:
                        result["extra"][which] = {"error": str(e)[:300]}
                    torch.cuda.empty_cache()
                LEGS["extra_replays"] = round(time.perf_counter() - t_leg, 3)
        LEGS["import_and_context"] = round(T_CTX - T_START, 3)
        result["legs_s"] = dict(LEGS)
        print(json.dumps(result), flush=True)

    if hard_exit:                 # (see the product ring leg: a stuck ring thread on rank 0; the line above is already out)
        sys.stdout.flush()
        os._exit(0)
    plan.close()
    if DIST_ON:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
