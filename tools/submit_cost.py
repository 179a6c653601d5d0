"""What one dpx_stream_submit costs the producer thread, and what a ring of N contexts sustains from pinned memory
(no file I/O: the slabs are filled once).   python tools/submit_cost.py [--ctxs 1,2,4,8] [--slab-mib 16] [--slabs 200]
Every context is on device 0 (a one-GPU box): the figure of interest is the producer's cost per slab, which caps the
ring at slab_samples / cost whatever the number of GPUs behind it."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import doppler_amd  # noqa: E402
from doppler_amd.engine import Stream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ctxs", default="1,2,4,8")
ap.add_argument("--slab-mib", type=int, default=16)
ap.add_argument("--slabs", type=int, default=240)
ap.add_argument("--shifts", default="5000,5001")
args = ap.parse_args()
RATE = 1024000
slab = args.slab_mib << 20
n = slab // 4
rng = np.random.default_rng(3)
fill = rng.integers(-23170, 23171, size=2 * n, dtype=np.int16).view(np.uint8)
for shift in [float(x) for x in args.shifts.split(",")]:
    for nc in [int(x) for x in args.ctxs.split(",")]:
        ctxs = [doppler_amd.Context(0) for _ in range(nc)]
        per = 3
        st = Stream(ctxs, "i16", "i16", RATE, slab_bytes=slab, n_slabs=per)
        depth = per * nc
        for _ in range(depth):                          # fill every pinned slab once
            st.acquire()[:] = fill
            st.submit(slab, [(n, shift)])
        for _ in range(depth):
            st.next()
        s0 = st.stats()
        t0 = time.perf_counter()
        inflight = 0
        lib, h = st._lib, st._h
        import ctypes as C
        p, cap, nb = C.c_void_p(), C.c_size_t(), C.c_size_t()
        seg = (doppler_amd._lib.Segment * 1)()
        seg[0].n_samples, seg[0].shift_hz = n, shift
        for k in range(args.slabs):
            if inflight == depth:
                lib.dpx_stream_next(h, C.byref(p), C.byref(nb))
                lib.dpx_stream_release(h)
                inflight -= 1
            lib.dpx_stream_acquire(h, C.byref(p), C.byref(cap))
            assert lib.dpx_stream_submit(h, slab, seg, 1) == 0
            inflight += 1
        while inflight:
            lib.dpx_stream_next(h, C.byref(p), C.byref(nb))
            lib.dpx_stream_release(h)
            inflight -= 1
        dt = time.perf_counter() - t0
        s1 = st.stats()
        k = s1["slabs"] - s0["slabs"]
        print("shift %g Hz, %d context(s) on device 0, %d MiB slabs: %.0f Msamples/s; submit %.1f us per slab "
              "(plan %.1f, device image %.1f, enqueue %.1f), %d of %d plans reused -> the producer alone allows %.1f Gsamples/s"
              % (shift, nc, args.slab_mib, args.slabs * n / dt / 1e6, (s1["total_us"] - s0["total_us"]) / k,
                 (s1["plan_us"] - s0["plan_us"]) / k, (s1["upload_us"] - s0["upload_us"]) / k,
                 (s1["enqueue_us"] - s0["enqueue_us"]) / k, s1["plans_reused"] - s0["plans_reused"], k,
                 n / ((s1["total_us"] - s0["total_us"]) / k) / 1e3), flush=True)
        st.close()
        for c in ctxs:
            c.close()
