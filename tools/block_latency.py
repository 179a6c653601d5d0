"""Per-call latency of the host-pointer operator entry point dpx_shift_block (the per-8-KiB-block drop-in use)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import doppler_amd
from doppler_amd import dsp

ctx = doppler_amd.Context(0)
rng = np.random.default_rng(0)
for nbytes in (8192, 65536, 1 << 20, 1 << 24):
    x = rng.integers(-20000, 20000, size=nbytes // 2, dtype=np.int16).view(np.uint8)
    sn = 0
    for _ in range(20):
        o, c, sn = dsp.shift_block(x, "i16", "i16", sn, 5000.0, 1024000, ctx=ctx)
    k = 2000 if nbytes <= 65536 else 200
    t = time.perf_counter()
    for _ in range(k):
        o, c, sn = dsp.shift_block(x, "i16", "i16", sn, 5000.0, 1024000, ctx=ctx)
    dt = (time.perf_counter() - t) / k
    print("%9d bytes/call: %8.1f us/call  %8.1f Msamples/s" % (nbytes, dt * 1e6, nbytes / 4 / dt / 1e6))
