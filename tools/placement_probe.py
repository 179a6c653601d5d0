#!/usr/bin/env python3
"""Does the placement of the caller's buffers decide the rate?  One plan (track replay, f32->f32 or i16->i16), the buffers
re-allocated TRIALS times in one process behind paddings of different sizes; per trial: the buffers' addresses and the
median rate of ITERS launches after a warm-up.  Development tool:  python tools/placement_probe.py [pair] [seconds] [trials]"""
import os, sys, time, calendar
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, doppler_amd

pair = sys.argv[1] if len(sys.argv) > 1 else "f32:f32"
seconds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 8
it, ot = pair.split(":")
BPS = {"i16": 4, "f32": 8}
RATE = 1024000
segs = bench.track_segments(seconds, RATE, it, calendar.timegm((2015, 1, 22, 19, 48, 0)))
n = sum(c for c, _ in segs)
dev = torch.device("cuda:0")
ctx = doppler_amd.Context(0)
plan = ctx.plan_segments(segs, RATE)
stream = torch.cuda.current_stream()
for t in range(trials):
    pad = torch.empty((t * 37 + 1) << 20, dtype=torch.uint8, device=dev) if t % 2 else None
    x = (torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev) if it == "i16" else torch.rand(2 * n, dtype=torch.float32, device=dev) * 2 - 1)
    y = torch.empty(n * BPS[ot], dtype=torch.uint8, device=dev)
    for _ in range(30):
        plan.run(x.data_ptr(), it, y.data_ptr(), ot, stream.cuda_stream)
    torch.cuda.synchronize()
    ms = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            plan.run(x.data_ptr(), it, y.data_ptr(), ot, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 10)
    ms.sort()
    gbps = n * (BPS[it] + BPS[ot]) / ms[len(ms) // 2] / 1e6
    print("trial %d  in %#x (mod 2 MiB %#x)  out %#x (out - in = %.1f MiB)  %.1f %% of 8 TB/s  (min %.1f, max %.1f)" % (
        t, x.data_ptr(), x.data_ptr() % (2 << 20), y.data_ptr(), (y.data_ptr() - x.data_ptr()) / 2**20, gbps / 80,
        n * (BPS[it] + BPS[ot]) / ms[-1] / 1e6 / 80, n * (BPS[it] + BPS[ot]) / ms[0] / 1e6 / 80), flush=True)
    del x, y, pad
    torch.cuda.empty_cache()
