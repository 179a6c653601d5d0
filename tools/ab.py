"""A/B measurement of kernel shapes on one MI355X (development tool, not the bench contract).

    python tools/ab.py [--rounds R] [--iters K] [--only SUBSTR] [--set NAME]

Every case = (workload, format pair, dpx_set_tuning variant, dpx_options).  All plans are built first, then the cases are
timed round-robin (R rounds x K back-to-back launches each, one HIP event pair per burst) so that clock and thermal
drift hit every case alike; the median burst is reported as algorithmic GB/s and % of the 8 TB/s HBM peak.
"""
import argparse
import calendar
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
import doppler_amd  # noqa: E402

BPS = {"i16": 4, "f32": 8}
RATE = 1024000


def track_segs(seconds, fmt):
    start = calendar.timegm((2015, 1, 22, 19, 48, 0)) if seconds <= 600 else calendar.timegm((2015, 1, 22, 19, 23, 0))
    return bench.track_segments(seconds, RATE, fmt, start)


def const_segs(shift, n=268435456):
    return [(n, float(shift))]


def synth_segs(rows, total=614400000, P=65536):
    """stretches of exactly `rows` periods of P = 65536 (shift = odd multiples of rate / 65536)"""
    segs, m, left = [], 1, total
    while left > 0:
        cnt = min(left, rows * P)
        segs.append((cnt, RATE / 65536.0 * (m % 256)))   # m odd <= 255: m * n < 2^24 is exact in f32, so P = 65536 exactly
        m += 2
        left -= cnt
    return segs


def cases(which):
    c = []
    if which in ("synth",):
        c.append(("synth one matrix of P=65536", lambda f: [(614400000, RATE / 65536.0)], "i16:i16", 5, dict(walk_compute=0)))
        c.append(("synth one matrix of P=65536", lambda f: [(614400000, RATE / 65536.0)], "i16:i16", 5, dict(walk_compute=1)))
        for rows in (2, 3, 4, 5, 6, 8, 10, 13, 16, 20, 26, 40):
            for comp in (0, 1):
                for waves in (5,):
                    c.append(("synth %d rows of P=65536" % rows, lambda f, r=rows: synth_segs(r), "i16:i16", 3, dict(walk_compute=comp, walk_waves=waves)))
    if which == "synth2":        # chunking policy per matrix height: (wavefronts, most rows per wavefront)
        for rows in (3, 5, 7, 9, 10, 12, 14, 17, 20, 25):
            for waves, mr in ((4, 2), (4, 3), (4, 4), (5, 2), (5, 3), (3, 2), (3, 4)):
                c.append(("synth %d rows of P=65536" % rows, lambda f, r=rows: synth_segs(r, total=307200000), "i16:i16", 3, dict(walk_waves=waves, walk_rows=mr)))
    if which == "size":
        for n in (268435456, 614400000, 1073741824):
            c.append(("const 5000 Hz n=%d" % n, lambda f, n=n: const_segs(5000, n), "i16:i16", 3, {}))
            c.append(("const 5001 Hz n=%d" % n, lambda f, n=n: const_segs(5001, n), "i16:i16", 3, dict(walk_compute=1)))
        for secs in (150, 262, 600):
            for comp in (0, 1):
                c.append(("track %d s replay" % secs, lambda f, t=secs: track_segs(t, f), "i16:i16", 3, dict(walk_compute=comp)))
    if which == "rowsopt":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift, rate in ((100, RATE), (9876.543, RATE), (815000, 2400000), (5000, RATE), (2500, RATE), (200, RATE), (50, RATE), (25, RATE), (-15000, 256000)):
            for pair in ("i16:i16", "f32:f32"):
                for opts in (dict(), dict(rows_r=4)):
                    c.append(("const %g Hz @%d" % (shift, rate), lambda f, s=shift: const_segs(s), pair, 6, dict(opts, _rate=rate)))
    if which == "bigp":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift, rate in ((3, RATE), (1, RATE), (10, RATE), (7, 2400000), (25, RATE), (100, RATE)):
            for pair in ("i16:i16", "f32:f32", "f32:i16"):
                for variant in (6, 5):
                    c.append(("const %g Hz @%d" % (shift, rate), lambda f, s=shift: const_segs(s), pair, variant, dict(_rate=rate)))
    if which == "bigshape":      # long periods rich in factors of two (rows of one column share their low address bits)
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift in (3, 10):
            c.append(("const %g Hz rows" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 6, {}))
            for r in (2, 8):
                c.append(("const %g Hz rows" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 6, dict(rows_r=r)))
            for waves, rows in ((5, 2), (4, 2), (2, 2), (4, 1), (8, 1), (8, 2), (5, 4), (3, 1)):
                c.append(("const %g Hz walk" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 5, dict(walk_waves=waves, walk_rows=rows)))
    if which == "rcomp":         # rows kernel: plan-time table against correctors evaluated per wavefront for 8 (4) rows
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        NEVER = 0xffffffff
        for shift in (3, 10, 25, 100, 9876.543, 5000):
            for pair in ("i16:i16", "f32:f32", "f32:i16", "i16:f32"):
                c.append(("const %g Hz rows table" % shift, lambda f, s=shift: const_segs(s), pair, 6, dict(rows_compute=NEVER)))
                c.append(("const %g Hz rows compute" % shift, lambda f, s=shift: const_segs(s), pair, 6, dict(rows_compute=1)))
                c.append(("const %g Hz rows compute" % shift, lambda f, s=shift: const_segs(s), pair, 6, dict(rows_compute=1, rows_r=4)))
    if which == "rthresh":       # from which period on evaluating beats the table (i16 -> i16)
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        NEVER = 0xffffffff
        for shift, rate in ((815000, 2400000), (5000, RATE), (2500, RATE), (9876.543, RATE), (250, RATE), (200, RATE), (160, RATE), (125, RATE), (100, RATE), (50, RATE), (3, RATE))[int(os.environ.get("AB_FROM", "0")):]:
            c.append(("const %g Hz table" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 6, dict(rows_compute=NEVER, _rate=rate)))
            c.append(("const %g Hz table" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 6, dict(rows_compute=NEVER, rows_r=4, _rate=rate)))
            c.append(("const %g Hz compute" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 6, dict(rows_compute=1, _rate=rate)))
            c.append(("const %g Hz default" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, dict(_rate=rate)))
            c.append(("const %g Hz default" % shift, lambda f, s=shift: const_segs(s), "f32:i16", 3, dict(_rate=rate)))
    if which == "rowlen":        # row length alone: the headline's period (1024) with L = mult x 1024, table, 2 rows per wavefront
        for mult in (4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 25, 32, 40, 48, 64, 128):
            c.append(("const 5000 Hz L=%d" % (mult * 1024), lambda f: const_segs(5000), "i16:i16", 6, dict(rows_mult=mult, rows_compute=0xffffffff)))
    if which == "rowlen2":       # row length near 8192 in steps of 32 samples (period 32: 32 kHz at 1.024 Msps)
        for L in (8192 - 1024, 8192 - 256, 8192 - 64, 8192 - 32, 8192, 8192 + 32, 8192 + 64, 8192 + 256, 8192 + 1024, 16384 - 32, 16384, 16384 + 32, 16384 + 2048, 24576, 32768, 2048, 4096):
            c.append(("const 32 kHz L=%d" % L, lambda f: const_segs(32000), "i16:i16", 6, dict(rows_mult=L // 32, rows_compute=0xffffffff)))
        for L in (8192, 16384, 10240):
            for r in (4, 8):
                c.append(("const 32 kHz L=%d" % L, lambda f: const_segs(32000), "i16:i16", 6, dict(rows_mult=L // 32, rows_r=r, rows_compute=0xffffffff)))
    if which == "waves":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for waves in (2, 3, 4, 5, 6, 8):
            o = dict(walk_waves=waves)
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, o))
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "f32:i16", 3, o))
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "f32:f32", 3, o))
            c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, o))
            c.append(("const 12345 Hz", lambda f: const_segs(12345), "i16:i16", 3, o))
            c.append(("const 5001 Hz", lambda f: const_segs(5001), "f32:f32", 3, o))
    if which == "merge":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift in (5001, 12345, 777):
            c.append(("const %d Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, {}))
        for opts in (dict(), dict(walk_waves=8), dict(walk_waves=5), dict(walk_waves=4), dict(walk_compute=0)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, opts))
        for rows in (3, 5, 9, 13):
            c.append(("synth %d rows of P=65536" % rows, lambda f, r=rows: synth_segs(r), "i16:i16", 3, dict()))
    if which == "t600":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, {}))
        for opts in (dict(), dict(walk_waves=8), dict(walk_waves=5)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, opts))
    if which == "f32":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for pair in ("f32:f32", "i16:f32", "f32:i16", "i16:i16"):
            c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, 3, {}))
            c.append(("const 100 Hz", lambda f: const_segs(100), pair, 5, {}))
            c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, {}))
    if which == "route":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift, rate in ((100, RATE), (9876.543, RATE), (815000, 2400000), (3, RATE), (5000, RATE), (2500, RATE), (1000, RATE), (15000, 256000)):
            for pair in ("i16:i16", "f32:f32"):
                for variant in (6, 5):
                    c.append(("const %g Hz @%d" % (shift, rate), lambda f, s=shift: const_segs(s), pair, variant, dict(_rate=rate)))
    if which == "final":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for opts in (dict(), dict(walk_waves=5), dict(walk_waves=8), dict(walk_compute=0)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, opts))
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "f32:i16", 3, opts))
        for rows in (2, 3, 4, 6, 8, 12):
            c.append(("synth %d rows of P=65536" % rows, lambda f, r=rows: synth_segs(r), "i16:i16", 3, dict()))
        for shift in (5001, 777):
            for opts in (dict(), dict(walk_waves=5), dict(walk_compute=0)):
                c.append(("const %d Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, opts))
    if which == "hybrid":
        for opts in (dict(walk_compute=1), dict(walk_compute=0), dict(), dict(walk_table_rows=12), dict(walk_table_rows=16), dict(walk_table_rows=32),
                     dict(walk_table_rows=48), dict(walk_waves=8), dict(walk_waves=6), dict(walk_waves=8, walk_table_rows=16)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, opts))
        for opts in (dict(walk_compute=1), dict(walk_compute=0), dict(), dict(walk_table_rows=16), dict(walk_waves=8), dict(walk_waves=6)):
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "f32:i16", 3, opts))
        for opts in (dict(walk_compute=1), dict()):
            c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, opts))
    if which == "shape":
        for waves, rows in ((5, 2), (8, 2), (8, 3), (10, 2), (12, 2), (16, 2), (6, 2), (10, 1), (16, 1)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_compute=1, walk_waves=waves, walk_rows=rows)))
        for waves, rows in ((5, 2), (8, 2), (12, 2)):
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "f32:i16", 3, dict(walk_compute=1, walk_waves=waves, walk_rows=rows)))
            c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, dict(walk_compute=0, walk_waves=waves, walk_rows=rows)))
    if which in ("walk", "all"):
        for comp in (0, 1):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_compute=comp)))
        for comp in (0, 1):
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "f32:i16", 3, dict(walk_compute=comp)))
        for waves, rows in ((5, 2), (5, 3), (4, 4), (6, 4), (8, 4), (8, 2)):
            for comp in (0, 1):
                c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_compute=comp, walk_waves=waves, walk_rows=rows)))
    if which in ("const", "all"):
        for shift in (5001, 9999, 777, 1234):
            for comp in (0, 1):
                c.append(("const %d Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, dict(walk_compute=comp)))
        for shift in (3, 100):
            c.append(("const %d Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, {}))
            for comp in (0, 1):
                c.append(("const %d Hz (walk forced)" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 5, dict(walk_compute=comp)))
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
    if which in ("persample", "all"):
        for shift in (3, 5001):
            for pair in ("i16:i16", "f32:f32"):
                for geom in ((128, 2), (256, 1)):
                    c.append(("const %d Hz, sincos per sample" % shift, lambda f, s=shift: const_segs(s), pair, 1, dict(_geom=geom)))
    if which == "span":          # round 3: span kernel (a workgroup keeps its window for up to walk_span rows) against the walk kernel
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for span in (1, 8, 16, 24, 32, 48, 64, 128):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_span=span)))
        for span in (1, 16, 32, 64, 128, 256):
            c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, dict(walk_span=span)))
        for pair in ("f32:i16", "i16:f32", "f32:f32"):
            for span in (1, 16, 32, 64):
                c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, dict(walk_span=span)))
        for shift in (3, 100):
            c.append(("const %d Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, {}))
            for span in (32, 64):
                c.append(("const %d Hz (walk forced)" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 5, dict(walk_span=span)))
    if which == "exp1":          # what the sincos arithmetic costs where (run against a -DDPX_EXP_NOSINCOS build with tools/ab_libs.sh)
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for span in (1, 8, 32):
            c.append(("track 600 s replay span=%d" % span, lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_span=span)))
        for span in (1, 32):
            c.append(("const 5001 Hz span=%d" % span, lambda f: const_segs(5001), "i16:i16", 3, dict(walk_span=span)))
        c.append(("const 3 Hz", lambda f: const_segs(3), "i16:i16", 3, {}))
        c.append(("const 5001 Hz, sincos per sample", lambda f: const_segs(5001), "i16:i16", 1, {}))
        for rows in (3, 5, 9, 25):
            c.append(("synth %d rows of P=65536" % rows, lambda f, r=rows: synth_segs(r, total=307200000), "i16:i16", 3, dict(walk_span=1)))
    if which == "uni":           # one-matrix launches: the matrix in the kernel arguments (walk_flags=0) against descriptors from memory (1)
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift in (5001, 1234, 7777.77):
            c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, dict(walk_span=1)))
            for span in (8, 10, 16):
                for fl in (0, 1):
                    c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, dict(walk_span=span, walk_flags=fl)))
        for waves, span in ((5, 10), (8, 16), (2, 4)):
            c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, dict(walk_span=span, walk_waves=waves)))
        for shift in (3, 100):
            c.append(("const %d Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, {}))
            for span in (8, 16):
                c.append(("const %d Hz (walk forced)" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 5, dict(walk_span=span)))
        for pair in ("f32:f32", "i16:f32", "f32:i16"):
            c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, 3, dict(walk_span=1)))
            c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, 3, dict(walk_span=8)))
    if which == "pack":          # how a second's 9.06 rows are packed into wavefronts (walk kernel shapes, span kernel)
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        def alt_segs(f, a=5001.0, b=5002.0, secs=262):
            return [(RATE, a if k % 2 == 0 else b) for k in range(secs)]
        for waves, rows in ((4, 2), (2, 4), (3, 3), (3, 4), (4, 3), (4, 4), (5, 2), (2, 3)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_span=1, walk_waves=waves, walk_rows=rows)))
            c.append(("alternating 5001/5002 Hz seconds", alt_segs, "i16:i16", 3, dict(walk_span=1, walk_waves=waves, walk_rows=rows)))
        for waves, span in ((4, 8), (2, 4), (5, 10)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_span=span, walk_waves=waves)))
            c.append(("alternating 5001/5002 Hz seconds", alt_segs, "i16:i16", 3, dict(walk_span=span, walk_waves=waves)))
        c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, dict(walk_span=1)))
    if which == "minl":          # const-mode walks: row length = the multiple of the period that reaches walk_flags >> 8 KiSamples
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift in (5001, 7777.77, 12345, 1234):
            for minl in (0, 64, 128, 256, 512, 1024, 2048):
                c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, dict(walk_span=8, walk_flags=minl << 8)))
        for shift in (100, 3):
            for minl in (0, 128, 512, 2048):
                c.append(("const %g Hz (walk forced)" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 5, dict(walk_span=8, walk_flags=minl << 8)))
    if which == "policy":        # the planner's row-length rule (rows of ~1 MB / one span per second) against rows of one period (target 8 Ki)
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        OLD = 8 << 8
        for o in (dict(), dict(walk_flags=OLD), dict(walk_span=1), dict(walk_waves=5, walk_span=10), dict(walk_waves=8, walk_span=16)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, o))
        for pair in ("f32:i16", "i16:f32", "f32:f32"):
            for o in (dict(), dict(walk_flags=OLD), dict(walk_span=1)):
                c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, o))
        for shift in (5001, 7777.77, 12345, 1234, 9999, 777, -5234.17):
            for o in (dict(), dict(walk_flags=OLD), dict(walk_span=1), dict(walk_waves=5, walk_span=10)):
                c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, o))
        for pair in ("f32:i16", "i16:f32", "f32:f32"):
            for o in (dict(), dict(walk_span=1)):
                c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, 3, o))
    if which == "reg1":
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for o in (dict(), dict(walk_flags=8 << 8), dict(walk_span=1), dict(walk_span=1, walk_flags=8 << 8), dict(walk_waves=5, walk_span=10), dict(walk_flags=1)):
            c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, o))
        for o in (dict(), dict(walk_span=1)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, o))
    if which == "span3":         # span height on the replay (rows per matrix are 6-12 under the row-length rule) and in const mode
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for span in (1, 8, 10, 12, 16, 32):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_span=span)))
        for span in (1, 8, 12, 16):
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "f32:i16", 3, dict(walk_span=span)))
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "i16:f32", 3, dict(walk_span=span)))
        for span in (1, 8, 12, 16):
            c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, dict(walk_span=span)))
    if which == "pairs":         # every format pair: walk kernel against span kernel shapes, const mode and replay
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for pair in ("i16:i16", "f32:i16", "i16:f32", "f32:f32"):
            for o in (dict(walk_span=1), dict(), dict(walk_waves=5, walk_span=10), dict(walk_waves=8, walk_span=16), dict(walk_waves=2, walk_span=4)):
                c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, 3, o))
            for o in (dict(walk_span=1), dict(), dict(walk_waves=5), dict(walk_waves=8), dict(walk_waves=5, walk_span=10)):
                c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, o))
    if which == "pairs2":        # f32-output pairs: fewer wavefronts per workgroup
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for pair in ("i16:f32", "f32:f32", "f32:i16"):
            for o in (dict(), dict(walk_waves=2, walk_span=4), dict(walk_waves=2, walk_span=8), dict(walk_waves=2, walk_span=6), dict(walk_waves=4, walk_span=4), dict(walk_waves=4, walk_span=6)):
                c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, 3, o))
            for o in (dict(), dict(walk_waves=2), dict(walk_waves=2, walk_span=4), dict(walk_waves=2, walk_span=6)):
                c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, o))
    if which == "rowrule":       # the scored row-length rule against round 3's first rule (target 256 Ki samples) and rows of one period
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift in (5001, 7777.77, 12345, 1234, 9999, 777):
            for o in (dict(), dict(walk_flags=256 << 8), dict(walk_flags=8 << 8)):
                c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 3, o))
        for pair in ("f32:i16", "i16:f32", "f32:f32"):
            for shift in (5001, 7777.77):
                for o in (dict(), dict(walk_flags=256 << 8)):
                    c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), pair, 3, o))
        for o in (dict(), dict(walk_flags=256 << 8), dict(walk_flags=8 << 8)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, o))
        for pair in ("f32:i16", "i16:f32", "f32:f32"):
            for o in (dict(), dict(walk_flags=256 << 8)):
                c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, o))
    if which == "pairs3":        # const mode, f32 output: fewer rows and wavefronts per workgroup
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for pair in ("i16:f32", "f32:f32"):
            for shift in (5001, 7777.77, 1234, 12345):
                for o in (dict(), dict(walk_waves=2, walk_span=4), dict(walk_waves=2, walk_span=6), dict(walk_waves=4, walk_span=4), dict(walk_waves=4, walk_span=6)):
                    c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), pair, 3, o))
    if which == "shapes":        # every (wavefronts, rows per span) shape of the span kernel, per format pair: const mode and replay
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for pair in ("i16:i16", "f32:i16", "i16:f32", "f32:f32"):
            c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, 3, {}))
            c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, {}))
            for waves in (2, 4, 5, 8):
                for span in (2, 4, 6, 8, 10, 12, 16):
                    if span > 4 * waves:
                        continue
                    c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, 3, dict(walk_waves=waves, walk_span=span)))
                    if span >= 4:
                        c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, dict(walk_waves=waves, walk_span=span)))
    if which == "pairs4":        # const mode, f32 -> i16 and i16 -> i16: few rows, many wavefronts
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for pair in ("f32:i16", "i16:i16"):
            for shift in (5001, 7777.77, 1234, 12345):
                for o in (dict(), dict(walk_waves=8, walk_span=4), dict(walk_waves=5, walk_span=4), dict(walk_waves=4, walk_span=4), dict(walk_waves=8, walk_span=8), dict(walk_waves=4, walk_span=8)):
                    c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), pair, 3, o))
    if which == "tshape":        # track-shaped plans, i16 -> i16: span heights (identical plans included: the spread of the method)
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for o in (dict(), dict(walk_waves=4, walk_span=12), dict(walk_span=8), dict(walk_span=10), dict(walk_span=16), dict(), dict(walk_span=1)):
            c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, o))
            c.append(("track 300 s replay", lambda f: track_segs(300, f), "i16:i16", 3, o))
    if which == "longp":         # periods of a million samples: rows kernel (correctors per wavefront) against span kernel shapes
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift in (3, 1, 10):
            for pair in ("i16:i16", "f32:f32"):
                c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), pair, 3, {}))
                for o in (dict(), dict(walk_waves=4, walk_span=4), dict(walk_waves=2, walk_span=4), dict(walk_waves=4, walk_span=6), dict(walk_flags=2048 << 8), dict(walk_waves=4, walk_span=12)):
                    c.append(("const %g Hz (walk forced)" % shift, lambda f, s=shift: const_segs(s), pair, 5, o))
    if which == "route2":        # page-aligned periods from 2592 to a million samples: rows kernel against span kernel
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift in (2, 3, 4, 5, 8, 20, 25, 40, 50, 100, 200, 9876.543):
            for pair in ("i16:i16", "f32:f32", "f32:i16"):
                c.append(("const %g Hz" % shift, lambda f, s=shift: const_segs(s), pair, 6, {}))
                c.append(("const %g Hz (walk forced)" % shift, lambda f, s=shift: const_segs(s), pair, 5, {}))
    if which == "sincos1":       # where the sincos arithmetic shows: per-sample path, replay, const walk (two builds: tools/ab_libs.sh)
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for shift in (3, 5001):
            c.append(("const %d Hz, sincos per sample" % shift, lambda f, s=shift: const_segs(s), "i16:i16", 1, {}))
            c.append(("const %d Hz, sincos per sample" % shift, lambda f, s=shift: const_segs(s), "f32:f32", 1, dict(_geom=(256, 1))))
        c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, {}))
        c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, {}))
        c.append(("const 3 Hz", lambda f: const_segs(3), "i16:i16", 3, {}))
        for rows in (3, 5):
            c.append(("synth %d rows of P=65536" % rows, lambda f, r=rows: synth_segs(r, total=307200000), "i16:i16", 3, {}))
    if which == "span2":         # span kernel: wavefronts per workgroup
        c.append(("const 5000 Hz (headline)", lambda f: const_segs(5000), "i16:i16", 3, {}))
        for waves in (2, 4, 5, 8):
            for span in (16, 32, 64):
                c.append(("track 600 s replay", lambda f: track_segs(600, f), "i16:i16", 3, dict(walk_span=span, walk_waves=waves)))
                c.append(("const 5001 Hz", lambda f: const_segs(5001), "i16:i16", 3, dict(walk_span=span, walk_waves=waves)))
    if which == "pairshape":     # replays with an f32 side: plans finalized for another (wavefronts, rows per span) than i16 -> i16's
        cand = {"f32:f32": ((2, 4), (2, 6), (4, 10), (4, 6)), "f32:i16": ((5, 6), (5, 8), (8, 10), (4, 4), (4, 6)), "i16:f32": ((4, 10), (4, 12), (4, 16), (2, 8))}
        for pair, shapes in cand.items():
            c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, dict(_geom=(0, 0))))
            for waves, span in shapes:
                c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, 3, dict(_geom=(0, 0), walk_waves=waves, walk_span=span)))
    if which == "w1":            # i16 -> f32 under one-wavefront span workgroups (needed a build with a WAVES = 1 instantiation: round 4, dropped)
        for name, fn in (("track 300 s replay", lambda f: track_segs(300, f)), ("const 5001 Hz", lambda f: const_segs(5001))):
            c.append((name, fn, "i16:f32", 3, dict(_geom=(0, 0))))
            for waves, span in ((1, 2), (1, 4), (1, 3), (2, 4), (2, 2)):
                c.append((name, fn, "i16:f32", 3, dict(_geom=(0, 0), walk_waves=waves, walk_span=span)))
    if which == "route3":        # pairs with an f32 side: the default plan (span kernel) against every corrector per sample (tile kernel)
        for pair in ("f32:i16", "i16:f32", "f32:f32"):
            for variant in (3, 1):
                c.append(("track 300 s replay", lambda f: track_segs(300, f), pair, variant, dict(_geom=(0, 0))))     # (0, 0): the library's own choice
                c.append(("const 5001 Hz", lambda f: const_segs(5001), pair, variant, dict(_geom=(0, 0))))
    if which == "persample4":    # the per-sample tile path, every format pair and tile geometry
        for pair in ("i16:i16", "f32:f32", "f32:i16", "i16:f32"):
            for geom in ((128, 2), (256, 1)) + (((64, 4),) if pair == "i16:i16" else ()):
                for shift in (3, 5001):
                    c.append(("const %d Hz, sincos per sample" % shift, lambda f, s=shift: const_segs(s), pair, 1, dict(_geom=geom)))
    if which == "geom":          # tile-kernel geometry per format pair: sincos per sample and tile tables
        for pair in ("i16:i16", "f32:f32", "f32:i16", "i16:f32"):
            for geom in ((128, 2), (256, 1), (128, 1), (256, 2)):
                c.append(("const 5001 Hz, sincos per sample", lambda f: const_segs(5001), pair, 1, dict(_geom=geom)))
                c.append(("const 5001 Hz, tile tables", lambda f: const_segs(5001), pair, 4, dict(_geom=geom)))
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--shuffle", action="store_true", help="time the cases in a different order every round")
    ap.add_argument("--set", default="all", choices=["all", "walk", "const", "persample", "synth", "size", "shape", "hybrid", "final", "route", "f32", "t600", "merge", "rowsopt", "waves", "bigp", "geom", "bigshape", "rcomp", "rthresh", "rowlen", "rowlen2", "synth2", "span", "span2", "exp1", "uni", "pack", "minl", "policy", "reg1", "span3", "pairs", "pairs2", "rowrule", "pairs3", "shapes", "pairs4", "tshape", "longp", "route2", "sincos1", "persample4", "route3", "pairshape", "w1"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = doppler_amd.Context(0)
    stream = torch.cuda.current_stream()
    bufs = {}
    built = []
    seg_cache = {}
    for name, mk, pair, variant, opts in cases(args.set):
        if args.only and args.only not in name:
            continue
        it, ot = pair.split(":")
        key = (name.split(" (")[0].split(",")[0].replace("5001 Hz n", "5001 Hz  n"), it)
        if key not in seg_cache:
            seg_cache[key] = mk(it)
        segs = seg_cache[key]
        n = sum(c for c, _ in segs)
        opts = dict(opts)
        # sets written for rounds 2-3 name options of the walk kernel, which round 4 removed: such a case is skipped
        gone = [k for k in opts if not k.startswith("_") and k not in dict(doppler_amd._lib.Options._fields_)]
        if gone or opts.get("walk_span") == 1:
            print("skipped (options of the removed walk kernel): %s %s" % (name, opts), file=sys.stderr)
            continue
        block, vecs = opts.pop("_geom", (128, 2))
        rate = opts.pop("_rate", RATE)
        ctx.set_tuning(block, vecs, variant)
        ctx.set_options(**opts)
        plan = ctx.plan_segments(segs, rate)
        lay = doppler_amd.plan_layout(segs, rate, 0, block, vecs, variant, options=opts)
        if (it, "in", n) not in bufs:
            bufs[(it, "in", n)] = (torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev) if it == "i16"
                                   else torch.rand(2 * n, dtype=torch.float32, device=dev) * 2 - 1)
        if (ot, "out", n) not in bufs:
            bufs[(ot, "out", n)] = torch.empty(n * BPS[ot], dtype=torch.uint8, device=dev)
        built.append(dict(name=name, pair=pair, variant=variant, opts=opts, geom=(block, vecs), plan=plan, n=n, it=it, ot=ot, lay=lay, ms=[]))
    ctx.set_tuning(-1, -1, 3)
    ctx.set_options()
    for b in built:        # warm-up
        x, o = bufs[(b["it"], "in", b["n"])], bufs[(b["ot"], "out", b["n"])]
        for _ in range(3):
            b["plan"].run(x.data_ptr(), b["it"], o.data_ptr(), b["ot"], stream.cuda_stream)
    stream.synchronize()
    import random
    rng = random.Random(12345)
    for _ in range(args.rounds):
        order = list(built)
        if args.shuffle:        # a case's figure depends a little on what ran just before it (clocks, power): vary the neighbours
            rng.shuffle(order)
        for b in order:
            x, o = bufs[(b["it"], "in", b["n"])], bufs[(b["ot"], "out", b["n"])]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.iters):
                b["plan"].run(x.data_ptr(), b["it"], o.data_ptr(), b["ot"], stream.cuda_stream)
            e1.record(stream)
            stream.synchronize()
            b["ms"].append(e0.elapsed_time(e1) / args.iters)
    ref = None
    for b in built:
        if "headline" in b["name"]:
            ref = statistics.median(b["ms"]) / b["n"]
    for b in built:
        med = statistics.median(b["ms"])
        alg = b["n"] * (BPS[b["it"]] + BPS[b["ot"]])
        gbs = alg / med / 1e6
        lay = b["lay"]
        kern = "walk" if lay["walk_launches"] else ("rows" if lay["rows_launches"] else "tile")
        print(json.dumps({"case": b["name"], "pair": b["pair"], "variant": b["variant"], "opts": b["opts"], "geom": b["geom"], "kernel": kern,
                          "ms_med": round(med, 4), "ms_min": round(min(b["ms"]), 4), "GBps": round(gbs, 1), "pct_peak": round(gbs / 80, 1),
                          "table_MiB": round(lay["table_entries"] * 8 / 2**20, 1), "single_samples": lay["single_samples"],
                          "vs_headline": round(ref / (med / b["n"]) * (BPS[b["it"]] + BPS[b["ot"]]) / 8, 4) if ref else None}), flush=True)


if __name__ == "__main__":
    main()
