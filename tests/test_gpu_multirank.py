"""BASELINE.json configs[3] and configs[4] on the HIP path, rank by rank, on the one GPU a test box has.

The multi-GPU design is "independent time chunks, counter seeded from the closed form, no data-path collective"
(the reference's whole carried state is one u32, src/main.rs:60 + src/dsp.rs:125-130).  What a
rank computes therefore depends on (its chunk, its seed, its segments) only, so running the chunks of all eight ranks
one after the other on device 0 exercises exactly the plans, seeds and kernels an 8-GPU run uses:

  * the PRODUCT side derives each chunk's seed with the closed form (doppler_amd.shard: chunk_seed / seed_for_segments)
    and each chunk's segments with segments_for_chunk;
  * the ORACLE side carries its own counter through the whole stream with the sequential rule and never sees a seed
    computed by the product; the two counters are compared at every chunk boundary;
  * every chunk's output bytes are compared with the oracle's, tolerance 0.

Chunk boundaries at GiB offsets, boundaries that split a constant-shift segment, and seeds that land mid-period
(odd periods) are all covered.  The world-size-2 variant with real processes and the gloo ordered gather, ranks calling
the kernel, is tests/test_gpu_sharding_gloo.py.
"""
import calendar
import os
import sys

import numpy as np
import pytest

from helpers import BPS, assert_same_bytes, make_iq

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THREADS = min(os.cpu_count() or 1, 64)


class DeviceBuffers:
    """One input and one output allocation reused by all chunks of a test."""

    def __init__(self, ctx, in_bytes, out_bytes):
        self.ctx = ctx
        self.d_in, self.d_out = ctx.malloc(max(16, in_bytes)), ctx.malloc(max(16, out_bytes))

    def run(self, x, intype, outtype, segments, rate, seed):
        ctx = self.ctx
        n = sum(c for c, _ in segments)
        assert x.size == n * BPS[intype]
        ctx.h2d(self.d_in, x)
        got = np.empty(n * BPS[outtype], dtype=np.uint8)
        plan = ctx.plan_segments(segments, rate, seed)
        try:
            plan.run(self.d_in, intype, self.d_out, outtype)
            ctx.synchronize()
            ctx.d2h(got, self.d_out)
            return got, plan.final_samplenum
        finally:
            plan.close()

    def close(self):
        self.ctx.free(self.d_in)
        self.ctx.free(self.d_out)


def same_in_slabs(got, want, what):
    assert got.size == want.size, what
    for a in range(0, got.size, 1 << 28):
        if not np.array_equal(got[a:a + (1 << 28)], want[a:a + (1 << 28)]):
            bad = np.flatnonzero(got[a:a + (1 << 28)] != want[a:a + (1 << 28)])
            raise AssertionError("%s: %d bytes differ in the slab at byte %d, first at byte %d" % (what, bad.size, a, a + int(bad[0])))


@pytest.mark.parametrize("shift", [5000, 5001])
def test_config3_8gib_const_stream_in_eight_rank_chunks(ctx, orc, shift):
    """configs[3]: `doppler const -i i16` on 8 GiB (2 147 483 648 samples) of synthetic IQ, time-chunk sharded over 8
    ranks = 8 x 1 GiB.  5000 Hz is the headline ratio (period 1024: every chunk starts on a period boundary); 5001 Hz
    has the odd period 113 027, so the seeds at r * 2^28 land mid-period and the chunks run on the span kernel."""
    from doppler_amd import shard
    world, n, rate = 8, 1 << 31, 1024000
    bufs = DeviceBuffers(ctx, (n // world) * 4, (n // world) * 4)
    sn_oracle = 0                       # the reference's samplenr (main.rs:60), carried by the oracle alone
    seeds = []
    try:
        for r in range(world):
            lo, hi = shard.chunk_bounds(n, world, r)
            assert (lo, hi) == (r << 28, (r + 1) << 28)
            seed = shard.chunk_seed(float(shift), rate, lo)          # closed form, product side
            assert seed == sn_oracle, "rank %d: closed-form seed %d, sequential counter %d" % (r, seed, sn_oracle)
            seeds.append(seed)
            rng = np.random.default_rng(1000 * shift + r)
            x = rng.integers(-23170, 23171, size=2 * (hi - lo), dtype=np.int16).view(np.uint8)
            got, fin = bufs.run(x, "i16", "i16", [(hi - lo, float(shift))], rate, seed)
            want, sn_oracle = orc.segments_stream(x, "i16", "i16", [(hi - lo, float(shift))], rate, samplenum=sn_oracle,
                                                  threads=THREADS)
            assert fin == sn_oracle, "rank %d: final counter" % r
            same_in_slabs(got, want, "configs[3] shift %d, rank %d" % (shift, r))
    finally:
        bufs.close()
    if shift == 5001:
        assert len(set(seeds[1:])) == world - 1      # every rank really starts at its own place in the period


def hour_track_segments(rate):
    sys.path.insert(0, ROOT)
    import bench
    return bench.track_segments(3600, rate, "f32", calendar.timegm((2015, 1, 22, 19, 23, 0)))    # --offset 5000


def test_config4_one_hour_track_f32_to_i16_in_eight_rank_chunks(ctx, orc):
    """configs[4]: `doppler track -i f32 -o i16`, one hour at 1.024 Msps (3 686 400 000 samples, 29.5 GB in, 14.7 GB out),
    3600 one-second shifts from the host SGP4 + the reference's per-block schedule with --offset 5000, sharded in 8 time
    chunks.  Each chunk: segments cut with segments_for_chunk (the boundaries fall one block before a shift change, so
    they split a segment), seed from seed_for_segments (closed form per segment, 0..3150 segments deep)."""
    from doppler_amd import shard
    import doppler_amd
    world, rate = 8, 1024000
    segs = hour_track_segments(rate)
    total = sum(c for c, _ in segs)
    assert total == 3600 * rate and 3500 <= len(segs) <= 3700
    per = total // world
    bufs = DeviceBuffers(ctx, per * 8, per * 4)
    sn_oracle = 0
    split = 0
    try:
        for r in range(world):
            lo, hi = shard.chunk_bounds(total, world, r, bytes_per_sample=8)
            assert hi - lo == per and lo % 1024 == 0
            before, inside = shard.segments_for_chunk(segs, lo, hi)
            assert sum(c for c, _ in before) == lo and sum(c for c, _ in inside) == hi - lo
            if before and inside and before[-1][1] == inside[0][1]:
                split += 1                                            # this boundary cuts a constant-shift segment
            seed = shard.seed_for_segments(before, rate)
            assert seed == sn_oracle, "rank %d: closed-form seed %d, sequential counter %d" % (r, seed, sn_oracle)
            assert doppler_amd.plan_layout(inside, rate, seed)["walk_launches"] == 1
            rng = np.random.default_rng(40 + r)
            x = rng.random(2 * per, dtype=np.float32)
            x *= 2.0
            x -= 1.0
            xb = x.view(np.uint8)
            got, fin = bufs.run(xb, "f32", "i16", inside, rate, seed)
            want, sn_oracle = orc.segments_stream(xb, "f32", "i16", inside, rate, samplenum=sn_oracle, threads=THREADS)
            assert fin == sn_oracle, "rank %d: final counter" % r
            same_in_slabs(got, want, "configs[4] rank %d" % r)
            del x, xb, got, want
    finally:
        bufs.close()
    assert split >= 6


@pytest.mark.parametrize("intype,outtype", [("f32", "i16"), ("i16", "f32"), ("i16", "i16")])
@pytest.mark.parametrize("world", [3, 7])
def test_uneven_worlds_split_segments_anywhere(ctx, orc, world, intype, outtype):
    """Worlds that do not divide the stream: chunk boundaries land anywhere in a second-long segment and anywhere in a
    period; a ragged final block goes to the last rank.  The chunks' concatenation equals BOTH the single-plan GPU output
    and the oracle's whole-stream output."""
    from doppler_amd import shard
    rate = 256000
    rng = np.random.default_rng(world)
    segs = []
    spb = 8192 // BPS[intype]
    for k in range(23):
        hz = float(np.float32(-5200.0 + 431.7 * k)) if k % 5 else float(np.float32(1000 * (k - 11)))
        segs.append(((rate + int(rng.integers(-20, 20)) * spb) // spb * spb, hz))
    segs[-1] = (segs[-1][0] + 321, segs[-1][1])                     # ragged final block
    n = sum(c for c, _ in segs)
    x = make_iq(intype, n, 60 + world, full_scale=True)
    want, sn_w = orc.segments_stream(x, intype, outtype, segs, rate, threads=THREADS)
    bufs = DeviceBuffers(ctx, n * BPS[intype], n * BPS[outtype])
    try:
        whole, fin_whole = bufs.run(x, intype, outtype, segs, rate, 0)
        assert fin_whole == sn_w
        assert_same_bytes(whole, want, outtype, "single plan")
        parts, sn_oracle, split = [], 0, 0
        for r in range(world):
            lo, hi = shard.chunk_bounds(n, world, r, bytes_per_sample=BPS[intype])
            before, inside = shard.segments_for_chunk(segs, lo, hi)
            split += bool(before and inside and before[-1][1] == inside[0][1])
            seed = shard.seed_for_segments(before, rate)
            assert seed == sn_oracle
            got, fin = bufs.run(x[lo * BPS[intype]:hi * BPS[intype]], intype, outtype, inside, rate, seed)
            sn_oracle = seed
            for c, hz in inside:
                sn_oracle = orc.advance_samplenum(sn_oracle, hz, rate, c)
            assert fin == sn_oracle
            parts.append(got)
        assert split >= world - 2
        assert_same_bytes(np.concatenate(parts), want, outtype, "%d chunks vs the oracle's whole stream" % world)
    finally:
        bufs.close()
