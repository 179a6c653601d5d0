"""GPU parity tests: the HIP path (always through the C ABI) against the CPU oracle and the committed
golden vectors, bit for bit.

Tolerances: north_star allows 1 ulp (f32) / +-1 LSB (i16).  Every comparison below is exact equality of
the output bytes (tolerance 0), except that any NaN matches any NaN for f32 outputs.
"""
import os

import numpy as np
import pytest

from helpers import BPS, assert_same_bytes, load_golden, make_iq, shift_block_cases

pytestmark = pytest.mark.gpu


def run_bulk(ctx, x, intype, outtype, segments, rate, sn0=0):
    """Device-resident bulk path: upload, plan, launch, download."""
    n = sum(c for c, _ in segments)
    assert x.size == n * BPS[intype]
    d_in, d_out = ctx.malloc(max(16, x.size)), ctx.malloc(max(16, n * BPS[outtype]))
    try:
        ctx.h2d(d_in, x)
        got = np.full(n * BPS[outtype], 0x5A, dtype=np.uint8)
        ctx.h2d(d_out, got)
        plan = ctx.plan_segments(segments, rate, sn0)
        plan.run(d_in, intype, d_out, outtype)
        ctx.synchronize()
        ctx.d2h(got, d_out)
        fin = plan.final_samplenum
        plan.close()
        return got, fin
    finally:
        ctx.free(d_in)
        ctx.free(d_out)


# ------------------------------------------------------------------ A7: ccexpf / sincos
def test_ccexpf_matches_glibc_bit_for_bit(ctx, orc):
    """complex.c:33-39 for imaginary arguments: device sincos == restated glibc == this host's libm."""
    from doppler_amd import dsp
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 1 << 32, size=1 << 22, dtype=np.uint64).astype(np.uint32)
    disc = np.array([0x418a3adb, 0x418a3adc, 0x418a3add, 0x418a3ade, 0x41bc76d9, 0x4202eb4b, 0x4255b0a9,
                     0x42687a55, 0x4280ce28, 0x42870e40, 0x42a35c07, 0x42a35d44, 0x42a97360, 0x42c55faa,
                     0x42cf5854, 0x42d8d23e, 0x42e87a55], dtype=np.uint32)
    special = np.array([0, 0x80000000, 1, 0x007fffff, 0x00800000, 0x39800000, 0x397fffff, 0x3f400000,
                        0x3f3fffff, 0x3f490fdb, 0x42f00000, 0x42efffff, 0x7f7fffff, 0x7f800000, 0xff800000,
                        0x7fc00000, 0x4b000000, 0x5f000000], dtype=np.uint32)
    # wavefronts whose lanes lie in both fast ranges and below 2^-12, none beyond 2^29 (sincosf_mixed: a counter wrapping inside a tile)
    mixed = np.empty(1 << 20, dtype=np.uint32)
    mixed[0::4] = rng.integers(0x39800000, 0x42f00000, size=1 << 18)
    mixed[1::4] = rng.integers(0x42f00000, 0x4e000000, size=1 << 18)
    mixed[2::4] = rng.integers(0x42f00000, 0x4e000000, size=1 << 18) | 0x80000000
    mixed[3::4] = rng.integers(0, 0x39800000, size=1 << 18)
    bits = np.concatenate([bits, disc, disc | 0x80000000, special, mixed])
    theta = bits.view(np.float32)
    z = np.zeros(theta.size, dtype=dsp.complex32)
    z["im"] = theta
    got = dsp.ccexpf(z, ctx=ctx)
    assert_same_bytes(got, orc.ccexpf_imag_array(theta, mode=1), "f32", "device vs restated glibc sincosf")
    if orc.libm_variant() == 1:
        # this host's libm through the reference's own ccexpf (oracle/_ref when built)
        assert_same_bytes(got, orc.ccexpf_imag_array(theta, mode=0), "f32", "device vs libm cexpf")
    ctx.set_libm_contraction(False)
    try:
        got0 = dsp.ccexpf(z, ctx=ctx)
    finally:
        ctx.set_libm_contraction(True)
    assert_same_bytes(got0, orc.ccexpf_imag_array(theta, mode=2), "f32", "device vs restated glibc (no fma)")


def test_ccexpf_strided_sweep_of_all_floats(ctx, orc):
    """Every 64th float bit pattern (2^26 arguments, all exponents, both signs)."""
    from doppler_amd import dsp
    for part in range(4):
        bits = (np.arange(1 << 24, dtype=np.uint64) * 64 + part * 16 + (np.uint64(part) << np.uint64(30))).astype(np.uint32)
        theta = bits.view(np.float32)
        z = np.zeros(theta.size, dtype=dsp.complex32)
        z["im"] = theta
        got = dsp.ccexpf(z, ctx=ctx)
        assert_same_bytes(got, orc.ccexpf_imag_array(theta, mode=1), "f32", "sweep part %d" % part)


def test_reference_known_answers_on_device(ctx, orc):
    """The reference's own test_cexpf (src/dsp.rs:57-83) against the device ccexpf, same vectors, same tolerance;
    and bit-exact against the golden values produced through the reference's complex.c."""
    from doppler_amd import dsp

    def rel_close(a, b, delta):
        return abs((float(a) - float(b)) / float(b)) < delta

    z = load_golden("reference_tests.npz")
    kin = np.zeros(4, dtype=dsp.complex32)
    kin["re"], kin["im"] = z["kat_in"][:, 0], z["kat_in"][:, 1]
    got = dsp.ccexpf(kin, ctx=ctx)
    assert got["re"][0] == 1.0 and got["im"][0] == 0.0
    assert rel_close(got["re"][1], 1.468694, 1e-6) and rel_close(got["im"][1], 2.2873552, 1e-6)
    assert rel_close(got["re"][2], 1593075600000000000000000000000.0, 1e-6)
    assert rel_close(got["im"][2], 1946674600000000000000000000000.0, 1e-6)
    assert got["re"][3] == np.inf and got["im"][3] == -np.inf
    assert got["re"].tobytes() == z["kat_out"][:, 0].astype(np.float32).tobytes()
    assert got["im"].tobytes() == z["kat_out"][:, 1].astype(np.float32).tobytes()


def test_general_ccexpf_matches_libm(ctx, orc):
    """ccexpf with a real part (outside the streaming path, but it is what complex.c:33-39 is): 2 M random pairs over
    all exponents, a grid of overflow / underflow / inf / nan corners, both libm builds."""
    from doppler_amd import dsp
    rng = np.random.default_rng(3)

    def rand_floats(n):
        e = rng.integers(0, 255, size=n).astype(np.uint32)
        m = rng.integers(0, 1 << 23, size=n).astype(np.uint32)
        s = rng.integers(0, 2, size=n).astype(np.uint32)
        return ((s << 31) | (e << 23) | m).view(np.float32)

    n = 2000000
    re, im = rand_floats(n), rand_floats(n)
    re[: n // 2] = rng.uniform(-110, 270, size=n // 2).astype(np.float32)
    im[: n // 4] = rng.uniform(-200, 200, size=n // 4).astype(np.float32)
    spec = np.array([0.0, -0.0, 1.0, -1.0, 88.0, 88.5, 88.72284, 89.0, 176.0, 176.5, 177.0, 264.0, 264.5, 265.0, 300.0, -103.0,
                     -103.5, -103.97, -104.0, -150.0, 1e-45, -1e-45, 1.17549435e-38, np.inf, -np.inf, np.nan, 3.4028235e38,
                     -3.4028235e38, 70.0, 1e6], dtype=np.float32)
    gr, gi = np.meshgrid(spec, spec)
    z = np.empty(n + gr.size, dtype=dsp.complex32)
    z["re"] = np.concatenate([re, gr.ravel()])
    z["im"] = np.concatenate([im, gi.ravel()])
    got = dsp.ccexpf(z, ctx=ctx)
    assert_same_bytes(got, orc.ccexpf_array(z, mode=1), "f32", "device ccexpf vs restated glibc cexpf")
    if orc.libm_variant() == 1:
        assert_same_bytes(got, orc.ccexpf_array(z, mode=0), "f32", "device ccexpf vs libm through complex.c")
    ctx.set_libm_contraction(False)
    try:
        got0 = dsp.ccexpf(z, ctx=ctx)
    finally:
        ctx.set_libm_contraction(True)
    assert_same_bytes(got0, orc.ccexpf_array(z, mode=2), "f32", "device ccexpf vs restated glibc cexpf (SSE2 builds)")


# ------------------------------------------------------------------ A1-A6 through the operator entry points
def test_golden_shift_block_cases(ctx):
    """Fused kernel vs the committed golden vectors (generated with the reference's complex.c linked)."""
    from doppler_amd import dsp
    n = 0
    for c in shift_block_cases():
        got, cnt, sn = dsp.shift_block(c["x"], c["intype"], c["outtype"], c["sn0"], c["shift"], c["rate"], ctx=ctx)
        assert cnt == c["n"] and sn == c["sn1"], c["key"]
        assert_same_bytes(got, c["y"], c["outtype"], c["key"])
        n += 1
    assert n > 200


SHIFTS = [(5000.0, 1024000), (-15000.0, 256000), (815000.0, 2400000), (0.0, 1024000), (3.0, 1024000),
          (9876.543, 1024000), (-5234.17, 1024000)]


@pytest.mark.parametrize("intype", ["i16", "f32"])
@pytest.mark.parametrize("outtype", ["i16", "f32"])
def test_shift_block_matches_oracle(ctx, orc, intype, outtype):
    """The fused kernel vs the oracle's three-pass restatement, all format pairs, carried counter."""
    from doppler_amd import dsp
    for si, (shift_hz, rate) in enumerate(SHIFTS):
        for n in [0, 1, 3, 4, 5, 2047, 2048, 2049, 3 * 2048 + 5, 40000, 150000]:
            for sn0 in [0, 1, 7]:
                x = make_iq(intype, n, 100 * si + n % 97 + sn0, full_scale=True)
                cx = orc.convert_iqi16_to_complex(x) if intype == "i16" else orc.convert_iqf32_to_complex(x)
                o, sn_w = orc.shift_frequency(cx, sn0, shift_hz, rate)
                want = orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)
                got, cnt, sn_g = dsp.shift_block(x, intype, outtype, sn0, shift_hz, rate, ctx=ctx)
                assert cnt == n
                assert sn_g == sn_w, (shift_hz, rate, n, sn0)
                assert_same_bytes(got, want, outtype, "shift=%r rate=%d n=%d sn0=%d" % (shift_hz, rate, n, sn0))


def test_operator_functions_match_reference_semantics(ctx, orc):
    """convert_* / shift_frequency / pack as separate operators (dsp.rs:85,101,117; main.rs:72-87)."""
    from doppler_amd import dsp
    x = make_iq("i16", 5000, 1, full_scale=True)
    assert_same_bytes(dsp.convert_iqi16_to_complex(x, ctx=ctx), orc.convert_iqi16_to_complex(x), "f32", "convert_iqi16")
    y = make_iq("f32", 3000, 2)
    b = dsp.convert_iqf32_to_complex(y, ctx=ctx)
    assert_same_bytes(b, orc.convert_iqf32_to_complex(y), "f32", "convert_iqf32")
    sn = sn_o = 0
    for _ in range(3):   # carried counter across calls, like main.rs:60
        g, sn = dsp.shift_frequency(b, sn, 815000.0, 2400000, ctx=ctx)
        w, sn_o = orc.shift_frequency(b, sn_o, 815000.0, 2400000)
        assert sn == sn_o
        assert_same_bytes(g, w, "f32", "shift_frequency")
    big = np.zeros(8, dtype=dsp.complex32)
    big["re"] = [0.5, 1.0, 1.5, -1.0, -1.5, np.nan, np.inf, -np.inf]
    big["im"] = [-0.5, 1.00002, 40000.0, -1.00002, 1e30, 1e-30, -0.0, 3e38]
    assert_same_bytes(dsp.pack_iqi16(big, ctx=ctx), orc.pack_i16(big), "i16", "pack saturation / NaN")
    with pytest.raises(dsp.DspError) as e:      # the reference panics: dsp.rs:87
        dsp.convert_iqi16_to_complex(x[:-1], ctx=ctx)
    assert e.value.code == -2
    with pytest.raises(dsp.DspError):           # dsp.rs:103
        dsp.shift_block(y[:-3], "f32", "f32", 0, 1.0, 1000, ctx=ctx)
    # empty input is legal (the reference's last, empty read)
    o, cnt, sn2 = dsp.shift_block(np.zeros(0, np.uint8), "i16", "f32", 5, 1.0, 1000, ctx=ctx)
    assert o.size == 0 and cnt == 0 and sn2 == 5


def test_shift_blocks_batches_the_reference_loop(ctx, orc):
    """dpx_shift_blocks: N reference blocks per call with one shift per block (what the track loop of main.rs:160-183
    produces), counter carried through the blocks and across calls — equal to the golden track replay block by block,
    for batch sizes 1, 7 and 64, and to N single-block calls."""
    from doppler_amd import dsp
    t = load_golden("track_stream_case.npz")
    rate, _, _, sn_final = t["meta"]
    x, log = t["x"], t["shift_log"]
    nblocks = (x.size + 8191) // 8192
    for batch in (1, 7, 64):
        outs, sn = [], 0
        for b0 in range(0, nblocks, batch):
            b1 = min(nblocks, b0 + batch)
            o, cnt, sn = dsp.shift_blocks(x[b0 * 8192:b1 * 8192], "i16", "i16", sn, log[b0:b1], int(rate), ctx=ctx)
            outs.append(o)
        assert sn == int(sn_final)
        assert_same_bytes(np.concatenate(outs), t["y"], "i16", "shift_blocks, %d blocks per call" % batch)
    with pytest.raises(dsp.DspError):
        dsp.shift_blocks(x[: 8192 * 3], "i16", "i16", 0, log[:2], int(rate), ctx=ctx)       # 3 blocks, 2 shifts
    o, cnt, sn = dsp.shift_blocks(np.zeros(0, np.uint8), "f32", "i16", 9, np.zeros(0, np.float32), 1000, ctx=ctx)
    assert o.size == 0 and cnt == 0 and sn == 9


def test_reference_bench_configuration_on_device(ctx):
    """src/dsp.rs:136-157: 1 000 000 bytes of 0xAA as f32 IQ, 815 kHz at 2.4 Msps, counter carried over 301 calls."""
    from doppler_amd import dsp
    z = load_golden("reference_tests.npz")
    cx = dsp.convert_iqf32_to_complex(np.full(1000000, 0xAA, dtype=np.uint8), ctx=ctx)
    sn = 0
    digest = []
    for it in range(301):
        o, sn = dsp.shift_frequency(cx, sn, 815000.0, 2400000, ctx=ctx)
        if it in (0, 1, 150, 300):
            digest.append(int(o.view(np.uint32).astype(np.uint64).sum()))
    assert digest == [int(d) for d in z["bench_digest"]]
    assert sn == int(z["bench_final_samplenum"][0])


# ------------------------------------------------------------------ bulk path
def test_golden_streams_bulk(ctx):
    z = load_golden("const_stream_cases.npz")
    for k in range(4):
        shift, rate, sn, it, ot = z["s%d_meta" % k]
        fi, fo = ("i16", "f32")[int(it)], ("i16", "f32")[int(ot)]
        x = z["s%d_in" % k]
        got, fin = run_bulk(ctx, x, fi, fo, [(x.size // BPS[fi], float(shift))], int(rate))
        assert fin == int(sn)
        assert_same_bytes(got, z["s%d_out" % k], fo, "golden stream %d" % k)
    # track replay: one plan segment per reference block, shifts from the golden schedule
    t = load_golden("track_stream_case.npz")
    rate, freq, off, sn = t["meta"]
    x = t["x"]
    n = x.size // 4
    segs = []
    for b, hz in enumerate(t["shift_log"]):
        cnt = min(2048, n - b * 2048)
        if cnt > 0:
            segs.append((cnt, float(hz)))
    got, fin = run_bulk(ctx, x, "i16", "i16", segs, int(rate))
    assert fin == int(sn)
    assert_same_bytes(got, t["y"], "i16", "golden track stream")


@pytest.mark.parametrize("intype,outtype", [("i16", "i16"), ("f32", "f32"), ("i16", "f32"), ("f32", "i16")])
def test_const_stream_bulk_vs_oracle(ctx, orc, intype, outtype):
    """`doppler const` over 4 Mi samples, device-resident bulk path, every kernel variant."""
    n = (1 << 22) + 8192 // BPS[intype] * 3 + 1234 * (BPS[intype] // 4)
    x = make_iq(intype, n, 11, full_scale=True)
    want, sn_w = orc.const_stream(x, intype, outtype, 5000, 1024000, threads=8)
    sn_w = orc.advance_samplenum(0, 5000.0, 1024000, n)
    try:
        for variant, block, vecs in [(3, 256, 1), (4, 256, 1), (1, 256, 1), (2, 128, 2), (4, 128, 2), (1, 128, 2), (1, 64, 4), (4, 64, 4)]:
            ctx.set_tuning(block, vecs, variant)
            got, fin = run_bulk(ctx, x, intype, outtype, [(n, 5000.0)], 1024000)
            assert fin == sn_w
            assert_same_bytes(got, want, outtype, "variant=%d block=%d vecs=%d" % (variant, block, vecs))
    finally:
        ctx.set_tuning(-1, -1, 3)     # back to the geometry chosen per launch


@pytest.mark.parametrize("shift,rate", [(815000.0, 2400000), (9876.543, 1024000), (-5234.17, 1024000), (3.0, 1024000),
                                        (0.0, 1024000), (-15000.0, 256000)])
def test_periods_and_large_angles_bulk(ctx, orc, shift, rate):
    """Periods that are not powers of two, a period longer than any table, |theta| >= 120 (the
    32x96-bit reduction range of sincosf), shift 0."""
    n = 700000 + 13
    for intype, outtype in (("i16", "i16"), ("f32", "f32")):
        x = make_iq(intype, n, 21)
        cx = orc.convert_iqi16_to_complex(x) if intype == "i16" else orc.convert_iqf32_to_complex(x)
        o, sn_w = orc.shift_frequency(cx, 0, shift, rate)
        want = orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)
        got, fin = run_bulk(ctx, x, intype, outtype, [(n, shift)], rate)
        assert fin == sn_w
        assert_same_bytes(got, want, outtype, "shift=%r %s->%s" % (shift, intype, outtype))


@pytest.mark.parametrize("shift,rate", [(5001.0, 1024000), (3.0, 1024000), (-7777.77, 1024000), (1.0e9, 1000), (0.001, 1024000)])
def test_sincos_per_sample_paths_known_per_tile(ctx, orc, shift, rate):
    """sincos per sample (variant 1): a tile that does not wrap takes sincosf's argument path from the stretch's three
    counters (first n with |theta| >= 2^-12 / 120 / 2^33, bisected by the planner) instead of looking at its angles.
    The shifts put those counters inside the stream (5001 Hz: n = 4 and 3910), beyond it (0.001 Hz: never plain), or
    make |theta| pass 2^33 (1e9 Hz at 1 ksps); every format pair, with the geometry chosen per launch."""
    n = 300000 + 7
    try:
        ctx.set_tuning(-1, -1, 1)
        for intype, outtype in (("i16", "i16"), ("f32", "f32"), ("i16", "f32"), ("f32", "i16")):
            x = make_iq(intype, n, 77)
            cx = orc.convert_iqi16_to_complex(x) if intype == "i16" else orc.convert_iqf32_to_complex(x)
            o, sn_w = orc.shift_frequency(cx, 0, shift, rate)
            want = orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)
            got, fin = run_bulk(ctx, x, intype, outtype, [(n, shift)], rate)
            assert fin == sn_w
            assert_same_bytes(got, want, outtype, "per sample, shift=%r %s->%s" % (shift, intype, outtype))
    finally:
        ctx.set_tuning(-1, -1, 3)


def test_track_segments_bulk_vs_oracle(ctx, orc):
    """Track replay (main.rs:156-184) at 256 ksps for 12 s with an overpass-shaped range rate: the schedule
    comes from the oracle's log; the counter is carried across every shift change."""
    rate, freq, off = 256000, 437505000, 2500
    t = np.arange(16, dtype=np.float64)
    rr = 6.8 * np.tanh((t - 6.0) / 2.0)
    n = rate * 12 + 2048 * 2 + 321
    for intype, outtype in (("i16", "i16"), ("f32", "i16")):
        x = make_iq(intype, n, 31)
        want, sn_w, log = orc.track_stream(x, intype, outtype, rate, freq, rr, offset_hz=off)
        spb = 8192 // BPS[intype]
        segs = []
        for b, hz in enumerate(log):
            cnt = min(spb, n - b * spb)
            if cnt > 0:
                if segs and segs[-1][1] == float(hz):
                    segs[-1] = (segs[-1][0] + cnt, float(hz))     # merge equal neighbours: same arithmetic
                else:
                    segs.append((cnt, float(hz)))
        assert 10 <= len(segs) <= 20
        got, fin = run_bulk(ctx, x, intype, outtype, segs, rate)
        assert fin == sn_w
        assert_same_bytes(got, want, outtype, "track %s->%s" % (intype, outtype))


def oracle_segments(orc, x, intype, outtype, segs, rate, sn0=0):
    cx = orc.convert_iqi16_to_complex(x) if intype == "i16" else orc.convert_iqf32_to_complex(x)
    outs, sn, pos = [], sn0, 0
    for cnt, hz in segs:
        o, sn = orc.shift_frequency(cx[pos:pos + cnt], sn, hz, rate)
        outs.append(o)
        pos += cnt
    o = np.concatenate(outs)
    return (orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)), sn


@pytest.mark.parametrize("intype,outtype", [("i16", "i16"), ("f32", "f32"), ("i16", "f32"), ("f32", "i16")])
def test_span_kernel_plans_vs_oracle(ctx, orc, intype, outtype):
    """Track-shaped plans with more than eight periodic stretches run as ONE span-kernel launch (matrices with
    shifted, line-aligned rows; leftover blocks for heads, tails and lead-ins) plus tile launches for long
    untabulated gaps — byte for byte against the oracle, odd periods, short periods, carried counters."""
    import doppler_amd
    rng = np.random.default_rng(77)
    plans = [
        ([(120000 + 2048 * int(rng.integers(0, 9)), float(np.float32(rng.uniform(-9000, 9000)))) for _ in range(12)], 256000, 0),
        ([(30000, 1000.0 * (2 * k + 1)) for k in range(10)], 1024000, 0),
        ([(50000 + 17 * k, 333.0 + k) for k in range(10)], 48000, 5),
    ]
    for i, (segs, rate, sn0) in enumerate(plans):
        lay = doppler_amd.plan_layout(segs, rate, sn0)
        assert lay["walk_launches"] == 1 and lay["walk_matrices"] >= 8, lay
        n = sum(c for c, _ in segs)
        x = make_iq(intype, n, 900 + i, full_scale=True)
        want, sn = oracle_segments(orc, x, intype, outtype, segs, rate, sn0)
        got, fin = run_bulk(ctx, x, intype, outtype, segs, rate, sn0=sn0)
        assert fin == sn
        assert_same_bytes(got, want, outtype, "walk plan %d %s->%s" % (i, intype, outtype))
    # the same arithmetic when a single long stretch is forced onto the span kernel (a one-matrix launch)
    segs, rate = [(3000000 + 77, 5001.0)], 1024000
    x = make_iq(intype, segs[0][0], 950)
    want, sn = oracle_segments(orc, x, intype, outtype, segs, rate)
    ctx.set_tuning(0, 0, 5)
    try:
        assert doppler_amd.plan_layout(segs, rate, variant=5)["walk_matrices"] == 1
        got, fin = run_bulk(ctx, x, intype, outtype, segs, rate)
    finally:
        ctx.set_tuning(0, 0, 3)
    assert fin == sn
    assert_same_bytes(got, want, outtype, "forced walk %s->%s" % (intype, outtype))


@pytest.mark.parametrize("waves", [2, 4, 5, 8])
@pytest.mark.parametrize("span", [0, 2, 3, 7, 9, 16, 33, 4096])
def test_span_kernel_shapes(ctx, orc, waves, span):
    """The span kernel (round 3: a workgroup keeps its column window for up to `span` rows of a matrix, two rows per
    wavefront per turn): every workgroup size, spans from a single turn (2, 3 rows: wavefronts without rows still take
    part in the slice and the barrier) to whole matrices (4096), heights that leave the last turn with one row or with
    idle wavefronts, odd periods (every row shifted differently against the slice), all format pairs."""
    import doppler_amd
    rate = 256000
    segs = [(rate + 2048 * k, float(np.float32(-3000.0 + 517.3 * k))) for k in range(9)] + [(5 * rate // 2, 1000.0), (3 * rate, 7.0), (rate // 3, 1234.5)]
    n = sum(c for c, _ in segs)
    opts = dict(walk_waves=waves, walk_span=span)
    lay = doppler_amd.plan_layout(segs, rate, variant=5, options=opts)
    assert lay["walk_launches"] == 1 and lay["walk_matrices"] >= 9 and lay["rows_launches"] == 0 and lay["table_entries"] == 0, lay
    ctx.set_tuning(0, 0, 5)
    ctx.set_options(**opts)
    try:
        for intype, outtype in (("i16", "i16"), ("f32", "i16"), ("i16", "f32"), ("f32", "f32")):
            x = make_iq(intype, n, 1700 + waves, full_scale=True)
            want, sn = orc.segments_stream(x, intype, outtype, segs, rate, threads=16)
            got, fin = run_bulk(ctx, x, intype, outtype, segs, rate)
            assert fin == sn
            assert_same_bytes(got, want, outtype, "span kernel, %d waves, span %d, %s->%s" % (waves, span, intype, outtype))
    finally:
        ctx.set_options()
        ctx.set_tuning(0, 0, 3)


@pytest.mark.parametrize("waves", [0, 2, 5, 8])
def test_span_kernel_short_and_medium_matrices(ctx, orc, waves):
    """Round 4: spans of up to 4 rows give their workgroups 2 (up to 2 rows: 4) adjacent windows, WAVES / 2 (/ 4) wavefronts
    each and one contiguous slice; whole matrices of 9-12 rows give their first wavefronts a second turn.  Matrices of 2..13
    rows of one period (8192 samples at 262 144 Hz, and an odd period at 256 000 Hz: every row shifted differently against
    the slice), ragged last rows, heads, tails and lead-ins between them; the planner's own shape (waves = 0: 4 wavefronts,
    8 at launch for f32 -> i16) and the others a caller may name; all format pairs, byte for byte."""
    import doppler_amd
    plans = [
        ([((2 + (5 * k) % 12) * 8192 + 1000 * (k % 3), 32.0 * (2 * k + 1)) for k in range(12)], 262144),
        ([(int(2.2 * 256000 / (1 + k % 7)) + 2048 * k, float(np.float32(-3000.0 + 517.3 * k))) for k in range(16)], 256000),
    ]
    opts = dict(walk_waves=waves) if waves else {}
    ctx.set_options(**opts)
    try:
        for pi, (segs, rate) in enumerate(plans):
            lay = doppler_amd.plan_layout(segs, rate, options=opts)
            assert lay["walk_launches"] == 1 and lay["walk_matrices"] >= 6 and lay["rows_launches"] == 0, lay   # (a lead-in may leave a stretch under two rows)
            n = sum(c for c, _ in segs)
            for intype, outtype in (("i16", "i16"), ("f32", "i16"), ("i16", "f32"), ("f32", "f32")):
                x = make_iq(intype, n, 2300 + waves + pi, full_scale=True)
                want, sn = orc.segments_stream(x, intype, outtype, segs, rate, threads=16)
                got, fin = run_bulk(ctx, x, intype, outtype, segs, rate)
                assert fin == sn
                assert_same_bytes(got, want, outtype, "short/medium matrices, plan %d, %d waves, %s->%s" % (pi, waves, intype, outtype))
    finally:
        ctx.set_options()


@pytest.mark.parametrize("span,flags", [(0, 0), (0, 1), (3, 0), (40, 0), (4096, 0)])
def test_span_kernel_one_matrix_launch(ctx, orc, span, flags):
    """Const mode on the span kernel: a launch of ONE matrix takes the matrix from its kernel arguments and the span from
    blockIdx.y (walk_flags=1: from descriptors in memory, like a track-shaped plan); odd period, a period with rows of whole
    lines, a counter carried in, head and tail as leftover blocks in the last grid rows; all format pairs."""
    import doppler_amd
    # ... and periods shorter than a slice (480, 256) down to 4, 2 and 1 (shift 0): the slice index wraps many times
    cases = [((5001.0, 1024000), 0), ((777.0, 1024000), 12345), ((100.0, 1024000), 7), ((815000.0, 2400000), 77),
             ((-15000.0, 256000), 0), ((64000.0, 256000), 3), ((128000.0, 256000), 1), ((0.0, 48000), 0)]
    n = (1 << 22) + 4321
    opts = dict(walk_span=span, walk_flags=flags)
    ctx.set_tuning(0, 0, 5)
    ctx.set_options(**opts)
    try:
        for (shift, rate), sn0 in cases:
            lay = doppler_amd.plan_layout([(n, shift)], rate, sn0, variant=5, options=opts)
            assert lay["walk_launches"] == 1 and lay["walk_matrices"] == 1 and lay["rows_launches"] == 0, lay
            for intype, outtype in (("i16", "i16"), ("f32", "i16"), ("i16", "f32"), ("f32", "f32")):
                x = make_iq(intype, n, 5100 + sn0, full_scale=True)
                cx = orc.convert_iqi16_to_complex(x) if intype == "i16" else orc.convert_iqf32_to_complex(x)
                o, sn_w = orc.shift_frequency(cx, sn0, shift, rate)
                want = orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)
                got, fin = run_bulk(ctx, x, intype, outtype, [(n, shift)], rate, sn0)
                assert fin == sn_w
                assert_same_bytes(got, want, outtype, "one-matrix span launch, span=%d flags=%d shift=%r %s->%s" % (span, flags, shift, intype, outtype))
    finally:
        ctx.set_options()
        ctx.set_tuning(0, 0, 3)


@pytest.mark.parametrize("sub_lg", [12, 16, 19])
def test_span_launches_dealt_out_as_sub_launches(ctx, orc, sub_lg):
    """Round 5: a span launch over a long stream is dealt out as sub-launches of about 2^28 samples, back to back on the stream
    (csrc/dpx_kernels.hip, span_t: workgroups find their work from (offset + blockIdx)).  dpx_options.sub_lg makes the pieces
    small enough to exercise the cut on test-sized streams — hundreds of sub-launches, cuts inside spans' padding, between
    spans and leftover groups, and in the leftover rows of a one-matrix launch — for every format pair, descriptor-driven
    (track-shaped) and one-matrix (const mode) launches: the bytes are those of the oracle, i.e. of the uncut launch."""
    import doppler_amd
    rng = np.random.default_rng(77)                   # the plan of test_span_kernel_plans_vs_oracle: twelve matrices
    track = ([(120000 + 2048 * int(rng.integers(0, 9)), float(np.float32(rng.uniform(-9000, 9000)))) for _ in range(12)], 256000, 0)
    const = ([((1 << 21) + 4321, 5001.0)], 1024000, 12345)
    ctx.set_options(sub_lg=sub_lg)
    try:
        for variant, (segs, rate, sn0) in ((3, track), (5, const)):
            ctx.set_tuning(0, 0, variant)
            lay = doppler_amd.plan_layout(segs, rate, sn0, variant=variant)
            assert lay["walk_launches"] == 1 and (lay["walk_matrices"] >= 8 if variant == 3 else lay["walk_matrices"] == 1), lay
            n = sum(c for c, _ in segs)
            assert n >> sub_lg >= 2                       # really more than one piece
            for intype, outtype in (("i16", "i16"), ("f32", "i16"), ("i16", "f32"), ("f32", "f32")):
                if variant == 3 and intype != outtype:
                    continue                              # (the mixed pairs of a many-matrix plan run on the tile kernel)
                x = make_iq(intype, n, 7000 + sub_lg, full_scale=True)
                want, sn = oracle_segments(orc, x, intype, outtype, segs, rate, sn0)
                got, fin = run_bulk(ctx, x, intype, outtype, segs, rate, sn0=sn0)
                assert fin == sn
                assert_same_bytes(got, want, outtype, "sub-launches of 2^%d, variant %d, %s->%s" % (sub_lg, variant, intype, outtype))
    finally:
        ctx.set_options()
        ctx.set_tuning(0, 0, 3)


@pytest.mark.parametrize("rows_compute,rows_r", [(1, 0), (1, 4), (0, 0), (0xffffffff, 0)])
def test_rows_kernel_evaluates_its_correctors(ctx, orc, rows_compute, rows_r):
    """Rows launches that leave their table alone: every wavefront evaluates the correctors of its columns for its 4 (8)
    rows.  rows_compute = 1 forces that for every format pair, 0 is the planner's rule (i16 -> i16 and periods that do not
    give 8192- or 16384-sample rows), 0xffffffff never.  Periods: 10240 (rows of one period), 2592 and 480 (rows of several
    periods: the column index wraps inside a row), 6400; a stream that starts mid-period (the counter carried in)."""
    import doppler_amd
    cases = [((100.0, 1024000), 0), ((9876.543, 1024000), 0), ((815000.0, 2400000), 77), ((160.0, 1024000), 3001)]
    n = (1 << 21) + 4321
    opts = dict(rows_compute=rows_compute, rows_r=rows_r)
    ctx.set_tuning(0, 0, 6)
    ctx.set_options(**opts)
    try:
        for (shift, rate), sn0 in cases:
            lay = doppler_amd.plan_layout([(n, shift)], rate, sn0, variant=6, options=opts)
            assert lay["rows_launches"] == 1 and lay["walk_launches"] == 0, lay
            for intype, outtype in (("i16", "i16"), ("f32", "i16"), ("i16", "f32"), ("f32", "f32")):
                x = make_iq(intype, n, 4100 + sn0, full_scale=True)
                cx = orc.convert_iqi16_to_complex(x) if intype == "i16" else orc.convert_iqf32_to_complex(x)
                o, sn_w = orc.shift_frequency(cx, sn0, shift, rate)
                want = orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)
                got, fin = run_bulk(ctx, x, intype, outtype, [(n, shift)], rate, sn0)
                assert fin == sn_w
                assert_same_bytes(got, want, outtype, "rows, rows_compute=%d R=%d shift=%r %s->%s" % (rows_compute, rows_r, shift, intype, outtype))
    finally:
        ctx.set_options()
        ctx.set_tuning(0, 0, 3)


def test_span_kernel_random_plans(ctx, orc):
    """Seeded random track-shaped plans: 9-30 segments of 0.05-1.3 s with arbitrary f32 shifts, random rate,
    format pair and counter start — whatever mixture of walk matrices, leftover ranges and tile launches the
    planner picks (auto, or forced onto the span kernel) must reproduce the oracle byte for byte."""
    import doppler_amd
    rng = np.random.default_rng(4242)
    used_walk = 0
    for case in range(16):
        rate = int(rng.choice([48000, 96000, 256000, 300000]))
        segs = []
        for _ in range(int(rng.integers(9, 31))):
            kind = rng.integers(0, 3)
            hz = (float(np.float32(rng.uniform(-11000, 11000))) if kind == 0 else
                  float(np.float32(rng.integers(-400, 400) * 25)) if kind == 1 else float(np.float32(rng.uniform(-40, 40))))
            segs.append((int(rate * rng.uniform(0.05, 1.3)) // 2048 * 2048 + (0 if rng.random() < 0.8 else int(rng.integers(1, 2048))), hz))
        segs = [(c, h) for c, h in segs if c > 0]
        intype, outtype = [("i16", "i16"), ("f32", "i16"), ("i16", "f32"), ("f32", "f32")][case % 4]
        sn0 = int(rng.choice([0, 1, 12345]))
        variant = 5 if case % 3 == 0 else 3
        used_walk += doppler_amd.plan_layout(segs, rate, sn0, variant=variant)["walk_launches"]
        n = sum(c for c, _ in segs)
        x = make_iq(intype, n, 7000 + case, full_scale=(case % 2 == 0))
        want, sn = oracle_segments(orc, x, intype, outtype, segs, rate, sn0)
        ctx.set_tuning(0, 0, variant)
        try:
            got, fin = run_bulk(ctx, x, intype, outtype, segs, rate, sn0=sn0)
        finally:
            ctx.set_tuning(0, 0, 3)
        assert fin == sn, (case, segs[:3])
        assert_same_bytes(got, want, outtype, "random walk plan %d (%d segments, %s->%s)" % (case, len(segs), intype, outtype))
    assert used_walk >= 5      # many of these plans really run on the span kernel


@pytest.mark.parametrize("variant", [1, 4])
def test_tile_kernel_random_plans(ctx, orc, variant):
    """The tile kernel alone (variant 1: every corrector evaluated per sample; 4: tile tables where they fit) on seeded
    random plans of 3-12 segments whose shifts give periods below a tile (the per-entry modulo), periods that wrap inside
    tiles, million-sample periods and stretches that never reset; every format pair and every tile shape built for it.
    Which four samples a lane takes depends on the pair (four consecutive ones, two pairs half a block apart, two pairs
    inside the wavefront's 256 samples with an exchange through LDS): each must reproduce the oracle byte for byte,
    ragged segment boundaries and stream tail included."""
    rng = np.random.default_rng(977 + variant)
    shapes = {("i16", "i16"): [(0, 0), (64, 4), (128, 2), (256, 1)], ("f32", "i16"): [(0, 0), (128, 2), (256, 1)],
              ("i16", "f32"): [(0, 0), (128, 2), (256, 1)], ("f32", "f32"): [(0, 0), (128, 2), (256, 1)]}
    try:
        for case in range(12):
            rate = int(rng.choice([48000, 256000, 1024000]))
            segs = []
            for _ in range(int(rng.integers(3, 13))):
                kind = int(rng.integers(0, 5))
                hz = (float(np.float32(rng.uniform(-11000, 11000))) if kind == 0 else          # arbitrary: period anything
                      float(np.float32(rate / int(rng.choice([2, 3, 5, 64, 500, 1000])))) if kind == 1 else   # short periods
                      float(np.float32(rng.integers(1, 40))) if kind == 2 else                  # periods of rate / k samples
                      float(np.float32(rng.uniform(1e-4, 1e-2))) if kind == 3 else              # never resets within the segment
                      0.0)
                cnt = int(rng.integers(1, 40)) * 1024 + (0 if rng.random() < 0.5 else int(rng.integers(1, 1024)))
                segs.append((cnt, hz))
            intype, outtype = [("i16", "i16"), ("f32", "i16"), ("i16", "f32"), ("f32", "f32")][case % 4]
            sn0 = int(rng.choice([0, 1, 777, (1 << 24) - 5000]))
            n = sum(c for c, _ in segs)
            x = make_iq(intype, n, 8100 + case, full_scale=(case % 2 == 0))
            want, sn = oracle_segments(orc, x, intype, outtype, segs, rate, sn0)
            for block, vecs in shapes[(intype, outtype)]:
                ctx.set_tuning(block, vecs, variant)
                got, fin = run_bulk(ctx, x, intype, outtype, segs, rate, sn0=sn0)
                assert fin == sn, (case, segs[:3])
                assert_same_bytes(got, want, outtype, "tile plan %d variant %d %dx%d (%d segments, %s->%s)" % (case, variant, block, vecs, len(segs), intype, outtype))
    finally:
        ctx.set_tuning(-1, -1, 3)


def test_kernels_stay_inside_the_output_buffer(ctx, orc):
    """Guard bands of 64 KiB before and after the output stay untouched by every kernel choice (the span kernel masks the
    lanes past a row, the rows kernel pads its grid, tiles are masked)."""
    import doppler_amd
    guard = 65536
    cases = [
        ([(300000 + 13, 5000.0)], 1024000, 3),                                      # rows kernel + ragged head and tail
        ([(50000 + 17 * k, 333.0 + k) for k in range(10)], 48000, 3),               # span kernel, leftover ranges
        ([(120001, 9876.543), (5000, 3.0), (70000, -1234.5)], 1024000, 4),          # tile kernel only
        ([(3000000 + 77, 5001.0)], 1024000, 5),                                     # one long stretch forced onto the span kernel
    ]
    for segs, rate, variant in cases:
        for intype, outtype in (("i16", "i16"), ("f32", "i16"), ("i16", "f32")):
            n = sum(c for c, _ in segs)
            x = make_iq(intype, n, 77)
            nb_out = n * BPS[outtype]
            d_in, d_buf = ctx.malloc(max(16, x.size)), ctx.malloc(nb_out + 2 * guard)
            try:
                ctx.h2d(d_in, x)
                fill = np.full(nb_out + 2 * guard, 0xC3, dtype=np.uint8)
                ctx.h2d(d_buf, fill)
                ctx.set_tuning(0, 0, variant)
                try:
                    plan = ctx.plan_segments(segs, rate)
                    plan.run(d_in, intype, d_buf + guard, outtype)
                    ctx.synchronize()
                    plan.close()
                finally:
                    ctx.set_tuning(0, 0, 3)
                back = np.empty_like(fill)
                ctx.d2h(back, d_buf)
                assert (back[:guard] == 0xC3).all() and (back[guard + nb_out:] == 0xC3).all(), (segs[:2], variant, intype, outtype)
                want, _ = oracle_segments(orc, x, intype, outtype, segs, rate)
                assert_same_bytes(back[guard:guard + nb_out], want, outtype, "guarded run variant %d" % variant)
            finally:
                ctx.free(d_in)
                ctx.free(d_buf)


def test_chunked_equals_whole(ctx, orc):
    """Time-chunk sharding: 5 block-aligned chunks seeded from the closed form reproduce
    the single-pass output byte for byte."""
    from doppler_amd import shard
    n = 2048 * 301 + 77
    x = make_iq("i16", n, 41)
    whole, fin = run_bulk(ctx, x, "i16", "i16", [(n, 9876.543)], 1024000)
    parts = []
    for r in range(5):
        lo, hi = shard.chunk_bounds(n, 5, r)
        seed = shard.chunk_seed(9876.543, 1024000, lo)
        got, _ = run_bulk(ctx, x[4 * lo:4 * hi], "i16", "i16", [(hi - lo, 9876.543)], 1024000, sn0=seed)
        parts.append(got)
    assert_same_bytes(np.concatenate(parts), whole, "i16", "chunks vs whole")
    want, _ = orc.const_stream(x, "i16", "i16", 9876, 1024000)   # different shift: must differ (sanity)
    assert not np.array_equal(want, whole)


def test_full_size_headline_config(ctx, orc):
    """BASELINE.json configs[1] at full size: 268 435 456 samples of i16 IQ, 5000 Hz at 1.024 Msps, compared
    byte for byte with the oracle run on all host cores (chunks seeded by the oracle's own sequential counter)."""
    n = 268435456
    rng = np.random.default_rng(2)
    x = rng.integers(-23170, 23171, size=2 * n, dtype=np.int16).view(np.uint8)
    got, fin = run_bulk(ctx, x, "i16", "i16", [(n, 5000.0)], 1024000)
    threads = min(os.cpu_count() or 1, 64)
    want, _ = orc.const_stream(x, "i16", "i16", 5000, 1024000, threads=threads)
    assert fin == 1024
    assert got.size == want.size
    # compare in slabs to keep the peak memory of the comparison small
    for a in range(0, got.size, 1 << 28):
        assert np.array_equal(got[a:a + (1 << 28)], want[a:a + (1 << 28)]), "mismatch in slab at byte %d" % a


def test_full_size_config0_f32_stream(ctx, orc):
    """BASELINE.json configs[0] at full size: 64 MiB of f32 IQ (8 388 608 samples), -15000 Hz at 256 ksps, f32 out."""
    n = 8388608
    x = make_iq("f32", n, 1)
    got, fin = run_bulk(ctx, x, "f32", "f32", [(n, -15000.0)], 256000)
    want, sn = orc.const_stream(x, "f32", "f32", -15000, 256000)
    assert fin == sn
    assert_same_bytes(got, want, "f32", "configs[0]")


@pytest.mark.parametrize("which", ["track", "track_256k"])
def test_full_size_track_replay(ctx, orc, which):
    """BASELINE.json configs[2] at full size: the 10-minute replay of bench.py --workload track (614 400 000 samples of
    i16 IQ, 600 one-second shifts from the host SGP4 + the reference's schedule, main.rs:156-184), and the same overpass at the
    reference README's own recording rate (bench.py --workload track_256k: 256 ksps, --offset -2500, README.md:60 — a quarter
    of the samples per second, so three times the share of seconds with fewer than five periods).  One span-kernel launch each,
    compared byte for byte with the oracle: every segment evaluated by the oracle's convert / shift_frequency / pack from the
    sequential counter the oracle itself carries across the segments (segments run on all host cores)."""
    import calendar
    import sys
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import doppler_amd
    seconds, rate, it, ot, start, offset, _ = bench.REPLAYS[which]
    assert (it, ot) == ("i16", "i16")
    segs = bench.track_segments(seconds, rate, "i16", calendar.timegm(start), offset=offset)
    n = sum(c for c, _ in segs)
    assert n == 600 * rate and doppler_amd.plan_layout(segs, rate)["walk_launches"] == 1
    rng = np.random.default_rng(3)
    x = rng.integers(-23170, 23171, size=2 * n, dtype=np.int16).view(np.uint8)
    got, fin = run_bulk(ctx, x, "i16", "i16", segs, rate)
    seeds, sn, pos = [], 0, 0
    for cnt, hz in segs:
        seeds.append((pos, cnt, hz, sn))
        sn = orc.advance_samplenum(sn, hz, rate, cnt)
        pos += cnt
    assert fin == sn

    def check(seg):
        pos, cnt, hz, sn0 = seg
        o, _ = orc.shift_frequency(orc.convert_iqi16_to_complex(x[4 * pos:4 * (pos + cnt)]), sn0, hz, rate)
        return np.array_equal(orc.pack_i16(o), got[4 * pos:4 * (pos + cnt)])

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 64)) as pool:
        ok = list(pool.map(check, seeds))
    assert all(ok), "segments that differ: %r" % [i for i, o in enumerate(ok) if not o][:10]


def test_special_values_and_degenerate_ratios(ctx, orc):
    """f32 inputs with NaN / inf / subnormals / signed zeros / huge magnitudes, i16 full-scale corners, and
    ratios that are inf or NaN (samplerate 0): outputs equal the oracle's (any NaN matches any NaN)."""
    from doppler_amd import dsp
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, 5e-39, 1.0, -1.0, 0.99999994, 3.4028235e38,
                         -3.4028235e38, np.inf, -np.inf, np.nan, 32767.5 / 32767.0, -32768.9 / 32767.0, 1e-20, 65504.0],
                        dtype=np.float32)
    rng = np.random.default_rng(12)
    re = rng.choice(specials, size=20000)
    im = rng.choice(specials, size=20000)
    x = np.empty(40000, dtype=np.float32)
    x[0::2], x[1::2] = re, im
    xb = x.view(np.uint8)
    for shift, rate in [(5000.0, 1024000), (815000.0, 2400000), (0.0, 1024000), (5000.0, 0), (0.0, 0), (3.0e38, 1)]:
        for outtype in ("f32", "i16"):
            cx = orc.convert_iqf32_to_complex(xb)
            o, sn_w = orc.shift_frequency(cx, 0, shift, rate)
            want = orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)
            got, cnt, sn_g = dsp.shift_block(xb, "f32", outtype, 0, shift, rate, ctx=ctx)
            assert sn_g == sn_w
            assert_same_bytes(got, want, outtype, "specials shift=%r rate=%d out=%s" % (shift, rate, outtype))
    corners = np.array([32767, -32768, -32768, 32767, 0, -1, 1, 0, 32767, 32767, -32768, -32768], dtype=np.int16)
    xi = np.tile(corners, 1000).view(np.uint8)
    # The i16 -> i16 kernels keep the unscaled integers and fold / 32768 into the pack constant (exact unless an
    # intermediate is a denormal): shifts of 1e-30 ... 1e-42 Hz make the sin side of every product tiny or denormal
    # (theta itself is a denormal for the last two), 3.0e38 Hz makes theta overflow.  Block, bulk and per-sample paths.
    for shift, rate in [(5000.0, 1024000), (-15000.0, 256000), (12345.678, 48000), (1e-30, 1024000), (-3e-37, 48000),
                        (1e-39, 1), (1e-42, 1000), (3.0e38, 1)]:
        cx = orc.convert_iqi16_to_complex(xi)
        o, sn_w = orc.shift_frequency(cx, 0, shift, rate)
        got, _, _ = dsp.shift_block(xi, "i16", "i16", 0, shift, rate, ctx=ctx)
        assert_same_bytes(got, orc.pack_i16(o), "i16", "i16 corners (saturating cast), shift=%r" % shift)
    big = np.tile(corners, 30000).view(np.uint8)                # 360 000 samples: the bulk kernels
    for shift, rate in [(1e-30, 1024000), (1e-42, 1000), (7.0, 1024000)]:
        cx = orc.convert_iqi16_to_complex(big)
        o, sn_w = orc.shift_frequency(cx, 0, shift, rate)
        for variant in (3, 1):
            ctx.set_tuning(0, 0, variant)
            try:
                got, fin = run_bulk(ctx, big, "i16", "i16", [(big.size // 4, shift)], rate)
            finally:
                ctx.set_tuning(0, 0, 3)
            assert fin == sn_w
            assert_same_bytes(got, orc.pack_i16(o), "i16", "i16 corners bulk, shift=%r variant=%d" % (shift, variant))


def test_random_cases_against_oracle(ctx, orc):
    """Seeded random sweep through the bulk path: formats, arbitrary f32 shifts, rates, counter starts, 1-4 segments."""
    rng = np.random.default_rng(77)
    for case in range(60):
        rate = int(rng.choice([8000, 48000, 256000, 1024000, 2400000, 1000003]))
        intype, outtype = [("i16", "i16"), ("f32", "f32"), ("i16", "f32"), ("f32", "i16")][case % 4]
        segs = []
        for _ in range(int(rng.integers(1, 5))):
            hz = float(np.float32(rng.uniform(-15000, 15000))) if rng.random() < 0.6 else float(np.float32(rng.integers(-9999, 9999)))
            cnt = int(rng.integers(1, 5000)) if rng.random() < 0.5 else int(rng.integers(66000, 200000))
            segs.append((cnt, hz))
        sn0 = int(rng.choice([0, 1, 77, 4096]))
        n = sum(c for c, _ in segs)
        x = make_iq(intype, n, 500 + case, full_scale=True)
        cx = orc.convert_iqi16_to_complex(x) if intype == "i16" else orc.convert_iqf32_to_complex(x)
        outs, sn, pos = [], sn0, 0
        for cnt, hz in segs:
            o, sn = orc.shift_frequency(cx[pos:pos + cnt], sn, hz, rate)
            outs.append(o)
            pos += cnt
        o = np.concatenate(outs)
        want = orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)
        got, fin = run_bulk(ctx, x, intype, outtype, segs, rate, sn0=sn0)
        assert fin == sn, (case, segs)
        assert_same_bytes(got, want, outtype, "random case %d %r" % (case, segs))


def test_first_reset_beyond_2_24(ctx, orc):
    """A shift of 0.02 Hz at 1.024 Msps: the counter runs to 51 199 998 before its first reset — past 2^24, where it is rounded
    before the multiply (fl32(n): steps of 2, then 4) and the closed form works on its significand (dpx_planner.cpp,
    first_reset_exact).  60 M samples: the linear run, the reset, the periodic stretch after it — against the oracle's
    sequential counter, and the stretch list against the candidate scan."""
    import doppler_amd
    from doppler_amd import engine
    rate, hz, n = 1024000, 0.02, 60_000_000
    assert engine.find_reset(hz, rate, 1, 1 << 30) == engine.find_reset_scan(hz, rate, 1, 1 << 27) == 51199998
    st, fin = doppler_amd.plan_describe([(n, hz)], rate, 0)
    assert [(s["first"], s["count"], s["n_start"], s["period"]) for s in st] == [(0, 1, 0, 0), (1, n - 1, 1, 51199998)], st
    assert fin == (n - 1) % 51199998 + 1
    x = make_iq("i16", n, 2424)
    got, sn = run_bulk(ctx, x, "i16", "i16", [(n, hz)], rate)
    want, sn_want = orc.segments_stream(x, "i16", "i16", [(n, hz)], rate, threads=16)
    assert sn == sn_want == fin
    assert_same_bytes(got, want, "i16", "first reset beyond 2^24")


def test_c_abi_error_paths(ctx):
    """Bad arguments come back as error codes with a message, never as a crash or a silent fallback."""
    import ctypes as C
    import doppler_amd
    from doppler_amd import _lib
    lib = doppler_amd.lib
    h = ctx.handle
    sn = C.c_uint32(0)
    n = C.c_size_t(0)
    buf = np.zeros(64, dtype=np.uint8)
    assert lib.dpx_shift_block(h, buf.ctypes.data, 64, 7, buf.ctypes.data, 64, 0, C.byref(sn), 1.0, 1000, C.byref(n)) == _lib.ERR_ARG
    assert lib.dpx_shift_block(h, buf.ctypes.data, 64, 0, buf.ctypes.data, 16, 1, C.byref(sn), 1.0, 1000, C.byref(n)) == _lib.ERR_CAPACITY
    assert lib.dpx_shift_block(h, buf.ctypes.data, 62, 0, buf.ctypes.data, 64, 0, C.byref(sn), 1.0, 1000, C.byref(n)) == _lib.ERR_BLOCK_LEN
    assert b"whole number" in lib.dpx_last_error()
    assert lib.dpx_shift_block(None, buf.ctypes.data, 64, 0, buf.ctypes.data, 64, 0, C.byref(sn), 1.0, 1000, C.byref(n)) == _lib.ERR_ARG
    plan = ctx.plan_const(5000.0, 1024000, 4096)
    d = ctx.malloc(65536)
    try:
        with pytest.raises(doppler_amd.DspError) as e:
            plan.run(d + 4, "i16", d + 32768, "i16")           # misaligned device pointer
        assert e.value.code == _lib.ERR_ARG
        with pytest.raises(doppler_amd.DspError):
            plan.run(0, "i16", d, "i16")
        with pytest.raises(doppler_amd.DspError):
            plan.run(d, "u8", d + 32768, "i16")
        plan.run(d, "i16", d + 32768, "i16")                  # and the valid call still works afterwards
        ctx.synchronize()
    finally:
        ctx.free(d)
        plan.close()
    with pytest.raises(doppler_amd.DspError):
        ctx.set_tuning(100, 1, 3)
    empty = ctx.plan_const(5000.0, 1024000, 0)                # empty plans are legal and run as no-ops
    assert empty.n_samples == 0 and empty.final_samplenum == 0
    empty.run(0, "i16", 0, "i16")
    empty.close()
    assert lib.dpx_device_count(C.byref(C.c_int())) == 0
    with pytest.raises(doppler_amd.DspError) as e:
        doppler_amd.Context(99)
    assert e.value.code == _lib.ERR_NO_DEVICE


def test_stream_api_ring_of_slabs(ctx, orc):
    """dpx_stream_*: slabs of unequal size, shift changes inside and between slabs, the ring filled to capacity,
    outputs collected in order — equal to the oracle run over the concatenated stream with the counter carried."""
    import doppler_amd
    from doppler_amd import _lib
    rate = 256000
    rng = np.random.default_rng(31)
    st = doppler_amd.Stream(ctx, "f32", "i16", rate, samplenum=0, slab_bytes=1 << 20, n_slabs=3)
    plan = []       # (n_samples, [(cnt, hz), ...]) per slab
    for k in range(8):
        n = int(rng.integers(1, (1 << 20) // 8 + 1)) if k != 3 else (1 << 20) // 8
        if k == 5:
            n = 0
        cuts = sorted(set(int(c) for c in rng.integers(0, n + 1, size=2))) if n else []
        bounds = [0] + [c for c in cuts if 0 < c < n] + [n]
        segs = [(b - a, float(np.float32(rng.uniform(-9000, 9000)))) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        plan.append((n, segs))
    total = sum(n for n, _ in plan)
    x = make_iq("f32", total, 88)
    outs, pos, submitted = [], 0, 0
    for n, segs in plan:
        if st.pending() == 3:
            with pytest.raises(doppler_amd.DspError) as e:      # ring full: acquire must refuse, not overwrite
                st.acquire()
            assert e.value.code == _lib.ERR_PLAN
            outs.append(st.next())
        buf = st.acquire()
        buf[: n * 8] = x[pos * 8:(pos + n) * 8]
        st.submit(n * 8, segs)
        pos += n
        submitted += 1
    while st.pending():
        outs.append(st.next())
    got = np.concatenate(outs)
    cx = orc.convert_iqf32_to_complex(x)
    ref, sn, pos = [], 0, 0
    for n, segs in plan:
        for cnt, hz in segs:
            o, sn = orc.shift_frequency(cx[pos:pos + cnt], sn, hz, rate)
            ref.append(o)
            pos += cnt
    assert st.samplenum == sn
    assert_same_bytes(got, orc.pack_i16(np.concatenate(ref)), "i16", "stream ring")
    with pytest.raises(doppler_amd.DspError):
        st.submit(8, [(1, 1.0)])                                # submit without acquire
    buf = st.acquire()
    with pytest.raises(doppler_amd.DspError):
        st.submit(12, [(1, 1.0)])                               # 12 bytes is not a whole f32 sample (dsp.rs:103)
    with pytest.raises(doppler_amd.DspError):
        st.submit(16, [(1, 1.0)])                               # segments do not add up
    st.close()


def test_stream_ring_over_several_contexts(ctx, orc):
    """dpx_stream_create_multi: the slab ring dealt out over three contexts (all on this box's one GPU; slab k runs on
    context k mod 3), shifts changing inside and between slabs, counter carried on the host — the same bytes as one
    sequential oracle pass, and the same as the single-context ring."""
    import doppler_amd
    rate = 1024000
    rng = np.random.default_rng(99)
    others = [doppler_amd.Context(0), doppler_amd.Context(0)]
    try:
        st = doppler_amd.Stream([ctx] + others, "i16", "f32", rate, slab_bytes=1 << 18, n_slabs=2)      # 6 slabs in the ring
        n_slabs, per = 23, (1 << 18) // 4
        x = make_iq("i16", n_slabs * per - 1000, 71, full_scale=True)
        plan, pos = [], 0
        for k in range(n_slabs):
            n = min(per, x.size // 4 - pos)
            cut = int(rng.integers(1, n))
            plan.append((pos, n, [(cut, float(np.float32(5001.0 + 17.5 * k))), (n - cut, float(np.float32(-3000.0 - k)))]))
            pos += n
        outs = []
        for pos, n, segs in plan:
            if st.pending() == 6:
                outs.append(st.next())
            buf = st.acquire()
            buf[: n * 4] = x[pos * 4:(pos + n) * 4]
            st.submit(n * 4, segs)
        while st.pending():
            outs.append(st.next())
        got = np.concatenate(outs)
        want, sn = orc.segments_stream(x, "i16", "f32", [sg for _, _, segs in plan for sg in segs], rate, threads=8)
        assert st.samplenum == sn
        assert_same_bytes(got, want, "f32", "ring over three contexts")
        st.close()
    finally:
        for c in others:
            c.close()
