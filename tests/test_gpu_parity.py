"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Tolerances (north_star: f32 within 1 ulp, i16 within +-1 LSB) are met with margin 0:
every comparison below is exact equality of the output bytes, except that any NaN
matches any NaN (x86 and CDNA4 propagate different NaN payloads through a*c - b*s).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = {"i16": 4, "f32": 8}


def make_iq(fmt, n, seed, full_scale=False):
    rng = np.random.default_rng(seed)
    if fmt == "i16":
        lim = 32767 if full_scale else 23170
        a = rng.integers(-lim - (1 if full_scale else 0), lim + 1, size=2 * n, dtype=np.int16)
        return a.view(np.uint8)
    a = rng.uniform(-1.0, 1.0, size=2 * n).astype(np.float32)
    return a.view(np.uint8)


def assert_same_bytes(got, want, outfmt, what=""):
    got = np.asarray(got).view(np.uint8).reshape(-1)
    want = np.asarray(want).view(np.uint8).reshape(-1)
    assert got.size == want.size, "%s: %d bytes vs %d" % (what, got.size, want.size)
    if outfmt == "f32":
        g, w = got.view(np.uint32), want.view(np.uint32)
        bad = g != w
        if bad.any():
            gf, wf = got.view(np.float32), want.view(np.float32)
            bad &= ~(np.isnan(gf) & np.isnan(wf))
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            raise AssertionError("%s: %d f32 words differ, first at word %d: got %r want %r" % (
                what, int(bad.sum()), i, got.view(np.float32)[i], want.view(np.float32)[i]))
    else:
        bad = got != want
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            raise AssertionError("%s: %d bytes differ, first at byte %d (sample %d): got %d want %d" % (
                what, int(bad.sum()), i, i // 4, got[i], want[i]))


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
    bits = np.concatenate([bits, disc, disc | 0x80000000, special])
    theta = bits.view(np.float32)
    z = np.zeros(theta.size, dtype=dsp.complex32)
    z["im"] = theta
    got = dsp.ccexpf(z, ctx=ctx)
    # restated glibc sincosf, FMA build (host-independent)
    assert_same_bytes(got, orc.ccexpf_imag_array(theta, mode=1), "f32", "device vs restated glibc sincosf")
    # this host's libm through the reference's own ccexpf (oracle/_ref when built)
    if orc.libm_variant() == 1:
        assert_same_bytes(got, orc.ccexpf_imag_array(theta, mode=0), "f32", "device vs libm cexpf")
    # the SSE2 build of libm is reproduced as well
    ctx.set_libm_contraction(False)
    try:
        got0 = dsp.ccexpf(z, ctx=ctx)
    finally:
        ctx.set_libm_contraction(True)
    assert_same_bytes(got0, orc.ccexpf_imag_array(theta, mode=2), "f32", "device vs restated glibc (no fma)")


SHIFTS = [(5000.0, 1024000), (-15000.0, 256000), (815000.0, 2400000), (0.0, 1024000), (3.0, 1024000),
          (9876.543, 1024000), (-5234.17, 1024000)]


@pytest.mark.parametrize("intype", ["i16", "f32"])
@pytest.mark.parametrize("outtype", ["i16", "f32"])
def test_shift_block_matches_oracle(ctx, orc, intype, outtype):
    """The fused kernel vs the oracle's three-pass restatement, all format pairs, carried counter."""
    from doppler_amd import dsp
    for si, (shift_hz, rate) in enumerate(SHIFTS):
        for n in [0, 1, 3, 4, 5, 2047, 2048, 2049, 3 * 2048 + 5, 40000]:
            for sn0 in [0, 1, 7]:
                x = make_iq(intype, n, 100 * si + n % 97 + sn0)
                want, _, cnt, sn_w = None, None, None, None
                # oracle: whole buffer through the per-sample functions (no 8192 cap here)
                if intype == "i16":
                    cx = orc.convert_iqi16_to_complex(x)
                else:
                    cx = orc.convert_iqf32_to_complex(x)
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
    a = dsp.convert_iqi16_to_complex(x, ctx=ctx)
    assert_same_bytes(a, orc.convert_iqi16_to_complex(x), "f32", "convert_iqi16")
    y = make_iq("f32", 3000, 2)
    b = dsp.convert_iqf32_to_complex(y, ctx=ctx)
    assert_same_bytes(b, orc.convert_iqf32_to_complex(y), "f32", "convert_iqf32")
    sn = 0
    sn_o = 0
    for _ in range(3):   # carried counter across calls, like main.rs:60
        g, sn = dsp.shift_frequency(b, sn, 815000.0, 2400000, ctx=ctx)
        w, sn_o = orc.shift_frequency(b, sn_o, 815000.0, 2400000)
        assert sn == sn_o
        assert_same_bytes(g, w, "f32", "shift_frequency")
    big = np.zeros(8, dtype=dsp.complex32)
    big["re"] = [0.5, 1.0, 1.5, -1.0, -1.5, np.nan, np.inf, -np.inf]
    big["im"] = [-0.5, 1.00002, 40000.0, -1.00002, 1e30, 1e-30, -0.0, 3e38]
    assert_same_bytes(dsp.pack_iqi16(big, ctx=ctx), orc.pack_i16(big), "i16", "pack saturation / NaN")
    with pytest.raises(dsp.DspError) as e:
        dsp.convert_iqi16_to_complex(x[:-1], ctx=ctx)
    assert e.value.code == -2
    with pytest.raises(dsp.DspError):
        dsp.shift_block(y[:-3], "f32", "f32", 0, 1.0, 1000, ctx=ctx)


@pytest.mark.parametrize("intype,outtype", [("i16", "i16"), ("f32", "f32"), ("i16", "f32"), ("f32", "i16")])
def test_const_stream_bulk_vs_oracle(ctx, orc, intype, outtype):
    """`doppler const` over 4 Mi samples, device-resident bulk path, every tuning variant."""
    import doppler_amd
    n = (1 << 22) + 8192 // FS[intype] * 3 + 1234 * (FS[intype] // 4)
    x = make_iq(intype, n, 11)
    want, sn_w = orc.const_stream(x, intype, outtype, 5000, 1024000)
    nthreads = 8
    d_in = ctx.malloc(x.size)
    d_out = ctx.malloc(n * FS[outtype])
    try:
        ctx.h2d(d_in, x)
        for variant, block, vecs in [(3, 256, 1), (4, 256, 1), (1, 256, 1), (2, 128, 1), (4, 128, 2), (1, 256, 2)]:
            ctx.set_tuning(block, vecs, variant)
            plan = ctx.plan_const(5000.0, 1024000, n)
            got = np.zeros(n * FS[outtype], dtype=np.uint8)
            ctx.h2d(d_out, got)
            plan.run(d_in, intype, d_out, outtype)
            ctx.synchronize()
            ctx.d2h(got, d_out)
            assert plan.final_samplenum == sn_w
            assert_same_bytes(got, want, outtype, "variant=%d block=%d vecs=%d" % (variant, block, vecs))
            plan.close()
    finally:
        ctx.set_tuning(256, 1, 3)
        ctx.free(d_in)
        ctx.free(d_out)
