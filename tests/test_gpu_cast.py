"""GPU: the two meanings of `(x * 32767.0) as i16` outside the i16 range (reference src/main.rs:77-78) — Rust >= 1.45
(saturate; the default) and the x86-64 code of a 2016 rustc (wrap; dpx_set_i16_cast) — in every kernel, against the oracle."""
import numpy as np
import pytest

from helpers import BPS, assert_same_bytes, load_golden, make_iq
from test_gpu_parity import oracle_segments, run_bulk

pytestmark = pytest.mark.gpu


def test_legacy_i16_cast_wraps_like_a_2016_rustc(ctx, orc):
    """dpx_set_i16_cast(DPX_CAST_LEGACY_X86): `(x * 32767.0) as i16` as the x86-64 code of a 2016 rustc computed it
    (CVTTSS2SI, low 16 bits kept: clipping samples wrap; NaN and |x| >= 2^31 give 0) against the oracle's twin
    (orc.set_i16_cast(1), itself cross-checked by tests/test_restatement.py) — block-wise operator, the pack operator, and
    bulk plans on EVERY kernel (round 4: the cast is a launch-uniform flag of all of them, no longer a tile-kernel build):
    rows kernel (5000 Hz), one-matrix span launch (5001 Hz), tile kernel (variant 4), a track-shaped span launch with
    matrices of 2-13 rows (several windows per workgroup, second turns) and leftover blocks; and back."""
    import doppler_amd
    from doppler_amd import dsp
    rng = np.random.default_rng(99)
    n = 3 * 2048 + 77 + (1 << 20)
    xi = make_iq("i16", n, 777, full_scale=True)                      # rotated full-scale samples clip
    f = make_iq("f32", n, 778).view(np.float32).copy()
    f[: n // 2] *= rng.choice([1.5, 3.0, 70.0, 7e4, 3e9, 1e30], size=n // 2).astype(np.float32)
    f[5:13] = [np.inf, -np.inf, np.nan, 65536.0, -65536.0, 2147483648.0 / 32767, -2147483648.0 / 32767, 1e-40]
    xf = f.view(np.uint8)
    segs = [(20000 + 17 * k, 333.0 + 7 * k) for k in range(3)] + [(n - 3 * 20000 - 51, -4000.0)]
    try:
        for legacy in (1, 0):
            ctx.set_i16_cast(bool(legacy))
            orc.set_i16_cast(legacy)
            differs = False
            for intype, x in (("i16", xi), ("f32", xf)):
                # block-wise, as main.rs:113-118 calls the closure
                sn, sn_w, pos = 0, 0, 0
                while pos < 3 * 8192:
                    blk = x[pos:pos + 8192]
                    got, _, sn = dsp.shift_block(blk, intype, "i16", sn, 5000.0, 1024000, ctx=ctx)
                    want, _, _, sn_w = orc.shift_block(blk, intype, "i16", sn_w, 5000.0, 1024000)
                    assert sn == sn_w
                    assert_same_bytes(got, want, "i16", "legacy=%d block-wise %s->i16" % (legacy, intype))
                    pos += 8192
                # bulk: the rows kernel, a one-matrix span launch, the tile kernel — the same plans in both modes
                for shift, variant, kern in ((5000, 3, "rows_launches"), (5001, 3, "walk_launches"), (5001, 4, "tile_launches")):
                    assert doppler_amd.plan_layout([(n, float(shift))], 1024000, variant=variant)[kern] == 1
                    want, sn_w = orc.const_stream(x, intype, "i16", shift, 1024000)
                    ctx.set_tuning(0, 0, variant)
                    try:
                        got, fin = run_bulk(ctx, x, intype, "i16", [(n, float(shift))], 1024000)
                    finally:
                        ctx.set_tuning(0, 0, 3)
                    assert fin == sn_w
                    assert_same_bytes(got, want, "i16", "legacy=%d const bulk %s->i16 %d Hz variant %d" % (legacy, intype, shift, variant))
                # track-shaped: nine matrices of 2..13 rows of one period (8192 at 262 144 Hz) with ragged last rows
                tsegs = [((2 + (5 * k) % 12) * 8192 + 1000 * (k % 3), 32.0 * (2 * k + 1)) for k in range(9)]
                tn = sum(c for c, _ in tsegs)
                assert tn <= n and doppler_amd.plan_layout(tsegs, 262144)["walk_matrices"] >= 5
                xt = x[:tn * (4 if intype == "i16" else 8)]
                want, sn_w = orc.segments_stream(xt, intype, "i16", tsegs, 262144)
                got, fin = run_bulk(ctx, xt, intype, "i16", tsegs, 262144)
                assert fin == sn_w
                assert_same_bytes(got, want, "i16", "legacy=%d track-shaped %s->i16" % (legacy, intype))
                want, sn_w = orc.segments_stream(x, intype, "i16", segs, 48000)
                got, fin = run_bulk(ctx, x, intype, "i16", segs, 48000)
                assert fin == sn_w
                assert_same_bytes(got, want, "i16", "legacy=%d segments %s->i16" % (legacy, intype))
                orc.set_i16_cast(0)
                sat, _ = orc.segments_stream(x, intype, "i16", segs, 48000)
                orc.set_i16_cast(legacy)
                differs = differs or not np.array_equal(sat, want)
            assert differs == bool(legacy)                            # the inputs do exercise the corner
            # the un-fused pack operator
            z = np.zeros(16, dtype=orc.complex32)
            z["re"][:8] = [1.2, -1.3, 0.5, 70000.0 / 32767, np.nan, np.inf, -np.inf, 3e9]
            z["im"][:8] = [-1.2, 1.3, -0.5, -70000.0 / 32767, 0.0, 1.0, -1.0, -3e9]
            assert_same_bytes(dsp.pack_iqi16(z, ctx=ctx), orc.pack_i16(z), "i16", "legacy=%d pack operator" % legacy)
    finally:
        ctx.set_i16_cast(False)
        orc.set_i16_cast(0)
