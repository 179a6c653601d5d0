"""CPU: a second restatement of the reference's hot path (tests/golden/restate_numpy.py: numpy float32, written from the
Rust text alone, sharing no code with oracle/) against the C oracle and the committed golden vectors — a slip in the
reading of the unpack, the multiply, the counter rule or the pack would have to be made twice, in two languages, to go
unnoticed.  Both `as i16` meanings (Rust >= 1.45 saturating; the x86-64 code of a 2016 rustc: truncate and wrap)."""
import os
import sys

import numpy as np
import pytest

from helpers import assert_same_bytes, load_golden, make_iq, shift_block_cases

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import restate_numpy as rn  # noqa: E402


def test_restatement_reproduces_every_golden_operator_case():
    n = 0
    for c in shift_block_cases():
        got, sn = rn.shift_block(c["x"], c["intype"], c["outtype"], c["sn0"], c["shift"], c["rate"])
        assert sn == c["sn1"], c["key"]
        assert_same_bytes(got, c["y"], c["outtype"], "numpy restatement, golden case %s" % c["key"])
        n += 1
    assert n == 238


def test_restatement_reproduces_the_golden_streams():
    z = load_golden("const_stream_cases.npz")
    for k in sorted(f[:-5] for f in z.files if f.endswith("_meta")):
        shift, rate, sn1, it, ot = z[k + "_meta"][:5]
        intype, outtype = ("i16", "f32")[int(it)], ("i16", "f32")[int(ot)]
        got, sn = rn.const_stream(z[k + "_in"], intype, outtype, float(shift), int(rate))
        assert sn == int(sn1), k
        assert_same_bytes(got, z[k + "_out"], outtype, "numpy restatement, const stream %s" % k)
    t = load_golden("track_stream_case.npz")
    rate, freq, offset, _ = t["meta"]
    got, sn, log = rn.track_stream(t["x"], "i16", "i16", int(rate), int(freq), t["rr"], int(offset))
    assert np.array_equal(log.view(np.uint32), t["shift_log"].astype(np.float32).view(np.uint32))
    assert_same_bytes(got, t["y"], "i16", "numpy restatement, track replay")


@pytest.mark.parametrize("legacy", [False, True])
def test_restatement_and_oracle_agree_on_clipping_and_special_inputs(orc, legacy):
    """Full-scale i16 (the rotation pushes |I + jQ| past full scale: the cast clips — or, in 2016, wrapped), f32 inputs far
    outside [-1, 1), infinities, NaN, denormals, +-2^31 / 32767 boundaries; both cast meanings; fresh seeds."""
    orc.set_i16_cast(1 if legacy else 0)
    try:
        rng = np.random.default_rng(77)
        for intype, n in (("i16", 2048), ("f32", 1024)):
            x = make_iq(intype, n, 4242, full_scale=True)
            if intype == "f32":
                f = x.view(np.float32).copy()
                f[:400] *= rng.choice([3.0, 70.0, 7e4, 3e9, 1e30], size=400).astype(np.float32)
                f[400:420] = [np.inf, -np.inf, np.nan, 65536.0, -65536.0, 65535.9, 2147483648.0 / 32767, -2147483648.0 / 32767,
                              1e-40, -1e-40, 32768.0 / 32767, 1.0, -1.0, 32767.5 / 32767, -32768.5 / 32767, 0.0, -0.0, 2.0, -2.0, 1.00001]
                x = f.view(np.uint8)
            for shift, rate, sn0 in ((5000.0, 1024000, 0), (-15000.0, 256000, 17), (9876.543, 1024000, 2591), (0.0, 48000, 0)):
                for outtype in ("i16", "f32"):
                    want, _, _, sn_w = orc.shift_block(x, intype, outtype, sn0, shift, rate)
                    got, sn = rn.shift_block(x, intype, outtype, sn0, shift, rate, legacy_cast=legacy)
                    assert sn == sn_w
                    assert_same_bytes(got, want, outtype, "numpy restatement vs oracle, %s->%s shift %r legacy=%s" % (intype, outtype, shift, legacy))
        # the two meanings differ exactly where the product leaves the i16 range
        z = np.array([1.2, -1.3, 0.5, 70000.0 / 32767, np.nan, np.inf, -np.inf, 3e9], dtype=np.float32)
        sat = rn.pack_i16(z, z, legacy=False).view(np.int16)[::2]
        leg = rn.pack_i16(z, z, legacy=True).view(np.int16)[::2]
        assert list(sat) == [32767, -32768, 16383, 32767, 0, 32767, -32768, 32767]
        assert list(leg) == [39320 - 65536, -42597 + 65536, 16383, 70000 - 65536, 0, 0, 0, 0]
    finally:
        orc.set_i16_cast(0)
