"""CPU: the oracle against everything the reference itself pins for this path, and against the
committed golden vectors (which were generated with the reference's own complex.c linked)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import assert_same_bytes, load_golden, shift_block_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_close(a, b, delta):
    # dsp.rs:49-55 assert_eq_delta: |(a-b)/b| < delta
    return abs((float(a) - float(b)) / float(b)) < delta


def test_reference_known_answers_cexpf(orc):
    """The reference's own test_cexpf (src/dsp.rs:57-83), same inputs, same tolerances."""
    re, im = orc.ccexpf(0.0, 0.0)
    assert rel_close(re, 1.0, 1e-6) and im == 0.0      # (a.im - 0)/0 is NaN in the reference; exact 0 here
    re, im = orc.ccexpf(1.0, 1.0)
    assert rel_close(re, 1.468694, 1e-6) and rel_close(im, 2.2873552, 1e-6)
    re, im = orc.ccexpf(70.0, 70.0)
    assert rel_close(re, 1593075600000000000000000000000.0, 1e-6)
    assert rel_close(im, 1946674600000000000000000000000.0, 1e-6)
    re, im = orc.ccexpf(1e6, 1e6)
    assert re == np.inf and im == -np.inf
    z = load_golden("reference_tests.npz")
    for (a, b), (wr, wi) in zip(z["kat_in"], z["kat_out"]):
        gr, gi = orc.ccexpf(a, b)
        assert np.float32(gr).tobytes() == np.float32(wr).tobytes()
        assert np.float32(gi).tobytes() == np.float32(wi).tobytes()


def test_reference_bench_configuration(orc):
    """src/dsp.rs:136-157: 1 000 000 bytes of 0xAA as f32 IQ, 815 kHz at 2.4 Msps, counter carried over
    301 calls.  The reference asserts nothing; the digests pin the oracle against regressions."""
    z = load_golden("reference_tests.npz")
    cx = orc.convert_iqf32_to_complex(np.full(1000000, 0xAA, dtype=np.uint8))
    assert cx.size == 125000
    sn = 0
    digest = []
    for it in range(301):
        o, sn = orc.shift_frequency(cx, sn, 815000.0, 2400000)
        if it in (0, 1, 150, 300):
            digest.append(int(o.view(np.uint32).astype(np.uint64).sum()))
    assert digest == [int(d) for d in z["bench_digest"]]
    assert sn == int(z["bench_final_samplenum"][0])


def test_restated_sincosf_matches_libm(orc):
    """oracle/check_sincosf: restated glibc-2.35 sincosf == this host's libm, bit for bit, on a strided
    sweep of all float bit patterns (the exhaustive run is recorded in DESIGN.md)."""
    exe = os.path.join(ROOT, "oracle", "check_sincosf")
    r = subprocess.run([exe, "--stride", "509", "--cexp", "--threads", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    variant = orc.libm_variant()
    assert variant in (0, 1), "this host's libm sincosf matches neither restated build"
    disc = np.array([0x418a3adb, 0x41bc76d9, 0x4202eb4b, 0x42e87a55, 0xc2687a55], dtype=np.uint32).view(np.float32)
    a = orc.ccexpf_imag_array(disc, mode=0)
    b = orc.ccexpf_imag_array(disc, mode=1 if variant == 1 else 2)
    assert a.tobytes() == b.tobytes()
    c = orc.ccexpf_imag_array(disc, mode=2 if variant == 1 else 1)
    assert a.tobytes() != c.tobytes()     # these arguments are exactly where the two libm builds differ


def test_device_reductions_proved_by_enumeration_on_the_cpu_model(tmp_path):
    """The device sincos (doppler_amd/csrc/dpx_sincos.h, round 4) rounds the quadrant with a fused multiply-add against
    1.5 * 2^52 and reduces 120 <= |theta| < 2^30 with a three-term double-precision 2/pi instead of glibc's integer
    product.  Neither is glibc's operation sequence; both must give glibc's floats.  tests/extended/sincos_model.c
    restates the two reductions with the IEEE double operations the device executes and enumerates EVERY argument of
    their ranges (both signs, both libm builds) against the restated glibc sincosf: no mismatch allowed.  (The device
    itself is enumerated on the GPU box: tests/extended/exhaustive_device_sincos.py.)"""
    exe = str(tmp_path / "sincos_model")
    fma = ["-mfma"] if "fma" in open("/proc/cpuinfo").read().split() else []
    subprocess.check_call(["gcc", "-O2", *fma, "-ffp-contract=off", "-I" + os.path.join(ROOT, "oracle"), "-o", exe,
                           os.path.join(ROOT, "tests", "extended", "sincos_model.c"), os.path.join(ROOT, "oracle", "sincosf_glibc.c"),
                           "-lm", "-lpthread"])
    for args in (["--v", "1"], ["--v", "0"], ["--plain", "--v", "1", "--lo", "0x39800000", "--hi", "0x42f00000"],
                 ["--plain", "--v", "0", "--lo", "0x39800000", "--hi", "0x42f00000"], ["--mixed"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True)
        assert r.returncode == 0 and "mismatches=0 " in r.stdout, r.stdout + r.stderr


def test_restated_expf_and_cexpf_match_libm(orc):
    """Restated glibc expf (strided sweep of all floats; the exhaustive run is in DESIGN.md) and cexpf
    (random pairs + overflow/underflow/inf/nan corners) against this host's libm."""
    exe = os.path.join(ROOT, "oracle", "check_sincosf")
    r = subprocess.run([exe, "--stride", "1021", "--expf", "--threads", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert '"expf_checked": 1' in r.stdout
    variant = orc.libm_variant()
    rng = np.random.default_rng(8)
    n = 300000
    z = np.empty(n + 36, dtype=orc.complex32)
    z["re"][:n] = rng.uniform(-110, 270, size=n).astype(np.float32)
    z["im"][:n] = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, size=n)).astype(np.float32)
    spec = np.array([0.0, 88.5, 177.0, 265.0, -104.0, np.inf], dtype=np.float32)
    gr, gi = np.meshgrid(spec, np.array([0.0, -0.0, 1.0, 1e6, np.inf, np.nan], dtype=np.float32))
    z["re"][n:], z["im"][n:] = gr.ravel(), gi.ravel()
    a = orc.ccexpf_array(z, mode=0)
    b = orc.ccexpf_array(z, mode=1 if variant == 1 else 2)
    same = ((a["re"].view(np.uint32) == b["re"].view(np.uint32)) | (np.isnan(a["re"]) & np.isnan(b["re"]))) & \
           ((a["im"].view(np.uint32) == b["im"].view(np.uint32)) | (np.isnan(a["im"]) & np.isnan(b["im"])))
    assert same.all(), z[~same][:5]


def test_oracle_matches_golden_shift_cases(orc):
    n = 0
    for c in shift_block_cases():
        cx = orc.convert_iqi16_to_complex(c["x"]) if c["intype"] == "i16" else orc.convert_iqf32_to_complex(c["x"])
        o, sn1 = orc.shift_frequency(cx, c["sn0"], c["shift"], c["rate"])
        y = orc.pack_i16(o) if c["outtype"] == "i16" else orc.pack_f32(o)
        assert sn1 == c["sn1"], c["key"]
        assert_same_bytes(y, c["y"], c["outtype"], c["key"])
        n += 1
    assert n > 200


def test_oracle_matches_golden_streams(orc):
    z = load_golden("const_stream_cases.npz")
    for k in range(4):
        shift, rate, sn, it, ot = z["s%d_meta" % k]
        fi, fo = ("i16", "f32")[int(it)], ("i16", "f32")[int(ot)]
        y, sn_g = orc.const_stream(z["s%d_in" % k], fi, fo, int(shift), int(rate))
        assert sn_g == int(sn)
        assert_same_bytes(y, z["s%d_out" % k], fo, "stream %d" % k)
        # multi-threaded chunked run of the oracle agrees with the sequential one
        y2, _ = orc.const_stream(z["s%d_in" % k], fi, fo, int(shift), int(rate), threads=3)
        assert_same_bytes(y2, y, fo, "stream %d mt" % k)
    t = load_golden("track_stream_case.npz")
    rate, freq, off, sn = t["meta"]
    y, sn_g, log = orc.track_stream(t["x"], "i16", "i16", int(rate), int(freq), t["rr"], offset_hz=int(off))
    assert sn_g == int(sn)
    assert_same_bytes(y, t["y"], "i16", "track")
    assert np.array_equal(log, t["shift_log"])
    # the schedule of main.rs:156-184: first block uses dt=0, changes only at whole seconds, one block late
    spb = 2048
    n_blocks = log.size
    dts = [0] + [int(np.float32(np.float32(b * spb) / np.float32(rate))) for b in range(0, n_blocks - 1)]
    want = [np.float32(np.float32(-(t["rr"][min(d, 11)] * 1000.0 / 299792458.0) * freq) + np.float32(off)) for d in dts]
    assert np.array_equal(np.array(want, dtype=np.float32), log)


def test_oracle_edge_cases(orc):
    """Ragged byte counts panic in the reference (dsp.rs:87,103); saturating i16 cast; NaN -> 0."""
    with pytest.raises(orc.OracleError):
        orc.convert_iqi16_to_complex(np.zeros(6, np.uint8))
    with pytest.raises(orc.OracleError):
        orc.convert_iqf32_to_complex(np.zeros(12, np.uint8))
    with pytest.raises(orc.OracleError):
        orc.const_stream(np.zeros(8192 + 3, np.uint8), "i16", "i16", 1, 1000)
    big = np.zeros(4, dtype=orc.complex32)
    big["re"] = [1.0, -1.5, np.nan, np.inf]
    big["im"] = [1.00002, 40000.0, -np.inf, 0.99999]
    out = orc.pack_i16(big).view(np.int16)
    assert out.tolist() == [32767, 32767, -32768, 32767, 0, -32768, 32767, 32766]
    # empty stream: one empty block, nothing written, counter untouched
    y, sn = orc.const_stream(np.zeros(0, np.uint8), "f32", "i16", 5, 1000)
    assert y.size == 0 and sn == 0
    # shift 0: counter sticks at 1, corrector (1, -0.0): not the identity for i16 (x/32768*32767)
    x = np.array([32767, -32768, 1000, -1], dtype=np.int16).view(np.uint8)
    y, sn = orc.const_stream(x, "i16", "i16", 0, 1024000)
    assert sn == 1 and y.view(np.int16).tolist() == [32766, -32767, 999, 0]


def test_segments_stream_checker_equals_the_block_loop(orc):
    """oracle.segments_stream (the multi-threaded checker for sharded streams) against the block-by-block restatements
    it stands in for: the golden track replay (per-block shifts of main.rs:156-184), and `doppler const` with a carried
    counter, for several thread counts and starting counters."""
    t = load_golden("track_stream_case.npz")
    rate, _, _, sn = t["meta"]
    x = t["x"]
    n = x.size // 4
    segs = [(min(2048, n - b * 2048), float(hz)) for b, hz in enumerate(t["shift_log"]) if n - b * 2048 > 0]
    for threads in (1, 3, 8):
        got, fin = orc.segments_stream(x, "i16", "i16", segs, int(rate), threads=threads)
        assert fin == int(sn)
        assert_same_bytes(got, t["y"], "i16", "segments_stream vs golden track stream, %d threads" % threads)
    rng = np.random.default_rng(5)
    for intype, outtype in (("i16", "f32"), ("f32", "i16")):
        m = 2048 * 37 + 99
        xb = (rng.integers(-32768, 32768, size=2 * m, dtype=np.int16).view(np.uint8) if intype == "i16"
              else rng.uniform(-1, 1, size=2 * m).astype(np.float32).view(np.uint8))
        for sn0 in (0, 1, 2591, 77777):
            want, sn_w = orc.const_stream(xb, intype, outtype, 9876, 1024000, samplenum=sn0)
            got, sn_g = orc.segments_stream(xb, intype, outtype, [(m, 9876.0)], 1024000, samplenum=sn0, threads=5)
            assert sn_g == sn_w
            assert_same_bytes(got, want, outtype, "segments_stream vs const_stream sn0=%d" % sn0)
    out, fin = orc.segments_stream(np.zeros(0, np.uint8), "i16", "i16", [], 1000, samplenum=7, threads=4)
    assert out.size == 0 and fin == 7
