"""Generates tests/golden/*.npz — golden input/output vectors for the hot path.

Run in the build container, where /root/reference is mounted, so that the oracle calls the
reference's own src/complex.c (oracle/_ref/libcomplex.so) for every corrector:

    python tests/golden/make_golden.py

The vectors are DATA (inputs and expected outputs); no reference source text is stored.
Inputs come from a counter-based generator (numpy PCG64 with fixed seeds), outputs from the
oracle's restatement of src/dsp.rs:85-134 + src/main.rs:62-99.  The four known-answer values of
the reference's own test (src/dsp.rs:57-83) are stored verbatim in kat_cexpf.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

SHIFTS = [(0.0, 1024000), (5000.0, 1024000), (-15000.0, 256000), (815000.0, 2400000), (9876.543, 1024000),
          (3.0, 1024000)]
LENGTHS = [0, 1, 2047, 2048, 2049, 3 * 2048 + 5]


def make_iq(fmt, n, seed):
    rng = np.random.default_rng(seed)
    if fmt == "i16":
        return rng.integers(-32768, 32768, size=2 * n, dtype=np.int16).view(np.uint8)
    return rng.uniform(-1.0, 1.0, size=2 * n).astype(np.float32).view(np.uint8)


def period_of(shift, rate):
    n = orc.advance_samplenum(0, shift, rate, 1)
    cnt = 1
    while True:
        n2 = orc.advance_samplenum(n, shift, rate, 1)
        if n2 == 1:
            return cnt
        n = n2
        cnt += 1


def main():
    assert orc.have_ref(), "build oracle/_ref first (make -C oracle ref): golden vectors must come from the reference's complex.c"
    cases = {}
    idx = 0
    for shift, rate in SHIFTS:
        P = period_of(shift, rate)
        starts = sorted(set([0, 1, max(1, P - 1), P]))
        for intype in ("i16", "f32"):
            for outtype in ("i16", "f32"):
                for n in LENGTHS:
                    for sn0 in starts:
                        # keep the files small: full cross product only for the short lengths
                        if n >= 2047 and (sn0 not in (0, P) or shift not in (5000.0, 815000.0, 9876.543)):
                            continue
                        if n > 2049 and (sn0 != 0 or intype != outtype):
                            continue
                        if P > 100000 and sn0 not in (0, 1):
                            continue
                        x = make_iq(intype, n, 1000 + idx)
                        cx = orc.convert_iqi16_to_complex(x) if intype == "i16" else orc.convert_iqf32_to_complex(x)
                        o, sn1 = orc.shift_frequency(cx, sn0, shift, rate)
                        y = orc.pack_i16(o) if outtype == "i16" else orc.pack_f32(o)
                        key = "c%04d" % idx
                        cases[key + "_in"] = x
                        cases[key + "_out"] = y
                        cases[key + "_meta"] = np.array([shift, rate, sn0, sn1, {"i16": 0, "f32": 1}[intype],
                                                         {"i16": 0, "f32": 1}[outtype], n], dtype=np.float64)
                        idx += 1
    np.savez_compressed(os.path.join(HERE, "shift_block_cases.npz"), **cases)
    print("shift_block_cases.npz: %d cases" % idx)

    # const-mode stream with the reference's 8192-byte block loop (main.rs:102-119), ragged tail
    streams = {}
    for k, (intype, outtype, shift, rate, nbytes) in enumerate([
            ("i16", "i16", 5000, 1024000, 8192 * 5 + 1236), ("f32", "f32", -15000, 256000, 8192 * 4),
            ("i16", "f32", 815000, 2400000, 8192 * 3 + 4), ("f32", "i16", 12345, 1024000, 8192 * 6 + 800)]):
        x = make_iq(intype, nbytes // (4 if intype == "i16" else 8), 5000 + k)
        y, sn = orc.const_stream(x, intype, outtype, shift, rate)
        streams["s%d_in" % k] = x
        streams["s%d_out" % k] = y
        streams["s%d_meta" % k] = np.array([shift, rate, sn, {"i16": 0, "f32": 1}[intype], {"i16": 0, "f32": 1}[outtype]],
                                           dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "const_stream_cases.npz"), **streams)

    # track-mode replay (main.rs:156-184) against a synthetic range-rate table (orbit math is external)
    t = np.arange(12, dtype=np.float64)
    rr = 6.5 * np.tanh((t - 6.0) / 2.5)            # km/s, an overpass-shaped range rate
    rate = 8000                                    # low rate: whole seconds pass every ~4 blocks
    x = make_iq("i16", rate * 7 + 2048 * 3 + 77, 7001)
    y, sn, log = orc.track_stream(x, "i16", "i16", rate, 437505000, rr, offset_hz=-1200)
    np.savez_compressed(os.path.join(HERE, "track_stream_case.npz"), x=x, y=y, rr=rr,
                        meta=np.array([rate, 437505000, -1200, sn], dtype=np.float64), shift_log=log)

    # the reference's own known-answer test (dsp.rs:57-83) and its bench configuration (dsp.rs:136-157)
    kat_in = np.array([(0.0, 0.0), (1.0, 1.0), (70.0, 70.0), (1e6, 1e6)], dtype=np.float32)
    kat_out = np.array([orc.ccexpf(a, b) for a, b in kat_in], dtype=np.float32)
    bench_in = np.full(1000000, 0xAA, dtype=np.uint8)
    cx = orc.convert_iqf32_to_complex(bench_in)
    sn = 0
    digest = []
    for it in range(301):
        o, sn = orc.shift_frequency(cx, sn, 815000.0, 2400000)
        if it in (0, 1, 150, 300):
            digest.append(o.view(np.uint32).astype(np.uint64).sum())
    np.savez_compressed(os.path.join(HERE, "reference_tests.npz"), kat_in=kat_in, kat_out=kat_out,
                        bench_digest=np.array(digest, dtype=np.uint64), bench_final_samplenum=np.array([sn]),
                        bench_first_call_head=o[:64].copy())
    print("done; libm variant", orc.libm_variant())


if __name__ == "__main__":
    main()
