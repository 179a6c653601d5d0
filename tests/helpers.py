"""Shared helpers of the test suite."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FMT = {0: "i16", 1: "f32"}
BPS = {"i16": 4, "f32": 8}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def shift_block_cases():
    z = load_golden("shift_block_cases.npz")
    keys = sorted(k[:-5] for k in z.files if k.endswith("_meta"))
    for k in keys:
        shift, rate, sn0, sn1, it, ot, n = z[k + "_meta"]
        yield dict(key=k, shift=float(np.float32(shift)), rate=int(rate), sn0=int(sn0), sn1=int(sn1), intype=FMT[int(it)],
                   outtype=FMT[int(ot)], n=int(n), x=z[k + "_in"], y=z[k + "_out"])


def make_iq(fmt, n, seed, full_scale=False):
    rng = np.random.default_rng(seed)
    if fmt == "i16":
        if full_scale:
            return rng.integers(-32768, 32768, size=2 * n, dtype=np.int16).view(np.uint8)
        return rng.integers(-23170, 23171, size=2 * n, dtype=np.int16).view(np.uint8)
    return rng.uniform(-1.0, 1.0, size=2 * n).astype(np.float32).view(np.uint8)


def assert_same_bytes(got, want, outfmt, what=""):
    """Exact equality of output bytes; for f32 any NaN matches any NaN (x86 and CDNA4 propagate
    different NaN payloads through a*c - b*s)."""
    got = np.ascontiguousarray(got).view(np.uint8).reshape(-1)
    want = np.ascontiguousarray(want).view(np.uint8).reshape(-1)
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


def oracle_counters(orc, segments, rate, sn0):
    """The reference's sequential counter (dsp.rs:125-130), sample by sample."""
    out = []
    sn = sn0
    for n, hz in segments:
        a = np.empty(n, np.uint32)
        for k in range(n):
            a[k] = sn
            sn = orc.advance_samplenum(sn, hz, rate, 1)
        out.append(a)
    return (np.concatenate(out) if out else np.empty(0, np.uint32)), sn
