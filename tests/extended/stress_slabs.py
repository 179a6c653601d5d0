"""Stress: the track-replay stream of tests/test_gpu_cli.py (or, with a third argument N, `const --shift N`) cut into
slabs at random block boundaries (what pipe timing does to the `doppler` command), every cut pattern through the
dpx_stream_* ring, compared with the oracle.  python tests/extended/stress_slabs.py <seed> <trials> [shift]"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import doppler_amd
from helpers import make_iq
from oracle import oracle as orc

rate, freq, off = 256000, 437505000, -2500
rr = 6.8 * np.tanh((np.arange(16) - 6.0) / 2.0)
n = rate * 9 + 2048 * 2 + 55
x = make_iq("i16", n, 5)
const_shift = int(sys.argv[3]) if len(sys.argv) > 3 else None     # third argument: `doppler const --shift N` instead of track
if const_shift is None:
    want, _, log = orc.track_stream(x, "i16", "i16", rate, freq, rr, offset_hz=off)
else:
    want, _ = orc.const_stream(x, "i16", "i16", const_shift, rate)
    log = [float(const_shift)] * ((x.size + 8191) // 8192)
want = np.asarray(want)
spb = 2048
nblocks = (x.size + 8191) // 8192
ctx = doppler_amd.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
    k = int(rng.integers(0, 12))
    cuts = sorted(set(int(c) for c in rng.integers(1, nblocks, size=k))) if k else []
    bounds = [0] + cuts + [nblocks]
    st = doppler_amd.Stream(ctx, "i16", "i16", rate, 0, slab_bytes=64 << 20, n_slabs=3)
    outs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        lo, hi = a * 8192, min(b * 8192, x.size)
        if st.pending() == 3:
            outs.append(st.next())
        buf = st.acquire()
        buf[: hi - lo] = x[lo:hi]
        segs = []
        for blk in range(a, b):
            cnt = min(spb, (x.size - blk * 8192) // 4)
            hz = float(log[blk])
            if segs and segs[-1][1] == hz:
                segs[-1] = (segs[-1][0] + cnt, hz)
            else:
                segs.append((cnt, hz))
        st.submit(hi - lo, segs)
    while st.pending():
        outs.append(st.next())
    got = np.concatenate(outs)
    st.close()
    if got.size != want.size or not np.array_equal(got, want):
        bad += 1
        d = np.flatnonzero(got[: want.size] != want[: got.size])
        print("MISMATCH trial", trial, "cuts", cuts, "first diff byte", d[:1], "n diff", d.size, "sample", d[:1] // 4, flush=True)
print("trials done, bad =", bad)
