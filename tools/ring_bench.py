"""Same-process A/B of how the slab ring (dpx_stream_*) crosses PCIe: tools/ring_bench.py [options]

For every (path, slab size, slabs in flight, host flags) the ring is driven from pinned memory exactly as a producer with
data at hand would: prime the ring, then next -> release -> acquire -> submit until --gib GiB of input have gone through;
time = first submit to last output handed back.  `copy` rows are the same ring with DPX_STREAM_COPY_ONLY (no arithmetic):
what the link gives that configuration.  Rounds are interleaved (config after config, then again), medians reported.
Every arithmetic configuration's first lap is compared byte for byte with the device-resident plan's output (the product's
HBM path; the ring's parity against the oracle is tests/test_gpu_cli.py's and tests/test_gpu_ring.py's business).
"""
import argparse
import itertools
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import doppler_amd  # noqa: E402

FLAGS = {"none": 0, "noncoherent": 0x80000000, "wc": 0x4, "numauser": 0x20000000, "coherent": 0x40000000}
BPS = {"i16": 4, "f32": 8}
WARM_S = 0.0


def flags_of(spec):
    v = 0
    for part in spec.split("+"):
        v |= FLAGS[part]
    return v


def reference_output(ctx, slab, it, ot, shift, rate):
    """The device-resident plan on the same slab content (counter 0)."""
    n = slab.size // BPS[it]
    d_in, d_out = ctx.malloc(slab.size), ctx.malloc(n * BPS[ot])
    try:
        ctx.h2d(d_in, slab)
        plan = ctx.plan_const(float(shift), rate, n)
        plan.run(d_in, it, d_out, ot)
        ctx.synchronize()
        out = np.empty(n * BPS[ot], dtype=np.uint8)
        ctx.d2h(out, d_out)
        plan.close()
        return out
    finally:
        ctx.free(d_in)
        ctx.free(d_out)


def run_ring(ctxs, it, ot, shift, rate, slab_bytes, n_slabs, total_bytes, path, copy_only, fin, fout, pattern, want_first):
    st = doppler_amd.Stream(ctxs, it, ot, rate, slab_bytes=slab_bytes, n_slabs=n_slabs, path=path.split(":")[0], copy_only=copy_only,
                            in_host_flags=fin, out_host_flags=fout, unpaced=":unpaced" in path, no_probe=":noprobe" in path)
    try:
        total_slabs = n_slabs * len(ctxs)
        segs = [(slab_bytes // BPS[it], float(shift))]
        bufs = []
        for _ in range(total_slabs):             # fill every slab once (the timed laps re-submit the same pinned bytes)
            b = st.acquire()
            b[:] = pattern
            bufs.append(b)
        laps = max(2, total_bytes // slab_bytes)
        # first lap untimed: plans, device images, first-touch of the output slabs; checked against the HBM path
        for _ in range(total_slabs):
            st.submit(slab_bytes, segs)
        ok = True
        for k in range(total_slabs):
            v = st.next_view()
            if k == 0 and want_first is not None and not copy_only:
                ok = bool(np.array_equal(v, want_first))
            st.release()
        for _ in range(total_slabs):
            st.acquire()
        if WARM_S > 0:          # a process's first 0.2-0.3 s of heavy PCIe traffic hold one 30-50 ms stall (profiles/r06_ring.md)
            tw = time.perf_counter()
            for _ in range(total_slabs):
                st.submit(slab_bytes, segs)
            while time.perf_counter() - tw < WARM_S:
                st.next_view()
                st.release()
                st.acquire()
                st.submit(slab_bytes, segs)
            for _ in range(total_slabs):
                st.next_view()
                st.release()
            for _ in range(total_slabs):
                st.acquire()
        t0 = time.perf_counter()
        for _ in range(total_slabs):
            st.submit(slab_bytes, segs)
        done = 0
        submitted = total_slabs
        while done < laps:
            st.next_view()
            st.release()
            done += 1
            if submitted < laps:
                st.acquire()
                st.submit(slab_bytes, segs)
                submitted += 1
        dt = time.perf_counter() - t0
        while st.pending():
            try:
                st.next_view()
                st.release()
            except doppler_amd.DspError:
                break
        return laps * slab_bytes / dt, ok, st.describe()
    finally:
        st.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=4.0)
    ap.add_argument("--pair", default="i16:i16")
    ap.add_argument("--shift", type=float, default=5000.0)
    ap.add_argument("--rate", type=int, default=1024000)
    ap.add_argument("--slab-mib", type=float, nargs="+", default=[16])
    ap.add_argument("--slabs", type=int, nargs="+", default=[4])
    ap.add_argument("--paths", nargs="+", default=["staged", "staged_per_slab", "direct", "direct_in", "direct_out"])
    ap.add_argument("--in-flags", nargs="+", default=["none"])
    ap.add_argument("--out-flags", nargs="+", default=["none"])
    ap.add_argument("--copy", action="store_true", help="also run every configuration with DPX_STREAM_COPY_ONLY")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--contexts", type=int, default=1, help="contexts on device 0 (a multi-GPU ring over one device)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--warm-s", type=float, default=0.3, help="untimed cycling before every timed run")
    a = ap.parse_args()
    global WARM_S
    WARM_S = a.warm_s
    it, ot = a.pair.split(":")
    ctxs = [doppler_amd.Context(0) for _ in range(a.contexts)]
    rng = np.random.default_rng(7)
    configs = []
    for mib, ns, path, fi, fo in itertools.product(a.slab_mib, a.slabs, a.paths, a.in_flags, a.out_flags):
        for co in ([False, True] if a.copy else [False]):
            configs.append((int(mib * (1 << 20)) // 8192 * 8192, ns, path, fi, fo, co))
    res = {c: [] for c in configs}
    patterns, wants = {}, {}
    for sb in sorted({c[0] for c in configs}):
        if it == "i16":
            patterns[sb] = rng.integers(-23170, 23171, size=sb // 2, dtype=np.int16).view(np.uint8)
        else:
            patterns[sb] = rng.uniform(-1, 1, size=sb // 4).astype(np.float32).view(np.uint8)
        wants[sb] = reference_output(ctxs[0], patterns[sb], it, ot, a.shift, a.rate)
    bad = []
    for r in range(a.rounds):
        for c in configs:
            sb, ns, path, fi, fo, co = c
            bps, ok, desc = run_ring(ctxs, it, ot, a.shift, a.rate, sb, ns, int(a.gib * (1 << 30)), path, co,
                                     flags_of(fi), flags_of(fo), patterns[sb], wants[sb])
            res[c].append(bps)
            if not ok:
                bad.append(c)
    print("%-16s %-5s %8s %5s %-14s %-14s %9s %9s %9s   (GB/s of INPUT through the ring; x2 on the link for i16:i16)" %
          ("path", "kind", "slab MiB", "slabs", "in flags", "out flags", "median", "min", "max"))
    rows = []
    for c in configs:
        sb, ns, path, fi, fo, co = c
        v = [x / 1e9 for x in res[c]]
        rows.append({"path": path, "copy_only": co, "slab_bytes": sb, "slabs": ns, "in_flags": fi, "out_flags": fo,
                     "GBps_in_median": round(statistics.median(v), 2), "GBps_in_min": round(min(v), 2), "GBps_in_max": round(max(v), 2),
                     "Msamples_per_s": round(statistics.median(res[c]) / BPS[it] / 1e6, 1)})
        print("%-16s %-5s %8.1f %5d %-14s %-14s %9.2f %9.2f %9.2f" % (path, "copy" if co else "shift", sb / (1 << 20), ns, fi, fo,
                                                                        statistics.median(v), min(v), max(v)))
    print("byte-exact against the device-resident plan: %s" % ("ALL" if not bad else "MISMATCH in %r" % sorted(set(bad))))
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"pair": a.pair, "gib": a.gib, "rounds": a.rounds, "contexts": a.contexts, "rows": rows, "mismatches": len(bad)}, f, indent=1)
    for c in ctxs:
        c.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
