"""Measurement sweep for the fused kernel on one MI355X (development tool, not the bench contract).

python tools/sweep.py [--n SAMPLES] [--iters K]
Prints one line per (format pair, variant, workgroup geometry): average kernel ms (HIP events on the
launch stream) and algorithmic GB/s.  Also times the calibration copy kernel.
"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import doppler_amd  # noqa: E402

BPS = {"i16": 4, "f32": 8}


def time_launches(fn, iters, warmup=3):
    st = torch.cuda.current_stream()
    for _ in range(warmup):
        fn()
    st.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(st)
        fn()
        b.record(st)
    st.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts) / len(ts), ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=268435456)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--pairs", default="i16:i16")
    ap.add_argument("--shift", type=float, default=5000.0)
    ap.add_argument("--rate", type=int, default=1024000)
    ap.add_argument("--variants", default="3,4,1")
    ap.add_argument("--geoms", default="128x1,128x2,256x1,256x2")
    ap.add_argument("--track", type=int, default=0, help="seconds of a synthetic overpass at --rate (track-mode plan)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = doppler_amd.Context(0)
    n = args.n
    if args.track:
        import math
        import time
        n = args.track * args.rate
        segs = []
        for t in range(args.track):     # S-shaped Doppler of a LEO pass at 437.505 MHz, one value per second, f32
            rr = 6.9 * math.tanh((t - args.track / 2) / (args.track / 8.0))
            hz = float(torch.tensor(-(rr * 1000.0 / 299792458.0) * 437505000.0 + 5000.0, dtype=torch.float32))
            segs.append((args.rate, hz))
        for pair in args.pairs.split(","):
            it, ot = pair.split(":")
            x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev) if it == "i16" else (torch.rand(2 * n, device=dev) * 2 - 1)
            out = torch.empty(n * BPS[ot], dtype=torch.uint8, device=dev)
            variant = int(args.variants.split(",")[0])
            ctx.set_tuning(256, 1, variant)
            t0 = time.perf_counter()
            plan = ctx.plan_segments(segs, args.rate)
            t_plan = time.perf_counter() - t0
            st = doppler_amd.plan_describe(segs, args.rate)[0]
            kinds = {"rows_or_tile_table": sum(1 for s_ in st if s_["lut_len"]), "direct": sum(1 for s_ in st if not s_["lut_len"]),
                     "direct_samples": sum(s_["count"] for s_ in st if not s_["lut_len"])}
            stream = torch.cuda.current_stream().cuda_stream
            avg, med, mn = time_launches(lambda: plan.run(x.data_ptr(), it, out.data_ptr(), ot, stream), args.iters)
            alg = n * (BPS[it] + BPS[ot])
            print(json.dumps({"kernel": "track", "variant": variant, "pair": pair, "seconds": args.track, "samples": n, "plan_ms": round(t_plan * 1e3, 2),
                              "stretches": len(st), **kinds, "ms_avg": round(avg, 4), "GBps_avg": round(alg / avg / 1e6, 1),
                              "Msps_avg": round(n / avg / 1e3, 0)}), flush=True)
            plan.close()
        return
    stream = torch.cuda.current_stream().cuda_stream
    for pair in args.pairs.split(","):
        it, ot = pair.split(":")
        if it == "i16":
            x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev)
        else:
            x = (torch.rand(2 * n, dtype=torch.float32, device=dev) * 2 - 1)
        out = torch.empty(n * BPS[ot], dtype=torch.uint8, device=dev)
        alg = n * (BPS[it] + BPS[ot])
        if it == ot:
            avg, med, mn = time_launches(lambda: ctx.debug_copy(x.data_ptr(), out.data_ptr(), n * BPS[it], stream), args.iters)
            print(json.dumps({"kernel": "copy", "pair": pair, "ms_avg": round(avg, 4), "ms_min": round(mn, 4),
                              "GBps_avg": round(alg / avg / 1e6, 1), "GBps_best": round(alg / mn / 1e6, 1)}), flush=True)
        for variant in [int(v) for v in args.variants.split(",")]:
            for geom in (args.geoms.split(",") if variant != 3 else ["256x1"]):
                block, vecs = [int(t) for t in geom.split("x")]
                ctx.set_tuning(block, vecs, variant)
                plan = ctx.plan_const(args.shift, args.rate, n)
                avg, med, mn = time_launches(lambda: plan.run(x.data_ptr(), it, out.data_ptr(), ot, stream), args.iters)
                print(json.dumps({"kernel": "shift", "pair": pair, "variant": variant, "block": block, "vecs": vecs,
                                  "ms_avg": round(avg, 4), "ms_med": round(med, 4), "ms_min": round(mn, 4),
                                  "GBps_avg": round(alg / avg / 1e6, 1), "GBps_best": round(alg / mn / 1e6, 1),
                                  "Msps_avg": round(n / avg / 1e3, 0)}), flush=True)
                plan.close()
        del x, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
