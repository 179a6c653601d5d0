"""ONE workload of the hot path as a plan + device buffers: the unit every measurement script here runs.

    python tools/prof_case.py CASE [key=value ...]          (under rocprofv3: tools/table_rocprof.sh, tools/prof_pmc.sh)
CASE  track<seconds>[f]   `doppler track` replay of the synthetic ESTCUBE-1-like pass (rate=N: samples per second)
      config4r<rank>      that rank's chunk of BASELINE.json configs[4]: 1 h f32 -> i16 replay in 8 time chunks
      const<shift>        `doppler const --shift <shift>` (n=N samples, default 2^28)
      copy                dpx_debug_copy of 1 GiB (calibration of the HBM-traffic counters: tools/profile_round.sh)
keys  pair=i16:i16  variant=N (dpx_set_tuning)  cast=legacy  rate=N  iters=N  geom=BxV  and any dpx_options field
      (rows_r, walk_span, sub_lg ...).  tools/ab.py times several such cases against each other in one process."""
import calendar
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import doppler_amd  # noqa: E402

BPS = {"i16": 4, "f32": 8}


def make_case(ctx, tokens, buffers=None):
    """tokens: [CASE, 'key=value', ...] -> dict(plan, x, out, it, ot, n, bytes, iters).  buffers: a dict shared between
    cases so that cases of the same size and formats run over the same device memory."""
    case = tokens[0]
    kv = dict(a.split("=") for a in tokens[1:])
    pair = kv.pop("pair", "i16:i16")
    variant = int(kv.pop("variant", 3))
    iters = int(kv.pop("iters", 6))
    rate = int(kv.pop("rate", 1024000))
    geom = kv.pop("geom", None)
    cast = kv.pop("cast", "saturate")          # cast=legacy: dpx_set_i16_cast(DPX_CAST_LEGACY_X86)
    it, ot = pair.split(":")
    seed = 0
    m = re.fullmatch(r"track(\d+)f?", case)
    if m:
        segs = bench.track_segments(int(m.group(1)), rate, it, calendar.timegm((2015, 1, 22, 19, 48, 0)))
    elif case.startswith("config4r"):
        from doppler_amd import shard
        it, ot = "f32", "i16"
        allsegs = bench.track_segments(3600, rate, it, calendar.timegm((2015, 1, 22, 19, 23, 0)))
        lo, hi = shard.chunk_bounds(3600 * rate, 8, int(case[8:]), bytes_per_sample=8)
        before, segs = shard.segments_for_chunk(allsegs, lo, hi)
        seed = shard.seed_for_segments(before, rate)
    elif case.startswith("const"):
        segs = [(int(kv.pop("n", 268435456)), float(case[5:]))]
    else:
        raise SystemExit("unknown case %r" % case)
    n = sum(c for c, _ in segs)
    ctx.set_tuning(*([int(t) for t in geom.split("x")] if geom else [-1, -1]), variant)
    ctx.set_options(**{k: int(v) for k, v in kv.items()})
    ctx.set_i16_cast(cast == "legacy")
    plan = ctx.plan_segments(segs, rate, samplenum=seed)
    ctx.set_options()
    ctx.set_tuning(-1, -1, 3)
    ctx.set_i16_cast(False)
    dev = torch.device("cuda:0")
    buffers = {} if buffers is None else buffers
    key = (n, it, ot)
    if key not in buffers:
        x = (torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev) if it == "i16" else torch.rand(2 * n, device=dev) * 2 - 1)
        buffers[key] = (x, torch.empty(n * BPS[ot], dtype=torch.uint8, device=dev))
    x, out = buffers[key]
    return dict(plan=plan, x=x, out=out, it=it, ot=ot, n=n, bytes=n * (BPS[it] + BPS[ot]), iters=iters, opts=kv, variant=variant)


def launch(c, stream):
    c["plan"].run(c["x"].data_ptr(), c["it"], c["out"].data_ptr(), c["ot"], stream.cuda_stream)


if __name__ == "__main__":
    ctx = doppler_amd.Context(0)
    if sys.argv[1] == "copy":
        kv = dict(a.split("=") for a in sys.argv[2:])
        a, b = torch.zeros(1 << 28, dtype=torch.int32, device="cuda:0"), torch.empty(1 << 28, dtype=torch.int32, device="cuda:0")
        for _ in range(int(kv.get("iters", 5))):
            ctx.debug_copy(a.data_ptr(), b.data_ptr(), 1 << 30, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print("ran copy", 1 << 30)
        sys.exit(0)
    c = make_case(ctx, sys.argv[1:])
    st = torch.cuda.current_stream()
    for _ in range(c["iters"]):
        launch(c, st)
    st.synchronize()
    print("ran", sys.argv[1], "%s:%s" % (c["it"], c["ot"]), c["variant"], c["opts"], c["n"])
