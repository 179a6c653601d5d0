"""Run ONE plan a few times (for rocprofv3): python tools/prof_case.py CASE [key=value ...]
CASE: track600 | track300f | const<shift> | synth<rows>;  options: dpx_options fields, variant=N, pair=i16:i16, iters=N, cast=legacy"""
import calendar
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import doppler_amd  # noqa: E402

RATE = 1024000
BPS = {"i16": 4, "f32": 8}
case = sys.argv[1]
kv = dict(a.split("=") for a in sys.argv[2:])
pair = kv.pop("pair", "i16:i16")
variant = int(kv.pop("variant", 3))
iters = int(kv.pop("iters", 6))
RATE = int(kv.pop("rate", RATE))
geom = kv.pop("geom", None)
cast = kv.pop("cast", "saturate")          # cast=legacy: dpx_set_i16_cast(DPX_CAST_LEGACY_X86)
it, ot = pair.split(":")
if case == "track600":
    segs = bench.track_segments(600, RATE, it, calendar.timegm((2015, 1, 22, 19, 48, 0)))
elif case == "track300f":
    segs = bench.track_segments(300, RATE, it, calendar.timegm((2015, 1, 22, 19, 48, 0)))
elif case.startswith("const"):
    segs = [(int(kv.pop("n", 268435456)), float(case[5:]))]
else:
    raise SystemExit("unknown case")
n = sum(c for c, _ in segs)
ctx = doppler_amd.Context(0)
ctx.set_tuning(*([int(t) for t in geom.split("x")] if geom else [0, 0]), variant)
ctx.set_options(**{k: int(v) for k, v in kv.items()})
ctx.set_i16_cast(cast == "legacy")
plan = ctx.plan_segments(segs, RATE)
dev = torch.device("cuda:0")
x = (torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev) if it == "i16" else torch.rand(2 * n, device=dev) * 2 - 1)
out = torch.empty(n * BPS[ot], dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream()
for _ in range(iters):
    plan.run(x.data_ptr(), it, out.data_ptr(), ot, st.cuda_stream)
st.synchronize()
print("ran", case, pair, variant, kv, n)
