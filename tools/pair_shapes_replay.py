import calendar, os, statistics, sys
sys.path.insert(0, os.getcwd())
import torch, bench, doppler_amd
RATE = 1024000
ctx = doppler_amd.Context(0); dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
BPS = {"i16": 4, "f32": 8}
res = []
for pair in (("i16", "f32"), ("f32", "f32"), ("f32", "i16"), ("i16", "i16")):
    it, ot = pair
    segs = bench.track_segments(300, RATE, it, calendar.timegm((2015, 1, 22, 19, 48, 0)))
    n = sum(c for c, _ in segs)
    x = (torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev) if it == "i16" else torch.rand(2 * n, device=dev) * 2 - 1)
    out = torch.empty(n * BPS[ot], dtype=torch.uint8, device=dev)
    built = []
    for opts in ({}, {"walk_waves": 2}, {"walk_waves": 4}, {"walk_waves": 5}, {"walk_waves": 8}):
        ctx.set_options(**opts)
        built.append((opts, ctx.plan_segments(segs, RATE), []))
    ctx.set_options()
    for _ in range(2):
        for o, p, ms in built:
            for _ in range(5): p.run(x.data_ptr(), it, out.data_ptr(), ot, st.cuda_stream)
    st.synchronize()
    for _ in range(5):
        for o, p, ms in built:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10): p.run(x.data_ptr(), it, out.data_ptr(), ot, st.cuda_stream)
            e1.record(st); st.synchronize(); ms.append(e0.elapsed_time(e1) / 10)
    for o, p, ms in built:
        m = statistics.median(ms)
        print("%s->%s %-20s %.1f %%" % (it, ot, o, n * (BPS[it] + BPS[ot]) / m / 1e6 / 80), flush=True)
    del x, out
