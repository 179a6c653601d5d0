"""The 600 s replay's one-second segments in stream order, sorted by rows per matrix, and shuffled: does the ORDER of the
matrices in the launch matter (each variant is a plan of its own over the same buffers; the bytes differ, the work does not)?"""
import calendar, os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, doppler_amd
RATE = 1024000
segs = bench.track_segments(600, RATE, "i16", calendar.timegm((2015, 1, 22, 19, 48, 0)))
def rows_of(seg):
    st, _ = doppler_amd.plan_describe([seg], RATE, samplenum=1)
    P = max(s["period"] for s in st)
    return seg[0] / P if P else 0
keyed = [(rows_of(s), s) for s in segs]
orders = {"stream order": segs, "sorted by rows": [s for _, s in sorted(keyed, key=lambda t: t[0])],
          "sorted, descending": [s for _, s in sorted(keyed, key=lambda t: -t[0])]}
rng = random.Random(5)
sh = list(segs); rng.shuffle(sh); orders["shuffled"] = sh
ctx = doppler_amd.Context(0)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
n = sum(c for c, _ in segs)
x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev)
out = torch.empty(2 * n, dtype=torch.int16, device=dev)
built = {k: dict(plan=ctx.plan_segments(v, RATE), ms=[], lay=doppler_amd.plan_layout(v, RATE)) for k, v in orders.items()}
for b in built.values():
    for _ in range(5):
        b["plan"].run(x.data_ptr(), "i16", out.data_ptr(), "i16", st.cuda_stream)
st.synchronize()
for _ in range(7):
    for b in built.values():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            b["plan"].run(x.data_ptr(), "i16", out.data_ptr(), "i16", st.cuda_stream)
        e1.record(st); st.synchronize()
        b["ms"].append(e0.elapsed_time(e1) / 10)
for k, b in built.items():
    med = statistics.median(b["ms"])
    print("%-20s %.1f us  %.1f %%   single_samples %d  workgroups %d" % (k, med * 1e3, n * 8 / med / 1e6 / 80, b["lay"]["single_samples"], b["lay"]["walk_workgroups"]))
