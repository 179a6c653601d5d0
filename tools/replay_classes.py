"""Where the 600 s replay loses: its one-second segments grouped by the number of rows of their walk matrix
(= samples / period), each group timed as a plan of its own (same shifts, same lengths, concatenated)."""
import calendar, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, doppler_amd
RATE = 1024000
segs = bench.track_segments(600, RATE, "i16", calendar.timegm((2015, 1, 22, 19, 48, 0)))
classes = {"rows<5": [], "5-9": [], "10-19": [], "20-39": [], ">=40": []}
for n, hz in segs:
    st, _ = doppler_amd.plan_describe([(n, hz)], RATE, samplenum=1)
    P = max(s["period"] for s in st)
    rows = n / P if P else 0
    key = "rows<5" if rows < 5 else "5-9" if rows < 10 else "10-19" if rows < 20 else "20-39" if rows < 40 else ">=40"
    classes[key].append((n, hz))
ctx = doppler_amd.Context(0)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
built = []
for key, sg in classes.items():
    if not sg:
        continue
    sg = (sg * (1 + 60 // len(sg)))[:max(len(sg), 60)]          # at least 60 seconds of stream per class
    n = sum(c for c, _ in sg)
    plan = ctx.plan_segments(sg, RATE)
    x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev)
    out = torch.empty(2 * n, dtype=torch.int16, device=dev)
    built.append([key, len(classes[key]), n, plan, x, out, [], doppler_amd.plan_layout(sg, RATE)])
for b in built:
    for _ in range(30):
        b[3].run(b[4].data_ptr(), "i16", b[5].data_ptr(), "i16", st.cuda_stream)
st.synchronize()
for _ in range(9):
    for b in built:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            b[3].run(b[4].data_ptr(), "i16", b[5].data_ptr(), "i16", st.cuda_stream)
        e1.record(st); st.synchronize()
        b[6].append(e0.elapsed_time(e1) / 20)
tot = sum(b[1] for b in built)
for b in built:
    med = statistics.median(b[6])
    print(json.dumps({"class": b[0], "segments_in_replay": b[1], "share_pct": round(100 * b[1] / tot, 1), "samples_timed": b[2],
                      "pct_peak": round(b[2] * 8 / med / 1e6 / 80, 1), "single_samples_pct": round(100 * b[7]["single_samples"] / b[2], 2)}))
