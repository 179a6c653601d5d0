"""(RATE=N in the environment: samples per second; an option set may carry "variant": 1.)
The 600 s replay's one-second segments grouped by the rows of their walk matrix, each group timed as a plan of its own
under several option sets (JSON list of dpx_options dicts in OPTS; default: the planner's own shape against spans of 8 / 16)."""
import calendar, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, doppler_amd
RATE = int(os.environ.get("RATE", "1024000"))     # samples per second of the replay
segs = bench.track_segments(600, RATE, "i16", calendar.timegm((2015, 1, 22, 19, 48, 0)))
edges = [(0, 3), (3, 5), (5, 7), (7, 9), (9, 12), (12, 17), (17, 25), (25, 1 << 30)]
if os.environ.get("EDGES"):     # e.g. EDGES=0,2,3,4,5,6,7,9 : finer classes (upper edge of the last one open)
    e = [float(v) for v in os.environ["EDGES"].split(",")]
    edges = [(e[i], e[i + 1]) for i in range(len(e) - 1)] + [(e[-1], 1 << 30)]
classes = {e: [] for e in edges}
for n, hz in segs:
    st, _ = doppler_amd.plan_describe([(n, hz)], RATE, samplenum=1)
    P = max(s["period"] for s in st)
    rows = n / P if P else 0
    for e in edges:
        if e[0] <= rows < e[1]:
            classes[e].append((n, hz))
shapes = [tuple(sorted(o.items())) for o in json.loads(os.environ.get("OPTS", '[{},{"walk_span":8},{"walk_span":16}]'))]
ctx = doppler_amd.Context(0)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
built = []
for key, sg in classes.items():
    if not sg:
        continue
    want = max(60, int(61440000 // RATE))                        # at least 60 seconds and 61 M samples of stream per class
    rep = (sg * (1 + want // len(sg)))[:max(len(sg), want)]
    n = sum(c for c, _ in rep)
    x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev)
    out = torch.empty(2 * n, dtype=torch.int16, device=dev)
    for sh in shapes:
        o = dict(sh)
        ctx.set_tuning(0, 0, int(o.pop("variant", 3)))      # "variant": 1 in an option set = every corrector per sample (tile kernel)
        ctx.set_options(**o)
        built.append(dict(key=key, nseg=len(sg), n=n, shape=sh, plan=ctx.plan_segments(rep, RATE), x=x, out=out, ms=[]))
ctx.set_options()
ctx.set_tuning(0, 0, 3)
for b in built:
    for _ in range(10):
        b["plan"].run(b["x"].data_ptr(), "i16", b["out"].data_ptr(), "i16", st.cuda_stream)
st.synchronize()
for _ in range(7):
    for b in built:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            b["plan"].run(b["x"].data_ptr(), "i16", b["out"].data_ptr(), "i16", st.cuda_stream)
        e1.record(st); st.synchronize()
        b["ms"].append(e0.elapsed_time(e1) / 10)
tot = sum(len(v) for v in classes.values())
print("%-10s %6s  " % ("rows", "share") + " ".join("%6s" % ("o%d" % i) for i, s in enumerate(shapes)))
t_def = t_best = 0.0
for key in classes:
    row = [b for b in built if b["key"] == key]
    if not row:
        continue
    pct = {b["shape"]: b["n"] * 8 / statistics.median(b["ms"]) / 1e6 / 80 for b in row}
    share = row[0]["nseg"] / tot
    t_def += share / pct[shapes[0]]
    t_best += share / max(pct.values())
    print("%-10s %5.1f%%  " % ("%g-%s" % (key[0], ("%g" % key[1]) if key[1] < 1 << 29 else ""), 100 * share) + " ".join("%6.1f" % pct[s] for s in shapes))
print("options:", [dict(s) for s in shapes])
print("replay composed from the classes: o0 everywhere %.1f %%, best per class %.1f %%" % (1 / t_def, 1 / t_best))
