"""Per-second segmentation cost: the same five odd-period shifts as one long matrix each, and as one-second segments."""
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import doppler_amd
RATE = 1024000
shifts = [5001.0, -5234.17, 9999.0, 12345.0, 7777.77]
SEC = 1024000
cases = []
for s in shifts:
    cases.append(("one matrix, %g Hz" % s, [(60 * SEC, s)]))
    cases.append(("60 one-second segments, %g Hz / %g Hz alternating" % (s, s + 1), [(SEC, s + (k % 2)) for k in range(60)]))
cases.append(("60 one-second segments cycling the five", [(SEC, shifts[k % 5]) for k in range(60)]))
cases.append(("30 two-second segments cycling the five", [(2 * SEC, shifts[k % 5]) for k in range(30)]))
cases.append(("15 four-second segments cycling the five", [(4 * SEC, shifts[k % 5]) for k in range(15)]))
ctx = doppler_amd.Context(0)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
n = 60 * SEC
x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev)
out = torch.empty(2 * n, dtype=torch.int16, device=dev)
built = []
for name, sg in cases:
    for variant in ((3, 5) if len(sg) == 1 else (3,)):
        ctx.set_tuning(0, 0, variant)
        built.append([name + (" (walk forced)" if variant == 5 else ""), ctx.plan_segments(sg, RATE), [], doppler_amd.plan_layout(sg, RATE, variant=variant)])
ctx.set_tuning(0, 0, 3)
for b in built:
    for _ in range(20):
        b[1].run(x.data_ptr(), "i16", out.data_ptr(), "i16", st.cuda_stream)
st.synchronize()
for _ in range(9):
    for b in built:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            b[1].run(x.data_ptr(), "i16", out.data_ptr(), "i16", st.cuda_stream)
        e1.record(st); st.synchronize()
        b[2].append(e0.elapsed_time(e1) / 20)
for b in built:
    med = statistics.median(b[2]); lay = b[3]
    print("%-72s %5.1f %%  wg %d  matrices %d  single %d  launches rows/walk/tile %d/%d/%d" % (b[0], n * 8 / med / 1e6 / 80, lay["walk_workgroups"], lay["walk_matrices"],
          lay["single_samples"], lay["rows_launches"], lay["walk_launches"], lay["tile_launches"]))
