import sys, math
sys.path.insert(0, ".")
import torch, doppler_amd
from tools.sweep import time_launches
ctx = doppler_amd.Context(0)
rate = 1024000; T = 300; n = T * rate
dev = torch.device("cuda:0")
x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev)
out = torch.empty(2 * n, dtype=torch.int16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(name, segs):
    for variant, label in ((4, "tile"), (3, "auto")):
        run1(name + " [" + label + "]", segs, variant)
def run1(name, segs, variant):
    ctx.set_tuning(128, 2, variant)
    plan = ctx.plan_segments(segs, rate)
    d = doppler_amd.plan_describe(segs, rate)[0]
    tab = sum(s["period"] for s in d if s["lut_len"])
    direct = sum(s["count"] for s in d if not s["lut_len"])
    avg, med, mn = time_launches(lambda: plan.run(x.data_ptr(), "i16", out.data_ptr(), "i16", st), 30)
    print("%-40s stretches %4d  tables %9d  direct %8d  med %.4f ms %.1f GB/s  (best %.1f)" % (name, len(d), tab, direct, med, n * 8 / med / 1e6, n * 8 / mn / 1e6))
    plan.close()
run("300 x P=1024 (k*1000 Hz, k odd)", [(rate, 1000.0 * (2 * k + 1)) for k in range(T)])
run("300 x same 5000 Hz (one stretch)", [(rate, 5000.0) for k in range(T)])
run("300 x P=4096-ish (k*250 Hz)", [(rate, 250.0 * (2 * k + 1)) for k in range(T)])
run("300 x P=16384 (k*62.5 Hz)", [(rate, 62.5 * (2 * k + 1)) for k in range(T)])
run("300 x P=65536 (k*15.625 Hz)", [(rate, 15.625 * (2 * k + 1)) for k in range(T)])
segs = []
for t in range(T):
    rr = 6.9 * math.tanh((t - T / 2) / (T / 8.0))
    segs.append((rate, float(torch.tensor(-(rr * 1000.0 / 299792458.0) * 437505000.0 + 5000.0, dtype=torch.float32))))
run("300 x overpass Doppler (f32 Hz)", segs)
