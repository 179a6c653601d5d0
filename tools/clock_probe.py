"""Sustained launches of one plan for a few seconds while sampling rocm-smi (clocks, power): python tools/clock_probe.py CASE [opts]"""
import calendar, os, subprocess, sys, threading, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, doppler_amd
RATE = 1024000
case = sys.argv[1]
kv = dict(a.split("=") for a in sys.argv[2:])
secs = float(kv.pop("secs", 3))
if case.startswith("track"):
    segs = bench.track_segments(int(case[5:]), RATE, "i16", calendar.timegm((2015, 1, 22, 19, 48, 0)))
else:
    segs = [(int(kv.pop("n", 268435456)), float(case[5:]))]
n = sum(c for c, _ in segs)
ctx = doppler_amd.Context(0)
ctx.set_options(**{k: int(v) for k, v in kv.items()})
plan = ctx.plan_segments(segs, RATE)
dev = torch.device("cuda:0")
x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev)
out = torch.empty(2 * n, dtype=torch.int16, device=dev)
st = torch.cuda.current_stream()
samples = []
stop = False
def smi():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            j = json.loads(r.stdout)["card0"]
            samples.append({k: v for k, v in j.items() if "sclk" in k or "mclk" in k or "fclk" in k or "ower" in k})
        except Exception as e:
            samples.append({"err": str(e)[:80]})
        time.sleep(0.05)
th = threading.Thread(target=smi); th.start()
t0 = time.time(); k = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.time() - t0 < secs:
    e0.record(st)
    for _ in range(50):
        plan.run(x.data_ptr(), "i16", out.data_ptr(), "i16", st.cuda_stream)
    e1.record(st); st.synchronize(); k += 50
    last = e0.elapsed_time(e1) / 50
stop = True; th.join()
print(case, kv, "launches", k, "last burst ms/launch", round(last, 4), "GB/s", round(n * 8 / last / 1e6, 1))
for s in samples[:2] + samples[len(samples)//2:len(samples)//2+2] + samples[-2:]:
    print(s)
