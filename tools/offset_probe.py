"""Does the relative placement of the input and output buffers matter?  Times the headline plan with the output
buffer shifted by various byte offsets against a 2 MiB-aligned base (HIP events, median of 30 launches)."""
import sys
sys.path.insert(0, ".")
import torch, doppler_amd
from tools.sweep import time_launches
ctx = doppler_amd.Context(0)
TRACK = len(sys.argv) > 1 and sys.argv[1] == "track"     # the 10-minute replay instead of the headline stream
n = 614400000 if TRACK else 268435456
dev = torch.device("cuda:0")
x = torch.randint(-23170, 23171, (2 * n,), dtype=torch.int16, device=dev)
big = torch.empty(4 * n + (64 << 20), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
if TRACK:
    import calendar, bench
    segs = bench.track_segments(600, 1024000, "i16", calendar.timegm((2015, 1, 22, 19, 48, 0)))
    plan = ctx.plan_segments(segs, 1024000, samplenum=0)
else:
    plan = ctx.plan_const(5000.0, 1024000, n)
base = (big.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)
print("in ptr %x  out base %x" % (x.data_ptr(), base))
for off in [0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 3 << 19, 2 << 20, 5 << 20, 16 << 20]:
    o = base + off
    avg, med, mn = time_launches(lambda: plan.run(x.data_ptr(), "i16", o, "i16", st), 30)
    print("offset %9d  med %.4f ms  %.1f GB/s  (best %.1f)" % (off, med, n * 8 / med / 1e6, n * 8 / mn / 1e6))
