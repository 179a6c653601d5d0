import time, sys, os
sys.path.insert(0, os.getcwd())
import torch, doppler_amd
from doppler_amd import shard
torch.cuda.init(); x=torch.zeros(16,device="cuda")
t=time.perf_counter(); ctx=doppler_amd.Context(0); print("ctx create %.2f ms"%((time.perf_counter()-t)*1e3))
for i in range(3):
    t=time.perf_counter(); sn=shard.chunk_seed(5000.0,1024000,0); t1=time.perf_counter(); p=ctx.plan_const(5000.0,1024000,268435456,samplenum=sn); t2=time.perf_counter()
    print("seed %.3f ms plan %.3f ms"%((t1-t)*1e3,(t2-t1)*1e3)); p.close()
