import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import bench
from clair3_amd import synthetic as syn
kind, ch, indel, batch = syn.FULL_ALIGNMENT, 8, True, 256
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
model, sd = bench.build_model(kind, ch, indel, 0)
x_host = syn.make_windows(kind, batch, seed=1000, channels=ch)
def ring(tag):
    el, y, els = bench.host_leg_median(model, x_host, 100, 5)
    print(f"{tag}: ring {batch*100/el:,.0f} {[round(batch*100/e) for e in els]}  {model.describe()[-60:]}", flush=True)
ring("fresh handle, nothing else run")
x = torch.from_numpy(x_host).to(dev)
model(x); torch.cuda.synchronize()
ring("after one device-resident call")
t = time.perf_counter() + 0.06
while time.perf_counter() < t:
    model(x)
torch.cuda.synchronize()
ring("after 60 ms of device-resident calls")
for _ in range(300): model(x)
torch.cuda.synchronize()
ring("after 300 more")
m2, _ = bench.build_model(kind, ch, indel, 0)
el, y, els = bench.host_leg_median(m2, x_host, 100, 5)
print(f"a second fresh handle: ring {batch*100/el:,.0f}")
