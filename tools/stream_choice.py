"""The device-resident entry (one batch in flight) on the legacy default stream against a stream of the caller's own: does the stream matter?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from clair3_amd import synthetic as syn
for name in ("full_alignment", "pileup"):
    kind, ch, indel, batch = (syn.FULL_ALIGNMENT, 8, True, 256) if name == "full_alignment" else (syn.PILEUP, 18, False, 1024)
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    model, _ = bench.build_model(kind, ch, indel, 0)
    x = torch.from_numpy(syn.make_windows(kind, batch, seed=1000, channels=ch)).to(dev)
    side = torch.cuda.Stream(device=dev)
    def run(stream, k=400):
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream(dev)):
            for _ in range(20): model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k): model(x)
            torch.cuda.synchronize()
            return batch * k / (time.perf_counter() - t0)
    for rep in range(3):
        print(f"{name}: default stream {run(None):,.0f}  side stream {run(side):,.0f} windows/s", flush=True)
