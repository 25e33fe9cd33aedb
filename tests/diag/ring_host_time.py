"""Where the submitting thread's time goes in the submit / wait ring (one handle, three slots in flight): seconds inside submit() and wait()
per batch next to the wall time per batch -- is the ring bound by the host thread or by the device?
python tests/diag/ring_host_time.py [full_alignment|pileup] [batch] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clair3_amd import synthetic as syn  # noqa: E402
from tests.test_parity_gpu import make_model  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "full_alignment"
kind, ch, indel = (syn.FULL_ALIGNMENT, 8, True) if name == "full_alignment" else (syn.PILEUP, 18, False)
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if kind == syn.FULL_ALIGNMENT else 1024)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
if os.environ.get("RING_DIAG_TORCH"):  # as bench.py does: torch imported first, and the device-resident entry run before the ring
    import torch
wseed, xseed = int(os.environ.get("RING_DIAG_WSEED", "0")), int(os.environ.get("RING_DIAG_XSEED", "1000"))  # (bench.py's: 0 and 1000)
m = make_model(kind, ch, indel, syn.make_state_dict(kind, ch, indel, seed=wseed))
x = syn.make_windows(kind, batch, seed=xseed, channels=ch)
if os.environ.get("RING_DIAG_TORCH"):
    xd = torch.from_numpy(x).cuda()
    t0 = time.perf_counter()
    for _ in range(steps):
        m(xd)
    torch.cuda.synchronize()
    print(f"device-resident entry, one in flight: {batch * steps / (time.perf_counter() - t0):,.0f} windows/s")
for slots in (3, 2, 1):
    tickets, ts, tw = [], 0.0, 0.0
    def run(k, timed):
        global ts, tw
        for i in range(k):
            if len(tickets) == slots:
                t0 = time.perf_counter(); m.wait(tickets.pop(0)); tw += (time.perf_counter() - t0) if timed else 0
            t0 = time.perf_counter(); tickets.append(m.submit(x, slot=i % slots)); ts += (time.perf_counter() - t0) if timed else 0
        while tickets:
            t0 = time.perf_counter(); m.wait(tickets.pop(0)); tw += (time.perf_counter() - t0) if timed else 0
    run(10, False)
    t0 = time.perf_counter(); run(steps, True); el = time.perf_counter() - t0
    print(f"{name} B={batch}, {slots} slot(s) in flight: {batch * steps / el:,.0f} windows/s = {1e6 * el / steps:.0f} us per batch; inside submit() {1e6 * ts / steps:.0f} us, "
          f"inside wait() {1e6 * tw / steps:.0f} us, the rest of the loop {1e6 * (el - ts - tw) / steps:.0f} us")
# the staging copy alone (what submit() does first): numpy copy of the same bytes into a page-aligned buffer, one thread
dst = np.empty_like(x)
t0 = time.perf_counter()
for _ in range(50):
    np.copyto(dst, x)
print(f"one-thread copy of a batch ({x.nbytes / 1e6:.2f} MB): {1e6 * (time.perf_counter() - t0) / 50:.0f} us")
