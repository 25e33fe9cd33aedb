"""How much HOST time a forward pass costs (launch overhead), next to its device time: the device-resident entry enqueued K
times without waiting, and the submit / wait ring.  python tests/diag/enqueue_cost.py [full_alignment|pileup] [batch]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

from clair3_amd import _lib, synthetic as syn
from clair3_amd.model import Clair3_F, Clair3_P

kind = syn.PILEUP if len(sys.argv) > 1 and sys.argv[1] == "pileup" else syn.FULL_ALIGNMENT
B = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if kind == syn.PILEUP else 256)
ch = 18 if kind == syn.PILEUP else 8
sd = syn.make_state_dict(kind, ch, True, seed=1)
m = (Clair3_P if kind == syn.PILEUP else Clair3_F)(add_indel_length=True, predict=True, input_channels=ch).to("cuda:0")
m.load_state_dict(sd)
x = syn.make_windows(kind, B, seed=2)
xd = torch.from_numpy(x).cuda()
y = torch.empty((B, m.row_size), dtype=torch.float32, device="cuda")
lib = _lib.lib()
stream = torch.cuda.current_stream().cuda_stream
K = 200


def enqueue():
    _lib.check(lib.c3_predict_device(m._handle, xd.data_ptr(), _lib.DTYPE_I8, B, y.data_ptr(), C.c_void_p(stream)), "c3_predict_device")


for _ in range(20):
    enqueue()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    enqueue()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{m.describe()}")
print(f"device-resident B={B}: host enqueue {1e6 * (t1 - t0) / K:.1f} us per pass, whole {1e6 * (t2 - t0) / K:.1f} us per pass "
      f"({B * K / (t2 - t0):.0f} windows/s)")
# strictly one pass at a time: enqueue + wait
t0 = time.perf_counter()
for _ in range(K):
    enqueue()
    torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"one pass at a time (enqueue, synchronize): {1e6 * (t1 - t0) / K:.1f} us per pass")
# the ring
tk = []
ts = tw = 0.0
t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter()
    tk.append(m.submit(x, slot=i % 4))
    b = time.perf_counter()
    ts += b - a
    if len(tk) == 4:
        m.wait(tk.pop(0))
        tw += time.perf_counter() - b
while tk:
    m.wait(tk.pop(0))
t1 = time.perf_counter()
print(f"ring of 4 slots: {1e6 * (t1 - t0) / K:.1f} us per batch ({B * K / (t1 - t0):.0f} windows/s): submit {1e6 * ts / K:.1f} us, wait {1e6 * tw / K:.1f} us of host time per batch")
