"""Can a memory-mapped tensor file (np.load(..., mmap_mode='r'): a read-only, file-backed mapping) be page-locked for the GPU, so
that the DMA engine reads the page cache and the staging copy disappears?  hipHostRegister on a FRESH mapping of a fresh file per
variant (default / read-only flag, with and without populating the mapping first), one H2D copy through it, data compared.
python tests/diag/register_mmap_probe.py

Result (round 3): it can -- a fresh read-only mapping registers in ~1 ms per 94 MB and copies at 56 GB/s with the read-only flag
(11 GB/s for the first transfer with the default flags, which is why c3_host_register asks for read-only first).  The drop-in
transport does NOT register the tensor files it maps all the same: hipHostUnregister waits for the device to drain (4.8 ms in
the middle of a job), and with one registration per file the full-alignment loop went from 680 k to 670 k windows/s, the pileup
loop (6 MB files) from 4.7 M to 3.5 M -- with the unregistration on a thread of its own; 490 k / 3.3 M with it in the loop."""
import ctypes as C
import os
import tempfile
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipGetErrorString.restype = C.c_char_p
torch.zeros(1).cuda()
d = tempfile.mkdtemp()
x = np.random.randint(-100, 100, size=(4000, 89, 33, 8), dtype=np.int8)
dev = torch.empty(x.nbytes, dtype=torch.int8, device="cuda")
pinned = torch.empty(x.nbytes, dtype=torch.int8).pin_memory()
k = 0
for populate in (False, True):
    for flags, name in ((0, "hipHostRegisterDefault"), (8, "hipHostRegisterReadOnly")):
        k += 1
        path = os.path.join(d, f"t{k}.npy")
        np.save(path, x)
        m = np.load(path, mmap_mode="r")
        ptr = m.ctypes.data
        t0 = time.perf_counter()
        if populate:
            m._mmap.madvise(22)  # MADV_POPULATE_READ
        t1 = time.perf_counter()
        rc = hip.hipHostRegister(C.c_void_p(ptr), m.nbytes, flags)
        t2 = time.perf_counter()
        msg = hip.hipGetErrorString(rc).decode() if rc else "ok"
        line = f"populate={populate} {name}: madvise {1e3 * (t1 - t0):.2f} ms, register rc={rc} ({msg}) {1e3 * (t2 - t1):.2f} ms"
        if rc == 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc2 = hip.hipMemcpy(C.c_void_p(dev.data_ptr()), C.c_void_p(ptr), m.nbytes, 1)
            t1 = time.perf_counter()
            rc3 = hip.hipMemcpy(C.c_void_p(dev.data_ptr()), C.c_void_p(ptr), m.nbytes, 1)
            t2 = time.perf_counter()
            same = bool((dev.cpu().numpy() == x.reshape(-1)).all())
            line += f"; first H2D {m.nbytes / (t1 - t0) / 1e9:.1f} GB/s, second {m.nbytes / (t2 - t1) / 1e9:.1f} GB/s, identical {same}"
            hip.hipHostUnregister(C.c_void_p(ptr))
        else:
            hip.hipGetLastError()
        print(line)
# the staged way, for scale: memcpy into a pinned buffer (one thread) + H2D
k += 1
path = os.path.join(d, f"t{k}.npy")
np.save(path, x)
m = np.load(path, mmap_mode="r")
t0 = time.perf_counter()
pinned.numpy()[:] = m.reshape(-1)
t1 = time.perf_counter()
hip.hipMemcpy(C.c_void_p(dev.data_ptr()), C.c_void_p(pinned.data_ptr()), m.nbytes, 1)
t2 = time.perf_counter()
print(f"staged: copy into pinned memory {m.nbytes / (t1 - t0) / 1e9:.1f} GB/s (one thread), H2D from it {m.nbytes / (t2 - t1) / 1e9:.1f} GB/s")
