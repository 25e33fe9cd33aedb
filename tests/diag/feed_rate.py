#!/usr/bin/env python3
"""Rate of the handle's feeder thread (c3_feed_push / c3_feed_wait) next to the submit / wait ring driven from this thread, on groups
of 2000 full-alignment windows in pageable memory (what the drop-in loop's transport moves) -- alone, with the sources memory-mapped
from files, with forked children alive, and with this thread creating / closing shared memory meanwhile (what the reference's loop
does per batch).  Needs an MI355X.   python tests/diag/feed_rate.py [groups] [windows per group]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clair3_amd import synthetic as syn  # noqa: E402
from tests.test_parity_gpu import make_model  # noqa: E402

groups, n = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 2000
m = make_model(syn.FULL_ALIGNMENT, 8, True, syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=1))
m.decode_columns(True)
x = syn.make_fa_windows(n, seed=2)
xs = [x.copy() for _ in range(8)]
m.predict_numpy(x)


def ring(src):
    t = [m.submit(src[i % 8], slot=i % 3) for i in range(3)]
    for i in range(3, groups):
        m.wait(t[i % 3])
        t[i % 3] = m.submit(src[i % 8], slot=i % 3)
    for i in range(groups, groups + 3):
        m.wait(t[i % 3])


def fed(src, meanwhile=None):
    t = [m.feed(src[i % 8]) for i in range(groups)]
    for k in t:
        if meanwhile:
            meanwhile()
        m.feed_wait(k)
    m.feed_drain()


def report(name, fn):
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    print(f"{name:64s} {groups * n / dt:10,.0f} windows/s  ({1e3 * dt / groups:.2f} ms per group of {n})  {m.describe().split()[-1]}", flush=True)


report("submit / wait ring from this thread", lambda: ring(xs))
report("feeder thread", lambda: fed(xs))
d = tempfile.mkdtemp(prefix="c3_feed_")
for i in range(8):
    np.save(os.path.join(d, f"t{i}.npy"), xs[i])
maps = [np.load(os.path.join(d, f"t{i}.npy"), mmap_mode="r") for i in range(8)]
report("ring, sources memory-mapped from files", lambda: ring(maps))
report("feeder, sources memory-mapped from files", lambda: fed(maps))


def shm_traffic():
    from multiprocessing import shared_memory
    for _ in range(2):
        s = shared_memory.SharedMemory(create=True, size=1000 * 121 * 4)
        np.ndarray((1000, 121), np.float32, buffer=s.buf)[:] = 1.0
        s.close()
        s.unlink()


report("feeder, this thread creating / closing shared memory meanwhile", lambda: fed(xs, shm_traffic))
kids = []
for _ in range(8):
    pid = os.fork()
    if pid == 0:
        time.sleep(30)
        os._exit(0)
    kids.append(pid)
if os.environ.get("C3_FEED_SLEEP_AFTER_FORK"):
    time.sleep(float(os.environ["C3_FEED_SLEEP_AFTER_FORK"]))
if os.environ.get("C3_FEED_PER_GROUP"):
    t = [m.feed(xs[i % 8]) for i in range(groups)]
    t0 = time.perf_counter()
    lat = []
    for k in t:
        m.feed_wait(k)
        lat.append(time.perf_counter() - t0)
    print("  rows of group i ready at (ms):", " ".join(f"{1e3 * v:.0f}" for v in lat), flush=True)
report("feeder, eight forked children asleep", lambda: fed(xs))
report("feeder again, eight forked children asleep", lambda: fed(xs))
report("ring, eight forked children asleep", lambda: ring(xs))
for pid in kids:
    os.kill(pid, 9)
    os.waitpid(pid, 0)
busy = []
for _ in range(8):
    pid = os.fork()
    if pid == 0:
        t_end = time.time() + 20
        while time.time() < t_end:
            sum(i * i for i in range(10000))
        os._exit(0)
    busy.append(pid)
report("feeder, eight forked children computing", lambda: fed(xs))
report("feeder again, eight forked children computing", lambda: fed(xs))
report("ring, eight forked children computing", lambda: ring(xs))
for pid in busy:
    os.kill(pid, 9)
    os.waitpid(pid, 0)
