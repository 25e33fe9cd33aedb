#!/usr/bin/env python3
"""Diagnostic: host time of one submit / wait pair (the ring of worker.predict_batches), pileup and full alignment, B = 1000."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clair3_amd import synthetic as syn  # noqa: E402
from tests.test_parity_gpu import make_model  # noqa: E402

for kind, ch, indel, mk in ((syn.PILEUP, 18, False, syn.make_pileup_windows), (syn.FULL_ALIGNMENT, 8, True, syn.make_fa_windows)):
    sd = syn.make_state_dict(kind, ch, indel, seed=1)
    m = make_model(kind, ch, indel, sd)
    x = mk(1000, seed=2)
    for _ in range(5):
        m.wait(m.submit(x, slot=0))
    ts, tw, n = 0.0, 0.0, 200
    slots = int(os.environ.get("SLOTS", "3"))
    tickets = [m.submit(x, slot=k) for k in range(slots)]
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        m.wait(tickets.pop(0))
        b = time.perf_counter()
        tickets.append(m.submit(x, slot=i % slots))
        c = time.perf_counter()
        tw += b - a
        ts += c - b
    el = time.perf_counter() - t0
    for t in tickets:
        m.wait(t)
    print(f"{'pileup' if kind == syn.PILEUP else 'full alignment'}: {n * 1000 / el:,.0f} windows/s; per batch {el / n * 1e6:.0f} us = wait {tw / n * 1e6:.0f} + submit {ts / n * 1e6:.0f} us of host time")
