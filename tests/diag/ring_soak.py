"""Soak of the host side (not a test): random batch sizes through the blocking call (one piece, or pieces through the ring), through
submit / wait on random free slots with several batches in flight, and through the device-resident entry -- every row compared, bit
for bit, with the row the same window got in one reference pass (a window's row does not depend on the batch it travels in).
python tests/diag/ring_soak.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clair3_amd import synthetic as syn  # noqa: E402
from tests.test_parity_gpu import make_model  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
cases = {}
for kind, ch, indel, pool in ((syn.PILEUP, 18, False, 20000), (syn.FULL_ALIGNMENT, 8, True, 3000)):
    sd = syn.make_state_dict(kind, ch, indel, seed=11)
    m = make_model(kind, ch, indel, sd)
    x = syn.make_windows(kind, pool, seed=12, channels=ch)
    y = np.concatenate([m.predict_numpy(x[i:i + 256]) for i in range(0, pool, 256)])  # the reference pass: single pieces
    cases[kind] = (m, x, y)
t_end = time.time() + budget
n_calls = n_rows = 0
while time.time() < t_end:
    kind = syn.PILEUP if rng.random() < 0.5 else syn.FULL_ALIGNMENT
    m, x, y = cases[kind]
    big = 12000 if kind == syn.PILEUP else 1800
    mode = int(rng.choice([int(v) for v in os.environ['MODES'].split(',')])) if 'MODES' in os.environ else int(rng.integers(0, 3))
    if os.environ.get('VERBOSE'):
        print('call', n_calls, kind, 'mode', mode, flush=True)
    if mode == 0:  # blocking call, sizes on both sides of the piece threshold
        n = int(rng.integers(1, big))
        lo = int(rng.integers(0, len(x) - n + 1))
        got = m.predict_numpy(x[lo:lo + n])
        assert np.array_equal(got, y[lo:lo + n]), ("blocking", kind, lo, n)
        n_rows += n
    elif mode == 1:  # ring: up to four batches in flight on random slots, waited for in random order
        k = int(rng.integers(1, 5))
        slots = rng.permutation(4)[:k]
        tickets = []
        for s in slots:
            n = int(rng.integers(1, 1100 if kind == syn.PILEUP else 300))
            lo = int(rng.integers(0, len(x) - n + 1))
            tickets.append((m.submit(x[lo:lo + n], slot=int(s)), lo, n))
        for j in rng.permutation(k):
            t, lo, n = tickets[j]
            assert np.array_equal(m.wait(t), y[lo:lo + n]), ("ring", kind, lo, n)
            n_rows += n
    else:  # device-resident entry
        n = int(rng.integers(1, 700 if kind == syn.PILEUP else 260))
        lo = int(rng.integers(0, len(x) - n + 1))
        got = m(torch.from_numpy(x[lo:lo + n]).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), y[lo:lo + n]), ("device", kind, lo, n)
        n_rows += n
    n_calls += 1
print(f"ok: {n_calls} calls, {n_rows} rows, every row bit-identical to the single-piece reference pass")
