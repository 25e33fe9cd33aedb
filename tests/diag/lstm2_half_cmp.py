#!/usr/bin/env python3
"""Diagnostic (not a test): LSTM2 half tiles vs full tiles on the same windows -- lstm1_out / lstm2_out / rows bit by bit."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    from clair3_amd import synthetic as syn
    from tests.test_parity_gpu import make_model
    sd = syn.make_state_dict(syn.PILEUP, 18, False, seed=93)
    m = make_model(syn.PILEUP, 18, False, sd, keep=True)
    x = syn.make_pileup_windows(64, seed=94)
    y = m.predict_numpy(x)
    np.savez(sys.argv[1], y=y, h1=m.debug_fetch("lstm1_out", (64, 33, 256)), h2=m.debug_fetch("lstm2_out", (64, 33, 320)))
else:
    out = []
    for v in ("0", "1"):
        fn = f"/tmp/l2h_{v}.npz"
        subprocess.check_call([sys.executable, __file__, fn], env=dict(os.environ, C3HIP_LSTM2_HALF=v))
        out.append(np.load(fn))
    for k in ("h1", "h2", "y"):
        a, b = out[0][k], out[1][k]
        d = a != b
        print(k, "differing", int(d.sum()), "of", d.size, "max", float(np.abs(a - b).max()))
        if d.any() and k == "h2":
            idx = np.argwhere(d)
            print(" first", idx[:10].tolist())
            print(" windows", sorted(set(idx[:, 0].tolist()))[:20], "steps", sorted(set(idx[:, 1].tolist()))[:40])
            for t in (0, 1, 2):
                print(f" fwd t={t}: units differing", sorted(set(np.argwhere(d[:, t, :160])[:, 1].tolist())))
            for t in (32, 31, 30):
                print(f" bwd t={t}: units differing", sorted(set(np.argwhere(d[:, t, 160:])[:, 1].tolist())))
