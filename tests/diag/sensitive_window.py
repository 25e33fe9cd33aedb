#!/usr/bin/env python3
"""One ill-conditioned pileup window (found by tests/diag/fuzz_parity.py, SEED=7: weights of synthetic._trained_like with a few +-8 LSTM
entries, window 549 of the batch) looked at from every side: the library's fp16x3 forms and its fp32-MFMA forms against the exact (fp64
oracle) rows and against the fp32 rows of the reference arithmetic, layer by layer.  Needs an MI355X.  usage: sensitive_window.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clair3_amd import synthetic as syn  # noqa: E402
from oracle import oracle, torch_port  # noqa: E402
from tests.test_parity_gpu import make_model  # noqa: E402

seed, w = 925999917, 549
sd = syn.make_state_dict(syn.PILEUP, 18, True, seed=seed, peaked=False, trained_like=True)
x = syn.make_pileup_windows(920, seed=seed, recipe="realistic")
lo = w - w % 16
xs = x[lo:lo + 16]
y_o, d_o = oracle.pileup_forward(sd, xs, True, debug=True)
y_t = torch_port.forward(syn.PILEUP, torch_port.to_torch(sd), xs, True).numpy()
k = w - lo
print(f"window {w}: |Y_fp32_pytorch - Y_exact| = {np.abs(y_t[k] - y_o[k]).max():.3e}")
for fp32 in ("0", "1"):
    os.environ["C3HIP_FP32"] = fp32
    m = make_model(syn.PILEUP, 18, True, sd, keep=True)
    y = m.predict_numpy(xs)
    h1 = m.debug_fetch("lstm1_out", (16, 33, 256))
    h2 = m.debug_fetch("lstm2_out", (16, 33, 320))
    l4 = m.debug_fetch("l4_out", (16, 128))
    print(f"C3HIP_FP32={fp32}: |Y - Y_exact| = {np.abs(y[k] - y_o[k]).max():.3e}   |Y - Y_fp32_pytorch| = {np.abs(y[k] - y_t[k]).max():.3e}   "
          f"lstm1_out {np.abs(h1[k] - d_o['lstm1_out'][k]).max():.2e}  lstm2_out {np.abs(h2[k] - d_o['lstm2_out'][k]).max():.2e}  "
          f"l4_out {np.abs(l4[k] - d_o['l4_out'][k]).max():.2e}   (other 15 windows of the tile: rows {np.abs(np.delete(y, k, 0) - np.delete(y_o, k, 0)).max():.2e})")
