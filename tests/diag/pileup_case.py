"""One pileup case layer by layer against the oracle: python tests/diag/pileup_case.py <weight seed> <window seed> <n> [trained_like] [indel]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clair3_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.test_parity_gpu import make_model  # noqa: E402

wseed, xseed, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
trained = len(sys.argv) < 5 or sys.argv[4] == "1"
indel = len(sys.argv) < 6 or sys.argv[5] == "1"
sd = syn.make_state_dict(syn.PILEUP, 18, indel, seed=wseed, trained_like=trained)
x = syn.make_pileup_windows(n, seed=xseed, recipe="realistic")
y_o, d = oracle.pileup_forward(sd, x, indel, debug=True)
for fp32 in ("0", "1"):
    os.environ["C3HIP_FP32"] = fp32
    m = make_model(syn.PILEUP, 18, indel, sd, keep=True)
    y = m.predict_numpy(x)
    print(f"C3HIP_FP32={fp32}: {m.describe()}")
    print("  rows: max |dY| per row (worst five):", sorted(((float(np.abs(y[i] - y_o[i]).max()), i) for i in range(n)), reverse=True)[:5])
    for key in ("lstm1_out", "lstm2_out", "l4_out"):
        a = m.debug_fetch(key, d[key].shape)
        e = np.abs(a - d[key])
        per_row = e.reshape(n, -1).max(axis=1)
        print(f"  {key}: max |d| {e.max():.3e} (|ref| max {np.abs(d[key]).max():.3e}); worst rows {np.argsort(per_row)[-3:][::-1].tolist()} "
              f"{np.sort(per_row)[-3:][::-1]}")
        if key != "l4_out":
            t = e.reshape(n, 33, -1).max(axis=(0, 2))
            print(f"     by position: first {t[:3]}, last {t[-3:]}, argmax {int(t.argmax())}")
