"""writes a synthetic pileup state dict as <base>.bin + <base>.txt for tools/fork_stall_lib.cpp"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clair3_amd import synthetic as syn  # noqa: E402

base = sys.argv[1]
sd = syn.make_state_dict(syn.PILEUP, 18, False, seed=1)
off = 0
with open(base + ".bin", "wb") as fb, open(base + ".txt", "w") as ft:
    for k, v in sd.items():
        v = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
        if v.dtype != np.float32 or v.ndim > 4:
            continue
        sh = list(v.shape) + [0] * (4 - v.ndim)
        ft.write(f"{k} {v.ndim} {sh[0]} {sh[1]} {sh[2]} {sh[3]} {off}\n")
        fb.write(v.tobytes())
        off += v.nbytes
