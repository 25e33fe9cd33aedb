#!/usr/bin/env python3
"""Compare two kernel selections of the full-alignment path tensor by tensor (run on the GPU box).
usage: cmp_variants.py "ENV_A=..,ENV_B=.." "ENV_A=..,.."   (comma-separated VAR=value lists; empty string = defaults)
Prints, per activation, how many elements differ bitwise and the lane/tile structure of the first differences."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from clair3_amd import synthetic as syn  # noqa: E402
from clair3_amd.model import Clair3_F  # noqa: E402

SHAPES = [(45, 17, 64)] * 3 + [(23, 9, 128)] * 3 + [(12, 5, 256)] * 3


def run(envs, sd, x):
    keys = []
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
        keys.append(k)
    m = Clair3_F(add_indel_length=True, predict=True, input_channels=8).keep_activations(True).to("cuda:0")
    m.load_state_dict(sd)
    y = m.predict_numpy(x)
    acts = [m.debug_fetch(f"act{l}", (len(x),) + SHAPES[l]) for l in range(9)]
    for k in keys:
        del os.environ[k]
    return acts, y


def main():
    a_env, b_env = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 70
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=0)
    x = syn.make_fa_windows(n, seed=0, channels=8)
    A, ya = run(a_env, sd, x)
    B, yb = run(b_env, sd, x)
    for l in range(9):
        a, b = A[l], B[l]
        diff = a.view(np.uint32) != b.view(np.uint32)
        msg = f"act{l}: {diff.sum()} of {diff.size} differ bitwise, max |d| = {np.abs(a - b).max():.3e}"
        if diff.any():
            idx = np.argwhere(diff)
            msg += "\n   first: " + "; ".join(f"{i.tolist()} A={a[tuple(i)]:.5f} B={b[tuple(i)]:.5f}" for i in idx[:6])
            msg += f"\n   images {np.unique(idx[:, 0])[:10].tolist()} rows {np.unique(idx[:, 1])[:12].tolist()} cols {np.unique(idx[:, 2])[:12].tolist()} chans {np.unique(idx[:, 3])[:16].tolist()}"
            H, W, C = SHAPES[l]
            tw, th = (W + 1) // 2, (H + 1) // 2
            tile = (idx[:, 0] * th + idx[:, 1] // 2) * tw + idx[:, 2] // 2
            msg += f"\n   tile%32 {np.unique(tile % 32)[:32].tolist()}  tile//32 {np.unique(tile // 32)[:16].tolist()}  pixel-in-tile {np.unique((idx[:, 1] % 2) * 2 + idx[:, 2] % 2).tolist()}"
        print(msg, flush=True)
    print(f"y: max |d| = {np.abs(ya - yb).max():.3e}")


if __name__ == "__main__":
    main()
