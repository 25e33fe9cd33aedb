#!/usr/bin/env python3
"""Diagnostic: where the host time of the drop-in loop goes (worker.lookahead_batches over tensor files + one _hip_predict per
batch of 1000), pileup and full alignment, under cProfile.  usage: dropin_profile.py [pileup|full_alignment] [group_windows]"""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clair3_amd import predict, synthetic as syn, worker  # noqa: E402
from tests.test_parity_gpu import make_model  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "pileup"
ch, indel = (18, False) if kind == syn.PILEUP else (8, True)
m = make_model(kind, ch, indel, syn.make_state_dict(kind, ch, indel, seed=1))
gw = int(sys.argv[2]) if len(sys.argv) > 2 else worker.group_windows_for(m)
per_file, n_files = (10000, 20) if kind == syn.PILEUP else (4000, 8)
xb = syn.make_windows(kind, 1000, seed=2, channels=ch)
d = tempfile.mkdtemp(prefix="c3_dropin_")
tiled = np.concatenate([xb] * (per_file // 1000))
info = "\n".join("chrS:%d:%s\t30-RA 30 " % (i + 1, "ACGT" * 8 + "A") for i in range(per_file)) + "\n"
for fi in range(n_files):
    np.save(os.path.join(d, f"t{fi}.npy"), tiled)
    open(os.path.join(d, f"t{fi}.info"), "w").write(info)
lst = os.path.join(d, "list")
open(lst, "w").write("\n".join(f"t{i}" for i in range(n_files)) + "\n")


def loop():
    n = 0
    for X, _, _ in worker.lookahead_batches(m, worker.iter_tensor_files(lst), 1000, predict._PENDING, depth=2, group_windows=gw):
        n += len(predict._hip_predict(m, None, X))
    return n


loop()
t0 = time.perf_counter()
n = loop()
el = time.perf_counter() - t0
print(f"{kind}: group {gw}: {n / el:,.0f} windows/s ({el * 1e3:.1f} ms for {n} windows in {n_files} files)")
pr = cProfile.Profile()
pr.enable()
loop()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
shutil.rmtree(d, ignore_errors=True)
