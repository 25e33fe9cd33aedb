"""Weights after THOUSANDS of optimizer steps (tests/test_sgd_trained_gpu.py trains a few hundred on the CPU inside the suite): the reference's
own modules trained through PyTorch-ROCm on the MI355X -- AdamW at the reference's learning rate (and at 3x it), its focal loss, BatchNorm and
dropout in training mode, teacher-labelled synthetic windows (tests/sgd_trained.py) -- then the library on the resulting state dict against the
reference's fp32 CPU rows on 2048 fresh windows, every row: max |dY|, labels of the four heads, what the load-time precision decision and the
range guard said, how large the weights and BatchNorm statistics have grown.  Still not a checkpoint trained on real data (none exists
offline); the closest this image can get.   python tests/diag/long_training_parity.py [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: one HIP runtime for torch and the library)

from clair3_amd import synthetic as syn  # noqa: E402
from tests import refmodels, sgd_trained, util  # noqa: E402
from tests.test_parity_gpu import make_model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
root = refmodels.reference_root_or_skip()
device = "cuda" if torch.cuda.is_available() else "cpu"
out = []
for kind, channels, batch, lr_scale in ((syn.PILEUP, 18, 256, 1.0), (syn.PILEUP, 18, 256, 3.0), (syn.FULL_ALIGNMENT, 8, 64, 1.0), (syn.FULL_ALIGNMENT, 8, 64, 3.0)):
    t0 = time.perf_counter()
    sd, losses = sgd_trained.train_reference(root, kind, channels, True, steps=steps, batch=batch, seed=5, device=device, lr_scale=lr_scale)
    t_train = time.perf_counter() - t0
    n = 2048
    x = syn.make_windows(kind, n, seed=777, channels=channels)
    m = make_model(kind, channels, True, sd)
    y = m.predict_numpy(x)
    y_ref = refmodels.reference_rows(refmodels.reference_model(root, kind, sd, True, channels), x)
    err = float(np.abs(y.astype(np.float64) - y_ref).max())
    heads = ((0, 21), (21, 24), (24, 57), (57, 90))
    labels_differ = [int((y[:, lo:hi].argmax(1) != y_ref[:, lo:hi].argmax(1)).sum()) for lo, hi in heads]
    # rows whose label differs in a head: how far apart the reference's OWN two largest probabilities of that head are there (a dead or
    # saturated head gives exact ties: the arg-max of a tie is decided by the last bit of either side)
    tie_gap = []
    for lo, hi in heads:
        rows_ = np.flatnonzero(y[:, lo:hi].argmax(1) != y_ref[:, lo:hi].argmax(1))
        top2 = np.sort(y_ref[rows_, lo:hi], axis=1)[:, -2:]
        tie_gap.append(float((top2[:, 1] - top2[:, 0]).max()) if len(rows_) else None)
    try:
        util.assert_rows_match(y, y_ref, tol=util.PROB_TOL, what=f"{kind} after {steps} steps")  # 1e-4 + labels outside near-ties, every row
        gate = "passed"
    except AssertionError as e:
        gate = "FAILED: " + str(e)[:300]
    w = {k: float(np.abs(v).max()) for k, v in sd.items() if v.ndim >= 1 and v.size > 1}
    rec = {"network": kind, "steps": steps, "batch": batch, "learning_rate_x": lr_scale, "trained_on": device, "training_seconds": round(t_train, 1),
           "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)], "peak_probability": float(y_ref.max()),
           "rows": n, "max_abs_dy_vs_reference_fp32_cpu_rows": err, "labels_differing_per_head": labels_differ,
           "largest_top2_gap_of_the_reference_row_where_a_label_differs": tie_gap, "gate_1e-4_and_labels": gate,
           "largest_weight": max(w.values()), "largest_weight_in": max(w, key=w.get),
           "largest_running_var": max([float(v.max()) for k, v in sd.items() if k.endswith("running_var")] or [0.0]),
           "smallest_running_var": min([float(v.min()) for k, v in sd.items() if k.endswith("running_var")] or [0.0]),
           "range_status": list(m.range_status()), "describe": m.describe()}
    out.append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "long_training_parity.json"), "w"), indent=1)
